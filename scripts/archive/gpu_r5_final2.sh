# round 5, last call: the measurement set of the FINAL build (the InstanceNorm-backward fusion changed the kernel sources after gpu_r5_final.sh):
# GPU suite (-m gpu; the property half of the slow set), SQ counters + PMC fetch/write of this build, rocprofv3 stats, the bench lines
OUT=gpurun_out/${1:-r5final2}; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider ) > $OUT/pytest_gpu.txt 2>&1; tail -n 4 $OUT/pytest_gpu.txt
( time timeout 600 python -m pytest tests -q -m "gpu and slow" -k "not 1024" -p no:cacheprovider ) > $OUT/pytest_gpu_slow.txt 2>&1; tail -n 4 $OUT/pytest_gpu_slow.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-config2 --single-stream"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sqA -o pmc -- $B > $GRAFT_REPO_ROOT/$OUT/sqA.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/sqB -o pmc -- $B > $GRAFT_REPO_ROOT/$OUT/sqB.log 2>&1)
AVC_GIT_HEAD=${2:-unknown} python scripts/sq_step_summary.py $OUT/sq_step.json /tmp/sqA /tmp/sqB | tee $OUT/sq_classes.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-config2 > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1)
done
AVC_GIT_HEAD=${2:-unknown} python scripts/pmc_summary.py $OUT/pmc_fetch_write_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_multi -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 > /dev/null 2>&1); cp /tmp/rp_multi/trace_kernel_stats.csv $OUT/rocprof_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --single-stream > /dev/null 2>&1); cp /tmp/rp_single/trace_kernel_stats.csv $OUT/rocprof_kernel_stats_single_stream.csv
cp $OUT/pmc_fetch_write_summary.json profiles/r05_pmc_fetch_write_summary.json; cp $OUT/sq_step.json profiles/r05_sq_step.json
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
python bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-config2 > $OUT/train_b4.json 2>/dev/null
python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 > $OUT/infer_b1024.json 2>/dev/null
python bench.py --batch 64 --frames 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-config2 > $OUT/train_t1024_b64.json 2>/dev/null
for f in train_b4 infer_b1024 train_t1024_b64; do python -c "import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $OUT/configs.txt; done
