# round 6 final measurement set (one call): GPU suite (-m gpu with durations, then its slow half), SQ counters of the whole single-stream
# step, PMC fetch/write of THIS build, rocprofv3 kernel stats (multi-/single-stream, bf16), the default bench line (with its config2 / 3 / 4
# sub-records and the CPU baseline) and the other configurations.  $1 = output name, $2 = git head
OUT=gpurun_out/${1:-r6final}; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_suite.sh $1 slow
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
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bf16 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --dtype bf16 > /dev/null 2>&1); cp /tmp/rp_bf16/trace_kernel_stats.csv $OUT/rocprof_kernel_stats_bf16.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bf16s -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --dtype bf16 --single-stream > /dev/null 2>&1); cp /tmp/rp_bf16s/trace_kernel_stats.csv $OUT/rocprof_kernel_stats_bf16_single_stream.csv
cp $OUT/pmc_fetch_write_summary.json profiles/r06_pmc_fetch_write_summary.json; cp $OUT/sq_step.json profiles/r06_sq_step.json
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 > $OUT/infer_b1024.json 2>/dev/null
python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 --dtype bf16r > $OUT/infer_b1024_bf16r.json 2>/dev/null
python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 --dtype bf16 > $OUT/infer_b1024_bf16s.json 2>/dev/null
python bench.py --batch 64 --frames 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-config2 > $OUT/train_t1024_b64.json 2>/dev/null
python bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-config2 > $OUT/train_b4.json 2>/dev/null
python bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-config2 --dtype bf16 > $OUT/train_b4_bf16.json 2>/dev/null
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_bf16_b256.json 2>/dev/null
python bench.py --dtype f32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-config2 > $OUT/train_f32x3_b256.json 2>/dev/null
python bench.py --mels 512 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-config2 > $OUT/train_m512_b128.json 2>/dev/null
python bench.py --mode ragged --steps 20 --warmup 3 > $OUT/infer_ragged_32pairs.json 2>/dev/null
python bench.py --mode infer --batch 1 --frames 400 --steps 50 --warmup 10 > $OUT/infer_b1_t400.json 2>/dev/null
for f in bench infer_b1024 infer_b1024_bf16r infer_b1024_bf16s train_t1024_b64 train_b4 train_b4_bf16 train_bf16_b256 train_f32x3_b256 train_m512_b128 infer_ragged_32pairs infer_b1_t400; do python -c "import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $OUT/configs.txt; done
python scripts/event_timeline.py > $OUT/timeline_f32.txt 2>&1; tail -9 $OUT/timeline_f32.txt
python scripts/event_timeline.py --dtype bf16s > $OUT/timeline_bf16.txt 2>&1; tail -9 $OUT/timeline_bf16.txt
