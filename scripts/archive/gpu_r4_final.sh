# round 4 final measurement set (one call; produced profiles/r04_bench.json (first version), r04_pmc_fetch_write_summary.json, r04_rocprof_kernel_stats*.csv, r04_train_*.json, r04_infer_*.json): full GPU test suite, PMC fetch/write summary of THIS build, rocprofv3 kernel stats
# (multi- and single-stream), the default bench line and the other BASELINE configs
OUT=gpurun_out/${1:-r4final}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-config2 > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1)
done
AVC_GIT_HEAD=${2:-unknown} python scripts/pmc_summary.py $OUT/pmc_fetch_write_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_multi -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 > /dev/null 2>&1); cp /tmp/rp_multi/trace_kernel_stats.csv $OUT/rocprof_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --single-stream > /dev/null 2>&1); cp /tmp/rp_single/trace_kernel_stats.csv $OUT/rocprof_kernel_stats_single_stream.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bf16 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --dtype bf16 > /dev/null 2>&1); cp /tmp/rp_bf16/trace_kernel_stats.csv $OUT/rocprof_kernel_stats_bf16.csv
cp $OUT/pmc_fetch_write_summary.json profiles/r04_pmc_fetch_write_summary.json
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 > $OUT/infer_b1024.json 2>/dev/null
python bench.py --batch 64 --frames 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-config2 > $OUT/train_t1024_b64.json 2>/dev/null
python bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-config2 > $OUT/train_b4.json 2>/dev/null
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_bf16_b256.json 2>/dev/null
python bench.py --dtype f32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-config2 > $OUT/train_f32x3_b256.json 2>/dev/null
python bench.py --mels 512 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-config2 > $OUT/train_m512_b128.json 2>/dev/null
python bench.py --mode ragged --steps 20 --warmup 3 > $OUT/infer_ragged_32pairs.json 2>/dev/null
for f in infer_b1024 train_t1024_b64 train_b4 train_bf16_b256 train_f32x3_b256 train_m512_b128 infer_ragged_32pairs; do python -c "import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), round(d['value'],1))"; done
