#!/bin/bash
# round 3: final artefacts on the final build -> gpurun_out/r3final (copied into profiles/r03_* afterwards)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3final}; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# 1. PMC passes FIRST (bench.py reads the summary of THIS build): separate --pmc runs, kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/pmc_$c.log 2>&1)
done
AVC_GIT_HEAD=${AVC_GIT_HEAD:-unknown} python scripts/pmc_summary.py profiles/r03_pmc_fetch_write_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
cp profiles/r03_pmc_fetch_write_summary.json $OUT/
# 2. the bench line (with cpu baseline and the PMC traffic of this build)
timeout 900 python bench.py --steps 20 --warmup 5 --profile-json $OUT/kernel_classes.json > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
# 3. other configurations
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1), d['unit'])"; }
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_bf16_b256.json 2>/dev/null; j bf16 < $OUT/train_bf16_b256.json
timeout 300 python bench.py --dtype bf16r --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_bf16r_b256.json 2>/dev/null; j bf16r < $OUT/train_bf16r_b256.json
timeout 300 python bench.py --dtype bf16 --batch 64 --frames 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train_bf16_t1024_b64.json 2>/dev/null; j bf16_t1024 < $OUT/train_bf16_t1024_b64.json
timeout 300 python bench.py --dtype bf16 --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/infer_bf16_b1024.json 2>/dev/null; j bf16_infer1024 < $OUT/infer_bf16_b1024.json
timeout 300 python bench.py --dtype f32x3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_f32x3_b256.json 2>/dev/null; j f32x3 < $OUT/train_f32x3_b256.json
timeout 300 python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/infer_b1024.json 2>/dev/null; j infer1024 < $OUT/infer_b1024.json
timeout 300 python bench.py --batch 64 --frames 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train_t1024_b64.json 2>/dev/null; j t1024 < $OUT/train_t1024_b64.json
timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/train_b4.json 2>/dev/null; j b4 < $OUT/train_b4.json
timeout 300 python bench.py --mels 512 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/train_m512_b128.json 2>/dev/null; j m512 < $OUT/train_m512_b128.json
timeout 300 python bench.py --mode ragged --steps 20 --warmup 3 > $OUT/infer_ragged_32pairs.json 2>/dev/null; j ragged < $OUT/infer_ragged_32pairs.json
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2> $OUT/bench_gloo2.err | tail -1 > $OUT/bench_gloo2.json; j gloo2 < $OUT/bench_gloo2.json
# 4. rocprofv3 kernel stats: multi-stream (normal) and single-stream
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_multi -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile > $OUT/rocprof_multi.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $OUT/rocprof_single.log 2>&1)
find $OUT/rocprof_multi -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
find $OUT/rocprof_single -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats_single_stream.csv \;
rm -rf $OUT/rocprof_multi $OUT/rocprof_single
head -8 $OUT/rocprof_kernel_stats_single_stream.csv | cut -c1-200
# 5. the GPU suite (SKIP_TESTS=1: it ran in its own call, scripts/gpu_suite.sh)
if [ -z "$SKIP_TESTS" ]; then
timeout 2400 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
grep -o "\[gpu[^]]*\][^[]*" $OUT/tests.log | grep -v "x3 dgrad\|x3 fwd" > $OUT/gpu_parity_report.txt
fi
