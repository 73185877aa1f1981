#!/bin/bash
# round 3, last build (dense stack rebuilt): PMC passes, the bench line, the other configurations, rocprofv3 kernel stats, smoke
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3final2}; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/pmc_$c.log 2>&1)
done
AVC_GIT_HEAD=${AVC_GIT_HEAD:-unknown} python scripts/pmc_summary.py profiles/r03_pmc_fetch_write_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
cp profiles/r03_pmc_fetch_write_summary.json $OUT/
( time timeout 600 python bench.py --steps 20 --warmup 5 --profile-json $OUT/kernel_classes.json > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -c 600 $OUT/bench.json; cat $OUT/bench.time
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['value'],1), d['unit'])"; }
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_bf16_b256.json 2>/dev/null; j bf16 < $OUT/train_bf16_b256.json
timeout 300 python bench.py --dtype f32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/train_f32x3_b256.json 2>/dev/null; j f32x3 < $OUT/train_f32x3_b256.json
timeout 300 python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/infer_b1024.json 2>/dev/null; j infer1024 < $OUT/infer_b1024.json
timeout 300 python bench.py --batch 64 --frames 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/train_t1024_b64.json 2>/dev/null; j t1024 < $OUT/train_t1024_b64.json
timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-profile > $OUT/train_b4.json 2>/dev/null; j b4 < $OUT/train_b4.json
timeout 300 python bench.py --mode ragged --steps 20 --warmup 3 > $OUT/infer_ragged_32pairs.json 2>/dev/null; j ragged < $OUT/infer_ragged_32pairs.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_multi -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile > $OUT/rocprof_multi.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $OUT/rocprof_single.log 2>&1)
find $OUT/rocprof_multi -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
find $OUT/rocprof_single -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats_single_stream.csv \;
rm -rf $OUT/rocprof_multi $OUT/rocprof_single
grep -i dense $OUT/rocprof_kernel_stats_single_stream.csv | cut -c1-140
timeout 300 python -c "from adaptive_voice_conversion_amd import _lib; _lib.load(); import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
