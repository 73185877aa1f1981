#!/bin/bash
# round 3, run E: tuning sweeps on the final kernels (A/B in separate processes, same box), full GPU suite
OUT=gpurun_out/${1:-r3e}; mkdir -p $OUT; export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.3f ms' % ('$*', d['ms_per_step']))"; }
{
run
run --tune wgrad_batch=8
run --tune wgrad_batch=16
run --tune wgrad_batch_wgs=512
run --tune wgrad_batch=6 --tune wgrad_batch_wgs=512
run --tune kg_wgs=0
run --tune kg_wgs=512
run --tune tile_thr11=1000
run --tune tile_thr11=500
run --tune dec_split_min=100000
run --tune ck16_wgs=0
run --tune ck16_wgs=512
run
} | tee $OUT/sweep.log
timeout 1800 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
grep -o "\[gpu[^]]*\][^[]*" $OUT/tests.log | grep -v "x3 dgrad\|x3 fwd" > $OUT/parity_report.txt
