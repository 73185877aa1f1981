#!/bin/bash
# round 3, run G: 64 x 128 column tiles (WN = 2)
OUT=gpurun_out/${1:-r3g}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_conv.py -q -m gpu -x -k "fwd_matches or dgrad_matches" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 300 python scripts/conv_micro.py 2>&1 | grep "B=" | cut -c1-400 | tee $OUT/conv_micro.log
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %.3f ms' % ('$*', d['ms_per_step']))"; }
{
run
run --tune tile12_wgs=512
run --tune tile12_wgs=256
run
run --tune tile12_wgs=512
run --mode infer --batch 1024
run --mode infer --batch 1024 --tune tile12_wgs=512
run --mode infer --batch 1024 --tune tile12_wgs=1024
run --dtype bf16
run --dtype bf16 --tune tile12_wgs=512
run --batch 64 --frames 1024
run --batch 64 --frames 1024 --tune tile12_wgs=512
} | tee $OUT/sweep.log
