#!/bin/bash
# run H: the build that is shipped (AVC_CONV_STAGES as compiled) in the latency-bound regimes
OUT=gpurun_out/${1:-r3h}; mkdir -p $OUT; export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %.3f ms' % ('$*', d['ms_per_step']))"; }
{
run --dtype bf16
run --batch 4
run --batch 16
run --batch 64
run --dtype bf16 --batch 64
run
run --mode ragged
run --mode infer --batch 1024
} | tee $OUT/sweep.log
