#!/bin/bash
# round-2: split-bf16 conv kernel inside the plan (opt-in) -- parity + step time
OUT=gpurun_out/${1:-r2t}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine.py tests/test_ops_conv.py -x -q -m gpu -k "x3 or m80-4-128" -s > $OUT/tests.log 2>&1; grep -E "grad rel-L2|passed|failed|^E" $OUT/tests.log | cut -c1-260 | tail -8
EXTRA="" bash scripts/gpu_tune.sh ${1:-r2t} default "conv_x3=1" default "conv_x3=1"
EXTRA="--single-stream" bash scripts/gpu_tune.sh ${1:-r2t}_ss default "conv_x3=1"
EXTRA="--mode infer --batch 1024 --steps 10 --warmup 3" bash scripts/gpu_tune.sh ${1:-r2t}_inf default "conv_x3=1"
