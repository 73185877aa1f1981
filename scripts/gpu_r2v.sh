#!/bin/bash
# split-bf16 products in the weight-gradient kernel (opt-in): parity + A/B
OUT=gpurun_out/${1:-r2v}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_conv.py -x -q -m gpu -k "wgrad_x3" -s > $OUT/tests.log 2>&1; grep -E "wgrad x3|passed|failed|^E" $OUT/tests.log | cut -c1-200 | tail -6
EXTRA="" bash scripts/gpu_tune.sh ${1:-r2v} default "wgrad_x3=1" "wgrad_x3=1 conv_x3=1" default "wgrad_x3=1" "wgrad_x3=1 conv_x3=1"
EXTRA="--single-stream" bash scripts/gpu_tune.sh ${1:-r2v}_ss default "wgrad_x3=1" "wgrad_x3=1 conv_x3=1"
