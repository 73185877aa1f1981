#!/bin/bash
# round-2 A/B: one-shot short-row conv kernel (conv_small.hip)
OUT=gpurun_out/${1:-r2h}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_conv.py tests/test_engine.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 300 python scripts/conv_micro.py small > $OUT/conv_micro_small.log 2>&1; cut -c1-330 $OUT/conv_micro_small.log
EXTRA="" bash scripts/gpu_tune.sh ${1:-r2h} default "conv_small=0" default "conv_small=0"
EXTRA="--single-stream" bash scripts/gpu_tune.sh ${1:-r2h}_ss default "conv_small=0"
