#!/bin/bash
# round-2 A/B: stride-2 dgrad with one column parity per wave
OUT=gpurun_out/${1:-r2k}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_conv.py tests/test_engine.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 300 python scripts/conv_micro.py s2 > $OUT/conv_micro_s2.log 2>&1; cut -c1-330 $OUT/conv_micro_s2.log
EXTRA="" bash scripts/gpu_tune.sh ${1:-r2k} default "dgrad_par=0" default "dgrad_par=0"
EXTRA="--single-stream" bash scripts/gpu_tune.sh ${1:-r2k}_ss default "dgrad_par=0"
