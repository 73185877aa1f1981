#!/bin/bash
OUT=gpurun_out/${1:-r2r}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_conv.py -x -q -m gpu -k x3 -s > $OUT/tests.log 2>&1; grep -E "x3 fwd|x3 dgrad|passed|failed|^E" $OUT/tests.log | cut -c1-200 | tail -12
timeout 300 python scripts/conv_micro.py x3 > $OUT/conv_micro_x3.log 2>&1; cut -c1-330 $OUT/conv_micro_x3.log
