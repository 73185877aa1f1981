#!/bin/bash
OUT=gpurun_out/${1:-r2o}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dsp.py -x -q -m gpu -s > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 600 python bench.py --mode dsp --steps 5 --warmup 2 > $OUT/dsp.json 2>$OUT/dsp.err; cut -c1-1500 $OUT/dsp.json; tail -3 $OUT/dsp.err
