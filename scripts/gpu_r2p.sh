#!/bin/bash
OUT=gpurun_out/${1:-r2p}; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_dsp -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode dsp --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
python scripts/trace_by_shape.py /tmp/tr_dsp > $OUT/dsp_shapes.txt 2>&1; head -14 $OUT/dsp_shapes.txt | cut -c1-190
cp $(find /tmp/tr_dsp -name "*kernel_stats.csv" | head -1) $OUT/dsp_kernel_stats.csv
timeout 600 python -m pytest tests/test_dsp.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
