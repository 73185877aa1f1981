#!/bin/bash
OUT=gpurun_out/${1:-r2u}; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_x3 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream --tune conv_x3=1 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
python scripts/trace_by_shape.py /tmp/tr_x3 conv_ > $OUT/shapes_x3.txt 2>&1
grep -v wgrad $OUT/shapes_x3.txt | head -30 | cut -c1-200
