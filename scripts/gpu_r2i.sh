#!/bin/bash
OUT=gpurun_out/${1:-r2i}; mkdir -p $OUT; export TMPDIR=/tmp
for v in default conv_small=0; do
  args=""; [ "$v" != "default" ] && args="--tune $v"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream $args > $GRAFT_REPO_ROOT/$OUT/trace_$v.log 2>&1)
  python scripts/trace_by_shape.py /tmp/tr_$v conv_ > $OUT/shapes_$v.txt 2>&1
done
head -70 $OUT/shapes_default.txt | cut -c1-200
