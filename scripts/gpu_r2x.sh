#!/bin/bash
OUT=gpurun_out/${1:-r2x}; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_x3 -o trace -- python $GRAFT_REPO_ROOT/bench.py --dtype f32x3 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $GRAFT_REPO_ROOT/$OUT/rocprof_x3.log 2>&1)
cp $(find /tmp/rp_x3 -name "*kernel_stats.csv" | head -1) $OUT/rocprof_kernel_stats_f32x3_single_stream.csv
python scripts/trace_by_shape.py /tmp/rp_x3 conv_ > $OUT/shapes_f32x3.txt 2>&1; head -24 $OUT/shapes_f32x3.txt | cut -c1-190
