#!/bin/bash
# per-stream timeline of a bf16-storage train step (rocprofv3 kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o trace -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16s --steps 4 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2>&1)
python scripts/trace_summary.py /tmp/rp/trace_kernel_trace.csv 40 > $OUT/trace_summary.txt
python scripts/trace_timeline.py /tmp/rp/trace_kernel_trace.csv > $OUT/trace_timeline.txt
head -8 $OUT/trace_timeline.txt
