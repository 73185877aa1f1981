#!/bin/bash
# round 3, call t: where does the B = 4 step go?  kernel trace (span / union-busy / sum), single-stream vs multi-stream, host-issue probe
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3t; mkdir -p $O; export TMPDIR=/tmp
for v in "" "--single-stream" "--dtype bf16" "--dtype bf16 --single-stream"; do
  timeout 300 python bench.py --batch 4 $v --steps 50 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=4 $v', round(d['ms_per_step'],3))" >> $O/b4.log
done
cat $O/b4.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp4 -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 6 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2>&1)
python scripts/trace_summary.py /tmp/rp4/trace_kernel_trace.csv 60 > $O/b4_trace_summary.txt 2>&1; head -45 $O/b4_trace_summary.txt
python scripts/trace_timeline.py /tmp/rp4/trace_kernel_trace.csv > $O/b4_trace_timeline.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp4s -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --single-stream --steps 6 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2>&1)
python scripts/trace_summary.py /tmp/rp4s/trace_kernel_trace.csv 10 > $O/b4_trace_summary_single.txt 2>&1; head -3 $O/b4_trace_summary_single.txt
