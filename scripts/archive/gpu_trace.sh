# one traced step of the default bench: per-kernel summary + per-stream timeline.  $1 = output name, $2 = GPU-side pre-sleep per step in ms
# (the host is then a whole step ahead: issue order is out of the picture), $3 = extra bench arguments (e.g. "--tune side_prio=1")
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-trace}; mkdir -p $OUT
(cd /tmp && rm -rf /tmp/rp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-config2 --presleep-ms ${2:-0} $3 > /dev/null 2>&1)
python $GRAFT_REPO_ROOT/scripts/trace_summary.py /tmp/rp/trace_kernel_trace.csv 40 > $OUT/trace_summary.txt
python $GRAFT_REPO_ROOT/scripts/trace_timeline.py /tmp/rp/trace_kernel_trace.csv > $OUT/trace_timeline.txt
head -6 $OUT/trace_timeline.txt
