export TMPDIR=/tmp; mkdir -p gpurun_out/r1y
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --single-stream > /dev/null 2>&1)
python scripts/trace_summary.py /tmp/rp/trace_kernel_trace.csv 45 | tee gpurun_out/r1y/trace_single_stream.txt
