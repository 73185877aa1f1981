# rocprofv3 kernel stats of the train step with every kernel on ONE stream (the mode bench.py's kernel_classes are measured in)
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-single}; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $OUT/rocprof_single.log 2>&1)
python $GRAFT_REPO_ROOT/scripts/trace_summary.py $OUT/rocprof_single/trace_kernel_trace.csv 60 > $OUT/trace_single_stream.txt
head -5 $OUT/trace_single_stream.txt
