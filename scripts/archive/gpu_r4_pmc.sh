# refresh the build-locked artefacts after a kernel-source change: PMC fetch/write summary of THIS build, rocprofv3 kernel stats
# (multi- and single-stream) and the default bench line.  $1 = output name, $2 = git head
OUT=gpurun_out/${1:-r4pmc}; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-config2 > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1)
done
AVC_GIT_HEAD=${2:-unknown} python scripts/pmc_summary.py $OUT/pmc_fetch_write_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_multi -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 > /dev/null 2>&1); cp /tmp/rp_multi/trace_kernel_stats.csv $OUT/rocprof_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --single-stream > /dev/null 2>&1); cp /tmp/rp_single/trace_kernel_stats.csv $OUT/rocprof_kernel_stats_single_stream.csv
cp $OUT/pmc_fetch_write_summary.json profiles/r04_pmc_fetch_write_summary.json
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json
