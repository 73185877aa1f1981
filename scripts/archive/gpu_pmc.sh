OUT=gpurun_out/r1p; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1)
done
python scripts/pmc_summary.py $OUT/pmc_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
ls /tmp/pmc_FETCH_SIZE | head; tail -2 $OUT/pmc_FETCH_SIZE.log | cut -c1-200
