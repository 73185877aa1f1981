#!/bin/bash
# usage: bash scripts/gpu_check.sh <tag> [tests] [bench] [prof]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests)
      for f in test_ops_rowops test_ops_conv test_engine test_model test_feed_infer; do
        timeout 900 python -m pytest tests/$f.py -m gpu -q --timeout 300 -s 2>&1 | tail -80 > $OUT/$f.log
        tail -2 $OUT/$f.log
      done
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log ;;
    bench)
      timeout 900 python bench.py --profile-json $OUT/prof_classes.json > $OUT/bench.log 2>&1; tail -2 $OUT/bench.log ;;
    benchq)
      timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-json $OUT/prof_classes.json > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1)
      find $OUT/rocprof -name "*kernel_stats*" | head -3; find $OUT/rocprof -name "*.csv" -size +2M -delete ;;
  esac
done
