#!/bin/bash
# usage: bash scripts/gpu_r2_check.sh <tag> [tests] [smoke] [bench] [benchq] [prof] [profs] [pmc] [cfgs]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -s -x > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
      grep -E "^[.s]*\[(gpu|emu)" $OUT/tests.log | sed -E "s/^[.s]*//" | cut -c1-400 > $OUT/tests_lines.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log ;;
    bench)
      timeout 900 python bench.py --profile-json $OUT/prof_classes.json > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-1500 ;;
    benchq)
      timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-json $OUT/prof_classes.json > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-1200 ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1)
      find $OUT/rocprof -name "*kernel_stats*" | head -3; find $OUT/rocprof -name "*.csv" -size +2M -delete ;;
    profs)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream > $GRAFT_REPO_ROOT/$OUT/rocprof_single.log 2>&1)
      find $OUT/rocprof_single -name "*kernel_stats*" | head -3; find $OUT/rocprof_single -name "*.csv" -size +2M -delete ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/$OUT/pmc_$c.log 2>&1)
      done
      python scripts/pmc_summary.py $OUT/pmc_fetch_write_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE ;;
    cfgs)
      timeout 600 python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 > $OUT/infer_b1024.json 2>$OUT/infer_b1024.err; cut -c1-300 $OUT/infer_b1024.json
      timeout 600 python bench.py --frames 1024 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/train_t1024_b64.json 2>$OUT/train_t1024.err; cut -c1-300 $OUT/train_t1024_b64.json
      timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/train_bf16_b256.json 2>$OUT/train_bf16.err; cut -c1-300 $OUT/train_bf16_b256.json
      timeout 600 python bench.py --feed --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $OUT/train_feed_b256.json 2>$OUT/train_feed.err; cut -c1-300 $OUT/train_feed_b256.json
      timeout 600 python bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-profile > $OUT/train_b4.json 2>$OUT/train_b4.err; cut -c1-300 $OUT/train_b4.json
      timeout 600 python bench.py --dtype f32x3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_f32x3_b256.json 2>$OUT/train_f32x3.err; cut -c1-300 $OUT/train_f32x3_b256.json
      timeout 600 python bench.py --dtype f32x3 --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/infer_f32x3_b1024.json 2>$OUT/infer_f32x3.err; cut -c1-300 $OUT/infer_f32x3_b1024.json
      timeout 600 python bench.py --dtype f32x3 --frames 1024 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/train_f32x3_t1024_b64.json 2>/dev/null; cut -c1-300 $OUT/train_f32x3_t1024_b64.json
      timeout 600 python bench.py --mode dsp --steps 5 --warmup 2 > $OUT/dsp.json 2>$OUT/dsp.err; cut -c1-300 $OUT/dsp.json ;;
  esac
done
