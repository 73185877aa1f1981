#!/bin/bash
OUT=gpurun_out/${1:-r2c}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/conv_micro.py rs > $OUT/conv_micro_rs.log 2>&1; cat $OUT/conv_micro_rs.log | cut -c1-400
timeout 1300 python -m pytest tests -m gpu -q --timeout 600 -s > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
grep -E "^\[(gpu|emu)" $OUT/tests.log | cut -c1-260 > $OUT/tests_lines.log
for v in "conv_rs=0" "conv_rs=1"; do
  tag=$(echo "$v" | tr ' =' '__')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof_$tag -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --single-stream --tune $v > $GRAFT_REPO_ROOT/$OUT/rocprof_$tag.log 2>&1)
  find $OUT/rocprof_$tag -name "*.csv" -size +2M -delete
  f=$(find $OUT/rocprof_$tag -name "*kernel_stats*" | head -1); echo "== $v $f"; head -25 $f | cut -c1-160
done
