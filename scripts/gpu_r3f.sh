#!/bin/bash
# round 3, run F: decoder backward as two half-batch chains (A/B), lrelu + fp32x3-parametrised GPU tests
OUT=gpurun_out/${1:-r3f}; mkdir -p $OUT; export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.3f ms' % ('$*', d['ms_per_step']))"; }
{
run
run --tune dec_split_min=100000
run
run --tune dec_split_min=100000
run --dtype bf16
run --dtype bf16 --tune dec_split_min=100000
run --batch 64
run --batch 64 --tune dec_split_min=100000
run --batch 4
} | tee $OUT/sweep.log
timeout 1800 python -m pytest tests/test_engine.py tests/test_dsp.py tests/test_graded_configs.py tests/test_feed_infer.py tests/test_ops_rowops.py -q -m gpu -s -k "not matches_oracle_at_graded or 256" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
grep -o "\[gpu[^]]*\][^[]*" $OUT/tests.log | grep -v "x3 dgrad\|x3 fwd" > $OUT/parity_report.txt
