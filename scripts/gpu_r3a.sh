#!/bin/bash
# round 3, run A: full GPU suite on the 16-byte-fragment conv kernel, conv micro-benchmarks, bench line
OUT=gpurun_out/${1:-r3a}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -s > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
grep -E "^\[" $OUT/tests.log > $OUT/parity_report.txt
timeout 300 python scripts/conv_micro.py 2>&1 | grep "B=" | cut -c1-330 | tee $OUT/conv_micro.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json
for d in bf16 f32x3; do timeout 300 python bench.py --dtype $d --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d['ms_per_step'],3))"; done
timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=4', round(d['ms_per_step'],3))"
