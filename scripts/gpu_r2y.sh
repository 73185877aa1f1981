#!/bin/bash
OUT=gpurun_out/${1:-r2y}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_graded_configs.py tests/test_engine.py tests/test_ops_conv.py -x -q -m gpu -k "x3" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 300 python scripts/conv_micro.py x3 2>&1 | tail -5 | cut -c1-330 | tee $OUT/conv_micro_x3_1x1.log
for d in f32 f32x3 f32x3; do timeout 300 python bench.py --dtype $d --steps 30 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d['ms_per_step'],3))"; done
timeout 300 python bench.py --dtype f32x3 --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer f32x3', round(d['ms_per_step'],3))"
