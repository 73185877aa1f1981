#!/bin/bash
# fp32x3 mode (opt-in split-bf16 products): whole-model parity at the graded shapes + bench lines
OUT=gpurun_out/${1:-r2w}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_graded_configs.py tests/test_engine.py tests/test_ops_conv.py -x -q -m gpu -k "x3" -s > $OUT/tests.log 2>&1; grep -E "fp32x3|grad rel-L2|passed|failed|^E" $OUT/tests.log | cut -c1-330 | tail -10
timeout 600 python bench.py --dtype f32x3 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/train_f32x3_b256.json 2>$OUT/train_f32x3.err; cut -c1-900 $OUT/train_f32x3_b256.json
timeout 600 python bench.py --dtype f32x3 --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/infer_f32x3_b1024.json 2>$OUT/infer_f32x3.err; cut -c1-300 $OUT/infer_f32x3_b1024.json
timeout 600 python bench.py --dtype f32x3 --frames 1024 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/train_f32x3_t1024_b64.json 2>/dev/null; cut -c1-300 $OUT/train_f32x3_t1024_b64.json
