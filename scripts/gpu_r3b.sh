#!/bin/bash
# round 3, run B: full GPU suite, wgrad micro (16-byte fragments), conv ablation, 2 ranks on one GPU over gloo
OUT=gpurun_out/${1:-r3b}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
grep -E "^\[" $OUT/tests.log > $OUT/parity_report.txt
timeout 300 python scripts/conv_micro.py 2>&1 | grep "B=" | cut -c1-330 | tee $OUT/conv_micro.log
timeout 300 python scripts/conv_ablate.py 2>&1 | grep "B=" | tee $OUT/conv_ablate.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('f32', d['ms_per_step'], {k:(round(v['ms_per_step'],3), v['tflops'] and round(v['tflops'],1)) for k,v in d['kernel_classes'].items()})"
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2> $OUT/bench_gloo2.err | tail -1 | cut -c1-600 | tee $OUT/bench_gloo2.json
