#!/bin/bash
# round 3, run D: three-stage LDS-DMA pipeline (prefetch distance 2, partial vmcnt wait, bare s_barrier)
OUT=gpurun_out/${1:-r3d}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_conv.py tests/test_engine.py tests/test_feed_infer.py -q -m gpu -s -x > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
grep -o "\[gpu[^]]*\][^[]*" $OUT/tests.log | grep -v "x3 dgrad\|x3 fwd" > $OUT/parity_report.txt
timeout 300 python scripts/conv_micro.py 2>&1 | grep "B=" | cut -c1-330 | tee $OUT/conv_micro.log
timeout 300 python scripts/conv_ablate.py 2>&1 | grep "B=" | tee $OUT/conv_ablate.log
timeout 300 python scripts/wgrad_ablate.py 2>&1 | grep "B=" | tee $OUT/wgrad_ablate.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('f32', d['ms_per_step'], {k:(round(v['ms_per_step'],3), v['tflops'] and round(v['tflops'],1)) for k,v in d['kernel_classes'].items()})"
for d in bf16 f32x3; do timeout 300 python bench.py --dtype $d --steps 20 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d['ms_per_step'],3))"; done
timeout 300 python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer B=1024', round(d['ms_per_step'],3))"
timeout 300 python bench.py --mode ragged --steps 20 --warmup 3 2>$OUT/ragged.err | tail -1 | tee $OUT/ragged.json | cut -c1-900
