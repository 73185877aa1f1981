# round 4, call B: the stream-K weight gradient (one launch per kernel instance + one chip-wide reduce launch per flush; first run of this script: the in-kernel last-arriver reduce) on the GPU: parity tests, micro, bench
OUT=gpurun_out/${1:-r4b}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_conv.py tests/test_engine.py tests/test_bf16_pairs.py -x -q -m gpu -k "wgrad or golden or gradient or determin" > $OUT/pytest_wgrad.log 2>&1; tail -5 $OUT/pytest_wgrad.log
python scripts/wgrad_ablate.py > $OUT/wgrad_ablate.log 2>&1; cat $OUT/wgrad_ablate.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config2 > $OUT/bench_f32.json 2> $OUT/bench_f32.err; python - $OUT <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+'/bench_f32.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['kernel_classes']))
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config2 --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; python - $OUT <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+'/bench_bf16.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['kernel_classes']))
PY
