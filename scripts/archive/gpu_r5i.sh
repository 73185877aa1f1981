# round 5, call I: fused InstanceNorm epilogue on bf16 pair tensors -- GPU parity, then config2 (bf16 storage engine) with conv_in_fuse = 0 / 1
OUT=gpurun_out/${1:-r5i}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_in_fuse.py tests/test_bf16_pairs.py tests/test_engine_random_configs.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_graded_configs.py -x -q -m gpu -k "bf16 and not 1024" 2>&1 | tail -3 | tee -a $OUT/tests.txt
for rep in 1 2 3; do
for f in 0 1; do
  python bench.py --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-profile --tune conv_in_fuse=$f 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('train bf16 B=256 conv_in_fuse=$f', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/ab.log
done
done
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_bf16.json
python - $OUT/bench_bf16.json <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.loads(open(sys.argv[1]).read())
kc = d.get("kernel_classes") or {}
print(round(d["ms_per_step"], 3), {k: (round(v["ms_per_step"], 3), v.get("launches_per_step")) for k, v in kc.items()})
print(d.get("roofline_instnorm", {}).get("all_shapes_per_step"))
PY
