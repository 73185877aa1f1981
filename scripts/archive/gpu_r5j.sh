# round 5, call J: InstanceNorm BACKWARD inside the dgrad epilogue -- GPU parity, then A/B of the train step with conv_in_fuse = 0 / 1
OUT=gpurun_out/${1:-r5j}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_in_fuse.py tests/test_engine.py tests/test_model.py tests/test_graded_configs.py -x -q -m gpu -k "not 1024" 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2 3; do
for f in 0 1; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 --tune conv_in_fuse=$f 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('train B=256 conv_in_fuse=$f', 'ms/step', round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/ab.log
  python bench.py --batch 4 --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-config2 --tune conv_in_fuse=$f 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('train B=4   conv_in_fuse=$f', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/ab.log
done
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config2 > $OUT/bench_fused.json 2>/dev/null
python - $OUT/bench_fused.json <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
kc = d.get("kernel_classes") or {}
print(round(d["ms_per_step"], 3), {k: (round(v["ms_per_step"], 3), v.get("launches_per_step")) for k, v in kc.items()})
ri = d.get("roofline_instnorm", {})
print("instnorm dominant", ri.get("frac"), "all shapes:", ri.get("all_shapes_per_step"), "back to back:", (ri.get("all_shapes_back_to_back") or {}).get("gbs"))
PY
