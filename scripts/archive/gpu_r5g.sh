# round 5, call G: the 16-byte-group weight pack (parity + class time), per-kernel durations of the fused-InstanceNorm conv instances
OUT=gpurun_out/${1:-r5g}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_conv.py tests/test_bf16_pairs.py tests/test_engine.py tests/test_model.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2>/dev/null
python - $OUT/bench.json <<'PY' | tee -a $OUT/ab.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
kc = d.get("kernel_classes") or {}
print(round(d["ms_per_step"], 3), {k: (round(v["ms_per_step"], 3), v.get("launches_per_step")) for k, v in kc.items()})
c2 = d.get("config2_bf16", {})
print("bf16:", c2.get("ms_per_step"), {k: (round(v["ms_per_step"], 3), v.get("launches_per_step")) for k, v in (c2.get("kernel_classes") or {}).items()})
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_single -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-config2 --single-stream > /dev/null 2>&1); cp /tmp/rp_single/trace_kernel_stats.csv $OUT/rocprof_kernel_stats_single_stream.csv
head -40 $OUT/rocprof_kernel_stats_single_stream.csv | cut -c1-200
