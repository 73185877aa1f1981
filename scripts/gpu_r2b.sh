#!/bin/bash
# round-2 A/B run: full gpu tests, conv_rs micro, train-step variants
OUT=gpurun_out/${1:-r2b}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -q --timeout 600 -s > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
grep -E "^\[(gpu|emu)" $OUT/tests.log | cut -c1-260 > $OUT/tests_lines.log
timeout 300 python scripts/conv_micro.py rs > $OUT/conv_micro_rs.log 2>&1; cat $OUT/conv_micro_rs.log | cut -c1-400
for v in "default" "conv_rs=0" "wgrad_batch=1" "wgrad_batch=100" "wgrad_batch=12" "wgrad_batch_wgs=256" "conv_rs=0 wgrad_batch=1"; do
  args=""; for kv in $v; do [ "$kv" != "default" ] && args="$args --tune $kv"; done
  tag=$(echo "$v" | tr ' =' '__')
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $args > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python - "$OUT/bench_$tag.json" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    kc=d.get("kernel_classes",{})
    print(sys.argv[2], "ms/step", round(d["ms_per_step"],3), {k:(round(v["ms_per_step"],3), v["launches_per_step"]) for k,v in kc.items()})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
