#!/bin/bash
# usage: bash scripts/gpu_tune.sh <tag> "<knob=val knob=val>" ...   ("default" = no knob): one train-step bench line per variant
OUT=gpurun_out/$1; shift; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  args=""; for kv in $v; do [ "$kv" != "default" ] && args="$args --tune $kv"; done
  tag=$(echo "$v" | tr ' =' '__')
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $EXTRA $args > $OUT/bench_$tag.json 2>$OUT/bench_$tag.err
  python - "$OUT/bench_$tag.json" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    kc=d.get("kernel_classes",{})
    print(sys.argv[2], "ms/step", round(d["ms_per_step"],3), {k:(round(v["ms_per_step"],3), v["launches_per_step"]) for k,v in kc.items() if k in ("conv_wgrad","slab_reduce","conv_fwd","conv_dgrad")})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
