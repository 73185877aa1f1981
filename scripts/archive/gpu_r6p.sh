#!/bin/bash
# round 6, call p: schedule knobs of the bf16 step on the final kernels
OUT=gpurun_out/${1:-r6p}; mkdir -p $OUT; export TMPDIR=/tmp
one() { local label="$1"; shift
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(48), round(d['ms_per_step'],3))" | tee -a $OUT/sweep.log; }
for rep in 1 2; do
  one "bf16 default" --dtype bf16
  one "bf16 wgrad_batch_wgs=128" --dtype bf16 --tune wgrad_batch_wgs=128
  one "bf16 wgrad_batch_wgs=192" --dtype bf16 --tune wgrad_batch_wgs=192
  one "bf16 wgrad_batch=6" --dtype bf16 --tune wgrad_batch=6
  one "bf16 wgrad_batch=8" --dtype bf16 --tune wgrad_batch=8
  one "bf16 wgrad_batch=16" --dtype bf16 --tune wgrad_batch=16
  one "bf16 dec_wgrad_flush=4 dec_wgrad_wgs=128" --dtype bf16 --tune dec_wgrad_flush=4 --tune dec_wgrad_wgs=128
  one "bf16 dec_wgrad_flush=6 dec_wgrad_wgs=192" --dtype bf16 --tune dec_wgrad_flush=6 --tune dec_wgrad_wgs=192
  one "bf16 side_prio=0" --dtype bf16 --tune side_prio=0
  one "bf16 kg_wgs=0 ck16_wgs=0" --dtype bf16 --tune ck16_wgs=0
  one "f32 default"
done
