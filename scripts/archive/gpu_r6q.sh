#!/bin/bash
# round 6, call q: encoder weight-gradient batches under the backward chain as NARROW launches (avc_tuning.enc_wgrad_wgs) -- sweep
OUT=gpurun_out/${1:-r6q}; mkdir -p $OUT; export TMPDIR=/tmp
one() { local label="$1"; shift
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(56), round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/sweep.log; }
for rep in 1 2; do
  one "bf16 default (batch 16, full width)" --dtype bf16
  for nb in 4 6 8; do for w in 32 64 96 128; do
    one "bf16 wgrad_batch=$nb enc_wgrad_wgs=$w" --dtype bf16 --tune wgrad_batch=$nb --tune enc_wgrad_wgs=$w
  done; done
  one "bf16 wgrad_batch=3 enc_wgrad_wgs=64" --dtype bf16 --tune wgrad_batch=3 --tune enc_wgrad_wgs=64
  one "bf16 wgrad_batch=2 enc_wgrad_wgs=48" --dtype bf16 --tune wgrad_batch=2 --tune enc_wgrad_wgs=48
  one "f32 default (batch 12, full width)"
  for nb in 4 8; do for w in 64 128; do
    one "f32 wgrad_batch=$nb enc_wgrad_wgs=$w" --tune wgrad_batch=$nb --tune enc_wgrad_wgs=$w
  done; done
  one "bf16 B=4 default" --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 B=4 wgrad_batch=6 enc_wgrad_wgs=64" --dtype bf16 --batch 4 --steps 200 --warmup 20 --tune wgrad_batch=6 --tune enc_wgrad_wgs=64
done
