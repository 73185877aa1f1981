#!/bin/bash
OUT=gpurun_out/${1:-r6s}; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; ST3=$PWD/_w_ab/libavc_st3.so
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(64), round(d['ms_per_step'],3))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "bf16 default" $NEW --dtype bf16
  one "bf16 tile_thr11=8191 (bank on 128-row tiles)" $NEW --dtype bf16 --tune tile_thr11=8191
  one "bf16 three conv stages (AVC_CONV_STAGES_BH=3)" $ST3 --dtype bf16
  one "bf16 bh_ck5=16" $NEW --dtype bf16 --tune bh_ck5=16
  one "bf16 wgrad_batch_wgs=192" $NEW --dtype bf16 --tune wgrad_batch_wgs=192
  one "bf16 dec_wgrad_flush=4 dec_wgrad_wgs=256" $NEW --dtype bf16 --tune dec_wgrad_flush=4 --tune dec_wgrad_wgs=256
  one "bf16 dec_wgrad_flush=8 dec_wgrad_wgs=192" $NEW --dtype bf16 --tune dec_wgrad_flush=8 --tune dec_wgrad_wgs=192
  one "f32 dec_wgrad_flush=6 dec_wgrad_wgs=192 wgrad_batch=16" $NEW --tune dec_wgrad_flush=6 --tune dec_wgrad_wgs=192 --tune wgrad_batch=16
  one "f32 default" $NEW
done
