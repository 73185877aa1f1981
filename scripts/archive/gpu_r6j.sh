#!/bin/bash
# round 6, call j: does compiling WITHOUT packed-fp32 VALU ops (v_pk_*_f32) make the bf16 storage engine's backward bit-reproducible, and what does it cost?
OUT=gpurun_out/${1:-r6j}; mkdir -p $OUT; export TMPDIR=/tmp
NOPK=$PWD/_w_ab/libavc_nopk_all.so; DEF=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so
AVC_HIP_LIB=$NOPK python scripts/bf16_repro_probe2.py bf16s 2>&1 | grep -v amdgpu.ids | tee $OUT/repro2_nopk.txt | grep -E "tuning|consecutive"
AVC_HIP_LIB=$NOPK python scripts/bf16_repro_probe2.py fp32 2>&1 | grep -v amdgpu.ids | tee $OUT/repro2_fp32_nopk.txt | grep -E "tuning|consecutive"
python scripts/bf16_repro_probe2.py fp32 2>&1 | grep -v amdgpu.ids | tee $OUT/repro2_fp32_default.txt | grep -E "tuning|consecutive"
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(40), round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "f32 default build" $DEF
  one "f32 no packed fp32" $NOPK
  one "bf16 default build" $DEF --dtype bf16
  one "bf16 no packed fp32" $NOPK --dtype bf16
done
