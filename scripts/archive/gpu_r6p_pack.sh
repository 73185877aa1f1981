#!/bin/bash
# round 6, call p: one-launch weight pack through an LDS stage (contiguous source runs) vs the global-memory gather -- parity + same-box A/B
OUT=gpurun_out/${1:-r6p}; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; BASE=$PWD/_w_ab/libavc_base.so
timeout 900 python -m pytest tests/test_engine.py tests/test_model.py -q -m gpu -k "one_launch_weight_pack or packed or golden" -x 2>&1 | tail -3 | tee $OUT/pytest.txt
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); k=d.get('kernel_classes',{}).get('pack_weights',{}); print('$label'.ljust(48), round(d['ms_per_step'],3), 'pack us', k.get('avg_us'), 'GB/s', k.get('gbs'), d['config'].get('final_losses'))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "f32 gather from global (previous)" $BASE
  one "f32 staged through LDS" $NEW
  one "bf16 previous" $BASE --dtype bf16
  one "bf16 staged" $NEW --dtype bf16
  one "f32 B=4 previous" $BASE --batch 4 --steps 200 --warmup 20
  one "f32 B=4 staged" $NEW --batch 4 --steps 200 --warmup 20
  one "bf16 B=4 previous" $BASE --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 B=4 staged" $NEW --dtype bf16 --batch 4 --steps 200 --warmup 20
done
one "f32x3 previous" $BASE --dtype f32x3
one "f32x3 staged" $NEW --dtype f32x3
one "512 mel B=128 previous" $BASE --mels 512 --batch 128 --steps 10 --warmup 3
one "512 mel B=128 staged" $NEW --mels 512 --batch 128 --steps 10 --warmup 3
