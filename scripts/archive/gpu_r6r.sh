#!/bin/bash
# round 6, call r: InstanceNorm backward inside the input-gradient epilogue on bf16 pair tensors -- parity, reproducibility, same-box A/B
OUT=gpurun_out/${1:-r6r}; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; HEAD=$PWD/_w_ab/libavc_head.so
timeout 900 python -m pytest tests/test_conv_in_fuse.py tests/test_bf16_pairs.py -q -m gpu -x 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 900 python -m pytest tests/test_graded_configs.py tests/test_engine_random_configs.py -q -m gpu -k "storage or bf16" -x 2>&1 | tail -3 | tee -a $OUT/pytest.txt
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(64), round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "bf16 previous commit + the new schedule defaults" $HEAD --dtype bf16 --tune dec_wgrad_flush=6 --tune dec_wgrad_wgs=192 --tune wgrad_batch=16
  one "bf16 working tree (IN backward fused on pair tensors)" $NEW --dtype bf16
  one "bf16 working tree, conv_in_fuse=0" $NEW --dtype bf16 --tune conv_in_fuse=0
  one "bf16 B=4 previous commit" $HEAD --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 B=4 working tree" $NEW --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 B=64 previous commit" $HEAD --dtype bf16 --batch 64 --steps 100 --warmup 20
  one "bf16 B=64 working tree" $NEW --dtype bf16 --batch 64 --steps 100 --warmup 20
  one "f32 working tree" $NEW
done
python scripts/event_timeline.py --dtype bf16s 2>&1 | tail -9 | tee $OUT/timeline_marks.txt
