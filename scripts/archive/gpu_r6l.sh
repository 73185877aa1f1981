#!/bin/bash
# round 6, call l: 16-byte staged bf16 weight gradient (csrc/conv_wgrad.hip, BHX) -- parity, steady-state per-chunk time, same-box A/B of the step
OUT=gpurun_out/${1:-r6l}; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; B3=$PWD/_w_ab/libavc_bhx3.so; HEAD=$PWD/_w_ab/libavc_head.so
timeout 900 python -m pytest tests/test_bf16_pairs.py -q -m gpu -x 2>&1 | tail -4 | tee $OUT/pytest_pairs.txt
timeout 900 python -m pytest tests/test_graded_configs.py -q -m gpu -k "storage" -x 2>&1 | tail -4 | tee -a $OUT/pytest_pairs.txt
for l in $HEAD $NEW; do AVC_HIP_LIB=$l python scripts/wgrad_bh_steady.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wgrad_steady.log; done
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(48), round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "bf16 HEAD (previous commit)" $HEAD --dtype bf16
  one "bf16 working tree" $NEW --dtype bf16
  one "bf16 16-byte staging, wgrad_batch_wgs=512" $NEW --dtype bf16 --tune wgrad_batch_wgs=512
  one "bf16 B=4 16-byte staging" $NEW --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "f32 (unchanged)" $NEW
done
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16.json 2>$OUT/bench_bf16.err
python -c "
import json; d=json.loads(open('$OUT/bench_bf16.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('kernel_classes')))" | tee $OUT/classes_bf16.txt
python scripts/event_timeline.py --dtype bf16s 2>&1 | tail -9 | tee $OUT/timeline_marks.txt
