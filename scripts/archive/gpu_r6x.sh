#!/bin/bash
# round 6, call x: bf16 weight gradient in chunk GROUPS (one barrier per G chunks) -- parity, steady state, same-box A/B incl. fp32
OUT=gpurun_out/${1:-r6x}; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; HEAD=$PWD/_w_ab/libavc_head.so
timeout 900 python -m pytest tests/test_bf16_pairs.py tests/test_ops_conv.py -q -m gpu -k "wgrad or reproducible or storage" -x 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 900 python -m pytest tests/test_graded_configs.py tests/test_engine.py -q -m gpu -k "storage or optin or determinis or golden" -x 2>&1 | tail -3 | tee -a $OUT/pytest.txt
for l in $HEAD $NEW; do AVC_HIP_LIB=$l python scripts/wgrad_bh_steady.py 2048 8192 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wgrad_steady.log; done
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(48), round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "bf16 previous commit" $HEAD --dtype bf16
  one "bf16 working tree (chunk groups)" $NEW --dtype bf16
  one "bf16 B=4 previous commit" $HEAD --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 B=4 working tree" $NEW --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "f32 previous commit" $HEAD
  one "f32 working tree" $NEW
  one "T=1024 B=64 bf16 previous commit" $HEAD --dtype bf16 --batch 64 --frames 1024 --steps 10 --warmup 3
  one "T=1024 B=64 bf16 working tree" $NEW --dtype bf16 --batch 64 --frames 1024 --steps 10 --warmup 3
done
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16.json 2>$OUT/bench_bf16.err
python -c "
import json; d=json.loads(open('$OUT/bench_bf16.json').read().strip().splitlines()[-1]); print({k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernel_classes'].items()})" | tee $OUT/classes_bf16.txt
python scripts/event_timeline.py --dtype bf16s 2>&1 | tail -8 | tee $OUT/timeline_marks.txt
