#!/bin/bash
# round 6, call t: three LDS stages for the bf16 1x1 / 32-channel-chunk conv instance only -- parity, op-level time, same-box A/B
OUT=gpurun_out/${1:-r6t}; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; HEAD=$PWD/_w_ab/libavc_head.so
timeout 900 python -m pytest tests/test_bf16_pairs.py tests/test_conv_in_fuse.py -q -m gpu -x 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 900 python -m pytest tests/test_graded_configs.py -q -m gpu -k "storage or optin" -x 2>&1 | tail -8 | tee -a $OUT/pytest.txt
for l in $HEAD $NEW; do echo "lib $l" | tee -a $OUT/conv_1x1.log; AVC_HIP_LIB=$l python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv_1x1.log
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import conv_ablate_bh as c
c.run(256, 1104, 128, 128, 1, 21, 8)
c.run(256, 1104, 128, 128, 1, 21, 8, "d") if False else None
c.run(256, 128, 128, 16, 1, 11, 8)
c.run(256, 128, 80, 128, 1, 11, 8)
PY
done
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(64), round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "bf16 previous commit" $HEAD --dtype bf16
  one "bf16 working tree (3 stages in the 1x1 instance)" $NEW --dtype bf16
  one "bf16 B=4 previous commit" $HEAD --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 B=4 working tree" $NEW --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 infer B=1024 (bf16s) previous commit" $HEAD --dtype bf16 --mode infer --batch 1024 --steps 10 --warmup 3
  one "bf16 infer B=1024 (bf16s) working tree" $NEW --dtype bf16 --mode infer --batch 1024 --steps 10 --warmup 3
done
