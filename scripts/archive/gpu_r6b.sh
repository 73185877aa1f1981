#!/bin/bash
# round 6, call b: the bf16 weight gradient with a 4-stage producer ring (csrc/conv_wgrad.hip, AVC_WGRAD_STAGES_BH) -- parity first, then
# same-box A/B against 2 / 3 stages (alternative builds under _w_ab/), op-level ablation, per-class table
OUT=gpurun_out/${1:-r6b}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_pairs.py tests/test_graded_configs.py -q -m gpu -k "bf16 or pairs or storage" -x 2>&1 | tail -8 > $OUT/pytest_bf16.txt
timeout 600 python -m pytest tests/test_engine.py -q -m gpu -k "determinis or bf16" 2>&1 | tail -5 >> $OUT/pytest_bf16.txt
cat $OUT/pytest_bf16.txt
one() {  # label, lib, bench args...
  local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(56), round(d['ms_per_step'],3), d['config'].get('final_losses'))" | tee -a $OUT/ab.log
}
S4=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; S2=$PWD/_w_ab/libavc_s2.so; S3=$PWD/_w_ab/libavc_s3.so
for rep in 1 2 3; do
  one "bf16 stages=2 (round 5 behaviour)" $S2 --dtype bf16
  one "bf16 stages=3" $S3 --dtype bf16
  one "bf16 stages=4" $S4 --dtype bf16
  one "bf16 stages=4 wgrad_batch_wgs=512" $S4 --dtype bf16 --tune wgrad_batch_wgs=512
  one "bf16 stages=4 wgrad_batch=24" $S4 --dtype bf16 --tune wgrad_batch=24
  one "bf16 stages=4 B=4" $S4 --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "f32 (unchanged kernels)" $S4
done
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16.json 2>$OUT/bench_bf16.err
python -c "
import json; d=json.loads(open('$OUT/bench_bf16.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('kernel_classes'), indent=0))" > $OUT/classes_bf16.txt
python scripts/event_timeline.py --dtype bf16s > $OUT/timeline_bf16.txt 2>&1
tail -12 $OUT/timeline_bf16.txt
