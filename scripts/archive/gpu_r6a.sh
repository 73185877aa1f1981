#!/bin/bash
# round 6, call a: where the bf16 (configs[2]) step is -- knob sweep on the shipped kernels, op-level ablations on pair tensors, event timeline
OUT=gpurun_out/${1:-r6a}; mkdir -p $OUT; export TMPDIR=/tmp
one() {  # label, bench args...
  local label="$1"; shift
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(56), round(d['ms_per_step'],3))" | tee -a $OUT/sweep.log
}
for rep in 1 2; do
  one "f32 default"
  one "bf16 default" --dtype bf16
  one "bf16 tile12_wgs=256" --dtype bf16 --tune tile12_wgs=256
  one "bf16 tile12_wgs=128" --dtype bf16 --tune tile12_wgs=128
  one "bf16 tile_thr11=255 (128x64 above 255 wgs)" --dtype bf16 --tune tile_thr11=255
  one "bf16 bh_ck5=16" --dtype bf16 --tune bh_ck5=16
  one "bf16 dec_split_min=100000" --dtype bf16 --tune dec_split_min=100000
  one "bf16 conv_in_fuse=0" --dtype bf16 --tune conv_in_fuse=0
  one "bf16 wgrad_batch=24" --dtype bf16 --tune wgrad_batch=24
  one "bf16 wgrad_batch_wgs=512" --dtype bf16 --tune wgrad_batch_wgs=512
  one "bf16 single_stream" --dtype bf16 --single-stream
  one "bf16 B=4" --dtype bf16 --batch 4 --steps 200 --warmup 20
  one "bf16 B=64" --dtype bf16 --batch 64 --steps 100 --warmup 20
  one "f32 B=4" --batch 4 --steps 200 --warmup 20
done
python scripts/conv_ablate_bh.py > $OUT/conv_ablate_bh.log 2>&1
python scripts/event_timeline.py --dtype bf16s > $OUT/timeline_bf16.txt 2>&1
python scripts/event_timeline.py --dtype bf16s --batch 4 > $OUT/timeline_bf16_b4.txt 2>&1
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16.json 2>$OUT/bench_bf16.err
tail -3 $OUT/sweep.log
