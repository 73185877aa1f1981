OUT=gpurun_out/${1:-r5sweep2}; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for t in "" "side_prio=1" "side_prio=1 wgrad_batch=16" "side_prio=1 wgrad_batch=24" "wgrad_batch=24" "side_prio=1 wgrad_batch=16 dec_split_min=64"; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 $args 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('${t:-default}'.ljust(48), 'f32 ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/sweep.log
done
for t in "" "side_prio=1" "side_prio=1 wgrad_batch=16"; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  python bench.py --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-profile $args 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('${t:-default}'.ljust(48), 'bf16 ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/sweep.log
  python bench.py --batch 4 --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-config2 $args 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('${t:-default}'.ljust(48), 'B=4 ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/sweep.log
done
done
