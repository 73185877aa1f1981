# round 5: a last look at the schedule knobs on the final build (the fused InstanceNorm epilogues shortened the chains): one box, alternating
OUT=gpurun_out/${1:-r5sweep}; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for t in "" "dec_split_min=512" "dec_split_min=64" "wgrad_batch=8" "wgrad_batch=16" "kg_wgs=512" "side_prio=1"; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 $args 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('${t:-default}'.ljust(22), 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/sweep.log
done
done
