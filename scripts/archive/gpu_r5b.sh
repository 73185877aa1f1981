# round 5, call B: tile walk + scheduling knobs -- op-level matrix, then the train step under the promising ones
OUT=gpurun_out/${1:-r5b}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_walk.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/walk_tests.txt
timeout 900 python scripts/walk_micro.py > $OUT/walk_micro.log 2>&1; tail -5 $OUT/walk_micro.log
for t in "" "conv_walk=4" "conv_walk=2" "conv_sched=2" "conv_sched=1" "conv_sched=6" "conv_walk=4 conv_sched=2" "conv_walk=2 conv_sched=2048"; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-config2 $args > $OUT/bench_tmp.json 2>> $OUT/bench.err
  python - "$t" $OUT/bench_tmp.json <<'PY' | tee -a $OUT/bench_ab.log
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print(f"{sys.argv[1] or 'default':32s} ms/step {d['ms_per_step']:.3f}  loss {d.get('final_losses')}")
PY
done
