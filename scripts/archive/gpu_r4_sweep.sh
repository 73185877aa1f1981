# command that produced profiles/r04_tune_sweep.log, r04_infer_b1_t400.json and the last r04_bench.json
# round 4: tuning sweep on the final build, every setting beside the default on ONE box (boxes differ by ~4 %), two rounds
OUT=gpurun_out/${1:-r4sweep}; mkdir -p $OUT
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config2 --no-profile $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s step %.4f ms' % ('$1', d['ms_per_step']))"; }
for i in 1 2; do
run "default" ""
run "wgrad_batch_wgs=192" "--tune wgrad_batch_wgs=192"
run "wgrad_batch_wgs=224" "--tune wgrad_batch_wgs=224"
run "wgrad_batch=8" "--tune wgrad_batch=8"
run "wgrad_batch=16" "--tune wgrad_batch=16"
run "default" ""
run "side_prio=0" "--tune side_prio=0"
run "side_prio=1" "--tune side_prio=1"
run "dec_split_min=512" "--tune dec_split_min=512"
run "tile12_wgs=512" "--tune tile12_wgs=512"
run "kg_wgs=512" "--tune kg_wgs=512"
run "kg_wgs=128" "--tune kg_wgs=128"
run "wgrad_batch=6" "--tune wgrad_batch=6"
done 2>&1 | tee $OUT/sweep.log
python bench.py --mode infer --batch 1 --frames 400 --steps 50 --warmup 10 > $OUT/infer_b1_t400.json 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/infer_b1_t400.json').read().strip().splitlines()[-1]); print('infer B=1 T=400', round(d['ms_per_step'],4), 'ms')"
python bench.py --mode infer --batch 1 --frames 400 --dtype bf16 --steps 50 --warmup 10 > $OUT/infer_b1_t400_bf16.json 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/infer_b1_t400_bf16.json').read().strip().splitlines()[-1]); print('infer B=1 T=400 bf16', round(d['ms_per_step'],4), 'ms')"
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_instnorm']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['spread'])"
