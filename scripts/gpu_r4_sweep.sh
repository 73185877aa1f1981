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
