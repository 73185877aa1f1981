# round 5, call E: same-box A/B of the round-4 tree vs the working tree (WALK as a compile-time flag: the default path must be the r4 kernel),
# then the GPU suite with durations
OUT=gpurun_out/${1:-r5e}; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do
  for tree in _w_r4 .; do
    (cd $tree && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 2>/dev/null | tail -1 > /tmp/b.json)
    python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$tree', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/bench_ab.log
  done
done
bash scripts/gpu_suite.sh $1 slow
