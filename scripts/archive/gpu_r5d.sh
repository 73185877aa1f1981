# round 5, call D: same-box A/B of the round-4 tree (_w_r4/, its own library) against the working tree: op-level layers + train step, twice each
OUT=gpurun_out/${1:-r5d}; mkdir -p $OUT; export TMPDIR=/tmp
M='import sys; sys.argv=["x"]; import conv_micro as m
for rep in range(2):
    m.run(256,128,128,128,5,1,tiles=(0,),which="fd"); m.run(256,128,128,64,5,1,tiles=(0,),which="fd"); m.run(256,128,128,32,5,1,tiles=(0,),which="fd")
    m.run(256,128,128,128,5,2,tiles=(0,),which="fd"); m.run(256,80,128,128,8,1,tiles=(0,),which="f"); m.run(256,1104,128,128,1,1,tiles=(0,),which="fd")
    m.run(1024,128,128,128,5,1,tiles=(0,),which="fd"); m.run(64,128,128,1024,5,1,tiles=(0,),which="fd"); m.run(256,128,256,64,5,1,tiles=(0,),which="f")'
for rep in 1 2; do
  echo "## r4 tree" | tee -a $OUT/micro.log; (cd _w_r4/scripts && python -c "$M" 2>&1 | grep "B=" | tail -9) | tee -a $OUT/micro.log
  echo "## working tree" | tee -a $OUT/micro.log; (cd scripts && python -c "$M" 2>&1 | grep "B=" | tail -9) | tee -a $OUT/micro.log
done
for rep in 1 2 3; do
  for tree in _w_r4 .; do
    (cd $tree && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 2>/dev/null | tail -1 > /tmp/b.json)
    python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$tree', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/bench_ab.log
  done
done
(cd _w_r4 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config2 2>/dev/null | tail -1 > $GRAFT_REPO_ROOT/$OUT/bench_r4.json)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config2 2>/dev/null | tail -1 > $OUT/bench_new.json
python - $OUT/bench_r4.json $OUT/bench_new.json <<'PY' | tee -a $OUT/bench_ab.log
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read())
    kc = d.get("kernel_classes") or {}
    print(f, round(d["ms_per_step"], 3), {k: (round(v["ms_per_step"], 3), v.get("launches_per_step")) for k, v in kc.items()})
PY
