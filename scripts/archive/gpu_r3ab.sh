#!/bin/bash
# round 3, call ab: InstanceNorm forward with every load in front of the reductions, float4 clip+Adam, 16 slab loads in flight (on top of
# the phased conv epilogue of call aa): A/B, the GPU suite minus the graded file, then the artefacts of this build
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3ab; mkdir -p $O; export TMPDIR=/tmp
b() { timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 40 --warmup 10 "$@" 2>> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'], 4), 'ms', round(d['value']), d['unit'], d['config']['final_losses'])" | tee -a $O/ab.log; }
b --dtype f32
b --dtype bf16
b --dtype f32 --batch 4
b --dtype f32x3
b --dtype f32
( time timeout 400 python -m pytest tests -x -q -m gpu --ignore=tests/test_graded_configs.py ) > $O/tests_rest.log 2>&1; tail -4 $O/tests_rest.log
grep -o "\[gpu[^]]*\][^[]*" $O/tests_rest.log | grep -v "x3 dgrad\|x3 fwd" > $O/gpu_parity_report_rest.txt
bash scripts/gpu_r3_final2.sh r3ab
