#!/bin/bash
# round 3, call aa: conv epilogue in load / compute / store phases (7 instead of 114 vector-memory drains per fragment):
# A/B against the previous build's numbers (same bench flags as call y: final losses must be IDENTICAL digit for digit), then the GPU suite minus the graded file
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3aa; mkdir -p $O; export TMPDIR=/tmp
b() { timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 40 --warmup 10 "$@" 2>> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'], 4), 'ms', round(d['value']), d['unit'], d['config']['final_losses'])" | tee -a $O/ab.log; }
b --dtype f32
b --dtype bf16
b --dtype f32 --batch 4
b --dtype bf16 --batch 4
b --dtype f32x3
b --dtype f32
b --batch 64 --frames 1024 --steps 10 --warmup 3
b --batch 128
b --batch 16
timeout 100 python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer1024', round(d['ms_per_step'], 4), round(d['value']))" | tee -a $O/ab.log
timeout 100 python bench.py --mode ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ragged', round(d['ms_per_step'], 4), round(d['value']))" | tee -a $O/ab.log
( time timeout 400 python -m pytest tests -x -q -m gpu --ignore=tests/test_graded_configs.py ) > $O/tests_rest.log 2>&1; tail -4 $O/tests_rest.log
