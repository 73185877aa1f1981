#!/bin/bash
# round 3, call y: Solver's half-batch pipeline (two half-batch plans on two streams, the second a forward pass behind the first):
# parity + race check on the GPU, then A/B of the step with and without it, skew variants, the other precisions, a traced step
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3y; mkdir -p $O; export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_model.py -x -q -m gpu -k "half_batch_pipeline" ) > $O/tests_halves.log 2>&1; tail -4 $O/tests_halves.log
b() { timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 40 --warmup 10 "$@" 2>> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'], 4), 'ms', round(d['value']), d['unit'], d['config']['final_losses'])" | tee -a $O/ab.log; }
b --halves off
b --halves on
b --halves off
b --halves on
b --halves on --skew none
b --halves on --tune dec_split_min=256
b --halves on --tune dec_split_min=64
b --dtype f32x3 --halves off
b --dtype f32x3 --halves on
b --dtype bf16 --halves off
b --dtype bf16 --halves on
b --batch 128 --halves off
b --batch 128 --halves on
b --batch 512 --halves off
b --batch 512 --halves on
b --batch 64 --frames 1024 --halves off --steps 10 --warmup 3
b --batch 64 --frames 1024 --halves on --steps 10 --warmup 3
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o trace -- python $GRAFT_REPO_ROOT/bench.py --halves on --steps 4 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2>&1)
python scripts/trace_summary.py /tmp/rp/trace_kernel_trace.csv 25 > $O/trace_summary.txt
python scripts/trace_timeline.py /tmp/rp/trace_kernel_trace.csv > $O/trace_timeline.txt
head -12 $O/trace_timeline.txt
