#!/bin/bash
# round 3, call w: the latency-oriented dense stack (16x16x4 tiles, double-buffered weight images, loads in front of / stores behind the
# products) -- parity on the GPU, then the step with and without it is NOT comparable in one build, so: bench lines (fp32, bf16 storage, B = 4),
# side-stream priority A/B, and rocprofv3 kernel stats (multi-stream) for the dense kernels' own durations
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3w; mkdir -p $O; export TMPDIR=/tmp
( time timeout 420 python -m pytest tests/test_engine.py -x -q -m gpu -k "forward_loss_backward_vs_oracle or matches_reference_goldens or two_half_batch" ) > $O/tests_engine.log 2>&1; tail -3 $O/tests_engine.log
( time timeout 420 python -m pytest tests/test_graded_configs.py -x -q -m gpu -k "train_step_matches_oracle_at_graded_shape or bf16_storage_mode_at_graded_shape" ) > $O/tests_graded.log 2>&1; tail -3 $O/tests_graded.log
b() { timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 40 --warmup 10 "$@" 2>> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'], 4), 'ms', round(d['value']), d['unit'])"; }
b --dtype f32
b --dtype f32 --tune side_prio=1
b --dtype f32
b --dtype f32 --tune side_prio=1
b --dtype bf16
b --dtype bf16 --tune side_prio=1
b --dtype bf16
b --dtype f32 --batch 4
b --dtype f32 --batch 4 --tune side_prio=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $O/prof.log 2>&1)
f=$(ls /tmp/prof_w/*/*kernel_stats.csv /tmp/prof_w/*kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/kernel_stats.csv; grep -i "dense\|timepool" $O/kernel_stats.csv | cut -c1-160
