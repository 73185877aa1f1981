#!/bin/bash
# round 3, call ac: the graded-shape file on the last build, heaviest cases last, under a hard cap (the driver runs the whole file at round end)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3ac; mkdir -p $O
( time timeout 330 python -m pytest tests/test_graded_configs.py -x -q -m gpu -s --durations=20 -k "inference_batch or relu_branches or bf16_storage or train_step" ) > $O/tests_graded.log 2>&1; tail -28 $O/tests_graded.log | cut -c1-180
grep -o "\[gpu[^]]*\][^[]*" $O/tests_graded.log > $O/gpu_parity_report_graded.txt
