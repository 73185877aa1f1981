#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-suite}; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu -s > $O/tests.log 2>&1 ) 2> $O/tests.time; tail -4 $O/tests.log; cat $O/tests.time
grep -o "\[gpu[^]]*\][^[]*" $O/tests.log | grep -v "x3 dgrad\|x3 fwd" > $O/gpu_parity_report.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
