#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3p
mkdir -p $O
for v in "--dtype bf16s" "--dtype bf16s --tune dec_split_min=1000000" "--dtype bf16s" "" "--dtype f32x3" "--dtype bf16" "--dtype bf16s --batch 64 --frames 1024" "--batch 64 --frames 1024" "--dtype bf16s --batch 4" "--batch 4"; do
  timeout 300 python bench.py $v --steps 30 --warmup 10 --no-profile --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3))" >> $O/ab.log
done
timeout 600 python -m pytest tests/test_graded_configs.py -x -q -m gpu -k "bf16_storage or train_step_matches" -s 2>&1 | grep -E "^\[|passed|failed" > $O/graded.log
cat $O/ab.log $O/graded.log
