#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3r
mkdir -p $O
for v in "" "--dtype bf16" "" "--dtype bf16" "--dtype f32x3" "--dtype bf16r"; do
  timeout 300 python bench.py $v --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kc=d.get('kernel_classes',{})
print('$v', round(d['ms_per_step'],3), {k:(round(v['ms_per_step'],3)) for k,v in kc.items() if k in ('conv_fwd','conv_dgrad','conv_wgrad')})" >> $O/ab.log
done
cat $O/ab.log
