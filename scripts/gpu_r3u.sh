#!/bin/bash
# round 3, call u: the bench line with the back-to-back IN sequence (fp32 and bf16 storage); DSP tests on the new ragged entry point
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3u; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
python -c "
import json; d=json.loads(open('$O/bench_f32.json').read().strip().splitlines()[-1]); print(d['ms_per_step']); print(json.dumps(d['roofline_instnorm'], indent=1))"
timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
python -c "
import json; d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); print(d['ms_per_step']); print(json.dumps(d['roofline_instnorm'], indent=1)); print(json.dumps(d['roofline'], indent=1))"
timeout 900 python -m pytest tests/test_dsp.py tests/test_abi.py -q -m gpu 2>&1 | tail -3
