#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3m
mkdir -p $O
S3=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip_s3.so
AVC_HIP_LIB=$S3 timeout 600 python -m pytest tests/test_bf16_pairs.py -x -q -m gpu -k "conv_fwd or conv_dgrad" 2>&1 | tail -3 > $O/s3_tests.log
for i in 1 2; do
  timeout 300 python bench.py --dtype bf16s --steps 30 --warmup 10 --no-profile --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-stage', d['ms_per_step'])" >> $O/ab.log
  AVC_HIP_LIB=$S3 timeout 300 python bench.py --dtype bf16s --steps 30 --warmup 10 --no-profile --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3-stage', d['ms_per_step'])" >> $O/ab.log
done
timeout 300 python bench.py --dtype bf16s --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_bf16s.json 2> $O/bench_bf16s.err
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
cat $O/s3_tests.log $O/ab.log
python - <<'PY'
import json
for f in ("bench_bf16s", "bench_f32"):
    d = json.loads(open(f"gpurun_out/r3m/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], json.dumps(d.get("roofline_instnorm"))[:1500])
PY
