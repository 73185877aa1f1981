mkdir -p gpurun_out/r1h; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_conv.py -m gpu -q --timeout 300 2>&1 | tail -2
for w in 512 256 1024; do
  echo "== AVC_WGRAD_WGS=$w"
  AVC_WGRAD_WGS=$w timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== single stream"
AVC_SINGLE_STREAM=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-json gpurun_out/r1h/prof_single.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); [print(k,v) for k,v in d['kernel_classes'].items()]"
