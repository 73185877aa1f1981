mkdir -p gpurun_out/r1l; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_conv.py -m gpu -q --timeout 300 2>&1 | tail -1
timeout 600 python scripts/conv_micro.py 2>&1 | grep -v amdgpu | tee gpurun_out/r1l/conv_micro.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-json gpurun_out/r1l/prof.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); [print(k,v) for k,v in d['kernel_classes'].items()]"
