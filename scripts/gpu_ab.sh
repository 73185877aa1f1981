export TMPDIR=/tmp; mkdir -p gpurun_out/r1u
timeout 600 python -m pytest tests/test_ops_conv.py tests/test_engine.py -m gpu -q --timeout 300 2>&1 | tail -2
python -c "
import sys; sys.argv=['x']; sys.path.insert(0,'scripts'); import conv_micro as m
m.run(256,128,128,128,5,1,tiles=(),which='w'); m.run(256,128,128,64,5,1,tiles=(),which='w'); m.run(256,128,128,32,5,1,tiles=(),which='w'); m.run(256,128,128,16,5,1,tiles=(),which='w'); m.run(256,1104,128,128,1,1,tiles=(),which='w'); m.run(256,80,128,128,8,1,tiles=(),which='w')" 2>&1 | grep -v amdgpu
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-json gpurun_out/r1u/prof.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); [print(k,v) for k,v in d['kernel_classes'].items() if 'conv' in k or 'slab' in k]"
