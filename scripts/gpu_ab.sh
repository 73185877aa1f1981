export TMPDIR=/tmp; mkdir -p gpurun_out/r1v
echo "== config4 inference B=1024"; timeout 300 python bench.py --mode infer --batch 1024 --steps 10 --warmup 3 2>&1 | tail -1 | tee gpurun_out/r1v/infer_b1024.json | cut -c1-260
echo "== config5 T=1024 B=64 train"; timeout 300 python bench.py --frames 1024 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | tee gpurun_out/r1v/train_t1024_b64.json | cut -c1-260
echo "== M=512 B=128"; timeout 300 python bench.py --mels 512 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | tee gpurun_out/r1v/train_m512_b128.json | cut -c1-260
echo "== B=256 headline"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-json gpurun_out/r1v/prof.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); [print(k,v) for k,v in d['kernel_classes'].items() if 'conv' in k or 'slab' in k]"
