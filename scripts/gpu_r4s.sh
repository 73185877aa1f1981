# round 4: eight-consumer-wave 128x64 weight-gradient tile (avc_tuning.wgrad_cw8) A/B on one box: parity, lone layer, step
OUT=gpurun_out/${1:-r4s}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_conv.py tests/test_engine.py -x -q -m gpu -k "wgrad or golden or gradient or determin" 2>&1 | tail -2
python - <<'PY'
import sys; sys.path.insert(0,'scripts')
import conv_micro as m
for cw in (0, 1):
    m.lib.avc_set_tuning(b"wgrad_cw8", cw)
    print("wgrad_cw8 =", cw)
    for T in (128, 64, 32):
        m.run(256, 128, 128, T, 5, 1, tiles=(), which="w")
    m.run(256, 128, 256, 64, 5, 1, tiles=(), which="w")
PY
for c in 0 1 0 1; do
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config2 --tune wgrad_cw8=$c | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes']; print('cw8 $c: step', round(d['ms_per_step'],4), 'wgrad', k['conv_wgrad']['ms_per_step'], k['conv_wgrad']['tflops'], 'reduce', k['slab_reduce']['ms_per_step'])"
done
