# round 4: InstanceNorm rows per lane group (avc_tuning.in_rows_per_group) -- op-level bandwidth at the step's shapes + the step
OUT=gpurun_out/${1:-r4q}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ops_rowops.py -x -q -m gpu 2>&1 | tail -2
python - > $OUT/in_rpl.log 2>&1 <<'PY'
import ctypes, torch, sys
sys.path.insert(0, '.')
import bench
from adaptive_voice_conversion_amd import _lib
lib = _lib.load()
for (B, C, T) in ((256, 128, 128), (256, 128, 64), (256, 128, 32), (256, 128, 16), (64, 128, 1024)):
    for rpl in (1, 2, 4):
        lib.avc_set_tuning(b"in_rows_per_group", rpl)
        r = bench.instnorm_dominant_shape(B, C, T)
        print(f"[{B},{C},{T}] rows/group {rpl}: fwd {r['fwd']['avg_launch_us']:.2f} us {r['fwd']['gbs']:.0f} GB/s | bwd {r['bwd']['avg_launch_us']:.2f} us {r['bwd']['gbs']:.0f} GB/s", flush=True)
    for rpl in (1, 2, 4):
        lib.avc_set_tuning(b"in_rows_per_group", rpl)
        r = bench.instnorm_dominant_shape(B, C, T, pairs=True)
        print(f"[{B},{C},{T}] PAIRS rows/group {rpl}: fwd {r['fwd']['avg_launch_us']:.2f} us {r['fwd']['gbs']:.0f} GB/s | bwd {r['bwd']['avg_launch_us']:.2f} us {r['bwd']['gbs']:.0f} GB/s", flush=True)
lib.avc_set_tuning(b"in_rows_per_group", 0)
PY
cat $OUT/in_rpl.log
for r in 1 2 0; do python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 --tune in_rows_per_group=$r | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step rows/group $r', round(d['ms_per_step'],4))"; done
