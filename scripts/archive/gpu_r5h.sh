# round 5, call H: occupancy experiment -- conv_gemm built with amdgpu_waves_per_eu(5,5) / (6,6) (alternative libraries, AVC_HIP_LIB) against the default build
OUT=gpurun_out/${1:-r5h}; mkdir -p $OUT; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/adaptive_voice_conversion_amd/csrc
for rep in 1 2; do
for v in "" _w5 _w6; do
  export AVC_HIP_LIB=$L/libavc_hip$v.so
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-profile --no-config2 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('lib$v train f32 B=256', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/ab.log
  python bench.py --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('lib$v train bf16 B=256', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/ab.log
  python bench.py --batch 4 --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-config2 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('lib$v train f32 B=4', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/ab.log
  python bench.py --mode infer --batch 1024 --steps 20 --warmup 5 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('lib$v infer B=1024', 'ms/step', round(d['ms_per_step'],3))" | tee -a $OUT/ab.log
done
done
