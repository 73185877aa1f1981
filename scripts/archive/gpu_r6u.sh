OUT=gpurun_out/r6v; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; PK=$PWD/_w_ab/libavc_r5.so
one() { local label="$1"; local lib="$2"; shift; shift
  AVC_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-profile --no-config2 "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$label'.ljust(64), round(d['ms_per_step'],3))" | tee -a $OUT/ab.log; }
for rep in 1 2 3; do
  one "infer B=1024 f32, round-6 library" $NEW --mode infer --batch 1024 --steps 10 --warmup 3
  one "infer B=1024 f32, round-5 library" $PK --mode infer --batch 1024 --steps 10 --warmup 3
  one "T=1024 B=64 f32, round-6 library" $NEW --batch 64 --frames 1024 --steps 10 --warmup 3
  one "T=1024 B=64 f32, round-5 library" $PK --batch 64 --frames 1024 --steps 10 --warmup 3
  one "B=256 f32, round-6 library" $NEW --steps 30 --warmup 8
  one "B=256 f32, round-5 library" $PK --steps 30 --warmup 8
done
