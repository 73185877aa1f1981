#!/bin/bash
# round 6, call c: (1) steady-state per-chunk time of the bf16 weight gradient per producer ring depth; (2) cross-process reproducibility of bf16
OUT=gpurun_out/${1:-r6c}; mkdir -p $OUT; export TMPDIR=/tmp
S4=$PWD/adaptive_voice_conversion_amd/csrc/libavc_hip.so; S2=$PWD/_w_ab/libavc_s2.so; S3=$PWD/_w_ab/libavc_s3.so
for l in $S2 $S3 $S4; do AVC_HIP_LIB=$l python scripts/wgrad_bh_steady.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wgrad_steady.log; done
for i in 1 2 3; do python scripts/bf16_repro_probe.py bf16s 256 2>/dev/null > $OUT/repro_bf16s_$i.txt; done
for i in 1 2; do python scripts/bf16_repro_probe.py fp32 256 2>/dev/null > $OUT/repro_fp32_$i.txt; done
for i in 1 2; do python scripts/bf16_repro_probe.py bf16r 256 2>/dev/null > $OUT/repro_bf16r_$i.txt; done
for i in 1 2; do AVC_HIP_LIB=$S2 python scripts/bf16_repro_probe.py bf16s 256 2>/dev/null > $OUT/repro_bf16s_s2_$i.txt; done
echo "bf16s 1 vs 2: $(diff $OUT/repro_bf16s_1.txt $OUT/repro_bf16s_2.txt | grep -c '^<') differing lines"; diff $OUT/repro_bf16s_1.txt $OUT/repro_bf16s_2.txt | head -20
echo "bf16s 1 vs 3: $(diff $OUT/repro_bf16s_1.txt $OUT/repro_bf16s_3.txt | grep -c '^<')"
echo "fp32: $(diff $OUT/repro_fp32_1.txt $OUT/repro_fp32_2.txt | grep -c '^<')"; echo "bf16r: $(diff $OUT/repro_bf16r_1.txt $OUT/repro_bf16r_2.txt | grep -c '^<')"
echo "bf16s stages=2 build: $(diff $OUT/repro_bf16s_s2_1.txt $OUT/repro_bf16s_s2_2.txt | grep -c '^<')"; echo "s2 vs s4: $(diff $OUT/repro_bf16s_s2_1.txt $OUT/repro_bf16s_1.txt | grep -c '^<')"
grep -c . $OUT/repro_bf16s_1.txt
python - <<'PY'
import sys
a=open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r6c/repro_bf16s_1.txt').read().splitlines()
r0=[l.split(' ',1)[1] for l in a if l.startswith('rep0')]; r1=[l.split(' ',1)[1] for l in a if l.startswith('rep1')]
print('within process, rep0 vs rep1 differing:', sum(x!=y for x,y in zip(r0,r1)))
PY
