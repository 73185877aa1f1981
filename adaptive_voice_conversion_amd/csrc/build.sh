#!/bin/bash
# Builds libavc_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${AVC_OUT:-$HERE/libavc_hip.so}"      # (AVC_OUT / AVC_BUILD_DIR + AVC_EXTRA_FLAGS: a second build of the same sources for A/B runs)
BUILD="${AVC_BUILD_DIR:-$HERE/build}"
mkdir -p "$BUILD"
# NO packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in device code.  Round 6: beside workgroups that keep the
# bf16 matrix pipes busy (the BF = 2 conv kernels of another stream), row kernels whose reductions the compiler had packed returned
# run-to-run DIFFERENT sums in the upper lane of the pack -- constant inputs, no LDS, no atomics (scripts/pairs_race_probe.py: 46-103 of
# 300 launches; profiles/r06_pairs_race_*.txt) -- and the bf16 storage engine's backward pass was not bit-reproducible in multi-stream mode
# (scripts/bf16_repro_probe2.py).  Without the packed forms: 0 of 300, every configuration reproducible, step time unchanged
# (profiles/r06_nopk_ab.log).  The feature switch is a device-target feature; the host pass of hipcc does not know it and says so (filtered).
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
OBJS=""
for f in "$HERE"/*.hip; do
  o="$BUILD/$(basename "$f" .hip).o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/conv_shared.h" -nt "$o" ] || [ "$HERE/conv_x3_shared.h" -nt "$o" ] || [ "$HERE/bf16_pairs.h" -nt "$o" ] || [ "$HERE/avc_common.h" -nt "$o" ] || [ "$HERE/avc_internal.h" -nt "$o" ] || [ "$ROOT/include/avc_hip.h" -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $NOPK $AVC_EXTRA_FLAGS -I"$HERE" -I"$ROOT/include" -c "$f" -o "$o" 2> >(grep -v "is not a recognized feature for this target" >&2) &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $OUT"
