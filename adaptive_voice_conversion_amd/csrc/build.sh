#!/bin/bash
# Builds libavc_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${AVC_OUT:-$HERE/libavc_hip.so}"      # (AVC_OUT / AVC_BUILD_DIR + AVC_EXTRA_FLAGS: a second build of the same sources for A/B runs)
BUILD="${AVC_BUILD_DIR:-$HERE/build}"
mkdir -p "$BUILD"
OBJS=""
for f in "$HERE"/*.hip; do
  o="$BUILD/$(basename "$f" .hip).o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/conv_shared.h" -nt "$o" ] || [ "$HERE/conv_x3_shared.h" -nt "$o" ] || [ "$HERE/bf16_pairs.h" -nt "$o" ] || [ "$HERE/avc_common.h" -nt "$o" ] || [ "$HERE/avc_internal.h" -nt "$o" ] || [ "$ROOT/include/avc_hip.h" -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $AVC_EXTRA_FLAGS -I"$HERE" -I"$ROOT/include" -c "$f" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $OUT"
