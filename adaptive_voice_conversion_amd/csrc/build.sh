#!/bin/bash
# Builds libavc_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/libavc_hip.so"
mkdir -p "$HERE/build"
OBJS=""
for f in "$HERE"/*.hip; do
  o="$HERE/build/$(basename "$f" .hip).o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/conv_shared.h" -nt "$o" ] || [ "$HERE/conv_x3_shared.h" -nt "$o" ] || [ "$HERE/bf16_pairs.h" -nt "$o" ] || [ "$HERE/avc_common.h" -nt "$o" ] || [ "$HERE/avc_internal.h" -nt "$o" ] || [ "$ROOT/include/avc_hip.h" -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $AVC_EXTRA_FLAGS -I"$HERE" -I"$ROOT/include" -c "$f" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $OUT"
