#!/bin/bash
# Builds the CPU lane-level simulation of the kernels (tests only; see tests/emu/hip_emu.h).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/tests/emu/libavc_emu.so"
SRCS=$(ls "$HERE"/*.hip)
/opt/rocm/lib/llvm/bin/clang++ -O2 -g -std=c++17 -fPIC -shared -x c++ -Wno-unused-value \
  -I"$ROOT/tests/emu" -I"$HERE" -I"$ROOT/include" $SRCS "$ROOT/tests/emu/hip_emu.cpp" -o "$OUT"
echo "built $OUT"
