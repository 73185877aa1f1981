#!/bin/bash
# Builds the CPU lane-level simulation of the kernels (tests only; see tests/emu/hip_emu.h).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/tests/emu/libavc_emu.so"
OBJ="$ROOT/tests/emu/build"
mkdir -p "$OBJ"
CXX=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-O2 -g -std=c++17 -fPIC -Wno-unused-value -I$ROOT/tests/emu -I$HERE -I$ROOT/include"
OBJS=""
for f in "$HERE"/*.hip "$ROOT/tests/emu/hip_emu.cpp"; do
  o="$OBJ/$(basename "$f").o"
  $CXX $FLAGS -x c++ -c "$f" -o "$o" &
  OBJS="$OBJS $o"
done
wait
$CXX -shared -fPIC $OBJS -o "$OUT.tmp.$$"
mv -f "$OUT.tmp.$$" "$OUT"   # atomic: concurrent test processes never map a half-written library
echo "built $OUT"
