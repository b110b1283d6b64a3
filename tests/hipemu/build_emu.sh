#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the product .hip sources with g++ against the fiber emulator
# (tests/hipemu/hip/hip_runtime.h) into tests/hipemu/libg2pc_emu.so.  Never loaded by the product.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
SRC="$ROOT/3dgs-to-pc_amd/g2pc/csrc"
OUT="$HERE/libg2pc_emu.so"
OBJS=""
for f in prims geom alloc sampler raster clean project; do
  [ -f "$SRC/$f.hip" ] || continue
  if [ ! -f "$HERE/$f.emu.o" ] || [ "$SRC/$f.hip" -nt "$HERE/$f.emu.o" ] || [ "$SRC/g2pc_internal.h" -nt "$HERE/$f.emu.o" ] \
     || [ "$SRC/g2pc_device.inl" -nt "$HERE/$f.emu.o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$HERE/$f.emu.o" ] \
     || [ "$ROOT/include/g2pc.h" -nt "$HERE/$f.emu.o" ]; then
    g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -x c++ -I"$HERE" -I"$ROOT/include" -Wno-attributes -Wno-unknown-pragmas \
        -c "$SRC/$f.hip" -o "$HERE/$f.emu.o"
  fi
  OBJS="$OBJS $HERE/$f.emu.o"
done
g++ -shared -fPIC $OBJS -o "$OUT"
echo "$OUT"
