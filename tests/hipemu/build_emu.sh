#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the product .hip sources with g++ against the fiber emulator
# (tests/hipemu/hip/hip_runtime.h) into tests/hipemu/libg2pc_emu.so.  Never loaded by the product.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
SRC="$ROOT/3dgs-to-pc_amd/g2pc/csrc"
# EMU_TAG / EMU_DEFS: a second build with other build-time switches (e.g. EMU_TAG=_r8 EMU_DEFS=-DG2PC_BK_EMIT_RMAX=8), its own objects
TAG="${EMU_TAG:-}"
OUT="$HERE/libg2pc_emu$TAG.so"
OBJS=""
for f in prims geom alloc sampler raster raster_cu clean project; do
  [ -f "$SRC/$f.hip" ] || continue
  O="$HERE/$f$TAG.emu.o"
  if [ ! -f "$O" ] || [ "$SRC/$f.hip" -nt "$O" ] || [ "$SRC/g2pc_internal.h" -nt "$O" ] \
     || [ "$SRC/g2pc_device.inl" -nt "$O" ] || [ "$SRC/py_project.inl" -nt "$O" ] || [ "$SRC/raster_common.h" -nt "$O" ] || [ -n "$(find "$SRC/experiments" -newer "$O" 2>/dev/null)" ] || [ "$HERE/hip/hip_runtime.h" -nt "$O" ] \
     || [ "$ROOT/include/g2pc.h" -nt "$O" ]; then
    g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -x c++ -I"$HERE" -I"$ROOT/include" -Wno-attributes -Wno-unknown-pragmas \
        ${EMU_DEFS:-} -c "$SRC/$f.hip" -o "$O"
  fi
  OBJS="$OBJS $O"
done
# relink only when an object is newer than the library, and atomically: several ranks of a torch.distributed test may call this
# at once, and a loader must never meet a half-written file
NEED=0
[ -f "$OUT" ] || NEED=1
for O in $OBJS; do [ "$O" -nt "$OUT" ] && NEED=1; done
if [ "$NEED" = 1 ]; then
  TMP="$OUT.$$.tmp"
  g++ -shared -fPIC $OBJS -o "$TMP"
  mv -f "$TMP" "$OUT"
fi
echo "$OUT"
