#!/bin/bash
# TEST INFRASTRUCTURE (CPU): the product's .hip sources compiled with g++ against the fibre emulator AND AddressSanitizer, then the
# emulator tests (or the command given) run against that library: every kernel's loads and stores are checked against the
# bounds of the torch (malloc) buffers they were handed.  Round 3: the whole emulator suite and the fuzzers are clean; the
# harness reports the `cell_start` word the outlier-removal driver used to be short of (k_tile_ranges, WRITE of size 4).
# (Limitation of a PRELOADED libasan: a C++ exception thrown inside torch -- e.g. the shape error the reference's own
#  cull_large_gaussians raises -- aborts in ASan's __cxa_throw interceptor; that is not a finding about the kernels.)
#   tools/emu_asan.sh                       -> pytest tests/test_emu_*.py
#   tools/emu_asan.sh python tools/experiments/quadtree_fuzz.py 1 40
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${G2PC_ASAN_DIR:-/tmp/g2pc_asan}
mkdir -p "$OUT"
SRC="$ROOT/3dgs-to-pc_amd/g2pc/csrc"
for f in prims geom alloc sampler raster raster_cu clean project; do
  if [ ! -f "$OUT/$f.o" ] || [ "$SRC/$f.hip" -nt "$OUT/$f.o" ] || [ "$ROOT/tests/hipemu/hip/hip_runtime.h" -nt "$OUT/$f.o" ] || [ "$ROOT/include/g2pc.h" -nt "$OUT/$f.o" ]; then
    # (asan-stack=0: the kernels run on the emulator's own fibre stacks, which ASan's stack instrumentation does not know)
    g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address --param asan-stack=0 -fno-omit-frame-pointer -x c++ \
        -I"$ROOT/tests/hipemu" -I"$ROOT/include" -Wno-attributes -Wno-unknown-pragmas -c "$SRC/$f.hip" -o "$OUT/$f.o" &
  fi
done
wait
g++ -shared -fPIC -fsanitize=address "$OUT"/*.o -o "$OUT/libg2pc_emu_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0:log_path=$OUT/report
export G2PC_EMU_LIB="$OUT/libg2pc_emu_asan.so"
rm -f "$OUT"/report.*
cd "$ROOT"
if [ $# -gt 0 ]; then "$@"; else python -m pytest tests/test_emu_*.py -x -q; fi
rc=$?
ls "$OUT"/report.* >/dev/null 2>&1 && { echo "AddressSanitizer reports:"; head -20 "$OUT"/report.*; exit 1; }
exit $rc
