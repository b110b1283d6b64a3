#!/bin/bash
# A/B aid: build another variant of libg2pc.so from the CURRENT sources with extra -D switches (the build-time A/B switches of
# raster.hip / prims.hip), in-tree so that it travels with the gpurun snapshot:
#   tools/experiments/build_variant.sh <tag> -DG2PC_FUSED_EMIT=0 -DG2PC_PREPROCESS_MULTI=0   ->  3dgs-to-pc_amd/g2pc/libg2pc_<tag>.so
# then alternate `python tools/experiments/ab_lib.py 3dgs-to-pc_amd/g2pc/libg2pc_<tag>.so <bench args>` with `python bench.py <bench args>`.
set -e
cd "$(dirname "$0")/../../3dgs-to-pc_amd/g2pc/csrc"
tag=$1; shift
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -I../../../include"
mkdir -p ../../../_ab_old/$tag
objs=""
for f in prims geom alloc sampler raster raster_cu clean project; do
  $HIPCC $FLAGS "$@" -c $f.hip -o ../../../_ab_old/$tag/$f.o &
  objs="$objs ../../../_ab_old/$tag/$f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o ../libg2pc_$tag.so
echo "$(cd .. && pwd)/libg2pc_$tag.so"
