#!/usr/bin/env bash
# Like build_ab.sh for flags that only touch compositor.hip / tile_kernels.hpp: recompiles that one translation unit and links it with the production objects
# of video-stitcher_amd/build/ (run build.sh first):   bash tools/build_ab_fast.sh prio_w1 -DMS_PRIO_WARP=1
set -euo pipefail
NAME=$1; shift
cd "$(dirname "$0")/../video-stitcher_amd/csrc"
mkdir -p ../../ab /tmp/ab_fast
FLAGS="$(cat ../build/flags.txt) $*"
/opt/rocm/bin/hipcc $FLAGS -x hip -c compositor.hip -o /tmp/ab_fast/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/$NAME.so ../build/prims.o /tmp/ab_fast/$NAME.o ../build/mesh_solver.o ../build/matcher.o ../build/features.o ../build/calib.o ../build/api.o ../build/geometry.o ../build/dist.o -ldl -lrt
echo "built ab/$NAME.so"
