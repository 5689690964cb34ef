#!/usr/bin/env bash
# Build an A/B variant of libmsstitch.so with extra compiler flags into ab/<name>.so (git-ignored, shipped to the GPU box):
#   bash tools/build_ab.sh nocanvas -DMS_PROBE_CANVAS=2     then  gpurun -- 'bash tools/ab_warp.sh'
set -euo pipefail
NAME=$1; shift
cd "$(dirname "$0")/../video-stitcher_amd/csrc"
B=/tmp/ab_build_$NAME; mkdir -p $B ../../ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wno-unused-function $*"
pids=()
for f in prims.hip compositor.hip mesh_solver.hip matcher.hip features.hip calib.hip api.cpp geometry.cpp dist.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c "$f" -o $B/${f%.*}.o & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/$NAME.so $B/*.o -ldl -lrt
echo "built ab/$NAME.so"
