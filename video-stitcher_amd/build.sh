#!/usr/bin/env bash
# Builds libmsstitch.so (HIP, gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wall -Wno-unused-function"
mkdir -p ../build
pids=()
for f in prims.hip compositor.hip; do
  if [ ! -f ../build/${f%.hip}.o ] || [ $f -nt ../build/${f%.hip}.o ] || [ common.hpp -nt ../build/${f%.hip}.o ] || [ launchers.hpp -nt ../build/${f%.hip}.o ] || [ ../../include/ms_stitch.h -nt ../build/${f%.hip}.o ]; then
    $HIPCC $FLAGS -c $f -o ../build/${f%.hip}.o & pids+=($!)
  fi
done
for f in api.cpp geometry.cpp; do
  if [ ! -f ../build/${f%.cpp}.o ] || [ $f -nt ../build/${f%.cpp}.o ] || [ common.hpp -nt ../build/${f%.cpp}.o ] || [ launchers.hpp -nt ../build/${f%.cpp}.o ] || [ ../../include/ms_stitch.h -nt ../build/${f%.cpp}.o ]; then
    $HIPCC $FLAGS -x hip -c $f -o ../build/${f%.cpp}.o & pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || { echo "compile failed" >&2; exit 1; }; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libmsstitch.so ../build/prims.o ../build/compositor.o ../build/api.o ../build/geometry.o
echo "built $(cd .. && pwd)/libmsstitch.so"
