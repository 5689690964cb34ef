#!/usr/bin/env bash
# Builds libmsstitch.so (hand-written HIP, gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wall -Wno-unused-function ${MS_EXTRA_FLAGS:-}"
mkdir -p ../build
# a change of flags (e.g. a -DMS_DEV_KNOBS developer build before, the production flags now) rebuilds every object
if [ "$(cat ../build/flags.txt 2>/dev/null)" != "$FLAGS" ]; then rm -f ../build/*.o; echo "$FLAGS" > ../build/flags.txt; fi
newest_hdr=$(ls -t *.hpp *.inc ../../include/ms_stitch.h ../../include/ms_dist.h | head -1)
pids=()
for f in prims.hip compositor.hip mesh_solver.hip matcher.hip features.hip calib.hip api.cpp geometry.cpp dist.cpp; do
  o=../build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
    $HIPCC $FLAGS -x hip -c "$f" -o "$o" & pids+=($!)
  fi
done
# the hand-declared RCCL prototypes of dist.cpp against the installed header (syntax only; skipped where the header is absent)
if [ -f /opt/rocm/include/rccl/rccl.h ] && { [ ! -f ../build/rccl_abi.ok ] || [ rccl_abi_check.cpp -nt ../build/rccl_abi.ok ] || [ dist.cpp -nt ../build/rccl_abi.ok ]; }; then
  $HIPCC -std=c++17 -fsyntax-only -x hip --offload-host-only rccl_abi_check.cpp || { echo "RCCL ABI check failed: the prototypes dist.cpp declares do not match the installed rccl.h" >&2; exit 1; }
  touch ../build/rccl_abi.ok
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || { echo "compile failed" >&2; exit 1; }; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libmsstitch.so ../build/prims.o ../build/compositor.o ../build/mesh_solver.o ../build/matcher.o ../build/features.o ../build/calib.o ../build/api.o ../build/geometry.o ../build/dist.o -ldl -lrt
# C++ host pipeline over the C-ABI (thread / queue graph of the reference's timed.cpp); host code only, links the library above
if [ ! -f ../stitch_app ] || [ ../host/stitch_app.cpp -nt ../stitch_app ] || [ ../shim/ms_shim.hpp -nt ../stitch_app ] || [ ../../include/ms_stitch.h -nt ../stitch_app ]; then
  $HIPCC -O2 -std=c++17 -Wall -Wno-unused-result -pthread ../host/stitch_app.cpp -I../../include -L.. -lmsstitch -Wl,-rpath,'$ORIGIN' -o ../stitch_app
fi
# the multi-GPU host pipeline (one thread per GPU over ms_dist: RCCL / host transport)
if [ ! -f ../stitch_dist ] || [ ../host/stitch_dist.cpp -nt ../stitch_dist ] || [ ../../include/ms_dist.h -nt ../stitch_dist ] || [ ../../include/ms_stitch.h -nt ../stitch_dist ]; then
  $HIPCC -O2 -std=c++17 -Wall -Wno-unused-result -pthread ../host/stitch_dist.cpp -I../../include -L.. -lmsstitch -Wl,-rpath,'$ORIGIN' -o ../stitch_dist
fi
echo "built $(cd .. && pwd)/libmsstitch.so"
