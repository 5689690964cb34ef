#!/usr/bin/env bash
# Build the AddressSanitizer + UBSan variants HERE (no GPU needed): ab/asanlib/libmsstitch.so (host code instrumented), ab/stitch_app_asan, ab/stitch_dist_asan,
# ab/libfake_rccl_asan.so (the loopback RCCL of the tests).  Then on the GPU box:  gpurun -- 'bash tools/asan_run.sh'
set -euo pipefail
cd "$(dirname "$0")/.."
bash tools/build_ab.sh asan -fsanitize=address,undefined -fno-omit-frame-pointer -g -Wno-option-ignored
mkdir -p ab/asanlib && cp ab/asan.so ab/asanlib/libmsstitch.so
for a in stitch_app stitch_dist; do
  /opt/rocm/bin/hipcc --offload-host-only -O1 -g -std=c++17 -fsanitize=address,undefined -Wno-unused-result -Wno-option-ignored -pthread video-stitcher_amd/host/$a.cpp -Iinclude -Lab/asanlib -lmsstitch -Wl,-rpath,'$ORIGIN/asanlib' -o ab/${a}_asan 2>&1 | grep -v 'warning\|^ *[0-9]* |\|^ *|' || true
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC -fvisibility=hidden -fsanitize=address,undefined -Wno-option-ignored -Wl,-Bsymbolic tests/fake_rccl.cpp -o ab/libfake_rccl_asan.so -lrt -lpthread
ls -la ab/asanlib/libmsstitch.so ab/stitch_app_asan ab/stitch_dist_asan ab/libfake_rccl_asan.so
