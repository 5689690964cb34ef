#!/usr/bin/env bash
# Build the ThreadSanitizer variants HERE (no GPU needed): ab/tsanlib/libmsstitch.so (host code of the library instrumented; device code as usual),
# ab/stitch_app_tsan, ab/stitch_dist_tsan.  Then on the GPU box:  gpurun -- 'bash tools/tsan_run.sh'   (ab/ is git-ignored but travels with the snapshot)
set -euo pipefail
cd "$(dirname "$0")/.."
bash tools/build_ab.sh tsan -fsanitize=thread -g -Wno-option-ignored
mkdir -p ab/tsanlib && cp ab/tsan.so ab/tsanlib/libmsstitch.so
for a in stitch_app stitch_dist; do
  /opt/rocm/bin/hipcc --offload-host-only -O1 -g -std=c++17 -fsanitize=thread -Wno-unused-result -Wno-option-ignored -pthread video-stitcher_amd/host/$a.cpp -Iinclude -Lab/tsanlib -lmsstitch -Wl,-rpath,'$ORIGIN/tsanlib' -o ab/${a}_tsan 2>&1 | grep -v 'warning\|^ *[0-9]* |\|^ *|' || true
done
# the loopback RCCL of the tests, instrumented too: its proxy threads against the rank threads (round 5)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC -fvisibility=hidden -fsanitize=thread -Wno-option-ignored -Wl,-Bsymbolic tests/fake_rccl.cpp -o ab/libfake_rccl_tsan.so -lrt -lpthread
ls -la ab/tsanlib/libmsstitch.so ab/stitch_app_tsan ab/stitch_dist_tsan ab/libfake_rccl_tsan.so
