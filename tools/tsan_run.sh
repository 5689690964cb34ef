#!/usr/bin/env bash
# Run ON THE GPU BOX: the host pipelines (stitch_app: capture / stitch / consume / recalibration threads; stitch_dist: one thread per rank over the host transport) with the
# library's host code under ThreadSanitizer (tools/tsan_build.sh).  The HIP / HSA runtimes are not instrumented: TSan reports hundreds of "races" between their own
# allocations and worker threads; what counts is a report with a frame of THIS repository on top of either access -- listed at the end (none = clean).
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/tsan; mkdir -p $O
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 exitcode=0"
M="--views 6 --size 640x360 --out 1280x640 --hfov 90 --bands 4"
run() { n=$1; shift; timeout 240 "$@" > $O/$n.out 2> $O/$n.err; echo "$n rc=$? tsan_reports=$(grep -c 'WARNING: ThreadSanitizer' $O/$n.err) ours=$(grep -E -c '^\s+#0 .*/root/repo/' $O/$n.err) $(tail -c 120 $O/$n.out | tr '\n' ' ')"; }
run app_plain ab/stitch_app_tsan $M --frames 60
run app_cpw ab/stitch_app_tsan $M --frames 300 --cpw
run app_mask ab/stitch_app_tsan $M --frames 300 --update-mask 12
run app_solve ab/stitch_app_tsan $M --frames 600 --solve-mesh
run app_consume ab/stitch_app_tsan $M --frames 40 --consume 512x256
run app_nv12 ab/stitch_app_tsan $M --frames 40 --nv12-direct
run dist2 ab/stitch_dist_tsan --gpus 2 --share-gpu --frames 64 --batch 4 $M
run dist4 ab/stitch_dist_tsan --gpus 4 --share-gpu --col-shards 2 --frames 32 --batch 4 $M --cpw --recalib-every 8
# round 5: the RCCL branch of csrc/dist.cpp over the loopback library (tests/fake_rccl.cpp, instrumented as well): rank threads, proxy threads, the shared-memory mailbox
export GPU_MAX_HW_QUEUES=32
run dist2_rccl ab/stitch_dist_tsan --gpus 2 --share-gpu --transport rccl --rccl-lib $PWD/ab/libfake_rccl_tsan.so --frames 64 --batch 4 $M
run dist4_rccl ab/stitch_dist_tsan --gpus 4 --share-gpu --transport rccl --rccl-lib $PWD/ab/libfake_rccl_tsan.so --col-shards 2 --frames 32 --batch 4 $M --cpw --recalib-every 8
echo "--- reports with this repository's code on top of an access:"
for f in $O/*.err; do grep -E '^\s+#0 .*/root/repo/' $f | sed 's/(.*//' | awk -v f=$(basename $f) '{$1=""; print f ":" $0}' | sort | uniq -c; done
