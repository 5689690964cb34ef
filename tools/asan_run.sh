#!/usr/bin/env bash
# Run ON THE GPU BOX: the host pipelines with the library's host code under AddressSanitizer + UndefinedBehaviorSanitizer
# (ab/asanlib/libmsstitch.so = tools/build_ab.sh asan -fsanitize=address,undefined ...; ab/stitch_app_asan, ab/stitch_dist_asan)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/asan; mkdir -p $O
export ASAN_OPTIONS="detect_leaks=0 halt_on_error=0 protect_shadow_gap=0" UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=0"
M="--views 6 --size 640x360 --out 1280x640 --hfov 90 --bands 4"
run() { n=$1; shift; timeout 240 "$@" > $O/$n.out 2> $O/$n.err; echo "$n rc=$? asan=$(grep -c 'ERROR: AddressSanitizer' $O/$n.err) ubsan=$(grep -c 'runtime error' $O/$n.err) $(tail -c 100 $O/$n.out | tr '\n' ' ')"; }
run app_plain ab/stitch_app_asan $M --frames 60
run app_cpw ab/stitch_app_asan $M --frames 300 --cpw
run app_mask ab/stitch_app_asan $M --frames 300 --update-mask 12
run app_solve ab/stitch_app_asan $M --frames 600 --solve-mesh
run app_consume ab/stitch_app_asan $M --frames 40 --consume 512x256
run app_nv12 ab/stitch_app_asan $M --frames 40 --nv12-direct
run app_refcalib ab/stitch_app_asan --reference-calib --frames 6
run dist2 ab/stitch_dist_asan --gpus 2 --share-gpu --frames 64 --batch 4 $M
run dist4 ab/stitch_dist_asan --gpus 4 --share-gpu --col-shards 2 --frames 32 --batch 4 $M --cpw --recalib-every 8
run dist_tables ab/stitch_dist_asan --gpus 2 --share-gpu --frames 32 --batch 4 $M --tables-from-rank0
export GPU_MAX_HW_QUEUES=32
run dist2_rccl ab/stitch_dist_asan --gpus 2 --share-gpu --transport rccl --rccl-lib $PWD/ab/libfake_rccl_asan.so --frames 64 --batch 4 $M
run dist4_rccl ab/stitch_dist_asan --gpus 4 --share-gpu --transport rccl --rccl-lib $PWD/ab/libfake_rccl_asan.so --col-shards 2 --frames 32 --batch 4 $M --cpw --recalib-every 8
echo "--- first lines of every report:"
for f in $O/*.err; do grep -E -A6 'ERROR: AddressSanitizer|runtime error' $f | head -40; done
