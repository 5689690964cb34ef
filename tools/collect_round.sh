#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): everything profiles/<tag>_* is made of, in dependency order.
#   gpurun --timeout 2400 -- 'MS_COMMIT=<short hash> bash tools/collect_round.sh r03'      (the GPU box has no .git: the caller passes the commit)
# 1. bench.py default + the other configurations      -> <tag>_bench.json, <tag>_other_configs.json
# 2. SQ / TCC / LDS counter passes, cfg2 and cfg3     -> <tag>_counters.txt, <tag>_cfg3_counters.txt
# 3. kernel trace + FETCH/WRITE PMC passes, cfg3 then cfg2 (traffic_latest.json = cfg2, the bench default) + the per-kernel report
# 4. live-path timeline (one frame per call)          -> <tag>_live_timeline.json
# 5. shards / features / update_mask / host enqueue    -> <tag>_shards.json, <tag>_features.txt, <tag>_update_mask.txt, <tag>_host_enqueue.txt
# Everything lands in gpurun_out/profiles_out/ (gpurun merges only gpurun_out/ back): copy into profiles/ and commit.
set -uo pipefail
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"
PO=gpurun_out/profiles_out; mkdir -p $PO
bash tools/refresh_profiles.sh $TAG > $PO/refresh.log 2>&1
cp gpurun_out/refresh/${TAG}_bench.json gpurun_out/refresh/${TAG}_other_configs.json profiles/ 2>/dev/null
bash tools/profile_counters.sh $TAG cfg2 32 > $PO/counters.log 2>&1
cp gpurun_out/counters_$TAG.txt profiles/${TAG}_counters.txt
bash tools/profile_counters.sh ${TAG}_cfg3 cfg3 32 > $PO/counters_cfg3.log 2>&1
cp gpurun_out/counters_${TAG}_cfg3.txt profiles/${TAG}_cfg3_counters.txt
bash tools/profile_traffic.sh ${TAG}_cfg3 cfg3 32 > $PO/traffic_cfg3.log 2>&1
bash tools/profile_traffic.sh ${TAG}_cfg5 cfg5 16 > $PO/traffic_cfg5.log 2>&1
bash tools/profile_traffic.sh ${TAG}_shipped shipped 64 > $PO/traffic_shipped.log 2>&1
bash tools/profile_counters.sh ${TAG}_cfg5 cfg5 16 > $PO/counters_cfg5.log 2>&1
cp gpurun_out/counters_${TAG}_cfg5.txt profiles/${TAG}_cfg5_counters.txt
bash tools/profile_counters.sh ${TAG}_shipped shipped 64 > $PO/counters_shipped.log 2>&1
cp gpurun_out/counters_${TAG}_shipped.txt profiles/${TAG}_shipped_counters.txt
bash tools/profile_traffic.sh $TAG cfg2 32 > $PO/traffic.log 2>&1
python tools/report.py $TAG > $PO/report.log 2>&1
cp profiles/${TAG}_* profiles/traffic_*.json $PO/ 2>/dev/null
bash tools/live_timeline.sh $TAG > /dev/null 2>&1
# 5. the two intra-frame sharding schemes on one GPU, the recalibration front-end, the enqueue-only mask update
python tools/refresh_view_shards.py $TAG > $PO/shards.log 2>&1 && cp gpurun_out/${TAG}_shards.json $PO/
python tools/bench_features.py $TAG > $PO/features.log 2>&1 && cp gpurun_out/${TAG}_features.txt $PO/
python tools/time_update_mask.py 2>&1 | grep margin > $PO/${TAG}_update_mask.txt
python tools/host_enqueue.py 2>&1 | grep calls > $PO/${TAG}_host_enqueue.txt
# 6. the bandwidth ceiling sweep, the from-the-inputs parity statistics, the multi-rank host pipeline (ranks share the GPU on a one-GPU box)
[ -x ab/copy_probe ] && ab/copy_probe 1024 > $PO/${TAG}_copy_probe.txt 2>&1
rm -f $PO/${TAG}_from_inputs.txt; MS_FROM_INPUTS_LOG=$PWD/$PO/${TAG}_from_inputs.txt python -m pytest tests/test_from_inputs_gpu.py -q > $PO/from_inputs.log 2>&1
{ video-stitcher_amd/stitch_dist --gpus 1 --frames 256 --batch 16 --no-checksum; video-stitcher_amd/stitch_dist --gpus 2 --share-gpu --frames 256 --batch 16 --no-checksum;
  video-stitcher_amd/stitch_dist --gpus 2 --share-gpu --col-shards 2 --frames 64 --batch 4 --views 12 --size 3840x2160 --out 7680x3840 --hfov 60 --no-checksum; } > $PO/${TAG}_stitch_dist.txt 2>&1
# 7. round 4: every frame of the default pass distinct (96 frame sets, 3.6 GB of source: nothing survives in the Infinity Cache) beside the 8-set pool; the NV12 ingest both ways;
#    the instruction-rate probe
timeout 900 python bench.py --no-cpu-baseline --no-pcie --no-live --distinct 96 > $PO/${TAG}_bench_distinct96.json 2> $PO/distinct96.err
{ python tools/time_nv12.py 32; python tools/time_nv12.py 1; python tools/time_nv12.py 32 cpw; } 2>/dev/null > $PO/${TAG}_nv12.txt
[ -x ab/valu_probe ] && ab/valu_probe > $PO/${TAG}_valu_probe.txt 2>&1
du -sh gpurun_out; ls $PO
