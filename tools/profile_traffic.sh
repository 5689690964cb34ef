#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): kernel trace + separate PMC passes for FETCH_SIZE and WRITE_SIZE of bench.py,
# with a known-size streaming copy in the same process for calibration.  Outputs under gpurun_out/prof_$TAG/.
#   gpurun -- 'bash tools/profile_traffic.sh r01 cfg2 16'
set -euo pipefail
TAG=${1:-r01}; CFG=${2:-cfg2}; F=${3:-16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
B="python bench.py --steps 12 --warmup 3 --no-live --no-pcie --no-verify --no-cpu-baseline --frames $F --streams 1 --config $CFG --calib"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $B > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- $B > $OUT/bench_write.log 2>&1
python tools/summarize_traffic.py $OUT $TAG $CFG $F
# the plain bench line of this configuration, run AFTER the summary so that its roofline.traffic reads this pass's traffic_latest.json
python bench.py --config $CFG > $OUT/bench_plain.json 2> $OUT/bench_plain.err && cp $OUT/bench_plain.json profiles/${TAG}_bench.json || true
# gpurun only merges gpurun_out/ back: park the summaries there (copy them into profiles/ and commit)
python tools/report.py $TAG > /dev/null 2>&1 || true
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* profiles/traffic_*.json gpurun_out/profiles_out/
rm -rf $OUT/trace $OUT/fetch $OUT/write      # the rocprofv3 databases are tens of MB each: only the summaries travel back (gpurun merges <= 64 MiB)
