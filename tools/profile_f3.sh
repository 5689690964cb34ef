#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): tools/time_f3.py plain (event timings, algorithmic bytes), then under rocprofv3 -- kernel trace + separate FETCH_SIZE / WRITE_SIZE
# passes, calibrated on the 1 GiB streaming copy of the same process -- and the per-kernel table profiles/<tag>_f3_report.md.
#   gpurun -- 'bash tools/profile_f3.sh r05'
set -uo pipefail
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/f3_$TAG; mkdir -p $OUT gpurun_out/profiles_out
python tools/time_f3.py > $OUT/time_f3.txt 2> $OUT/time_f3.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/time_f3.py --calib > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python tools/time_f3.py --calib > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python tools/time_f3.py --calib > $OUT/write.log 2>&1
python tools/summarize_f3.py $OUT $TAG > $OUT/summary.log 2>&1
cp profiles/${TAG}_f3_report.md $OUT/time_f3.txt gpurun_out/profiles_out/ 2>/dev/null
rm -rf $OUT/trace $OUT/fetch $OUT/write
cat profiles/${TAG}_f3_report.md
