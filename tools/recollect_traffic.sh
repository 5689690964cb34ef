#!/usr/bin/env bash
# Run ON THE GPU BOX: the kernel trace + PMC traffic passes and the per-kernel report of the four configurations only (after a change of csrc/ that leaves everything else
# of a round's collection valid), then REPEATS default bench lines on the same box.   gpurun --timeout 2400 -- 'MS_COMMIT=<hash> bash tools/recollect_traffic.sh r05 5'
set -uo pipefail
TAG=${1:-r05}; REPEATS=${2:-5}
cd "$GRAFT_REPO_ROOT"
PO=gpurun_out/profiles_out; mkdir -p $PO
bash tools/profile_traffic.sh ${TAG}_cfg3 cfg3 32 > $PO/traffic_cfg3.log 2>&1
bash tools/profile_traffic.sh ${TAG}_cfg5 cfg5 16 > $PO/traffic_cfg5.log 2>&1
bash tools/profile_traffic.sh ${TAG}_shipped shipped 32 > $PO/traffic_shipped.log 2>&1
bash tools/profile_traffic.sh $TAG cfg2 32 > $PO/traffic.log 2>&1
for t in $TAG ${TAG}_cfg3 ${TAG}_cfg5 ${TAG}_shipped; do python tools/report.py $t > /dev/null 2>&1; done
cp profiles/${TAG}_* profiles/traffic_*.json $PO/ 2>/dev/null
: > $PO/${TAG}_bench_repeats.txt
for i in $(seq $REPEATS); do
  python bench.py --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), (d.get('value_distinct') or {}).get('value'), d['ceiling']['copy_TBps'], d['roofline']['frac'], d['roofline'].get('traffic_stale'), d['frame_roofline']['wall_frac_traffic'], round(d['kernels_ms_per_call']['k_warp']*1e3,1), d['live']['us_per_frame_p50'])" >> $PO/${TAG}_bench_repeats.txt
done
cat $PO/${TAG}_bench_repeats.txt
