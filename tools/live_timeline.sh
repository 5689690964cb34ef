#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): kernel timeline of the live path (one frame per ms_stitch call) -> gpurun_out/profiles_out/<tag>_live_timeline.json
# Per kernel of the chain: mean duration and mean gap to the end of the previous kernel (launch + dependency latency).
set -uo pipefail
TAG=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/profiles_out; mkdir -p $OUT
D=/tmp/live_tl; rm -rf $D; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --frames 1 --streams 1 --steps 200 --warmup 3 --passes 1 \
    --no-cpu-baseline --no-pcie --no-verify --no-live --no-distinct ${BENCH_EXTRA:-} > $OUT/${TAG}_live_bench.json 2> /tmp/live_tl.err
python - "$D" "$OUT/${TAG}_live_timeline.json" <<'PY'
import csv, glob, json, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("ms::", "")))
rows.sort()
# frames: a chain starts at a warp kernel
frames, cur = [], []
for s, e, n in rows:
    if n.startswith("k_warp") and cur: frames.append(cur); cur = []
    cur.append((s, e, n))
if cur: frames.append(cur)
L = collections.Counter(len(f) for f in frames).most_common(1)[0][0]
frames = [f for f in frames if len(f) == L][-150:]
out = {"frames": len(frames), "kernels_per_frame": L, "chain": []}
for i in range(L):
    dur = [f[i][1] - f[i][0] for f in frames]
    gap = [f[i][0] - f[i - 1][1] for f in frames] if i else [0]
    out["chain"].append({"kernel": frames[0][i][2], "dur_us": round(sum(dur) / len(dur) / 1e3, 2), "gap_before_us": round(sum(gap) / len(gap) / 1e3, 2)})
span = [f[-1][1] - f[0][0] for f in frames]
out["gpu_span_us"] = round(sum(span) / len(span) / 1e3, 2)
out["sum_dur_us"] = round(sum(k["dur_us"] for k in out["chain"]), 2)
out["sum_gap_us"] = round(sum(k["gap_before_us"] for k in out["chain"]), 2)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $D
