#!/usr/bin/env bash
# Run ON THE GPU BOX: where does a recalibration's time go?  Kernel timeline of bench.py --config cfg3 (one context, 32 frames per call, a mesh update every 60 frames):
# the GPU time from the end of the stitch before an update to the start of the stitch after it (kernels of the update + idle), against the ordinary gap between two stitches.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
D=/tmp/recal_tl; rm -rf $D; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --config ${1:-cfg3} --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-verify --no-live --no-distinct > /tmp/recal_tl.json 2> /tmp/recal_tl.err; python -c "import json; d=json.loads(open('/tmp/recal_tl.json').read().strip().splitlines()[-1]); print('bench under rocprofv3:', round(d['value']), 'frames/s')"
python - "$D" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ms::", "")))
rows.sort()
# a stitch call starts at k_stage1_s and ends at k_blend8<true...> (level 0)
plain, recal = [], []
dur_plain, dur_after = [], []      # GPU span of a stitch call (first kernel's start -> level-0 band kernel's end): ordinary calls / the call right after an update
i = 0
last_end = None; pending = []; call_start = None; after_update = False
for s, e, n in rows:
    if n.startswith("k_stage1"):
        after_update = False
        if last_end is not None:
            gap = (s - last_end) / 1e3
            after_update = any(p[2].startswith("k_mesh") for p in pending)
            (recal if after_update else plain).append((gap, [(p[2][:28], round((p[1] - p[0]) / 1e3, 1), round((p[0] - last_end) / 1e3, 1)) for p in pending]))
        pending = []; last_end = None; call_start = s
    elif n.startswith("k_blend8<true"):
        last_end = e; pending = []
        if call_start is not None:
            (dur_after if after_update else dur_plain).append((e - call_start) / 1e3)
    elif last_end is not None:
        pending.append((s, e, n))
import statistics as st
print("ordinary gap between two stitch calls: n=%d median %.1f us" % (len(plain), st.median(g for g, _ in plain)))
print("gap with a mesh update in it: n=%d median %.1f us" % (len(recal), st.median(g for g, _ in recal)))
gp, gr = sorted(g for g, _ in plain), sorted(g for g, _ in recal)
print("gaps: ordinary sum %.0f us, max %.1f, p95 %.1f; with an update sum %.0f us, max %.1f, p95 %.1f; sum of stitch spans %.0f us" % (sum(gp), gp[-1], gp[int(len(gp) * .95)], sum(gr), gr[-1], gr[int(len(gr) * .95)], sum(dur_plain) + sum(dur_after)))
print("GPU span of a stitch call: ordinary median %.1f us (n=%d), right after an update %.1f us (n=%d)" % (st.median(dur_plain), len(dur_plain), st.median(dur_after), len(dur_after)))
for g, p in recal[2:6]:
    print("  %.1f us:" % g, p)
PY
rm -rf $D
