#!/usr/bin/env bash
# Run ON THE GPU BOX: how much of the timed region of the default bench.py run is the GPU idle?  rocprofv3 kernel timeline -> union of the kernel intervals over the
# three streams, per pass; the gaps between the last kernel of a pass (fork-join) and the first of the next.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-.}"
D=/tmp/idle_tl; rm -rf $D; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py ${BENCH_ARGS:-} --steps 4 --warmup 2 --no-cpu-baseline --no-pcie --no-verify --no-live --no-distinct > /tmp/idle_tl.json 2> /tmp/idle_tl.err
python - "$D" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_calib" in n or "rocclr" in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", "").replace("ms::", "")[:24]))
rows.sort()
# the timed region: the last 4 steps x 20 passes; take the last 60 % of the kernels as steady state
rows = rows[int(len(rows) * 0.4):]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(((s - cur_e) / 1e3, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print("span %.1f ms, GPU busy (union of kernels) %.1f ms = %.3f; %d idle gaps, sum %.1f ms" % (span / 1e6, busy / 1e6, busy / span, len(gaps), sum(g for g, _ in gaps) / 1e3))
gaps.sort(reverse=True)
big = [g for g in gaps if g[0] > 20]
print("gaps > 20 us: %d, sum %.1f ms; the ten largest:" % (len(big), sum(g for g, _ in big) / 1e3), [(round(g, 1), n) for g, n in gaps[:10]])
import collections
c = collections.Counter(n for g, n in gaps if g > 3)
print("kernel after a gap > 3 us:", c.most_common(8))
PY
rm -rf $D
