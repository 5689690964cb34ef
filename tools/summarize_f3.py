#!/usr/bin/env python3
"""rocprofv3 databases of tools/profile_f3.sh -> profiles/<tag>_f3_report.md: launch time, PMC bytes (FETCH_SIZE x f_r + WRITE_SIZE x f_w, calibrated on the 1 GiB copy of
the same process) and TB/s of the ingest / egress format kernels that live outside ms_stitch, next to the event timings and algorithmic bytes of tools/time_f3.py."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find_db(d):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(".db"):
                return os.path.join(root, f)
    return None


def counters(db, name):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, avg(value), count(*), avg(duration) from counters_collection where counter_name = ? group by kernel_name", (name,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def trace(db):
    cur = sqlite3.connect(db).cursor()
    return {r[0]: (r[1], r[2]) for r in cur.execute("select name, count(*), avg(end-start) from kernels group by name").fetchall()}


def short(n):
    return n.replace("void ", "").replace("ms::", "").split("(")[0]


def main(out, tag):
    tr = trace(find_db(os.path.join(out, "trace")))
    fetch = counters(find_db(os.path.join(out, "fetch")), "FETCH_SIZE")
    write = counters(find_db(os.path.join(out, "write")), "WRITE_SIZE")
    GiB = float(1 << 30)
    cal_r = [v for k, v in fetch.items() if "k_calib_copy" in k]
    cal_w = [v for k, v in write.items() if "k_calib_copy" in k]
    f_r = GiB / (cal_r[0][0] * 1024.0) if cal_r and cal_r[0][0] > 0 else 2.0
    f_w = GiB / (cal_w[0][0] * 1024.0) if cal_w and cal_w[0][0] > 0 else 1.0
    want = ("k_nv12_to_bgr", "k_bgr_to_i420", "k_consume_i420", "k_blend8<true", "k_blend8<false")
    lines = ["# Ingest / egress format kernels outside ms_stitch (%s): rocprofv3 kernel trace + PMC passes of `tools/time_f3.py`" % tag, "",
             "FETCH_SIZE x %.3f + WRITE_SIZE x %.3f (calibrated on the 1 GiB tuned copy of the same process).  Config-2 sizes: 32 frames x 6 cameras of 1920x1080 per NV12 launch group"
             " (3 launches of 64 images), 32 canvases of 3840 wide per I420 launch, one 3840x1920 canvas per consume()." % (f_r, f_w), "",
             "| kernel | launches | mean launch µs | PMC MB / launch | PMC TB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
    for k in sorted(set(fetch) | set(write)):
        s = short(k)
        if not s.startswith(want):
            continue
        fr, wr = fetch.get(k, (0, 0, 0)), write.get(k, (0, 0, 0))
        n, ns = tr.get(k, (fr[1], fr[2]))
        b = fr[0] * 1024 * f_r + wr[0] * 1024 * f_w
        lines.append("| `%s` | %d | %.1f | %.1f | %.2f | %.3f |" % (s, n, ns / 1e3, b / 1e6, b / (ns * 1e-9) / 1e12, b / (ns * 1e-9) / 8e12))
    t = os.path.join(out, "time_f3.txt")
    if os.path.exists(t):
        lines += ["", "Event timings of the plain run (median of 20; algorithmic bytes = every input byte read once, every output byte written once):", "", "```"] + \
                 [l.rstrip() for l in open(t) if l.startswith("{")] + ["```"]
    p = os.path.join(ROOT, "profiles", "%s_f3_report.md" % tag)
    open(p, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
