#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace [+ PMC]) into the text tables we commit under
profiles/: per-kernel calls / total / mean / min / max duration, and per-kernel mean counter values."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    lines = []
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines.append("# rocprofv3 --kernel-trace summary of %s" % path)
    lines.append("%-60s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "mean_ns", "min_ns", "max_ns", "pct"))
    for n, c, s, a, mn, mx in rows:
        lines.append("%-60s %8d %14d %12.1f %12d %12d %6.2f%%" % (n[:60], c, s, a, mn, mx, 100.0 * s / tot))
    try:
        pm = cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                         "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except Exception as ex:  # noqa
        pm = []
        lines.append("# (no PMC data: %s)" % ex)
    if pm:
        lines.append("")
        lines.append("%-60s %-28s %18s %8s %12s" % ("kernel", "counter", "mean_value", "samples", "mean_dur_ns"))
        for k, p, v, c, d in pm:
            lines.append("%-60s %-28s %18.1f %8d %12.1f" % (k[:60], p, v, c, d or 0))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
