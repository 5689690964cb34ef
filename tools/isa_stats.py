#!/usr/bin/env python3
"""Static instruction mix per kernel from hipcc's device assembly (no GPU needed).

  hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S compositor.hip -o comp.s ; python tools/isa_stats.py comp.s [filter]

Counts are static (every instruction once, loops not weighted): a guide to where VALU / memory instructions sit,
not a cycle model."""
import re
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2:] or [""]
    cur, stats = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); stats[cur] = {}
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or re.match(r"^\.Lfunc_end", line):
            cur = None
            continue
        m = re.match(r"^\t([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            cls = "valu"
        elif op.startswith("s_waitcnt"):
            cls = "waitcnt"
        elif op.startswith("s_"):
            cls = "salu"
        elif op.startswith("global_load") or op.startswith("buffer_load"):
            cls = "vmem_ld"
        elif op.startswith("global_store") or op.startswith("buffer_store"):
            cls = "vmem_st"
        elif op.startswith("ds_"):
            cls = "lds"
        elif op.startswith("scratch_"):
            cls = "scratch"
        else:
            cls = "other"
        d = stats[cur]
        d[cls] = d.get(cls, 0) + 1
        d.setdefault("ops", {})
        d["ops"][op] = d["ops"].get(op, 0) + 1
    for k, d in stats.items():
        if not any(f in k for f in flt) or not d:
            continue
        ops = d.pop("ops")
        top = sorted(ops.items(), key=lambda kv: -kv[1])[:14]
        print(k[:90])
        print("   ", {c: d[c] for c in sorted(d)})
        print("   ", " ".join("%s:%d" % kv for kv in top))


if __name__ == "__main__":
    main()
