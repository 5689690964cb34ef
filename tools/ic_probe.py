#!/usr/bin/env python3
"""Per-kernel time per frame against the frames per ms_stitch call (one context, one stream): do the kernels whose input was written by the kernel before them
(CPW first remap <- per-frame resize, mesh remap <- first remap, level-0 reduce <- warp) run faster when a call's intermediates fit the 256 MiB Infinity Cache?
Usage (GPU box): python tools/ic_probe.py [config:frames,frames,... ...]   ->  one line per run, us per frame of every kernel of the chain."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = sys.argv[1:] or ["shipped:3,6,9,12,18,32", "cfg3:3,6,12,32", "cfg2:3,6,12,32"]
for r in runs:
    cfg, fl = r.split(":")
    for F in [int(x) for x in fl.split(",")]:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--streams", "1", "--frames", str(F), "--steps", "5", "--warmup", "2", "--passes", str(max(20, 640 // F)),
               "--no-live", "--no-pcie", "--no-cpu-baseline", "--no-distinct", "--recalib-every", "0"]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(cfg, F, "FAILED", p.stderr[-400:]); continue
        d = json.loads(line[-1])
        k = d["kernels_ms_per_call"]
        print("%s F=%d: %.0f frames/s verified=%s | " % (cfg, F, d["value"], d.get("verified")) + "  ".join("%s %.2f" % (n, 1e3 * v / F) for n, v in k.items()) +
              " | sum %.2f us/frame" % (1e3 * sum(k.values()) / F), flush=True)
