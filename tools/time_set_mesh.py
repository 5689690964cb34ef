#!/usr/bin/env python3
"""GPU time of one recalibration's mesh -> map expansion (ms_set_mesh for every view, convertMeshesToMap) on the config-3 rig."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("video-stitcher_amd", "tests", "tools", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import msstitch as ms
import synth
from bench_mesh_solver import build

comp, cfg, warped, matches = build()
n = cfg["n"]
for M in (10, 40):
    meshes = [[synth.mesh(comp.view_geom(i).roi.width, comp.view_geom(i).roi.height, M, M, phase=0.1 * i + 0.7 * r) for i in range(n)] for r in range(4)]
    for r in range(3):
        for i in range(n):
            comp.set_mesh(i, *meshes[r][i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gpu, host = [], []
    for r in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for i in range(n):
            comp.set_mesh(i, *meshes[r % 4][i])
        e1.record()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1))
    print(json.dumps({"what": "ms_set_mesh x %d views" % n, "mesh": "%dx%d" % (M, M), "gpu_ms": round(sorted(gpu)[len(gpu) // 2], 3),
                      "host_enqueue_ms": round(1e3 * sorted(host)[len(host) // 2], 3)}))

# per view (the view straddling +-pi is four times as wide as the others)
M = 40
meshes = [synth.mesh(comp.view_geom(i).roi.width, comp.view_geom(i).roi.height, M, M, phase=0.1 * i) for i in range(n)]
for i in range(n):
    ts = []
    for r in range(10):
        torch.cuda.synchronize()
        e0.record(); comp.set_mesh(i, *meshes[i]); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(json.dumps({"view": i, "size": [comp.view_geom(i).roi.width, comp.view_geom(i).roi.height], "gpu_ms": round(sorted(ts)[5], 4)}))
