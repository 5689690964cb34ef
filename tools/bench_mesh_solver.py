#!/usr/bin/env python3
"""Recalibration-path timing of the descriptor matcher (ms_knn_match_hamming2) and the CPW mesh optimiser (ms_create_mesh) on the config-2/3 rig: 6 warped 1080p views, 100 matches per
seam (MAX_FEATURES_PER_IMAGE), the reference's default 10 x 10 mesh and BASELINE config 3's 40 x 40 mesh.  Prints one JSON line per mesh
size; the numpy oracle (oracle/mesh_oracle.py, CPU) is timed beside it on the 10 x 10 case when --cpu is given.

  python tools/bench_mesh_solver.py [--cpu] [--reps 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("video-stitcher_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np
import torch

import msstitch as ms
import synth


def build(rig="cfg2"):
    cfg = synth.CONFIGS[rig]
    comp = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"],
                         enable_cpw=True, out_size=(cfg["out_w"], cfg["out_h"]))
    for i in range(cfg["n"]):
        K, R = synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i)
        comp.set_camera(i, K, R)
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    n = cfg["n"]
    rois = [comp.view_geom(i).roi for i in range(n)]
    warped = [ms.remap(torch.from_numpy(synth.frame(cfg["w"], cfg["h"], i, 0)).cuda(), *comp.maps(i)) for i in range(n)]
    rng = np.random.default_rng(0)
    matches = []
    for i in range(n):
        d = (i + 1) % n
        off = (rois[d].x - rois[i].x) % cfg["out_w"]
        narrow = rois[i].width < cfg["out_w"] // 2 and rois[d].width < cfg["out_w"] // 2
        lst = []
        while narrow and len(lst) < 100:
            x1, y1 = rng.uniform(off + 2, rois[i].width - 2), rng.uniform(8, rois[i].height - 8)
            x2 = x1 - off + rng.normal(4, 2)
            if 0 <= x2 < rois[d].width:
                lst.append((x1, y1, x2, y1 + rng.normal(0, 2), d))
        matches.append(lst)
    return comp, cfg, warped, matches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    # descriptor matching of matchFeatures: cuda::ORB::create(2500, ...) -> 2500 x 2500 Hamming 2-NN per seam (featurefinder.cpp:15, :60)
    rng = np.random.default_rng(1)
    q = torch.from_numpy(rng.integers(0, 256, (2500, 32), dtype=np.uint8)).cuda()
    t = torch.from_numpy(rng.integers(0, 256, (2500, 32), dtype=np.uint8)).cuda()
    ms.knn_match_hamming2(q, t)
    ts = []
    for _ in range(max(3, a.reps)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ms.knn_match_hamming2(q, t)
        ts.append(time.perf_counter() - t0)
    line = {"what": "ms_knn_match_hamming2", "query": 2500, "train": 2500, "bytes": 32, "ms": round(1e3 * min(ts), 3)}
    if a.cpu:
        import features_oracle as fo
        t0 = time.perf_counter()
        fo.knn2(q.cpu().numpy(), t.cpu().numpy())
        line["cpu_oracle_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
    print(json.dumps(line), flush=True)
    comp, cfg, warped, matches = build()
    scale = synth.warp_scale(cfg["out_w"])
    for M in (10, 40):
        prm = ms.mesh_default_params(mesh_cols=M, mesh_rows=M, focal_length=scale, theta_rule=1)
        ms.create_mesh(warped, matches, prm)                      # warm-up (module load, allocator)
        ts = []
        for _ in range(a.reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mx, my, info = ms.create_mesh(warped, matches, prm)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for i in range(cfg["n"]):
            comp.set_mesh(i, mx[i], my[i])
        torch.cuda.synchronize()
        t_set = time.perf_counter() - t0
        line = {"what": "ms_create_mesh", "mesh": "%dx%d" % (M, M), "views": cfg["n"], "matches": sum(map(len, matches)), **info,
                "ms": round(1e3 * min(ts), 2), "us_per_iteration": round(1e6 * min(ts) / max(1, info["iterations"]), 2),
                "set_mesh_ms": round(1e3 * t_set, 2), "max_displacement_px": round(max(comp.mesh_displacement(i) for i in range(cfg["n"])), 2)}
        if a.cpu and M == 10:
            import mesh_oracle as mo
            imgs = [w.cpu().numpy() for w in warped]
            t0 = time.perf_counter()
            rx, ry, rinfo = mo.create_mesh(imgs, matches, M, M, focal=scale, theta_fn=lambda s, d: mo.generic_theta(s, d, cfg["n"]))
            line["cpu_oracle_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
            line["max_vertex_diff_px"] = float(max(np.abs(mx - rx).max(), np.abs(my - ry).max()))
            line["cpu_iterations"] = rinfo["iterations"]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
