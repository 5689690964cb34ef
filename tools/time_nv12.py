#!/usr/bin/env python3
"""Run ON THE GPU BOX: the NV12 ingest of config 2 both ways, 32 frames per call, inputs resident in HBM --
   (a) ms_nv12_to_bgr_batch + ms_stitch  (the round-3 path: the BGR image is written and read again)
   (b) ms_stitch_nv12                    (k_warp_nv12 samples the planes directly)
GPU time per frame from events around the calls (median of 20), and a byte-for-byte comparison of the canvases."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))
import msstitch as ms      # noqa: E402
import synth               # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CPW = len(sys.argv) > 2 and sys.argv[2] == "cpw"      # config 3: CPW on, 40 x 40 meshes (stage 1 then samples the NV12 planes)
cfg = synth.CONFIGS["cfg2"]
comp = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"], out_size=(cfg["out_w"], cfg["out_h"]), max_frames=F,
                     enable_cpw=CPW)
g = synth.gains(cfg["n"])
for i in range(cfg["n"]):
    comp.set_camera(i, *synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i)); comp.set_gain(i, g[i])
comp.build_maps(); comp.build_masks(1); comp.init_blender()
if CPW:
    for i in range(cfg["n"]):
        r = comp.view_geom(i).roi
        comp.set_mesh(i, *synth.mesh(r.width, r.height, 40, 40, phase=0.1 * i))
rng = np.random.default_rng(3)
pool = [[torch.from_numpy(np.ascontiguousarray(np.roll(synth.nv12_frame(cfg["w"], cfg["h"], i), 7 * t, axis=1))).cuda() for i in range(cfg["n"])] for t in range(8)]
nv = [pool[j % 8] for j in range(F)]
bgr = [[torch.empty((cfg["h"], cfg["w"], 3), dtype=torch.uint8, device="cuda") for _ in range(cfg["n"])] for _ in range(F)]
outa = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
outb = [torch.zeros_like(outa[0]) for _ in range(F)]
flat_nv = [t for fr in nv for t in fr]; flat_bgr = [t for fr in bgr for t in fr]
run_a = comp.prepared(bgr, out8u=outa)
run_b = comp.prepared_nv12(nv, out8u=outb)


def timed(fn, n=20):
    ts = []
    for it in range(n + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        if it >= 3:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


conv_run = ms.nv12_to_bgr_batch_prepared(flat_nv, flat_bgr)      # (descriptors marshalled once, like run_a / run_b)


def path_a():
    conv_run()
    run_a()


conv = timed(conv_run)
a = timed(path_a)
b = timed(run_b)
torch.cuda.synchronize()
same = all(torch.equal(x, y) for x, y in zip(outa, outb))
print(json.dumps({"config": "cfg3 (CPW 40x40)" if CPW else "cfg2", "frames_per_call": F, "convert_ms": round(conv, 4), "convert_then_stitch_ms": round(a, 4), "stitch_nv12_ms": round(b, 4),
                  "us_per_frame": {"convert": round(conv / F * 1e3, 2), "convert_then_stitch": round(a / F * 1e3, 2), "stitch_nv12": round(b / F * 1e3, 2)},
                  "bit_identical": bool(same)}))
