#!/usr/bin/env python
"""Start-up from a calibration blob (ms_save_tables / ms_load_tables; the reference recomputes its calibration at every start, timed.cpp:553).
Run on the GPU box:  python tools/time_tables_blob.py [config]      (config: a key of synth.CONFIGS, default cfg2)
One JSON line: ms to make a ready context from cameras, gains and seam masks (build_maps + build_masks + init_blender), ms to save the blob and its size,
ms for ms_load_tables to hand back a ready context, and whether that context stitches the same frame byte for byte."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))
import msstitch as ms      # noqa: E402
import synth               # noqa: E402


def main(name="cfg2"):
    cfg = synth.CONFIGS[name]
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comp = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"], out_size=(cfg["out_w"], cfg["out_h"]))
    gains = synth.gains(cfg["n"])
    for i in range(cfg["n"]):
        comp.set_camera(i, *synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i))
        comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    blob = comp.save_tables()
    t_save = time.perf_counter() - t0
    t0 = time.perf_counter()
    twin = ms.Compositor.from_tables(blob)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    frames = [[torch.from_numpy(synth.frame(cfg["w"], cfg["h"], i, 3)).cuda() for i in range(cfg["n"])]]
    outs = []
    for c in (comp, twin):
        o8 = torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda")
        c.stitch(frames, out8u=[o8])
        torch.cuda.synchronize()
        outs.append(o8)
    print(json.dumps({"config": name, "build_from_inputs_ms": round(t_build * 1e3, 1), "save_blob_ms": round(t_save * 1e3, 1), "blob_MB": round(len(blob) / 1e6, 2),
                      "load_blob_ms": round(t_load * 1e3, 1), "same_frame": bool(torch.equal(outs[0], outs[1]))}))


if __name__ == "__main__":
    main(*sys.argv[1:2])
