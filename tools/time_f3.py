#!/usr/bin/env python3
"""Run ON THE GPU BOX: the ingest / egress format kernels that are NOT inside ms_stitch, config-2 sizes, inputs resident in HBM --
     ms_nv12_to_bgr_batch   cvtColor(YUV2BGR_NV12) of the capture threads (APP/networking.cpp:45-47): 32 frames x 6 cameras of 1080p per call
     ms_bgr_to_i420_batch   cvtColor(BGR2YUV_I420) of consume() (APP/timed.cpp:308-316): the panorama rows of 32 canvases
     ms_consume_i420        consume()'s resize to 4096 x 2048 + black bars + I420 in one pass (APP/timed.cpp:251-316): one panorama
     ms_stitch_i420 / ms_stitch   the whole call with the level-0 band kernel writing planar I420 / the 8UC3 canvas
   GPU time from events (median of 20), ALGORITHMIC bytes (every input byte read once, every output byte written once) and the resulting TB/s.
   Under rocprofv3 (tools/profile_f3.sh) the same run gives the PMC bytes per kernel; --calib adds the 1 GiB calibration copy."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))
import msstitch as ms      # noqa: E402
import synth               # noqa: E402

F = 32
cfg = synth.CONFIGS["cfg2"]
W, H, N = cfg["w"], cfg["h"], cfg["n"]


def timed(fn, n=20):
    ts = []
    for it in range(n + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        if it >= 3:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


rows = []


def row(name, ms_t, nbytes, per):
    rows.append({"kernel": name, "ms": round(ms_t, 4), "us_per_frame": round(ms_t * 1e3 / per, 2), "alg_MB": round(nbytes / 1e6, 1), "TBps_alg": round(nbytes / (ms_t * 1e-3) / 1e12, 3),
                 "frac_of_8TBps": round(nbytes / (ms_t * 1e-3) / 8e12, 3)})


# ---- NV12 -> BGR, 32 x 6 images per call (8 distinct frame sets) ------------------------------------------------------------------------------
pool = [[torch.from_numpy(np.ascontiguousarray(np.roll(synth.nv12_frame(W, H, i), 7 * t, axis=1))).cuda() for i in range(N)] for t in range(8)]
nv = [pool[j % 8][i] for j in range(F) for i in range(N)]
bgr = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(F * N)]
conv_nv = ms.nv12_to_bgr_batch_prepared(nv, bgr)
row("ms_nv12_to_bgr_batch (%d x %d images of %dx%d, 3 launches of 64)" % (F, N, W, H), timed(conv_nv), F * N * W * H * 4.5, F)

# ---- the compositor, for real canvases ------------------------------------------------------------------------------------------------------------
comp = ms.Compositor(N, (W, H), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"], out_size=(cfg["out_w"], cfg["out_h"]), max_frames=F)
g = synth.gains(N)
for i in range(N):
    comp.set_camera(i, *synth.camera(N, W, H, cfg["hfov_deg"], i)); comp.set_gain(i, g[i])
comp.build_maps(); comp.build_masks(1); comp.init_blender()
frames = [[bgr[j * N + i] for i in range(N)] for j in range(F)]
canv = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
run8 = comp.prepared(frames, out8u=canv)
slabs = comp.new_i420(F)
runi = comp.prepared_i420(frames, slabs)
t8, ti = timed(run8), timed(runi)
pg = comp.pano_geom()
y0, rows_i = comp.i420_rows()
rows.append({"kernel": "ms_stitch (8UC3 canvas) vs ms_stitch_i420 (planar I420 by the level-0 band kernel), %d frames per call" % F, "ms": [round(t8, 4), round(ti, 4)],
             "us_per_frame": [round(t8 * 1e3 / F, 2), round(ti * 1e3 / F, 2)], "output_MB_per_frame": [round(3.0 * pg.dst_roi_final.width * pg.dst_roi_final.height / 1e6, 2), round(1.5 * cfg["out_w"] * rows_i / 1e6, 2)]})

# ---- BGR -> I420 of the panorama rows of every canvas (the per-op egress: what ms_stitch_i420 makes unnecessary) ---------------------------------------
srcs = [c[y0:y0 + rows_i] for c in canv]
dsts = [torch.empty((rows_i * 3 // 2, cfg["out_w"]), dtype=torch.uint8, device="cuda") for _ in range(F)]
conv = ms.bgr_to_i420_batch_prepared(srcs, dsts)
row("ms_bgr_to_i420_batch (%d canvases, %d rows of %d)" % (F, rows_i, cfg["out_w"]), timed(conv), F * cfg["out_w"] * rows_i * 4.5, F)

# ---- consume(): one 8UC3 canvas -> 4096 x 2048 I420 with black bars ---------------------------------------------------------------------------------------
out_w, out_h = 4096, 2048
res = {}


def cons():
    res["o"] = ms.consume_i420(canv[0], (out_w, out_h))


t = timed(cons)
ih = res["o"][1]
row("ms_consume_i420 (%dx%d canvas -> %dx%d I420, image height %d)" % (cfg["out_w"], cfg["out_h"], out_w, out_h, ih), t, 3.0 * cfg["out_w"] * cfg["out_h"] + 1.5 * out_w * out_h, 1)

if "--calib" in sys.argv:
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").random_(0, 255); b = torch.empty_like(a)
    for _ in range(3):
        ms.calib_copy(a, b)
    torch.cuda.synchronize()
for r in rows:
    print(json.dumps(r))
