#!/usr/bin/env python3
"""Generates tests/golden/*.npz FROM THE ORACLE (regression vectors; they pin the oracle against accidental
change).  The reference itself cannot be executed here (CUDA path: no nvcc/NVIDIA GPU; vendored OpenCV: needs its
cmake-generated headers), so these are NOT reference outputs -- the reference-derived known answers are the
integers in geometry_kats.json.   Usage: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))
import oracle as O  # noqa: E402
import synth  # noqa: E402

rng = np.random.default_rng(20260928)
src = rng.integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
x, y = np.meshgrid(np.arange(48), np.arange(36))
mx = (0.8 * x - 0.3 * y + 5 + rng.uniform(-0.5, 0.5, x.shape)).astype(np.float32)
my = (0.3 * x + 0.9 * y - 4).astype(np.float32)
mx[0, 0] = np.nan; mx[1, 1] = -1; my[1, 1] = -1
s16 = rng.integers(-32768, 32768, size=(21, 30, 3), dtype=np.int16)
wmap = (rng.integers(0, 256, size=(33, 27)).astype(np.float32) * np.float32(1 / 255.)).astype(np.float32)
mesh = (rng.random((6, 7), dtype=np.float32) * 50).astype(np.float32)
mesh_x, mesh_y = synth.mesh(64, 48, 5, 6, phase=0.7, amp=5.0)
dt_in = (rng.random((24, 31)) < 0.9).astype(np.uint8) * 255
mmx, mmy = O.convert_mesh_to_map(mesh_x, mesh_y, 64, 48)
np.savez_compressed(os.path.join(HERE, "prims_small.npz"), src=src, mx=mx, my=my, remap=O.remap_linear_8uc3(src, mx, my),
                    s16=s16, pyr_down=O.pyr_down_16s(s16), pyr_up=O.pyr_up_16s(s16), wmap=wmap, pyr_down_32f=O.pyr_down_32f(wmap),
                    mesh=mesh, custom_resize=O.custom_resize_32f(mesh, 61, 47), mesh_x=mesh_x, mesh_y=mesh_y,
                    mesh_map_x=mmx, mesh_map_y=mmy, dt_in=dt_in, dt_out=O.distance_transform_l1(dt_in))

# two overlapping views, half masks, 3 bands (the shape of MultiBandBlender.CanBlendTwoImages, test_blenders.cpp:47-78)
corners = [(0, 0), (40, 3)]; sizes = [(96, 64), (96, 61)]
imgs, masks = [], []
for i, (w, h) in enumerate(sizes):
    imgs.append(synth.frame(w, h, i, 0, noise=False))
    m = np.zeros((h, w), np.uint8)
    if i == 0:
        m[:, :68] = 255
    else:
        m[:, 28:] = 255
    masks.append(m)
b = O.Blender(corners, sizes, 3)
for i in range(2):
    b.init_view(i, masks[i])
for i in range(2):
    b.feed(i, imgs[i])
out, omask = b.blend()
plain = np.zeros_like(out)
plain[0:64, 0:68] = imgs[0][:, :68]
plain[3:64, 68:136] = imgs[1][:, 28:]
sel = omask != 0
mse = np.mean((out[sel].astype(float) - plain[sel].astype(float)) ** 2)
psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-9))
np.savez_compressed(os.path.join(HERE, "blender_2view.npz"), corners=np.array(corners), sizes=np.array(sizes), num_bands=3,
                    img0=imgs[0], img1=imgs[1], mask0=masks[0], mask1=masks[1], out=out, out_mask=omask, psnr_vs_feather_free=psnr)
print("wrote golden vectors; blend PSNR vs plain copy = %.1f dB" % psnr)

# recalibration path (regression vectors of the numpy oracles): cv::remap's CPU arithmetic, Hamming 2-NN, a small CPW mesh solve
sys.path.insert(0, os.path.join(ROOT, "tests"))
import features_oracle as FO  # noqa: E402
import mesh_oracle as MO  # noqa: E402
from test_mesh_oracle import rig  # noqa: E402
q = rng.integers(0, 256, size=(40, 32), dtype=np.uint8) & np.uint8(0x33)
t = rng.integers(0, 256, size=(57, 32), dtype=np.uint8) & np.uint8(0x33)
kidx, kdist = FO.knn2(q, t)
images, matches = rig(n=3, seed=21)
gx, gy, ginfo = MO.create_mesh(images, matches, 6, 5, focal=60.0, global_dist=8, theta_fn=lambda s, d: MO.generic_theta(s, d, 3))
np.savez_compressed(os.path.join(HERE, "recalibration_small.npz"), src=src, mx=mx, my=my, cv_remap=O.cv_remap_linear(src, mx, my),
                    q=q, t=t, knn_idx=kidx, knn_dist=kdist, sal0=MO.saliency(images[0], 6, 5),
                    mesh_images=np.stack(images), mesh_matches=np.array([m for l in matches for m in l], np.float64), mesh_counts=np.array([len(l) for l in matches]),
                    mesh_x=gx, mesh_y=gy, mesh_iterations=ginfo["iterations"], mesh_rows=ginfo["rows"], mesh_nnz=ginfo["nnz"])
print("wrote recalibration_small.npz: %d iterations" % ginfo["iterations"])
