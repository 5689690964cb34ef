"""Shared helpers of the parity tests."""
import numpy as np
import torch

import synth


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_dev_roi(a, rng, pad=(5, 15)):
    """cvtest::createMat(size, type, useRoi=true) (OCV/ts/src/cuda_test.cpp:92-104): embed the image in a
    larger allocation so the row step is not cols*elemSize."""
    a = np.ascontiguousarray(a)
    t0, t1, l0, l1 = [int(rng.integers(pad[0], pad[1])) for _ in range(4)]
    big = np.zeros((a.shape[0] + t0 + t1, a.shape[1] + l0 + l1) + a.shape[2:], a.dtype)
    big[t0:t0 + a.shape[0], l0:l0 + a.shape[1]] = a
    dev = torch.from_numpy(big).cuda()
    return dev[t0:t0 + a.shape[0], l0:l0 + a.shape[1]]


def host(t):
    return t.detach().cpu().numpy()


def make_rig(ms, name, enable_cpw=False, max_frames=1, mask_mode=1, projection=None, simple_kernels=False, lds_stage=None, shards=1, shard_index=0,
             col_shards=1, col_shard_index=0, update_mask_margin=0):
    """Compositor for one of synth.CONFIGS, calibrated end to end on the device."""
    cfg = synth.CONFIGS[name]
    proj = ms.PROJ_SPHERICAL if projection is None else projection
    comp = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), proj, synth.warp_scale(cfg["out_w"]),
                         num_bands=cfg["num_bands"], enable_cpw=enable_cpw, out_size=(cfg["out_w"], cfg["out_h"]),
                         max_frames=max_frames, simple_kernels=simple_kernels, lds_stage=lds_stage, shards=shards, shard_index=shard_index,
                         col_shards=col_shards, col_shard_index=col_shard_index, update_mask_margin=update_mask_margin)
    g = synth.gains(cfg["n"])
    for i in range(cfg["n"]):
        K, R = synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i)
        comp.set_camera(i, K, R)
        comp.set_gain(i, g[i])
    comp.build_maps()
    comp.build_masks(mask_mode)
    comp.init_blender()
    return comp, cfg, g


def oracle_blender_from(O, comp, cfg):
    """An oracle MultiBandBlender with the SAME geometry, maps-derived masks taken from the device."""
    rois = [comp.view_geom(i).roi.tuple() for i in range(cfg["n"])]
    b = O.Blender([r[:2] for r in rois], [r[2:] for r in rois], cfg["num_bands"])
    for i in range(cfg["n"]):
        b.init_view(i, host(comp.mask(i)))
    return b, rois
