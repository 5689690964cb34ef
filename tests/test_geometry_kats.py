"""Geometry known-answer tests (exact integer equality) against the reference-derived values of
SURVEY.md Appendix C -- for BOTH the oracle (pins it) and the product's host geometry (ms_warp_roi /
ms_result_roi run on the CPU; no GPU needed).  Blender padding of the product is checked on the GPU box
through ms_get_view_geom (test_tables_gpu.py)."""
import json
import math
import os

import numpy as np
import pytest

import synth

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "geometry_kats.json")))
PROJ = {"plane": 0, "cylindrical": 1, "spherical": 2}


def rig_rois(fn, case):
    sc = synth.warp_scale(case["out_w"])
    out = []
    for i in range(case["n"]):
        K, R = synth.camera(case["n"], case["w"], case["h"], case["hfov_deg"], i)
        out.append(fn(PROJ[case["proj"]], K, R, sc, case["w"], case["h"]))
    return out


@pytest.mark.parametrize("case", KATS["cases"], ids=lambda c: c["name"])
def test_oracle_rig_geometry(oracle, case):
    rois = rig_rois(oracle.warp_roi, case)
    for r, v in zip(rois, case["views"]):
        assert list(r) == v["tl"] + v["size"]
    rr = oracle.result_roi([r[:2] for r in rois], [r[2:] for r in rois])
    assert list(rr) == case["result_roi"]
    g = oracle.blender_prepare(rr, case["num_bands"])
    assert g.num_bands == case["num_bands"] and list(g.dst_roi.tuple()) == case["dst_roi"]
    for r, v in zip(rois, case["views"]):
        vg = oracle.blender_view_geom(g, r[:2], r[2:])
        assert [vg.top, vg.left, vg.bottom, vg.right] == v["tlbr"]
        assert [r[2] + vg.left + vg.right, r[3] + vg.top + vg.bottom] == v["padded"]
        assert [vg.x_tl, vg.y_tl] == v["xy_tl"]


@pytest.mark.parametrize("case", KATS["cases"], ids=lambda c: c["name"])
def test_product_rig_geometry(ms, case):
    rois = rig_rois(ms.warp_roi, case)
    for r, v in zip(rois, case["views"]):
        assert list(r) == v["tl"] + v["size"]
    assert list(ms.result_roi(rois)) == case["result_roi"]


@pytest.mark.parametrize("sv", KATS["single_views"], ids=lambda c: c["name"])
def test_single_view_rois(oracle, ms, sv):
    K = np.array([[sv["f"], 0, sv["w"] / 2.0], [0, sv["f"], sv["h"] / 2.0], [0, 0, 1]], np.float32)
    _, R = synth.camera(1, sv["w"], sv["h"], 90.0, 0, yaw=math.radians(sv["yaw_deg"]))
    scale = sv["scale"] if "scale" in sv else float(np.float32(sv["scale_num"] / (2 * math.pi)))
    assert list(oracle.warp_roi(PROJ[sv["proj"]], K, R, scale, sv["w"], sv["h"])) == sv["roi"]
    assert list(ms.warp_roi(PROJ[sv["proj"]], K, R, scale, sv["w"], sv["h"])) == sv["roi"]


def test_plane_and_pole_paths_agree(oracle, ms):
    """Paths the rig KATs do not reach: plane warper (all-pixel scan) and the spherical pole fix-up
    (camera looking straight up/down).  Product host code vs oracle, exact."""
    K = np.array([[300, 0, 160], [0, 300, 120], [0, 0, 1]], np.float32)
    a = math.radians(20)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float32)
    assert ms.warp_roi(0, K, R, 300.0, 320, 240) == oracle.warp_roi(0, K, R, 300.0, 320, 240)
    for pitch in (90.0, -90.0, 80.0):
        p = math.radians(pitch)
        Rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]], np.float32)
        got, ref = ms.warp_roi(2, K, Rx, 100.0, 320, 240), oracle.warp_roi(2, K, Rx, 100.0, 320, 240)
        assert got == ref
    p = math.radians(80.0)
    up = oracle.warp_roi(2, K, np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]], np.float32), 100.0, 320, 240)
    assert up[1] + up[3] - 1 == int(np.float32(math.pi * 100.0)), "the pole fix-up must extend the ROI to v = pi*scale"
