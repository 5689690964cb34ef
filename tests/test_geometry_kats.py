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


def test_calibrate_cameras_in_product_code_equals_the_test_side_rig_model(ms):
    """a17: calibrateCameras + stitch_calib's scale bookkeeping now live behind the ABI (ms_calibrate_cameras, csrc/geometry.cpp; msshim::calibrateCameras /
    stitch_calib use it).  synth.reference_rig stays as the independent test-side statement of calibration.cpp:28-68, 101-116, 147-181, 269-288:
    every scale and every fp32 matrix entry must agree exactly, for the shipped budgets (defs.h:51-53) and for the BASELINE full-resolution rig."""
    import synth
    for n, w, h, hfov, budgets in [(6, 1920, 1080, 90.0, (0.6, 0.01, 1.4)), (6, 1920, 1080, 90.0, (0.6, 0.01, -1.0)), (6, 1920, 1080, 90.0, (-1.0, 0.01, -1.0)),
                                   (12, 3840, 2160, 60.0, (0.6, 0.01, 1.4)), (4, 200, 150, 110.0, (0.6, 0.01, -1.0))]:
        a = ms.calibrate_cameras(n, w, h, hfov, *budgets)
        b = synth.reference_rig(n, w, h, budgets[0], budgets[1], budgets[2], hfov)
        for k in ("work_scale", "seam_scale", "seam_work_aspect", "compose_scale", "warped_image_scale", "seam_warp_scale", "compose_warp_scale"):
            assert a[k] == b[k], (k, a[k], b[k])
        for i in range(n):
            for k in ("K_seam", "K_compose", "R"):
                assert np.array_equal(a[k][i], b[k][i]), (k, i)
        resized = abs(a["compose_scale"] - 1) > 0.1
        assert bool(a["resize_input"]) == resized
        assert (a["compose_width"], a["compose_height"]) == ((int(np.rint(w * a["compose_scale"])), int(np.rint(h * a["compose_scale"]))) if resized else (w, h))
    # the num_bands rule of calibration.cpp:183-194 on the reference rig's panoramas (SURVEY App. C sizes): 6 bands, as the survey notes
    assert ms.num_bands_rule(3839, 627)[1] == 6 and ms.num_bands_rule(3839, 687)[1] == 6
    bw, nb = ms.num_bands_rule(10, 9)
    assert bw < 1.0 and nb == 0                                     # blend_width < 1: the reference falls back to Blender::NO
