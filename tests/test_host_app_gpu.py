"""The C++ host pipeline (video-stitcher_amd/host/stitch_app.cpp: the thread / queue graph of the reference's timed.cpp over the
C-ABI) must produce the same panorama as the Python binding for the same rig and frames."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

import synth
from helpers import make_rig, to_dev, host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "video-stitcher_amd", "stitch_app")


def run_app(tmp_path, *args):
    dump = str(tmp_path / "pano.bin")
    out = subprocess.check_output([APP, "--dump", dump] + [str(a) for a in args], timeout=300)
    line = [l for l in out.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line), dump


@pytest.mark.parametrize("rig", ["mini6", "cfg2"])
def test_host_app_matches_python_binding(ms, cuda, tmp_path, rig):
    cfg = synth.CONFIGS[rig]
    info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                         "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 12, "--i420")
    assert info["frames"] == 12 and info["frames_per_s"] > 0
    got = np.fromfile(dump, np.uint8).reshape(cfg["out_h"], cfg["out_w"], 3)
    comp, _, _ = make_rig(ms, rig)
    out8 = torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda)
    comp.stitch([[to_dev(synth.frame(cfg["w"], cfg["h"], i, 0, noise=False)) for i in range(cfg["n"])]], out8u=[out8])
    torch.cuda.synchronize()
    assert np.array_equal(got, host(out8))
    comp.close()


@pytest.mark.parametrize("flag", ["--nv12", "--nv12-direct"])
def test_host_app_nv12_ingest(ms, cuda, tmp_path, flag):
    """--nv12: cameras deliver NV12, the host uploads half the bytes and cvtColor(YUV2BGR_NV12) (networking.cpp:45-47) runs on the device;
    --nv12-direct: no conversion pass, the warp samples the planes (ms_stitch_nv12).  Same panorama either way."""
    cfg = synth.CONFIGS["mini6"]
    info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                         "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 8, flag)
    assert info["nv12"] is True and info["nv12_direct"] is (flag == "--nv12-direct")
    got = np.fromfile(dump, np.uint8).reshape(cfg["out_h"], cfg["out_w"], 3)
    comp, _, _ = make_rig(ms, "mini6")
    out8 = torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda)
    frames = [ms.nv12_to_bgr(to_dev(synth.nv12_frame(cfg["w"], cfg["h"], i))) for i in range(cfg["n"])]
    comp.stitch([frames], out8u=[out8])
    torch.cuda.synchronize()
    assert np.array_equal(got, host(out8))
    comp.close()


def test_host_app_cpw_with_concurrent_recalibration(ms, cuda, tmp_path):
    """CPW on, meshes swapped by the recalibration thread while frames flow: must run to completion and produce a covered panorama."""
    cfg = synth.CONFIGS["mini6"]
    info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                         "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 400, "--cpw")
    assert info["frames"] == 400 and info["cpw"] is True
    got = np.fromfile(dump, np.uint8).reshape(cfg["out_h"], cfg["out_w"], 3)
    assert (got.max(axis=2) > 0).mean() > 0.25
    assert info["recalibrations"] >= 1


def test_host_app_consume_on_the_device(ms, cuda, tmp_path):
    """--consume WxH: the consumer thread runs consume()'s resize + black bars + BGR2YUV_I420 (timed.cpp:251-316) as one kernel on its own stream and hands the
    encoder frame to the host; the frame of the last panorama must equal the binding's ms_consume_i420 of the dumped panorama (FNV-1a checksum)."""
    cfg = synth.CONFIGS["cfg2"]
    info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                         "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 40, "--consume", "1024x512")
    pano = np.fromfile(dump, np.uint8).reshape(cfg["out_h"], cfg["out_w"], 3)
    frame, ih = ms.consume_i420(to_dev(pano), (1024, 512))
    assert info["consume_image_height"] == ih == 512
    h = 0xcbf29ce484222325 * 0          # the app starts its FNV-1a variant at 0
    for b in host(frame).reshape(-1).tolist():
        h = ((h ^ b) * 1099511628211) & 0xffffffffffffffff
    assert info["consume_checksum"] == "%016x" % h


@pytest.mark.parametrize("rig", ["mini6", "cfg2"])
def test_host_app_updates_masks_on_the_recalibration_thread(ms, cuda, tmp_path, rig):
    """timed.cpp:598-605 re-enabled: after every mesh swap the recalibration thread calls mb->update_mask(idx) -- enqueue-only here (update_mask_margin),
    while the stitcher thread keeps stitching.  The run must complete, and the tables the context ends up with must equal what a fresh context builds from
    the final meshes with the synchronous update_mask (the app stitches one more frame through both and compares: `update_mask_equals_sync_rebuild`)."""
    cfg = synth.CONFIGS[rig]
    info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                         "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 400 if rig == "mini6" else 300, "--update-mask", 12)
    assert info["cpw"] is True and info["update_mask_margin"] == 12 and info["recalibrations"] >= 2
    assert info["max_mesh_displacement_px"] <= 12
    assert info["update_mask_equals_sync_rebuild"] is True


def test_host_app_solves_meshes_while_stitching(ms, cuda, tmp_path):
    """--solve-mesh: the recalibration thread uploads the current frames, remaps them, runs the device feature front-end (overlap masks, ORB, Hamming 2-NN +
    ratio test, RANSAC homographies: msshim::featurefinder) and msshim::MeshWarper::calibrateMeshWarp (ms_create_mesh: triangle statistics + least-squares CG
    on its own stream), and swaps the meshes in -- while the stitcher thread runs.  Small rig: the views are too small for ORB's 31-px border, so the solve
    sees no matches (global + smoothness terms only); it must still do real work every round."""
    cfg = synth.CONFIGS["mini6"]
    for attempt in range(4):       # the scenario that exposed the stream-ordered allocator losing a solve's upload about once in 30 solves
        info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                             "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 6000, "--solve-mesh")
        assert info["frames"] == 6000 and info["cpw"] is True
        # every solve must do real work: a lost upload shows as a solve that "converges" in 0 iterations (and a black panorama)
        assert info["recalibrations"] >= 1 and info["mesh_solver_iterations"] > info["recalibrations"] * 300, info
        assert info["max_mesh_displacement_px"] < 20.0
        got = np.fromfile(dump, np.uint8).reshape(cfg["out_h"], cfg["out_w"], 3)
        assert (got.max(axis=2) > 0).mean() > 0.25


def test_host_app_recalibrates_from_real_features_at_full_size(ms, cuda, tmp_path):
    """SURVEY 8 f4 end to end, no external feature code: config-2 cameras looking at one synthetic scene (a texture fixed on the viewing sphere, odd and
    even cameras seeing it +-3 panorama pixels apart); every recalibration round finds ORB keypoints in the six warped 1080p views, matches neighbours,
    fits RANSAC homographies and lets the mesh optimiser absorb the disparity."""
    cfg = synth.CONFIGS["cfg2"]
    info, dump = run_app(tmp_path, "--frames", 1500, "--solve-mesh")
    r = info["recalibrations"]
    assert r >= 1 and info["orb_keypoints"] >= r * 6 * 1000, info                 # thousands of corners per view
    assert info["ratio_matches"] >= r * 500 and info["ransac_inliers"] >= r * 200 and info["ransac_inliers"] <= info["ratio_matches"], info
    assert info["mesh_solver_iterations"] > r * 300 and 2.0 < info["max_mesh_displacement_px"] < 32.0, info
    got = np.fromfile(dump, np.uint8).reshape(cfg["out_h"], cfg["out_w"], 3)
    assert (got.max(axis=2) > 0).mean() > 0.25


def test_host_app_reference_calibration_matches_the_same_steps_through_the_binding(ms, cuda, tmp_path):
    """stitch_app --reference-calib = msshim::stitch_calib (calibration.cpp:252-311 in product code): rig and scales from ms_calibrate_cameras, the shipped
    cylindrical warper, compose-scale ROIs -> num_bands rule, seam-scale gains + Voronoi seams from the first frames, and (COMPOSE_MEGAPIX 1.4 on 1080p:
    compose_scale 0.82) cuda::resize of every frame before the remap.  The same steps written out through the Python binding must give the same panorama."""
    n, w, h = 6, 1920, 1080
    info, dump = run_app(tmp_path, "--reference-calib", "--frames", 6)
    ow, oh = [int(v) for v in info["out"].split("x")]
    got = np.fromfile(dump, np.uint8).reshape(oh, ow, 3)
    rig = ms.calibrate_cameras(n, w, h, 90.0, 0.6, 0.01, 1.4)
    assert rig["resize_input"] and (rig["compose_width"], rig["compose_height"]) == (1578, 887)
    cw, ch = rig["compose_width"], rig["compose_height"]
    rois = [ms.warp_roi(ms.PROJ_CYLINDRICAL, rig["K_compose"][i], rig["R"][i], rig["compose_warp_scale"], cw, ch) for i in range(n)]
    pano = ms.result_roi(rois)
    bw, nb = ms.num_bands_rule(pano[2], pano[3], 5.0)
    assert info["bands"] == min(nb, int(np.ceil(np.log2(max(pano[2], pano[3])))))
    fit_w = (2 * max(abs(pano[0]), abs(pano[0] + pano[2])) + 1) & ~1
    fit_h = (2 * max(abs(pano[1]), abs(pano[1] + pano[3])) + 1) & ~1
    assert (ow, oh) == (fit_w, fit_h)
    comp = ms.Compositor(n, (cw, ch), ms.PROJ_CYLINDRICAL, rig["compose_warp_scale"], num_bands=nb, out_size=(ow, oh))
    for i in range(n):
        comp.set_camera(i, rig["K_compose"][i], rig["R"][i])
    comp.build_maps()
    full = [to_dev(synth.frame(w, h, i, 0, noise=False)) for i in range(n)]
    gains = comp.calibrate_seam(full, np.stack(rig["K_seam"]), rig["seam_scale"], rig["seam_warp_scale"], dilate=False, estimate_gains=True)
    assert all(0.5 < g < 2.0 for g in gains)
    comp.init_blender()
    small = [ms.resize_linear(f, fx=rig["compose_scale"], fy=rig["compose_scale"]) for f in full]
    assert small[0].shape[:2] == (ch, cw)
    out8 = torch.zeros((oh, ow, 3), dtype=torch.uint8, device=cuda)
    comp.stitch([small], out8u=[out8])
    torch.cuda.synchronize()
    assert np.array_equal(got, host(out8))
    assert got.any()
    comp.close()
