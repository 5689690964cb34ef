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


def test_host_app_nv12_ingest(ms, cuda, tmp_path):
    """--nv12: cameras deliver NV12, the host uploads half the bytes and cvtColor(YUV2BGR_NV12) (networking.cpp:45-47) runs on the device."""
    cfg = synth.CONFIGS["mini6"]
    info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                         "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 8, "--nv12")
    assert info["nv12"] is True
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


def test_host_app_solves_meshes_while_stitching(ms, cuda, tmp_path):
    """--solve-mesh: the recalibration thread uploads the current frames, remaps them, runs msshim::MeshWarper::calibrateMeshWarp
    (ms_create_mesh: triangle statistics + least-squares CG on its own stream) and swaps the meshes in, while the stitcher thread runs."""
    cfg = synth.CONFIGS["mini6"]
    for attempt in range(4):       # the scenario that exposed the stream-ordered allocator losing a solve's upload about once in 30 solves
        info, dump = run_app(tmp_path, "--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                             "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"], "--frames", 6000, "--solve-mesh")
        assert info["frames"] == 6000 and info["cpw"] is True
        # every solve must do real work: a lost upload shows as a solve that "converges" in 0 iterations (and a black panorama)
        assert info["recalibrations"] >= 1 and info["mesh_solver_iterations"] > info["recalibrations"] * 300, info
        assert 1.0 < info["max_mesh_displacement_px"] < 20.0
        got = np.fromfile(dump, np.uint8).reshape(cfg["out_h"], cfg["out_w"], 3)
        assert (got.max(axis=2) > 0).mean() > 0.25
