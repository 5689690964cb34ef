"""Calibration-time tables built on the device (ms_build_maps / ms_build_masks / ms_init_blender / ms_set_mesh)
against the oracle: geometry exact (SURVEY App. C known answers), maps within the reference's own 1e-4-class
tolerance (device sinf/cosf), masks/seams and weight pyramids exact GIVEN the same masks, CPW mesh -> map exact."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from helpers import host, make_rig, to_dev

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KATS = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "geometry_kats.json")))["cases"]}


def test_config2_geometry_known_answers(ms, cuda):
    comp, cfg, _ = make_rig(ms, "cfg2", mask_mode=0)
    case = KATS["config2_spherical"]
    pg = comp.pano_geom()
    assert list(pg.dst_roi_final.tuple()) == case["result_roi"] and list(pg.dst_roi.tuple()) == case["dst_roi"]
    assert (pg.canvas_x, pg.canvas_y) == (1, 646)
    for i, v in enumerate(case["views"]):
        g = comp.view_geom(i)
        assert list(g.roi.tuple()) == v["tl"] + v["size"]
        assert [g.top, g.left, g.bottom, g.right] == v["tlbr"]
        assert [g.x_tl, g.y_tl] == v["xy_tl"]
        assert [g.x_br - g.x_tl, g.y_br - g.y_tl] == v["padded"]
    comp.close()


@pytest.mark.parametrize("proj", ["spherical", "cylindrical"])
def test_maps_and_masks_vs_oracle(ms, cuda, oracle, proj):
    pid = {"spherical": ms.PROJ_SPHERICAL, "cylindrical": ms.PROJ_CYLINDRICAL}[proj]
    comp, cfg, _ = make_rig(ms, "mini6", mask_mode=1, projection=pid)
    sc = synth.warp_scale(cfg["out_w"])
    rois, masks_ref = [], []
    for i in range(cfg["n"]):
        K, R = synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i)
        r = comp.view_geom(i).roi.tuple()
        assert r == oracle.warp_roi(pid, K, R, sc, cfg["w"], cfg["h"])
        rois.append(r)
        rx, ry = oracle.build_warp_maps(pid, r[0], r[1], r[3], r[2], oracle.k_rinv_gpu(K, R), sc)
        gx, gy = [host(t) for t in comp.maps(i)]
        # reference bound: 1e-4 at scale 2 (ocl/test_warpers.cpp:100).  Here: 1e-3 px for coordinates that can touch the
        # source image, relative 1e-4 for the far-out ones (x/z with z -> 0 amplifies the sinf/cosf ulp differences)
        for g, r_, lim in ((gx, rx, cfg["w"]), (gy, ry, cfg["h"])):
            near = np.abs(r_) < 4 * lim
            assert np.abs(g - r_)[near].max() < 1e-3, float(np.abs(g - r_)[near].max())
            assert np.allclose(g[~near], r_[~near], rtol=1e-4, atol=0), "far coordinates"
        # valid mask = warp(255, NEAREST, CONSTANT) evaluated on the DEVICE maps -> exact
        masks_ref.append(oracle.remap_nearest_8uc1(np.full((cfg["h"], cfg["w"]), 255, np.uint8), gx, gy))
    oracle.voronoi_seams([r[:2] for r in rois], masks_ref)
    for i in range(cfg["n"]):
        assert np.array_equal(host(comp.mask(i)), masks_ref[i]), "seam mask of view %d" % i
    # masks partition the covered area: no pixel belongs to two views after the seams
    pg = comp.pano_geom()
    cover = np.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width), np.int32)
    for i, r in enumerate(rois):
        y0, x0 = r[1] - pg.dst_roi_final.y, r[0] - pg.dst_roi_final.x
        cover[y0:y0 + r[3], x0:x0 + r[2]] += masks_ref[i] != 0
    assert cover.max() == 1
    comp.close()


def test_weight_pyramids_and_result_mask_vs_oracle(ms, cuda, oracle):
    comp, cfg, _ = make_rig(ms, "mini4")
    rois = [comp.view_geom(i).roi.tuple() for i in range(cfg["n"])]
    b = oracle.Blender([r[:2] for r in rois], [r[2:] for r in rois], cfg["num_bands"])
    assert b.num_bands == comp.pano_geom().num_bands
    for i in range(cfg["n"]):
        b.init_view(i, host(comp.mask(i)))
        g, og = comp.view_geom(i), b.view_geom(i)
        assert (g.top, g.left, g.bottom, g.right, g.x_tl, g.y_tl, g.x_br, g.y_br) == \
               (og.top, og.left, og.bottom, og.right, og.x_tl, og.y_tl, og.x_br, og.y_br)
        for l in range(b.num_bands + 1):
            assert np.array_equal(host(comp.weight_level(i, l)), b.weight_level(i, l)), "weights view %d level %d" % (i, l)
    b.close()
    comp.close()


def test_user_masks_override(ms, cuda, oracle):
    """init_gpu receives the mask from the caller (blenders.cpp:344): arbitrary 8-bit masks, incl. grey values."""
    comp, cfg, gains = make_rig(ms, "mini4", mask_mode=0)
    rng = np.random.default_rng(5)
    rois = [comp.view_geom(i).roi.tuple() for i in range(cfg["n"])]
    masks = []
    for i, r in enumerate(rois):
        m = (rng.random((r[3], r[2])) < 0.7).astype(np.uint8) * 255
        m[::5, ::3] = 128                                   # non-binary weights
        masks.append(m)
        comp.set_mask(i, m)
    comp.init_blender()
    frames = [synth.frame(cfg["w"], cfg["h"], i, 2) for i in range(cfg["n"])]
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    b = oracle.Blender([r[:2] for r in rois], [r[2:] for r in rois], cfg["num_bands"])
    for i in range(cfg["n"]):
        b.init_view(i, masks[i])
    for i in range(cfg["n"]):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames[i], xm, ym, gains[i])
    ref, refmask = b.blend()
    assert np.array_equal(host(out16), ref) and np.array_equal(host(comp.result_mask()), refmask)
    b.close(); comp.close()


def test_blocked_camera_and_uncovered_regions(ms, cuda, oracle):
    """A view whose mask is entirely zero (a blocked camera) contributes nothing and owns no work; where no mask covers the panorama the
    weight sum is 1e-5, the result mask is 0 and the output is 0 (blenders.cpp:803-810).  Also a view reduced to a one-pixel-wide strip."""
    comp, cfg, gains = make_rig(ms, "mini6", mask_mode=0)
    rois = [comp.view_geom(i).roi.tuple() for i in range(cfg["n"])]
    masks = []
    for i, r in enumerate(rois):
        m = np.full((r[3], r[2]), 255, np.uint8)
        if i == 1:
            m[:] = 0                                   # blocked camera
        if i == 3:
            m[:] = 0; m[:, r[2] // 2] = 255             # a single column survives
        if i == 4:
            m[: r[3] // 2] = 0                          # upper half missing: leaves a hole in the panorama where nothing else covers
        masks.append(m)
        comp.set_mask(i, m)
    comp.init_blender()
    frames = [synth.frame(cfg["w"], cfg["h"], i, 5) for i in range(cfg["n"])]
    pg = comp.pano_geom()
    out16 = torch.full((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), 77, dtype=torch.int16, device=cuda)
    out8 = torch.full((cfg["out_h"], cfg["out_w"], 3), 9, dtype=torch.uint8, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out8u=[out8], out16s=[out16])
    torch.cuda.synchronize()
    b = oracle.Blender([r[:2] for r in rois], [r[2:] for r in rois], cfg["num_bands"])
    for i in range(cfg["n"]):
        b.init_view(i, masks[i])
    for i in range(cfg["n"]):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames[i], xm, ym, gains[i])
    ref, refmask = b.blend()
    assert np.array_equal(host(out16), ref) and np.array_equal(host(comp.result_mask()), refmask)
    assert (refmask == 0).any() and (ref[refmask == 0] == 0).all()
    b.close(); comp.close()


def test_num_bands_is_capped_by_the_panorama_size(ms, cuda, oracle):
    """MultiBandBlender::prepare caps num_bands at ceil(log2(max(dst width, height))) (blenders.cpp:241-245): 7 asked on a 64-wide pano
    resolves to 6, and the tiny rig still stitches bit-exactly (all levels fall to the generic kernels)."""
    n, w, h, out = 4, 40, 30, (64, 32)
    comp = ms.Compositor(n, (w, h), ms.PROJ_SPHERICAL, synth.warp_scale(out[0]), num_bands=7, out_size=out)
    gains = synth.gains(n)
    for i in range(n):
        comp.set_camera(i, *synth.camera(n, w, h, 110.0, i)); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    rois = [comp.view_geom(i).roi.tuple() for i in range(n)]
    g = oracle.blender_prepare(oracle.result_roi([r[:2] for r in rois], [r[2:] for r in rois]), 7)
    pg = comp.pano_geom()
    assert pg.num_bands == g.num_bands == 6 and pg.dst_roi.tuple() == g.dst_roi.tuple()
    frames = [synth.frame(w, h, i, 1) for i in range(n)]
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    b = oracle.Blender([r[:2] for r in rois], [r[2:] for r in rois], 7)
    for i in range(n):
        b.init_view(i, host(comp.mask(i)))
    for i in range(n):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames[i], xm, ym, gains[i])
    ref, refmask = b.blend()
    assert np.array_equal(host(out16), ref) and np.array_equal(host(comp.result_mask()), refmask)
    b.close(); comp.close()


@pytest.mark.parametrize("nm", [(10, 10), (40, 40), (7, 13)])
def test_mesh_to_map_vs_oracle(ms, cuda, oracle, nm):
    """convertMeshesToMap (APP/meshwarper.cpp:823-886): custom_resize up, scatter-average at half resolution
    (holes -> NaN), custom_resize up again.  Device atomics vs the reference's sequential loop: the partial
    sums are integers < 2^24, so the result is order-independent and must match bit for bit."""
    comp, cfg, _ = make_rig(ms, "mini6", enable_cpw=True)
    for i in (0, 3):
        r = comp.view_geom(i).roi
        mx, my = synth.mesh(r.width, r.height, nm[0], nm[1], phase=0.5 + i, amp=6.0)
        comp.set_mesh(i, mx, my)
        gx, gy = [host(t) for t in comp.mesh_maps(i)]
        rx, ry = oracle.convert_mesh_to_map(mx, my, r.width, r.height)
        for g, ref in ((gx, rx), (gy, ry)):
            assert np.array_equal(np.isnan(g), np.isnan(ref))
            assert np.array_equal(g[~np.isnan(g)], ref[~np.isnan(ref)])
    comp.close()


@pytest.mark.parametrize("rig,nm", [("mini6", (10, 10)), ("mini4", (7, 12)), ("cfg2", (40, 40))])
def test_set_meshes_all_views_in_two_launches(ms, cuda, oracle, rig, nm):
    """ms_set_meshes = convertMeshesToMap for every image at once (meshwarper.cpp:823-886), blockIdx.z = view: bit-identical to the per-view ms_set_mesh and to
    the oracle, over repeated calls (its accumulators ping-pong and clear each other), interleaved with per-view updates, with the measured displacements and a
    stitched frame equal too."""
    comp, cfg, _ = make_rig(ms, rig, enable_cpw=True)
    ref, _, _ = make_rig(ms, rig, enable_cpw=True)
    n = cfg["n"]
    rois = [comp.view_geom(i).roi for i in range(n)]
    for rnd, amp in enumerate((6.0, 11.0, 3.0, 25.0)):
        meshes = [synth.mesh(rois[i].width, rois[i].height, nm[0], nm[1], phase=0.3 * i + 0.9 * rnd, amp=amp) for i in range(n)]
        comp.set_meshes(meshes)
        if rnd == 2:      # a per-view update in between must not disturb the batched scratch (and vice versa)
            comp.set_mesh(1, *meshes[1])
        for i in range(n):
            ref.set_mesh(i, *meshes[i])
        for i in range(n):
            got = [host(t) for t in comp.mesh_maps(i)]
            want = [host(t) for t in ref.mesh_maps(i)]
            assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True), (rnd, i)
            assert comp.mesh_displacement(i) == ref.mesh_displacement(i)
            if rig != "cfg2" or i in (0, 3):
                ox, oy = oracle.convert_mesh_to_map(meshes[i][0], meshes[i][1], rois[i].width, rois[i].height)
                assert np.array_equal(got[0], ox, equal_nan=True) and np.array_equal(got[1], oy, equal_nan=True)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 1)) for i in range(n)]]
    pg = comp.pano_geom()
    a = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    b = torch.zeros_like(a)
    comp.stitch(frames, out16s=[a]); ref.stitch(frames, out16s=[b]); torch.cuda.synchronize()
    assert torch.equal(a, b)
    with pytest.raises(ms.MsError):
        plain, _, _ = make_rig(ms, "mini4")
        plain.set_meshes([synth.mesh(50, 40, 5, 5)] * 4)          # a context without enable_cpw
    comp.close(); ref.close()


@pytest.mark.parametrize("batched", [True, False])
def test_first_mesh_update_is_ordered_on_its_own_stream(ms, cuda, batched):
    """The first mesh update of a context allocates and clears its scratch (accumulators, displacement words).  That has to happen ON THE UPDATE'S STREAM: a plain
    hipMemset runs on the NULL stream, asynchronously to the host, and a non-blocking stream does not wait for it -- with the NULL stream busy (here: a queue of large
    fills; in stitch_dist: the other ranks' host-transport copies) the scatter kernels ran on uncleared accumulators and the clear landed afterwards: 5 of 30
    four-rank runs delivered wrong frames for the batch after the first recalibration (tools/dist_repeat.py)."""
    quiet, cfg, _ = make_rig(ms, "mini6", enable_cpw=True)
    n = cfg["n"]
    rois = [quiet.view_geom(i).roi for i in range(n)]
    meshes = [synth.mesh(rois[i].width, rois[i].height, 9, 11, phase=0.4 * i, amp=7.0) for i in range(n)]
    for i in range(n):
        quiet.set_mesh(i, *meshes[i])
    torch.cuda.synchronize()
    want = [[host(t).copy() for t in quiet.mesh_maps(i)] for i in range(n)]
    busy = torch.zeros(1 << 29, dtype=torch.uint8, device=cuda)
    for trial in range(3):
        comp, _, _ = make_rig(ms, "mini6", enable_cpw=True)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        for _ in range(40):
            busy.add_(1)                                  # tens of milliseconds of work queued on the NULL stream
        with torch.cuda.stream(side):
            if batched:
                comp.set_meshes(meshes)
            else:
                for i in range(n):
                    comp.set_mesh(i, *meshes[i])
        side.synchronize()
        torch.cuda.synchronize()
        for i in range(n):
            got = [host(t) for t in comp.mesh_maps(i)]
            assert np.array_equal(got[0], want[i][0], equal_nan=True) and np.array_equal(got[1], want[i][1], equal_nan=True), (trial, i)
            assert comp.mesh_displacement(i) == quiet.mesh_displacement(i)
        comp.close()
    quiet.close()


def test_mesh_interpolation_equals_host_lerp(ms, cuda):
    """ms_set_mesh_interp = interpolateMesh (meshwarper.cpp:337-354: start + (end - start) * progress in fp32) + ms_set_mesh."""
    comp, cfg, _ = make_rig(ms, "mini6", enable_cpw=True)
    r = comp.view_geom(2).roi
    a = synth.mesh(r.width, r.height, 9, 7, phase=0.2, amp=5.0)
    b = synth.mesh(r.width, r.height, 9, 7, phase=1.1, amp=9.0)
    p = np.float32(0.37)
    comp.set_mesh_interp(2, a, b, float(p))
    got = [host(m).copy() for m in comp.mesh_maps(2)]
    lerp = [(s + (e - s) * p).astype(np.float32) for s, e in zip(a, b)]
    comp.set_mesh(2, *lerp)
    want = [host(m) for m in comp.mesh_maps(2)]
    assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True)
    comp.close()


def test_mesh_double_buffering(ms, cuda):
    """ms_set_mesh from the recalibration side takes effect at the next ms_stitch and never tears a frame."""
    comp, cfg, _ = make_rig(ms, "mini6", enable_cpw=True)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 0)) for i in range(cfg["n"])]]
    pg = comp.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)

    def stitch_with(phase):
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            comp.set_mesh(i, *synth.mesh(r.width, r.height, 10, 10, phase=phase, amp=5.0))
        o = torch.zeros(shape, dtype=torch.int16, device=cuda)
        comp.stitch(frames, out16s=[o])
        return o
    a = stitch_with(0.0)
    b = stitch_with(1.0)
    a2 = stitch_with(0.0)
    torch.cuda.synchronize()
    assert torch.equal(a, a2) and not torch.equal(a, b)
    comp.close()


def test_set_meshes_concurrent_with_stitch_never_tears_a_frame(ms, cuda):
    """The documented recalibration-thread use (timed.cpp:414-463; shim convertMeshesToMap): ms_set_meshes on its own thread and stream while another thread
    stitches.  Every stitched frame must be the frame of ONE complete mesh set -- set A or set B for every view -- never a mix of views and never a buffer the
    expansion kernels are still writing.  ADVICE r05: the batched call published its views one by one and recorded the chain's event only behind the last one, so
    a stitch that took the mutex in between waited for the PREVIOUS update's record and read the new buffer half written.  cfg2-sized views make the expansion
    long enough (tens of microseconds per launch) for a stitch to land inside it."""
    import threading
    comp, cfg, _ = make_rig(ms, "cfg2", enable_cpw=True)
    n = cfg["n"]
    rois = [comp.view_geom(i).roi for i in range(n)]
    sets = [[synth.mesh(rois[i].width, rois[i].height, 40, 40, phase=0.3 * i + 1.7 * k, amp=9.0) for i in range(n)] for k in range(2)]
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 0)) for i in range(n)]]
    pg = comp.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    want = []
    for k in range(2):
        comp.set_meshes(sets[k])
        o = torch.zeros(shape, dtype=torch.int16, device=cuda)
        comp.stitch(frames, out16s=[o]); torch.cuda.synchronize()
        want.append(o)
    assert not torch.equal(want[0], want[1])
    n_stitch, stop, errs = 150, threading.Event(), []

    def recalibrate():
        try:
            torch.cuda.set_device(cuda)
            side = torch.cuda.Stream()
            k = 0
            with torch.cuda.stream(side):
                while not stop.is_set():
                    comp.set_meshes(sets[k & 1]); k += 1
            side.synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = threading.Thread(target=recalibrate)
    th.start()
    outs = [torch.zeros(shape, dtype=torch.int16, device=cuda) for _ in range(n_stitch)]
    main = torch.cuda.Stream()
    try:
        with torch.cuda.stream(main):
            for j in range(n_stitch):
                comp.stitch(frames, out16s=[outs[j]])
        main.synchronize()
    finally:
        stop.set(); th.join()
    torch.cuda.synchronize()
    assert not errs, errs
    seen = [0, 0]
    for j, o in enumerate(outs):
        a, b = torch.equal(o, want[0]), torch.equal(o, want[1])
        assert a or b, "frame %d is neither the frame of mesh set A nor of mesh set B: a torn or half-written mesh" % j
        seen[0 if a else 1] += 1
    comp.close()


@pytest.mark.parametrize("dilate", [False, True])
def test_seam_scale_calibration_pipeline(ms, cuda, oracle, dilate):
    """stitch_calib's own mask/gain pipeline (APP/calibration.cpp:92-135, 224-237) through ms_calibrate_seam vs the same sequence
    composed from oracle primitives: resize -> seam-scale warp (image REFLECT/LINEAR, mask NEAREST) -> GainCompensator::feed ->
    Voronoi -> [dilate] -> resize up (INTER_LINEAR, grey values!) -> AND with the compose-scale valid mask."""
    n, w, h = 6, 480, 270
    rig = synth.reference_rig(n, w, h)                    # WORK 0.6 MP, SEAM 0.01 MP, compose at full size
    proj = ms.PROJ_CYLINDRICAL                           # what the app ships (calibration.cpp:100)
    comp = ms.Compositor(n, (w, h), proj, rig["compose_warp_scale"], num_bands=4, out_size=(0, 0))
    for i in range(n):
        comp.set_camera(i, rig["K_compose"][i], rig["R"][i])
    comp.build_maps()
    frames = [synth.frame(w, h, i, 0, noise=False) for i in range(n)]
    frames = [np.clip(f.astype(np.float32) * (0.9 + 0.04 * i), 0, 255).astype(np.uint8) for i, f in enumerate(frames)]   # exposure differences
    gains = comp.calibrate_seam([to_dev(f) for f in frames], rig["K_seam"], rig["seam_scale"], rig["seam_warp_scale"], dilate=dilate)

    # ---- the same pipeline from oracle primitives ----
    ss = rig["seam_scale"]
    rois, imgs_w, masks_w = [], [], []
    for i in range(n):
        seam = oracle.resize_linear_8u(frames[i], fx=ss, fy=ss)
        hs, ws = seam.shape[:2]
        r = oracle.warp_roi(proj, rig["K_seam"][i], rig["R"][i], rig["seam_warp_scale"], ws, hs)
        mx, my = oracle.build_warp_maps(proj, r[0], r[1], r[3], r[2], oracle.k_rinv_gpu(rig["K_seam"][i], rig["R"][i]), rig["seam_warp_scale"])
        rois.append(r)
        imgs_w.append(oracle.remap_linear_reflect_8uc3(seam, mx, my))
        masks_w.append(oracle.remap_nearest_8uc1(np.full((hs, ws), 255, np.uint8), mx, my))
    ref_gains = oracle.gain_compensator([r[:2] for r in rois], imgs_w, masks_w)
    oracle.voronoi_seams([r[:2] for r in rois], masks_w)
    assert np.allclose(gains, ref_gains, rtol=2e-3), (gains, ref_gains)        # device vs glibc sinf/cosf in the seam maps -> <= 1 LSB image diffs
    assert 0.7 < min(gains) and max(gains) < 1.3 and max(gains) - min(gains) > 0.02   # it does compensate the injected exposure ramp
    diff_px = 0
    for i in range(n):
        g = comp.view_geom(i).roi
        m = masks_w[i]
        if dilate:
            m = oracle.dilate3x3_8u(m)
        big = oracle.resize_linear_8u(m, dsize=(g.width, g.height))
        gx, gy = [host(t) for t in comp.maps(i)]
        valid = oracle.remap_nearest_8uc1(np.full((h, w), 255, np.uint8), gx, gy)
        ref = big & valid
        got = host(comp.mask(i))
        diff_px += int((got != ref).sum())
        assert ((got != 0) & (got != 255)).any(), "bilinear upsizing leaves grey seam pixels (non-binary weights)"
    assert diff_px <= 0.002 * sum(m.size for m in masks_w) * 16, diff_px       # a seam-scale border flip would show as a small block
    # the blender accepts these grey masks: end-to-end vs the oracle fed with the product's masks and gains
    comp.init_blender()
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    rois_c = [comp.view_geom(i).roi.tuple() for i in range(n)]
    b = oracle.Blender([r[:2] for r in rois_c], [r[2:] for r in rois_c], 4)
    for i in range(n):
        b.init_view(i, host(comp.mask(i)))
    for i in range(n):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames[i], xm, ym, gains[i])
    ref16, refmask = b.blend()
    assert np.array_equal(host(out16), ref16) and np.array_equal(host(comp.result_mask()), refmask)
    b.close(); comp.close()


@pytest.mark.parametrize("kind", ["wild", "folded", "collapsed", "shifted_out", "nan_vertices", "huge"])
def test_mesh_to_map_adversarial_meshes(ms, cuda, oracle, kind):
    """The two-launch expansion (LDS-aggregated integer scatter with a global fallback, fused mean / resize / displacement) on meshes that
    leave its fast paths: displacements far beyond the LDS window, folds, every pixel landing in one cell (counts far above any window,
    sums beyond 2^24 where the reference's float sums stop being exact), pixels scattered out of the view, NaN / Inf vertices.  Maps must
    equal the oracle's convertMeshesToMap wherever the reference itself is order-independent, the reported displacement must equal the maps'."""
    comp, cfg, _ = make_rig(ms, "cfg2" if kind == "huge" else "mini6", enable_cpw=True)
    rng = np.random.default_rng(len(kind))
    views = (3,) if kind == "huge" else (0, 3, 5)
    for i in views:
        r = comp.view_geom(i).roi
        N, M = (9, 11) if kind != "huge" else (40, 40)
        mx, my = synth.mesh(r.width, r.height, N, M, phase=0.3 * i, amp=3.0)
        if kind == "wild":
            mx = mx + rng.uniform(-90, 90, mx.shape).astype(np.float32); my = my + rng.uniform(-60, 60, my.shape).astype(np.float32)
        elif kind == "folded":
            mx = mx[:, ::-1].copy(); my = my[::-1].copy()                    # mirrored: every pixel lands far from where it starts
        elif kind == "collapsed":
            mx[:] = np.float32(r.width / 2 + 0.5); my[:] = np.float32(r.height / 2 + 0.5)
        elif kind == "shifted_out":
            mx = mx + np.float32(r.width * 0.8); my = my - np.float32(r.height * 0.6)
        elif kind == "nan_vertices":
            mx[2, 3] = np.nan; my[4, 5] = np.inf; mx[6, 7] = -np.inf; mx[1, 1] = 1e30
        elif kind == "huge":
            mx = mx + (40 * np.sin(np.arange(M) / 3.0)[None, :]).astype(np.float32)
        mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
        comp.set_mesh(i, mx, my)
        gx, gy = [host(t) for t in comp.mesh_maps(i)]
        rx, ry = oracle.convert_mesh_to_map(mx, my, r.width, r.height)
        if kind == "collapsed":
            # one cell receives every pixel: the reference's float sum is no longer exact there (order-dependent), ours is the exact sum;
            # everything else is a hole on both sides and the one mean agrees to float accuracy
            assert np.array_equal(np.isnan(gx), np.isnan(rx)) and np.array_equal(np.isnan(gy), np.isnan(ry))
            ok = ~np.isnan(rx)
            assert np.allclose(gx[ok], rx[ok], rtol=1e-3) and np.allclose(gy[~np.isnan(ry)], ry[~np.isnan(ry)], rtol=1e-3)
        else:
            assert np.array_equal(gx, rx, equal_nan=True) and np.array_equal(gy, ry, equal_nan=True)
        yy, xx = np.mgrid[0:r.height, 0:r.width].astype(np.float32)
        with np.errstate(invalid="ignore"):
            d = np.fmax(np.abs(gx - xx), np.abs(gy - yy))
        want = np.float32(np.nanmax(d)) if np.isfinite(d).any() else np.float32(0)
        assert comp.mesh_displacement(i) == want
        # a second, ordinary update right after must not see anything the adversarial one left in the accumulators
        sx, sy = synth.mesh(r.width, r.height, 10, 10, phase=0.9, amp=4.0)
        comp.set_mesh(i, sx, sy)
        g2 = [host(t) for t in comp.mesh_maps(i)]
        r2 = oracle.convert_mesh_to_map(sx, sy, r.width, r.height)
        assert np.array_equal(g2[0], r2[0], equal_nan=True) and np.array_equal(g2[1], r2[1], equal_nan=True)
    comp.close()


def test_tables_blob_round_trip_rebuilds_identical_tables_and_frames(ms, cuda):
    """ms_save_tables / ms_load_tables (the reference re-runs stitch_calib at every start, timed.cpp:553): a context rebuilt from the blob has the same masks,
    weight pyramids, result mask and work lists (band cell classes), and stitches the same frame bit for bit; corrupt blobs are refused."""
    cfg = synth.CONFIGS["mini6"]
    comp = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"], out_size=(cfg["out_w"], cfg["out_h"]))
    gains = synth.gains(cfg["n"])
    for i in range(cfg["n"]):
        comp.set_camera(i, *synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i))
        comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    blob = comp.save_tables()
    twin = ms.Compositor.from_tables(blob)
    assert twin.n == cfg["n"]
    for i in range(cfg["n"]):
        assert torch.equal(comp.mask(i), twin.mask(i))
        assert comp.view_geom(i).roi.tuple() == twin.view_geom(i).roi.tuple()
        for l in range(cfg["num_bands"] + 1):
            assert torch.equal(comp.weight_level(i, l), twin.weight_level(i, l))
    assert torch.equal(comp.result_mask(), twin.result_mask())
    for l in range(cfg["num_bands"]):
        assert comp.band_cells(l) == twin.band_cells(l)
    frames = [[torch.from_numpy(synth.frame(cfg["w"], cfg["h"], i, 3)).cuda() for i in range(cfg["n"])]]
    pg = comp.pano_geom()
    outs = []
    for c in (comp, twin):
        o16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device="cuda")
        o8 = torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda")
        c.stitch(frames, out8u=[o8], out16s=[o16])
        torch.cuda.synchronize()
        outs.append((o16, o8))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    bad = bytearray(blob); bad[len(bad) // 2] ^= 0x40
    with pytest.raises(ms.MsError):
        ms.Compositor.from_tables(bytes(bad))
    with pytest.raises(ms.MsError):
        ms.Compositor.from_tables(blob[:-5])
    # the checksum covers the HEADER too: a flipped bit inside the embedded ms_config -- here num_bands 3 -> 2 and out_width 640 -> 641, both of which
    # pass ms_create's range checks and would silently build other tables -- is refused
    import ctypes
    cfg_off = 48      # magic[8], header_bytes, config_bytes, n_views, blender_kind, feather_sharpness, reserved, total_bytes (8), checksum (8); checked by the asserts below
    assert int.from_bytes(blob[12:16], "little") == ctypes.sizeof(ms.Config) and int.from_bytes(blob[cfg_off:cfg_off + 4], "little") == ctypes.sizeof(ms.Config)
    raw = ms.Config.from_buffer_copy(blob[cfg_off:cfg_off + ctypes.sizeof(ms.Config)])
    assert raw.num_views == cfg["n"] and raw.num_bands == cfg["num_bands"] and raw.out_width == cfg["out_w"]
    for field, flip in (("num_bands", 1), ("out_width", 1), ("projection", 1)):
        off = cfg_off + getattr(ms.Config, field).offset
        bad = bytearray(blob); bad[off] ^= flip
        with pytest.raises(ms.MsError, match="checksum"):
            ms.Compositor.from_tables(bytes(bad))
    # a column / view shard does not save (its blob would hand every loader that shard's window)
    shard = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"], out_size=(cfg["out_w"], cfg["out_h"]),
                          col_shards=2, col_shard_index=1)
    for i in range(cfg["n"]):
        shard.set_camera(i, *synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i))
    shard.build_maps(); shard.build_masks(1); shard.init_blender()
    with pytest.raises(ms.MsError, match="shard"):
        shard.save_tables()
    shard.close()
