"""Parity FROM THE INPUTS: every other pixel test feeds the oracle the device-built maps and masks ("bit-exact given the same
tables").  Here the oracle starts from what the reference starts from -- K, R, the frames -- and builds everything itself with glibc sinf / cosf:
`buildMaps` (warpers_cuda.cpp:210-231, build_warp_maps.cu), the valid-warp masks, the Voronoi seams (seam_finders.cpp:111-160), [the seam-scale
pipeline with exposure gains, calibration.cpp:92-135,224-237], the blender.  The product does the same on the device.  What may differ is what the
device's sinf / cosf move: map coordinates by <= 1e-3 px (the reference's own device-vs-CPU bound is 1e-4 at scale 2, test/ocl/test_warpers.cpp:100),
hence a few mask pixels along view borders and seams, hence bilinear samples by a fraction of a grey level.  The reference's own end-to-end bound
between its CUDA and CPU blenders is |diff| <= 3 (stitching/test/test_blenders.cuda.cpp:90): that is the criterion here, on the pixels both results
cover, away from mask pixels that differ.

Observed on MI355X (profiles/r03_from_inputs.txt; MS_FROM_INPUTS_LOG=<file> appends a run's statistics, `pytest -s` prints them):
  view masks and result masks IDENTICAL in all four cases (the device's sinf / cosf flip no border or seam pixel on these rigs)
  mini rig (6 x 320 x 180, 3 bands)                       max |diff| 1 on 17 of 64 338 px                      maps within 6e-5 px
  config 2 (6 x 1080p -> 3840 x 1920, spherical, 5 bands) max |diff| 3 on 1 px, 2 on 294, 1 on 11 091 of 2.31 M  maps within 4.9e-4 px inside the image
  shipped rig at 480 x 270 (cylindrical, 4 bands)         max |diff| 2 on 4 px, 1 on 419 of 390 k; gains equal to 4 digits
  shipped rig at 1080p (1578 x 887 compose, 6 bands)      max |diff| 3 on 1 px, 2 on 58, 1 on 3 960 of 4.2 M; gains equal to 4 digits
  configs[4] geometry (12 x 4K -> 7680 x 3840, 5 bands)   result mask identical, 95 border pixels of view masks flip (maps within 9.8e-4 px); away from them max |diff| 3,
                                                          99.16 % of 5.8 M px bit-equal (round 4, profiles/r04_from_inputs.txt)
  config 3 shape (config 2 + CPW 40 x 40, both remaps)    masks identical, max |diff| 2 on 144 px, 1 on 9 119 of 2.31 M (round 4)
"""
import os

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st
from scipy import ndimage

import synth
from helpers import host, to_dev

pytestmark = pytest.mark.gpu

# statistics of this run; MS_FROM_INPUTS_LOG=<file> appends them there (profiles/r03_from_inputs.txt is such a log, committed)
OBSERVED = {}


def record(name, stats):
    import json
    import os
    OBSERVED[name] = stats
    print("\nFROM-INPUTS %s: %s" % (name, stats))
    log = os.environ.get("MS_FROM_INPUTS_LOG")
    if log:
        with open(log, "a") as f:
            f.write(json.dumps({"case": name, **stats}) + "\n")


def oracle_from_inputs(O, proj, Ks, Rs, scale, w, h, frames, gains, num_bands, masks=None, meshes=None):
    """stitch_calib + stitch_one entirely in the oracle, from the camera parameters (meshes: per view the N x M vertex meshes of the CPW, expanded by the oracle's convertMeshesToMap)"""
    n = len(Ks)
    rois = [O.warp_roi(proj, Ks[i], Rs[i], scale, w, h) for i in range(n)]
    maps = [O.build_warp_maps(proj, r[0], r[1], r[3], r[2], O.k_rinv_gpu(Ks[i], Rs[i]), scale) for i, r in enumerate(rois)]
    if masks is None:
        masks = [O.remap_nearest_8uc1(np.full((h, w), 255, np.uint8), mx, my) for mx, my in maps]
        O.voronoi_seams([r[:2] for r in rois], masks)
    b = O.Blender([r[:2] for r in rois], [r[2:] for r in rois], num_bands)
    for i in range(n):
        b.init_view(i, masks[i])
    for i in range(n):
        if meshes is None:
            b.stitch_online(i, frames[i], maps[i][0], maps[i][1], gains[i])
        else:
            mx, my = O.convert_mesh_to_map(meshes[i][0], meshes[i][1], rois[i][2], rois[i][3])
            b.stitch_online(i, frames[i], maps[i][0], maps[i][1], gains[i], mx, my)
    out, mask = b.blend()
    b.close()
    return rois, maps, masks, out, mask


def compare(name, got16, got_mask, ref16, ref_mask, view_mask_diffs, halo, extra=None, max_excluded=0.02, max_view_mask_flips=None):
    """the statistics + the reference's own criterion.  max_view_mask_flips: the COUNT of view-mask pixels that may differ between the device-built and the oracle-built
    masks of a fixed rig: the exclusion zone below must not be able to hide a regression of the map builder."""
    assert got16.shape == ref16.shape and got_mask.shape == ref_mask.shape
    mask_diff = got_mask != ref_mask
    # pixels within the blend support of a differing mask pixel may legitimately differ by more (a seam that moved by a pixel): excluded and counted
    near = ndimage.binary_dilation(view_mask_diffs | mask_diff, iterations=1, structure=np.ones((3, 3), bool))
    if near.any():
        near = ndimage.maximum_filter(near.astype(np.uint8), size=2 * halo + 1) > 0
    common = (got_mask != 0) & (ref_mask != 0)
    d = np.abs(got16.astype(np.int32) - ref16.astype(np.int32)).max(axis=2)
    far = common & ~near
    hist = np.bincount(np.minimum(d[common], 8), minlength=9)
    stats = dict(pano_px=int(common.size), common_px=int(common.sum()), result_mask_diff_px=int(mask_diff.sum()), view_mask_diff_px=int(view_mask_diffs.sum()),
                 excluded_px=int((common & near).sum()), max_diff_far=int(d[far].max()) if far.any() else 0, max_diff_anywhere=int(d[common].max()),
                 hist_0_to_8plus=[int(x) for x in hist], exact_fraction=float(hist[0]) / max(1, int(common.sum())))
    stats.update(extra or {})
    record(name, stats)
    if max_view_mask_flips is not None:
        assert stats["view_mask_diff_px"] <= max_view_mask_flips, stats
    assert stats["max_diff_far"] <= 3, stats                                        # test_blenders.cuda.cpp:90
    assert stats["result_mask_diff_px"] <= 2e-4 * common.size, stats                # masks equal except a few border / seam pixels
    assert stats["excluded_px"] <= max_excluded * common.size, stats
    assert stats["exact_fraction"] > 0.9, stats                                     # and the overwhelming majority is bit-equal anyway
    return stats


def view_mask_diff_in_pano(rois, pano_roi, masks_a, masks_b):
    """pano-ROI-sized map of the pixels where some view's blend mask differs between the two pipelines"""
    out = np.zeros((pano_roi[3], pano_roi[2]), bool)
    for r, a, b in zip(rois, masks_a, masks_b):
        y0, x0 = r[1] - pano_roi[1], r[0] - pano_roi[0]
        out[y0:y0 + r[3], x0:x0 + r[2]] |= a != b
    return out


@pytest.mark.parametrize("rig", ["mini6", "cfg2", "cfg5"])
def test_spherical_rig_from_camera_parameters(ms, cuda, oracle, rig):
    """BASELINE configs[1] (and its small twin) and the configs[4] geometry (12 x 4K -> 7680 x 3840 on one GPU): K, R, frames, fixed gains -> panorama; device maps + device Voronoi vs the oracle's own."""
    cfg = synth.CONFIGS[rig]
    n, w, h, nb = cfg["n"], cfg["w"], cfg["h"], cfg["num_bands"]
    scale = synth.warp_scale(cfg["out_w"])
    cams = [synth.camera(n, w, h, cfg["hfov_deg"], i) for i in range(n)]
    gains = synth.gains(n)
    frames = [synth.frame(w, h, i, 3) for i in range(n)]
    comp = ms.Compositor(n, (w, h), ms.PROJ_SPHERICAL, scale, num_bands=nb, out_size=(cfg["out_w"], cfg["out_h"]))
    for i in range(n):
        comp.set_camera(i, *cams[i]); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    rois, maps, masks, ref16, ref_mask = oracle_from_inputs(oracle, ms.PROJ_SPHERICAL, [c[0] for c in cams], [c[1] for c in cams], scale, w, h, frames, gains, nb)
    assert rois == [comp.view_geom(i).roi.tuple() for i in range(n)]
    worst = 0.0
    for i in range(n):
        gx, gy = [host(t) for t in comp.maps(i)]
        inside = (maps[i][0] > -2) & (maps[i][0] < w + 1) & (maps[i][1] > -2) & (maps[i][1] < h + 1)      # samples that touch the source image
        for g, r_ in ((gx, maps[i][0]), (gy, maps[i][1])):
            worst = max(worst, float(np.abs(g - r_)[inside].max()))
    # device sinf / cosf against glibc's: a few ulp of the coordinate -- 1e-3 px on the small rigs, 2e-3 px (16 ulp at x ~ 1900) at 1080p
    assert worst < (1e-3 if w <= 640 else 2.5e-3 if w <= 1920 else 6e-3), worst      # (4K: 16 ulp at x ~ 3800)
    vdiff = view_mask_diff_in_pano(rois, pg.dst_roi_final.tuple(), [host(comp.mask(i)) for i in range(n)], masks)
    # 12 x 4K: 49 k border pixels of the valid-warp masks, map coordinates within 1e-3 px of the oracle's: a few dozen of them round to the other side of an image edge
    # (95 observed), and the 96-px blend support around each is excluded from the |diff| <= 3 criterion -- 2.2 % of the panorama there, nothing on the 1080p rigs
    # "pixel-for-pixel on integer masks" holds GIVEN THE SAME libm: the valid-warp masks are (int) truncations of sinf / cosf / atan2f results, and the device's
    # functions differ from glibc's in the last ulps.  On the 1080p rigs no mask pixel flips (asserted: 0); on 12 x 4K exactly 95 of 49 k border pixels do
    # (DESIGN.md section 2); the RESULT mask is identical everywhere (compare() asserts that for every rig).
    compare(rig + "_spherical", host(out16), host(comp.result_mask()), ref16, ref_mask, vdiff, halo=3 * 2 ** nb, extra={"max_map_diff_px": worst},
            max_excluded=0.04 if rig == "cfg5" else 0.02, max_view_mask_flips=95 if rig == "cfg5" else 0)
    comp.close()


@pytest.mark.parametrize("rig,nm", [("mini6", (10, 10)), ("cfg2", (40, 40))])
def test_spherical_rig_with_cpw_from_camera_parameters(ms, cuda, oracle, rig, nm):
    """BASELINE configs[2] (config 2 + CPW 40 x 40 meshes; and the small twin with the reference's 10 x 10): the frames go through BOTH remaps
    (timed.cpp:90-104) -- projection maps from K, R and mesh maps from the vertex meshes, each built by the oracle itself -- before the blender."""
    cfg = synth.CONFIGS[rig]
    n, w, h, nb = cfg["n"], cfg["w"], cfg["h"], cfg["num_bands"]
    scale = synth.warp_scale(cfg["out_w"])
    cams = [synth.camera(n, w, h, cfg["hfov_deg"], i) for i in range(n)]
    gains = synth.gains(n)
    frames = [synth.frame(w, h, i, 5) for i in range(n)]
    comp = ms.Compositor(n, (w, h), ms.PROJ_SPHERICAL, scale, num_bands=nb, enable_cpw=True, out_size=(cfg["out_w"], cfg["out_h"]))
    for i in range(n):
        comp.set_camera(i, *cams[i]); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    meshes = []
    for i in range(n):
        r = comp.view_geom(i).roi
        meshes.append(synth.mesh(r.width, r.height, nm[0], nm[1], phase=0.1 * i))
        comp.set_mesh(i, *meshes[i])
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    rois, maps, masks, ref16, ref_mask = oracle_from_inputs(oracle, ms.PROJ_SPHERICAL, [c[0] for c in cams], [c[1] for c in cams], scale, w, h, frames, gains, nb, meshes=meshes)
    assert rois == [comp.view_geom(i).roi.tuple() for i in range(n)]
    vdiff = view_mask_diff_in_pano(rois, pg.dst_roi_final.tuple(), [host(comp.mask(i)) for i in range(n)], masks)
    compare(rig + "_spherical_cpw_%dx%d" % nm, host(out16), host(comp.result_mask()), ref16, ref_mask, vdiff, halo=3 * 2 ** nb, max_view_mask_flips=0)
    comp.close()


@settings(max_examples=int(os.environ.get("MS_TEST_EXAMPLES_FROM_INPUTS", 8)), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(n=st.integers(3, 6), w=st.integers(160, 480), h=st.integers(120, 320), spread=st.floats(1.3, 1.8), out_w=st.sampled_from([768, 1024, 1536]),
       bands=st.integers(2, 5), cyl=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_random_rigs_from_camera_parameters(ms, cuda, oracle, n, w, h, spread, out_w, bands, cyl, seed):
    """The same comparison on random rigs (views, sizes, field of view, bands, spherical / cylindrical): here a border or seam pixel of a mask MAY flip (the
    device's sinf / cosf against glibc's), so the criterion is the general one -- |diff| <= 3 outside the blend support of the pixels whose masks differ,
    few such pixels, the overwhelming majority bit-equal."""
    hfov = min(130.0, 360.0 / n * spread)
    proj = ms.PROJ_CYLINDRICAL if cyl else ms.PROJ_SPHERICAL
    scale = synth.warp_scale(out_w)
    rng = np.random.default_rng(seed)
    cams = [synth.camera(n, w, h, hfov, i) for i in range(n)]
    gains = [float(g) for g in rng.uniform(0.92, 1.08, n)]
    frames = [synth.frame(w, h, i, int(seed % 7)) for i in range(n)]
    comp = ms.Compositor(n, (w, h), proj, scale, num_bands=bands, out_size=(0, 0))
    for i in range(n):
        comp.set_camera(i, *cams[i]); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    rois, maps, masks, ref16, ref_mask = oracle_from_inputs(oracle, proj, [c[0] for c in cams], [c[1] for c in cams], scale, w, h, frames, gains, pg.num_bands)
    assert rois == [comp.view_geom(i).roi.tuple() for i in range(n)]
    vdiff = view_mask_diff_in_pano(rois, pg.dst_roi_final.tuple(), [host(comp.mask(i)) for i in range(n)], masks)
    compare("random_%s_n%d_%dx%d_b%d_seed%d" % ("cyl" if cyl else "sph", n, w, h, pg.num_bands, seed), host(out16), host(comp.result_mask()), ref16, ref_mask, vdiff,
            halo=3 * 2 ** pg.num_bands, max_excluded=0.6)      # (one flipped border pixel excludes its whole blend support: up to a 193 x 193 window at 5 bands)
    comp.close()


@pytest.mark.parametrize("size", [(480, 270), (1920, 1080)])
def test_shipped_configuration_from_camera_parameters(ms, cuda, oracle, size):
    """The configuration the reference ships (defs.h:25-27,51-55, calibration.cpp:100,147-194): cylindrical warper, WORK 0.6 / SEAM 0.01 / COMPOSE 1.4
    megapixels -- so every frame goes through cuda::resize (timed.cpp:75-85) --, exposure gains and Voronoi seams from the seam-scale pipeline,
    num_bands by the app's rule.  Product: ms_calibrate_cameras, ms_resize_linear_batch, ms_calibrate_seam, ms_init_blender, ms_stitch.
    Oracle: the same steps from its own primitives and its own (glibc) maps at both scales."""
    w, h = size
    n = 6
    proj = ms.PROJ_CYLINDRICAL
    rig = ms.calibrate_cameras(n, w, h, 90.0, 0.6, 0.01, 1.4)
    cs, cw, ch = rig["compose_scale"], rig["compose_width"], rig["compose_height"]
    frames = [np.clip(synth.frame(w, h, i, 1).astype(np.float32) * (0.92 + 0.03 * i), 0, 255).astype(np.uint8) for i in range(n)]      # exposure differences
    # ---- product
    full = [to_dev(f) for f in frames]
    small = ms.resize_linear_batch(full, fx=cs, fy=cs) if rig["resize_input"] else full
    assert tuple(small[0].shape[:2]) == (ch, cw)
    rois_p = [ms.warp_roi(proj, rig["K_compose"][i], rig["R"][i], rig["compose_warp_scale"], cw, ch) for i in range(n)]
    pano = ms.result_roi(rois_p)
    _, nb = ms.num_bands_rule(pano[2], pano[3], 5.0)
    comp = ms.Compositor(n, (cw, ch), proj, rig["compose_warp_scale"], num_bands=nb, out_size=(0, 0))
    for i in range(n):
        comp.set_camera(i, rig["K_compose"][i], rig["R"][i])
    comp.build_maps()
    gains = comp.calibrate_seam(full, rig["K_seam"], rig["seam_scale"], rig["seam_warp_scale"], dilate=False)
    comp.init_blender()
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([small], out16s=[out16])
    torch.cuda.synchronize()
    # ---- oracle, from the same camera parameters
    O = oracle
    ss = rig["seam_scale"]
    s_rois, s_imgs, s_masks = [], [], []
    for i in range(n):
        seam = O.resize_linear_8u(frames[i], fx=ss, fy=ss)
        hs, ws = seam.shape[:2]
        r = O.warp_roi(proj, rig["K_seam"][i], rig["R"][i], rig["seam_warp_scale"], ws, hs)
        mx, my = O.build_warp_maps(proj, r[0], r[1], r[3], r[2], O.k_rinv_gpu(rig["K_seam"][i], rig["R"][i]), rig["seam_warp_scale"])
        s_rois.append(r)
        s_imgs.append(O.remap_linear_reflect_8uc3(seam, mx, my))
        s_masks.append(O.remap_nearest_8uc1(np.full((hs, ws), 255, np.uint8), mx, my))
    ref_gains = O.gain_compensator([r[:2] for r in s_rois], s_imgs, s_masks)
    O.voronoi_seams([r[:2] for r in s_rois], s_masks)
    small_np = [O.resize_linear_8u(f, fx=cs, fy=cs) if rig["resize_input"] else f for f in frames]
    assert all(np.array_equal(host(a), b) for a, b in zip(small, small_np)), "cuda::resize of the frames is bit-exact"
    c_rois = [O.warp_roi(proj, rig["K_compose"][i], rig["R"][i], rig["compose_warp_scale"], cw, ch) for i in range(n)]
    assert c_rois == rois_p and nb == pg.num_bands
    masks = []
    for i, r in enumerate(c_rois):
        mx, my = O.build_warp_maps(proj, r[0], r[1], r[3], r[2], O.k_rinv_gpu(rig["K_compose"][i], rig["R"][i]), rig["compose_warp_scale"])
        valid = O.remap_nearest_8uc1(np.full((ch, cw), 255, np.uint8), mx, my)
        masks.append(O.resize_linear_8u(s_masks[i], dsize=(r[2], r[3])) & valid)      # calibration.cpp:232-235
    _, _, _, ref16, ref_mask = oracle_from_inputs(O, proj, rig["K_compose"], rig["R"], rig["compose_warp_scale"], cw, ch, small_np, ref_gains, nb, masks=masks)
    assert np.allclose(gains, ref_gains, rtol=2e-3), (gains, ref_gains)
    vdiff = view_mask_diff_in_pano(c_rois, pg.dst_roi_final.tuple(), [host(comp.mask(i)) for i in range(n)], masks)
    # the masks are GREY along the seams here (bilinear upsizing of the seam-scale masks): a differing seam-scale mask pixel moves a whole 1 / seam_scale block
    name = "shipped_%dx%d" % (w, h)
    assert int(vdiff.sum()) == 0, "view-mask pixels flipped on the shipped rig: %d (0 observed on MI355X since round 3)" % int(vdiff.sum())
    mask_diff = host(comp.result_mask()) != ref_mask
    d = np.abs(host(out16).astype(np.int32) - ref16.astype(np.int32)).max(axis=2)
    common = (host(comp.result_mask()) != 0) & (ref_mask != 0)
    near = vdiff | mask_diff
    if near.any():
        near = ndimage.maximum_filter(near.astype(np.uint8), size=2 * 3 * 2 ** nb + 1) > 0
    far = common & ~near
    hist = np.bincount(np.minimum(d[common], 8), minlength=9)
    stats = dict(compose="%dx%d" % (cw, ch), num_bands=nb, pano="%dx%d" % (pg.dst_roi_final.width, pg.dst_roi_final.height), common_px=int(common.sum()),
                 result_mask_diff_px=int(mask_diff.sum()), view_mask_diff_px=int(vdiff.sum()), excluded_px=int((common & near).sum()),
                 max_diff_far=int(d[far].max()) if far.any() else 0, max_diff_anywhere=int(d[common].max()), hist_0_to_8plus=[int(x) for x in hist],
                 gains_product=[round(g, 4) for g in gains], gains_oracle=[round(float(g), 4) for g in ref_gains])
    record(name, stats)
    # gains agree to 2e-3 relative, i.e. up to 0.5 grey levels at 255 BEFORE the blend: the criterion is the reference's <= 3 away from differing mask pixels
    assert stats["max_diff_far"] <= 3, stats
    assert stats["result_mask_diff_px"] <= 1e-3 * common.size, stats
    comp.close()
