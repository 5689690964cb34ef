"""Randomised rigs (hypothesis): number of views, source size, field of view, output width, bands, projection, mask mode, CPW on/off.
Every draw is calibrated on the device and one frame is compared with the oracle bit for bit (16S panorama + result mask)."""
import os

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import synth
from helpers import host, to_dev

pytestmark = pytest.mark.gpu


@settings(max_examples=int(os.environ.get("MS_TEST_EXAMPLES", 25)), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(n=st.integers(2, 6), w=st.integers(48, 150), h=st.integers(40, 110), spread=st.floats(1.25, 1.9), out_w=st.sampled_from([192, 256, 320, 448]),
       bands=st.integers(1, 4), cyl=st.booleans(), seams=st.booleans(), cpw=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_random_rig_matches_oracle(ms, cuda, oracle, n, w, h, spread, out_w, bands, cyl, seams, cpw, seed):
    check_rig(ms, cuda, oracle, n, w, h, spread, out_w, bands, cyl, seams, cpw, seed)


@settings(max_examples=int(os.environ.get("MS_TEST_EXAMPLES_MEDIUM", 6)), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(n=st.integers(3, 6), w=st.integers(300, 640), h=st.integers(200, 400), spread=st.floats(1.3, 1.8), out_w=st.sampled_from([1024, 1536, 2048]),
       bands=st.integers(3, 5), cyl=st.booleans(), seams=st.booleans(), cpw=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_random_medium_rig_matches_oracle(ms, cuda, oracle, n, w, h, spread, out_w, bands, cyl, seams, cpw, seed):
    """The same draw at sizes where every level takes the tiled / vectorised kernels, the fused tails, owner cells and XCD-ordered work lists."""
    check_rig(ms, cuda, oracle, n, w, h, spread, out_w, bands, cyl, seams, cpw, seed)


def check_rig(ms, cuda, oracle, n, w, h, spread, out_w, bands, cyl, seams, cpw, seed):
    hfov = min(130.0, 360.0 / n * spread)
    proj = ms.PROJ_CYLINDRICAL if cyl else ms.PROJ_SPHERICAL
    rng = np.random.default_rng(seed)
    margin = int(rng.choice([0, 6, 16])) if cpw else 0      # > 0: work lists planned for masks that move (enqueue-only update_mask); must not change a result
    comp = ms.Compositor(n, (w, h), proj, synth.warp_scale(out_w), num_bands=bands, enable_cpw=cpw, out_size=(out_w, out_w // 2), update_mask_margin=margin)
    gains = [float(g) for g in rng.uniform(0.9, 1.1, n)]
    for i in range(n):
        comp.set_camera(i, *synth.camera(n, w, h, hfov, i)); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1 if seams else 0); comp.init_blender()
    meshes, vmeshes = None, []
    if cpw:
        meshes = []
        for i in range(n):
            r = comp.view_geom(i).roi
            mesh = synth.mesh(r.width, r.height, int(rng.integers(3, 13)), int(rng.integers(3, 13)), phase=0.4 * i, amp=float(rng.uniform(0.5, 12.0)))
            comp.set_mesh(i, *mesh)
            vmeshes.append(mesh)
            meshes.append(tuple(host(m) for m in comp.mesh_maps(i)))
            want = oracle.convert_mesh_to_map(mesh[0], mesh[1], r.width, r.height)       # convertMeshesToMap, bit for bit (NaN holes included)
            assert np.array_equal(meshes[-1][0], want[0], equal_nan=True) and np.array_equal(meshes[-1][1], want[1], equal_nan=True)
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    rois = [comp.view_geom(i).roi.tuple() for i in range(n)]
    b = oracle.Blender([r[:2] for r in rois], [r[2:] for r in rois], bands)
    for i in range(n):
        b.init_view(i, host(comp.mask(i)))
    for i in range(n):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames[i], xm, ym, gains[i], *(meshes[i] if meshes else (None, None)))
    ref, refmask = b.blend()
    assert b.num_bands == pg.num_bands
    assert np.array_equal(host(out16), ref) and np.array_equal(host(comp.result_mask()), refmask)
    # a column shard of the same rig (SURVEY 8(e) pano-column split): its window must equal the oracle's frame too, from the views it asks for only
    S = int(rng.integers(2, 4)); k = int(rng.integers(0, S))
    shard = ms.Compositor(n, (w, h), proj, synth.warp_scale(out_w), num_bands=bands, enable_cpw=cpw, out_size=(out_w, out_w // 2), col_shards=S, col_shard_index=k)
    for i in range(n):
        shard.set_camera(i, *synth.camera(n, w, h, hfov, i)); shard.set_gain(i, gains[i])
    shard.build_maps(); shard.build_masks(1 if seams else 0)
    try:
        shard.init_blender()
        narrow = False
    except ms.MsError as e:
        narrow = "too narrow" in str(e)
        assert narrow, e
    if not narrow:
        for i in range(n if cpw else 0):
            shard.set_mesh(i, *vmeshes[i])
        b0, b1 = shard.col_window()
        need = shard.needed_views()
        got = torch.full_like(out16, -5)
        shard.stitch([[to_dev(f) if (need >> i) & 1 else None for i, f in enumerate(frames)]], out16s=[got])
        torch.cuda.synchronize()
        assert np.array_equal(host(got)[:, b0:b1], ref[:, b0:b1]), "column shard %d/%d, window [%d, %d)" % (k, S, b0, b1)
    shard.close()
    # the planar I420 output must equal canvas + cvtColor(BGR2YUV_I420) wherever it is supported (any parity of the canvas offset)
    canvas = torch.zeros((out_w // 2, out_w, 3), dtype=torch.uint8, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out8u=[canvas])
    y0, rows = comp.i420_rows()
    slabs = comp.new_i420(1)
    try:
        comp.stitch_i420([[to_dev(f) for f in frames]], slabs)
        supported = True
    except ms.MsError:
        supported = False
    if supported:
        torch.cuda.synchronize()
        assert np.array_equal(host(slabs[0]), host(ms.bgr_to_i420(canvas[y0:y0 + rows])))
    else:                         # small pyramids never reach the tiled level-0 kernel: the call must refuse, not approximate
        assert pg.dst_roi.width % 8 != 0 or pg.num_bands <= 2 or rows == 0
    if margin:
        # enqueue-only update_mask of one view: the oracle's update_mask if the mesh stays within the margin, nothing otherwise
        u = int(rng.integers(0, n))
        within = comp.mesh_displacement(u) <= margin
        comp.update_mask(u)
        comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
        torch.cuda.synchronize()
        if within:
            b.update_mask(u, *meshes[u])
        for i in range(n):
            xm, ym = [host(t) for t in comp.maps(i)]
            b.stitch_online(i, frames[i], xm, ym, gains[i], *meshes[i])
        ref2, refmask2 = b.blend()
        assert np.array_equal(host(out16), ref2) and np.array_equal(host(comp.result_mask()), refmask2), "update_mask of view %d (within margin: %s)" % (u, within)
    b.close(); comp.close()


@settings(max_examples=int(os.environ.get("MS_TEST_EXAMPLES", 15)), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(n=st.integers(2, 5), w=st.integers(48, 120), h=st.integers(40, 90), spread=st.floats(1.3, 1.8), out_w=st.sampled_from([192, 256, 320]),
       nf=st.integers(1, 3), sharp=st.sampled_from([0.02, 0.05, 0.2]), holes=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_random_feather_rig_batches_and_canvas(ms, cuda, oracle, n, w, h, spread, out_w, nf, sharp, holes, seed):
    """FeatherBlender mode on random rigs, batches of 1-3 frames, random holes in the masks, 16S result + mask + 8U canvas."""
    from test_compositor_gpu import canvas_from
    hfov = min(130.0, 360.0 / n * spread)
    out = (out_w, out_w // 2)
    comp = ms.Compositor(n, (w, h), ms.PROJ_SPHERICAL, synth.warp_scale(out_w), num_bands=0, out_size=out, max_frames=nf)
    rng = np.random.default_rng(seed)
    gains = [float(g) for g in rng.uniform(0.9, 1.1, n)]
    for i in range(n):
        comp.set_camera(i, *synth.camera(n, w, h, hfov, i)); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(0)
    masks = []
    for i in range(n):
        m = host(comp.mask(i)).copy()
        if holes:
            y0, x0 = rng.integers(0, max(1, m.shape[0] - 8)), rng.integers(0, max(1, m.shape[1] - 8))
            m[y0:y0 + 8, x0:x0 + 8] = 0
            comp.set_mask(i, m)
        masks.append(m)
    comp.init_feather(sharp)
    frames = [[rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)] for _ in range(nf)]
    pg = comp.pano_geom()
    out16 = [torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda) for _ in range(nf)]
    out8 = [torch.zeros((out[1], out[0], 3), dtype=torch.uint8, device=cuda) for _ in range(nf)]
    comp.stitch([[to_dev(f) for f in fr] for fr in frames], out8u=out8, out16s=out16)
    torch.cuda.synchronize()
    corners = [comp.view_geom(i).roi.tuple()[:2] for i in range(n)]
    for t in range(nf):
        warped = []
        for i in range(n):
            xm, ym = [host(m) for m in comp.maps(i)]
            warped.append(oracle.convert_scale_8u(oracle.remap_linear_8uc3(frames[t][i], xm, ym), gains[i]))
        ref16, refmask, roi = oracle.feather_blend(corners, warped, masks, sharp)
        assert np.array_equal(host(out16[t]), ref16) and np.array_equal(host(comp.result_mask()), refmask)
        assert np.array_equal(host(out8[t]), canvas_from(ref16, pg, out[0], out[1]))
    comp.close()


@settings(max_examples=int(os.environ.get("MS_TEST_EXAMPLES", 12)), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(n=st.integers(2, 3), w=st.integers(64, 160), h=st.integers(48, 120), step=st.floats(8.0, 24.0), hfov=st.floats(45.0, 75.0),
       scale=st.floats(60.0, 220.0), bands=st.integers(1, 4), seams=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_random_plane_rig_matches_oracle(ms, cuda, oracle, n, w, h, step, hfov, scale, bands, seams, seed):
    """PlaneWarperGpu rigs: 2-3 views fanned out by 8-24 degrees (a plane cannot hold a wide panorama), random scale; no equirect canvas."""
    import math
    comp = ms.Compositor(n, (w, h), ms.PROJ_PLANE, float(np.float32(scale)), num_bands=bands, out_size=(0, 0))
    rng = np.random.default_rng(seed)
    gains = [float(g) for g in rng.uniform(0.9, 1.1, n)]
    for i in range(n):
        comp.set_camera(i, *synth.camera(1, w, h, hfov, 0, yaw=math.radians((i - (n - 1) / 2.0) * step))); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1 if seams else 0); comp.init_blender()
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    rois = [comp.view_geom(i).roi.tuple() for i in range(n)]
    b = oracle.Blender([r[:2] for r in rois], [r[2:] for r in rois], bands)
    for i in range(n):
        b.init_view(i, host(comp.mask(i)))
    for i in range(n):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames[i], xm, ym, gains[i])
    ref, refmask = b.blend()
    assert np.array_equal(host(out16), ref) and np.array_equal(host(comp.result_mask()), refmask)
    b.close(); comp.close()


@pytest.mark.parametrize("n,bands,cpw", [(16, 7, False), (16, 5, True), (12, 6, False)])
def test_maximum_views_and_bands(ms, cuda, oracle, n, bands, cpw):
    """The limits of the C-ABI (MS_MAX_VIEWS = 16 views, num_bands up to 7) on a rig that is still cheap for the oracle: same checks as the random rigs."""
    check_rig(ms, cuda, oracle, n, 96, 72, 1.6, 2048, bands, False, True, cpw, 4242 + n)
