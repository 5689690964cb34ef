"""Parity of the fused per-frame path (ms_stitch) with the oracle's restatement of
stitch_online x N + MultiBandBlender::blend (APP/timed.cpp:56-152, blenders.cpp:700-832).

Integer outputs (16SC3 pano, 8UC3 canvas, 8UC1 mask) must be BIT-EXACT given identical maps and masks;
the device-built maps/weights themselves are compared with a float tolerance in test_tables_gpu.py."""
import numpy as np
import pytest
import torch

import synth
from helpers import host, make_rig, oracle_blender_from, to_dev

pytestmark = pytest.mark.gpu


def run_oracle(O, comp, cfg, gains, frames_np, meshes=None):
    b, rois = oracle_blender_from(O, comp, cfg)
    for i in range(cfg["n"]):
        xm, ym = [host(t) for t in comp.maps(i)]
        mx = my = None
        if meshes is not None:
            mx, my = meshes[i]
        b.stitch_online(i, frames_np[i], xm, ym, gains[i], mx, my)
    out, mask = b.blend()
    b.close()
    return out, mask


def canvas_from(out16, pg, out_w, out_h):
    ref = np.zeros((out_h, out_w, 3), np.uint8)
    fh, fw = out16.shape[:2]
    x0, y0 = pg.canvas_x, pg.canvas_y
    xs0, ys0 = max(0, -x0), max(0, -y0)
    xs1, ys1 = min(fw, out_w - x0), min(fh, out_h - y0)
    ref[y0 + ys0:y0 + ys1, x0 + xs0:x0 + xs1] = np.clip(out16[ys0:ys1, xs0:xs1], 0, 255).astype(np.uint8)
    return ref


@pytest.mark.parametrize("rig", ["mini6", "mini4"])
@pytest.mark.parametrize("mask_mode", [0, 1])
def test_stitch_matches_oracle(ms, cuda, oracle, rig, mask_mode):
    comp, cfg, gains = make_rig(ms, rig, mask_mode=mask_mode)
    frames_np = [synth.frame(cfg["w"], cfg["h"], i, 0) for i in range(cfg["n"])]
    pg = comp.pano_geom()
    fw, fh = pg.dst_roi_final.width, pg.dst_roi_final.height
    out16 = torch.full((fh, fw, 3), -7, dtype=torch.int16, device=cuda)
    out8 = torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda)
    comp.stitch([[to_dev(f) for f in frames_np]], out8u=[out8], out16s=[out16])
    torch.cuda.synchronize()
    ref16, refmask = run_oracle(oracle, comp, cfg, gains, frames_np)
    got16 = host(out16)
    assert np.array_equal(host(comp.result_mask()), refmask), "integer mask must be pixel-for-pixel"
    bad = np.argwhere(got16 != ref16)
    assert bad.size == 0, "first mismatches (y,x,c): %s got %s want %s" % (bad[:5], got16[tuple(bad[:5].T)], ref16[tuple(bad[:5].T)])
    assert np.array_equal(host(out8), canvas_from(ref16, pg, cfg["out_w"], cfg["out_h"]))
    comp.close()


def test_band_cell_classes_and_their_parity(ms, cuda, oracle):
    """k_blend8 picks its arithmetic per 64 x 16 cell (ms_get_band_cells): owned cells (one view, weight 1) and, at level 0, exclusive cells (a seam runs through the
    cell but every pixel has one contributing view with weight exactly 1: binary seam masks) use integers only; everything else the reference's float multiply and
    divide.  All three classes must occur in these rigs and give the oracle's pixels: Voronoi seam masks (owned + exclusive), overlapping all-255 masks (two views
    with weight 1 at a pixel: weight sum 2.00001 -> general) and masks with grey values (general)."""
    rng = np.random.default_rng(11)
    seen = {"owned": 0, "exclusive": 0, "general": 0}
    for variant in ("voronoi", "overlap", "grey"):
        comp, cfg, gains = make_rig(ms, "mini6", mask_mode=1 if variant == "voronoi" else 0)
        if variant == "grey":
            for i in range(cfg["n"]):
                r = comp.view_geom(i).roi
                m = np.full((r.height, r.width), 255, np.uint8)
                m[:, : r.width // 3] = 0
                m[::7, r.width // 2:] = 90                   # non-binary weights on some rows
                comp.set_mask(i, m)
            comp.init_blender()
        o, e, g = comp.band_cells(0)
        pg = comp.pano_geom()
        assert o + e + g == -(-pg.dst_roi.width // 64) * -(-pg.dst_roi.height // 16)
        if variant == "voronoi":
            assert o > 0 and e > 0, (o, e, g)
            inner = sum(comp.band_cells(l)[1] for l in range(1, pg.num_bands))
            assert inner == 0, "exclusive cells exist at level 0 only (the coarser weights are fractional along the seams)"
        else:
            assert g > 0, (variant, o, e, g)
        seen["owned"] += o; seen["exclusive"] += e; seen["general"] += g
        frames_np = [synth.frame(cfg["w"], cfg["h"], i, 4) for i in range(cfg["n"])]
        frames_np[0] = rng.integers(0, 256, frames_np[0].shape, dtype=np.uint8)      # noise: Laplacians of both signs and full range in every cell
        out16 = torch.full((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), 5, dtype=torch.int16, device=cuda)
        comp.stitch([[to_dev(f) for f in frames_np]], out16s=[out16])
        torch.cuda.synchronize()
        ref16, refmask = run_oracle(oracle, comp, cfg, gains, frames_np)
        assert np.array_equal(host(out16), ref16), variant
        comp.close()
    assert all(v > 0 for v in seen.values()), seen


def test_stitch_cpw_matches_oracle(ms, cuda, oracle):
    comp, cfg, gains = make_rig(ms, "mini6", enable_cpw=True)
    frames_np = [synth.frame(cfg["w"], cfg["h"], i, 3) for i in range(cfg["n"])]
    meshes = []
    for i in range(cfg["n"]):
        r = comp.view_geom(i).roi
        mx, my = synth.mesh(r.width, r.height, 10, 12, phase=0.3 * i, amp=4.0)
        comp.set_mesh(i, mx, my)
        dmx, dmy = comp.mesh_maps(i)
        meshes.append((host(dmx), host(dmy)))
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames_np]], out16s=[out16])
    torch.cuda.synchronize()
    ref16, _ = run_oracle(oracle, comp, cfg, gains, frames_np, meshes)
    assert np.array_equal(host(out16), ref16)
    comp.close()


@pytest.mark.parametrize("margin", [0, 12])
def test_update_mask_matches_oracle(ms, cuda, oracle, margin):
    """MultiBandBlender::update_mask (blenders.cpp:297-315): masks re-warped through the CPW meshes replace the blend weights.
    margin = 0: the synchronous rebuild; margin > 0: the enqueue-only path (double-buffered tables, work lists planned with the margin)."""
    comp, cfg, gains = make_rig(ms, "mini6", enable_cpw=True, update_mask_margin=margin)
    frames_np = [synth.frame(cfg["w"], cfg["h"], i, 5) for i in range(cfg["n"])]
    frames = [[to_dev(f) for f in frames_np]]
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    meshes = []
    for i in range(cfg["n"]):
        r = comp.view_geom(i).roi
        mx, my = synth.mesh(r.width, r.height, 9, 11, phase=0.4 * i, amp=6.0)
        comp.set_mesh(i, mx, my)
        dmx, dmy = comp.mesh_maps(i)
        meshes.append((host(dmx), host(dmy)))
    comp.stitch(frames, out16s=[out16]); torch.cuda.synchronize()
    before = host(out16).copy()
    updated = (0, 2, 3)
    for i in updated:
        comp.update_mask(i)
    comp.stitch(frames, out16s=[out16]); torch.cuda.synchronize()

    b, _ = oracle_blender_from(oracle, comp, cfg)       # comp.mask(i) stays the mask init_gpu received
    changed = 0
    for i in updated:
        warped = b.update_mask(i, *meshes[i])
        changed += int((warped != host(comp.mask(i))).sum())
    assert changed > 0, "the meshes must actually move the masks"
    for i in range(cfg["n"]):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames_np[i], xm, ym, gains[i], *meshes[i])
    ref16, refmask = b.blend()
    b.close()
    assert np.array_equal(host(comp.result_mask()), refmask)
    assert np.array_equal(host(out16), ref16)
    assert not np.array_equal(before, ref16)
    if margin:
        # a mesh that displaces further than the margin leaves the tables as they are (the work lists were not planned for it) ...
        r = comp.view_geom(1).roi
        comp.set_mesh(1, *synth.mesh(r.width, r.height, 9, 11, phase=0.9, amp=3.0 * margin))
        assert comp.mesh_displacement(1) > margin
        comp.update_mask(1)
        assert np.array_equal(host(comp.result_mask()), refmask)
        # ... and the next one within the margin takes effect again; many updates in a row alternate between the two copies of the tables
        comp.set_mesh(1, *synth.mesh(r.width, r.height, 9, 11, phase=0.4, amp=6.0))
        for _ in range(5):
            comp.update_mask(1)
        out_b = torch.zeros_like(out16)
        comp.stitch(frames, out16s=[out_b]); torch.cuda.synchronize()
        b2, _ = oracle_blender_from(oracle, comp, cfg)
        m1 = tuple(host(m) for m in comp.mesh_maps(1))
        for i in updated + (1,):
            b2.update_mask(i, *(m1 if i == 1 else meshes[i]))
        for i in range(cfg["n"]):
            xm, ym = [host(t) for t in comp.maps(i)]
            b2.stitch_online(i, frames_np[i], xm, ym, gains[i], *(m1 if i == 1 else meshes[i]))
        ref_b, refmask_b = b2.blend()
        b2.close()
        assert np.array_equal(host(out_b), ref_b) and np.array_equal(host(comp.result_mask()), refmask_b)
    # fresh masks drop the re-warped ones: back to the first result
    for i in range(cfg["n"]):
        comp.set_mask(i, host(comp.mask(i)))
    comp.init_blender()
    comp.stitch(frames, out16s=[out16]); torch.cuda.synchronize()
    assert np.array_equal(host(out16), before)
    comp.close()


def test_synchronous_update_mask_is_safe_beside_a_stitching_thread(ms, cuda):
    """with update_mask_margin = 0, ms_update_mask rebuilds (reallocates) the static tables.  Issued from a second thread while the first
    keeps calling ms_stitch it must neither crash nor corrupt a frame: the rebuild and the stitch enqueue exclude each other (ms_ctx::tables_mu), and
    the rebuild waits for the stitches already on the GPU.  Every frame stitched meanwhile must equal the result before OR after some prefix of the
    updates, and the final frame a sequential context's."""
    import threading
    comp, cfg, gains = make_rig(ms, "mini6", enable_cpw=True, update_mask_margin=0)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 2)) for i in range(cfg["n"])]]
    pg = comp.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    meshes = []
    for i in range(cfg["n"]):
        r = comp.view_geom(i).roi
        meshes.append(synth.mesh(r.width, r.height, 9, 11, phase=0.3 * i, amp=5.0))
        comp.set_mesh(i, *meshes[i])
    # the sequential answers: after 0, 1, 2, ... updates
    seq, _, _ = make_rig(ms, "mini6", enable_cpw=True, update_mask_margin=0)
    for i in range(cfg["n"]):
        seq.set_mesh(i, *meshes[i])
    answers = []
    o = torch.zeros(shape, dtype=torch.int16, device=cuda)
    seq.stitch(frames, out16s=[o]); torch.cuda.synchronize(); answers.append(host(o).copy())
    order = [0, 3, 1, 4]
    for v in order:
        seq.update_mask(v)
        seq.stitch(frames, out16s=[o]); torch.cuda.synchronize(); answers.append(host(o).copy())
    seq.close()
    errors, seen, stop = [], [], threading.Event()

    def stitcher():
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                out = torch.zeros(shape, dtype=torch.int16, device=cuda)
                while not stop.is_set():
                    comp.stitch(frames, out16s=[out]); st.synchronize()
                    seen.append(host(out).copy())
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=stitcher)
    t.start()
    try:
        st2 = torch.cuda.Stream()
        with torch.cuda.stream(st2):
            for v in order:
                comp.update_mask(v)
    finally:
        stop.set(); t.join()
    assert not errors, errors
    assert len(seen) >= 1
    for k, fr in enumerate(seen):
        assert any(np.array_equal(fr, a) for a in answers), "frame %d of %d matches no sequential state" % (k, len(seen))
    comp.stitch(frames, out16s=[o]); torch.cuda.synchronize()
    assert np.array_equal(host(o), answers[-1])
    comp.close()


@pytest.mark.parametrize("nf,rig,cpw", [(3, "mini6", False), (32, "mini6", False), (64, "mini6", False), (40, "mini6", True), (64, "mini4", True), (49, "mini4", False), (33, "mini6", True)])
def test_batched_frames_equal_single_frames(ms, cuda, nf, rig, cpw):
    """Frames per ms_stitch call up to the ABI's limit (64 since round 6).  The kernels that read the callers' frames take their pointers in a by-value table of 192 entries, so a
    call of more frames than 192 / views (32 for six views, 48 for four) sends those launches out in CHUNKS -- each with its own table, the per-frame buffers offset by the chunk's
    first frame -- while the reduce and band chains cover all frames at once: 64 and 40 frames (a full chunk + a short one), 49 and 33 (a full chunk + ONE frame: the one-frame
    instantiation of the kernels) must equal the same frames stitched one per call, without and with CPW (the first remap and the mesh remap are both chunked)."""
    comp, cfg, gains = make_rig(ms, rig, max_frames=nf, enable_cpw=cpw)
    if cpw:
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            comp.set_mesh(i, *synth.mesh(r.width, r.height, 9, 11, phase=0.3 * i, amp=5.0))
    pg = comp.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, t)) for i in range(cfg["n"])] for t in range(nf)]
    batch = [torch.zeros(shape, dtype=torch.int16, device=cuda) for _ in range(nf)]
    comp.stitch(frames, out16s=batch)
    with pytest.raises(ms.MsError):
        ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), max_frames=65)
    for t in range(nf):
        single = torch.zeros(shape, dtype=torch.int16, device=cuda)
        comp.stitch([frames[t]], out16s=[single])
        assert torch.equal(single, batch[t])
    assert not torch.equal(batch[0], batch[1])
    comp.close()


def test_full_size_config2_matches_oracle(ms, cuda, oracle):
    """BASELINE.json configs[1] at full size: 6x1080p -> 3840x1920, 5 bands, one frame, bit-exact."""
    comp, cfg, gains = make_rig(ms, "cfg2")
    frames_np = [synth.frame(cfg["w"], cfg["h"], i, 0) for i in range(cfg["n"])]
    pg = comp.pano_geom()
    assert pg.dst_roi.tuple() == (-1919, 646, 3840, 640) and pg.num_bands == 5
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    out8 = torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda)
    comp.stitch([[to_dev(f) for f in frames_np]], out8u=[out8], out16s=[out16])
    torch.cuda.synchronize()
    oracle.set_num_threads(8)
    ref16, refmask = run_oracle(oracle, comp, cfg, gains, frames_np)
    assert np.array_equal(host(comp.result_mask()), refmask)
    assert np.array_equal(host(out16), ref16)
    # size-independent properties: zero outside the mask, canvas = saturate(pano) at its spherical position
    got = host(out16)
    assert not got[refmask == 0].any()
    assert np.array_equal(host(out8), canvas_from(ref16, pg, cfg["out_w"], cfg["out_h"]))
    assert (refmask[313] == 255).all(), "the 6 views cover the full circle along the equator row"
    comp.close()


@pytest.mark.parametrize("rig,cpw", [("cfg5", False), ("cfg2", True)])
def test_full_size_config5_and_config3_match_oracle(ms, cuda, oracle, rig, cpw):
    """BASELINE.json configs[4] geometry (12 x 4K -> 7680 x 3840, SURVEY App. C: pano ROI 7680 x 768, two full-width views) and configs[2]
    (config 2 + CPW, 40 x 40 meshes) at full size against the oracle, one frame, bit-exact."""
    comp, cfg, gains = make_rig(ms, rig, enable_cpw=cpw)
    frames_np = [synth.frame(cfg["w"], cfg["h"], i, 1) for i in range(cfg["n"])]
    pg = comp.pano_geom()
    if rig == "cfg5":
        assert pg.dst_roi_final.tuple() == (-3839, 1536, 7680, 768) and pg.num_bands == 5
    meshes = None
    if cpw:
        meshes = []
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            comp.set_mesh(i, *synth.mesh(r.width, r.height, 40, 40, phase=0.1 * i + 0.7, amp=8.0))
            meshes.append(tuple(host(m) for m in comp.mesh_maps(i)))
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames_np]], out16s=[out16])
    torch.cuda.synchronize()
    oracle.set_num_threads(16)
    ref16, refmask = run_oracle(oracle, comp, cfg, gains, frames_np, meshes)
    assert np.array_equal(host(comp.result_mask()), refmask)
    assert np.array_equal(host(out16), ref16)
    comp.close()


@pytest.mark.parametrize("rig,cpw", [("cfg2", False), ("cfg2", True), ("cfg5", False), ("mini6", True), ("mini4", False)])
def test_tiled_kernels_equal_simple_kernels(ms, cuda, rig, cpw):
    """The work-list / multi-pixel-per-lane kernels against the one-pixel-per-lane kernels (same library,
    debug switch): identical 16S panorama, also where whole tiles are skipped as zero-weight."""
    outs = []
    for simple in (False, True):
        comp, cfg, gains = make_rig(ms, rig, enable_cpw=cpw, simple_kernels=simple)
        if cpw:
            for i in range(cfg["n"]):
                r = comp.view_geom(i).roi
                comp.set_mesh(i, *synth.mesh(r.width, r.height, 8, 9, phase=0.2 * i, amp=3.0))
        pg = comp.pano_geom()
        out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
        out8 = torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda)
        comp.stitch([[to_dev(synth.frame(cfg["w"], cfg["h"], i, 1)) for i in range(cfg["n"])]], out8u=[out8], out16s=[out16])
        torch.cuda.synchronize()
        outs.append((out16, out8))
        comp.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("amp", [20.0, 60.0])
def test_cpw_stage1_short_list_and_fallback(ms, cuda, oracle, amp):
    """CPW stage 1 only warps the tiles stage 2 can reach while the mesh displaces by <= 32 px (measured at ms_set_mesh); a mesh that
    moves samples further falls back to whole views.  Both sides of the threshold against the oracle, with a mesh change in between
    (the stage buffer then holds stale tiles of the previous mesh, which must never be sampled)."""
    comp, cfg, gains = make_rig(ms, "mini6", enable_cpw=True)
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    frames = [synth.frame(cfg["w"], cfg["h"], i, 3) for i in range(cfg["n"])]
    for a in (60.0 if amp < 32 else 5.0, amp):          # first the other regime, then the one under test
        meshes = []
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            mx, my = synth.mesh(r.width, r.height, 9, 11, phase=0.3 * i, amp=a)
            comp.set_mesh(i, mx, my)
            dmx, dmy = comp.mesh_maps(i)
            meshes.append((host(dmx), host(dmy)))
        comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
        torch.cuda.synchronize()
        d = max(comp.mesh_displacement(i) for i in range(cfg["n"]))
        assert (d > 32.0) == (a > 32.0) and 0.5 * a < d < 1.5 * a          # measured on the device at ms_set_mesh
    ref16, refmask = run_oracle(oracle, comp, cfg, gains, frames, meshes=meshes)
    assert np.array_equal(host(out16), ref16)
    comp.close()


@pytest.mark.parametrize("rig", ["cfg2", "mini4"])
def test_lds_staged_warp_equals_direct(ms, cuda, rig):
    """Opt-in variant of the warp kernel that stages each tile's source bounding box through LDS (16-byte chunk copy,
    v_alignbyte extraction): same pixels as the default direct-gather path."""
    outs = []
    for lds in (False, True):
        comp, cfg, gains = make_rig(ms, rig, lds_stage=lds)
        pg = comp.pano_geom()
        out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
        comp.stitch([[to_dev(synth.frame(cfg["w"], cfg["h"], i, 4)) for i in range(cfg["n"])]], out16s=[out16])
        torch.cuda.synchronize()
        outs.append(out16)
        comp.close()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("rig,shards", [("mini6", 2), ("mini6", 3), ("cfg2", 2), ("cfg5", 2)])       # cfg5 = BASELINE configs[4]: 12 x 4K, views 6 + 6
def test_view_sharding_equals_single_context(ms, cuda, rig, shards):
    """BASELINE configs[4] mechanism on one GPU: S contexts, each owning a block of views, write partial int16 sums;
    the sink context adds them (wrap-around) and finishes.  Must equal the single-context frame bit for bit."""
    full, cfg, _ = make_rig(ms, rig, max_frames=2)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, t)) for i in range(cfg["n"])] for t in range(2)]
    pg = full.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    want16 = [torch.zeros(shape, dtype=torch.int16, device=cuda) for _ in range(2)]
    want8 = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda) for _ in range(2)]
    full.stitch(frames, out8u=want8, out16s=want16)
    comps, parts = [], []
    for k in range(shards):
        c, _, _ = make_rig(ms, rig, max_frames=2, shards=shards, shard_index=k)
        lo, hi = k * cfg["n"] // shards, (k + 1) * cfg["n"] // shards
        mine = [[fr[v] if lo <= v < hi else None for v in range(cfg["n"])] for fr in frames]
        part = torch.full((2 * c.partial_bytes() // 2,), 12345, dtype=torch.int16, device=cuda)
        c.stitch_partial(mine, part)
        comps.append(c); parts.append(part)
    with pytest.raises(ms.MsError, match="view shard"):
        comps[0].stitch(frames, out16s=want16)
    got16 = [torch.zeros(shape, dtype=torch.int16, device=cuda) for _ in range(2)]
    got8 = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda) for _ in range(2)]
    comps[0].stitch_finish(2, parts[::-1], out8u=got8, out16s=got16)       # order of the partials must not matter
    torch.cuda.synchronize()
    for t in range(2):
        assert torch.equal(got16[t], want16[t]) and torch.equal(got8[t], want8[t])
    for c in comps:
        c.close()
    full.close()


@pytest.mark.parametrize("rig,shards,cpw", [("mini6", 2, False), ("mini6", 3, True), ("mini4", 2, False), ("cfg2", 4, False), ("cfg3", 2, True), ("cfg5", 2, False)])
def test_column_sharding_equals_single_context(ms, cuda, rig, shards, cpw):
    """SURVEY 8(e), pano-column split: S contexts, each compositing a window of panorama columns from work lists cut down to that window plus
    its halo; every window must equal the unsharded frame bit for bit (16S ROI, 8U canvas, I420), the windows must tile the panorama, and a
    shard must not even look at the views that do not reach its window (they are handed over as None)."""
    name = "cfg2" if rig == "cfg3" else rig
    full, cfg, _ = make_rig(ms, name, max_frames=2, enable_cpw=cpw)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, t)) for i in range(cfg["n"])] for t in range(2)]
    meshes = None
    if cpw:
        meshes = [synth.mesh(full.view_geom(i).roi.width, full.view_geom(i).roi.height, 12, 9, phase=0.3 * i, amp=6.0) for i in range(cfg["n"])]
        for i, (mx, my) in enumerate(meshes):
            full.set_mesh(i, mx, my)
    pg = full.pano_geom()
    fw, fh = pg.dst_roi_final.width, pg.dst_roi_final.height
    want16 = [torch.zeros((fh, fw, 3), dtype=torch.int16, device=cuda) for _ in range(2)]
    want8 = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda) for _ in range(2)]
    full.stitch(frames, out8u=want8, out16s=want16)
    wanti = full.new_i420(2)
    full.stitch_i420(frames, wanti)
    assert full.col_window() == (0, fw) and full.needed_views() == (1 << cfg["n"]) - 1
    y0, rows = full.i420_rows()
    W = cfg["out_w"]
    edges, read = [], 0
    for k in range(shards):
        c, _, _ = make_rig(ms, name, max_frames=2, enable_cpw=cpw, col_shards=shards, col_shard_index=k)
        if cpw:
            for i, (mx, my) in enumerate(meshes):
                c.set_mesh(i, mx, my)
        b, e = c.col_window()
        edges.append((b, e))
        need = c.needed_views()
        read += bin(need).count("1")
        mine = [[fr[v] if (need >> v) & 1 else None for v in range(cfg["n"])] for fr in frames]
        got16 = [torch.full((fh, fw, 3), -7, dtype=torch.int16, device=cuda) for _ in range(2)]
        got8 = [torch.full((cfg["out_h"], cfg["out_w"], 3), 9, dtype=torch.uint8, device=cuda) for _ in range(2)]
        c.stitch(mine, out8u=got8, out16s=got16)
        goti = c.new_i420(2)
        c.stitch_i420(mine, goti)
        torch.cuda.synchronize()
        cb, ce = b + pg.canvas_x, e + pg.canvas_x
        for t in range(2):
            assert torch.equal(got16[t][:, b:e], want16[t][:, b:e]), "16S window of shard %d" % k
            r0, r1 = max(pg.canvas_y, 0), min(pg.canvas_y + fh, cfg["out_h"])          # (canvas rows outside the panorama ROI are never written)
            assert torch.equal(got8[t][r0:r1, max(cb, 0):ce], want8[t][r0:r1, max(cb, 0):ce]), "8U window of shard %d" % k
            gi, wi = goti[t].view(-1), wanti[t].view(-1)
            Y = lambda a: a[:W * rows].view(rows, W)
            U = lambda a: a[W * rows:W * rows + (W // 2) * (rows // 2)].view(rows // 2, W // 2)
            V = lambda a: a[W * rows + (W // 2) * (rows // 2):].view(rows // 2, W // 2)
            assert torch.equal(Y(gi)[:, max(cb, 0):ce], Y(wi)[:, max(cb, 0):ce])
            lo, hi = (max(cb, 0) + 1) // 2, ce // 2            # chroma samples whose 2 x 2 block starts inside the window
            assert torch.equal(U(gi)[:, lo:hi], U(wi)[:, lo:hi]) and torch.equal(V(gi)[:, lo:hi], V(wi)[:, lo:hi])
        c.close()
    assert edges[0][0] == 0 and edges[-1][1] == fw and all(edges[i][1] == edges[i + 1][0] for i in range(shards - 1)), edges
    assert all(b % 16 == 0 for b, _ in edges)
    if rig in ("cfg2", "cfg5"):
        assert read < shards * cfg["n"], "every shard read every view: the window does not cut the work lists"
    full.close()


def test_config1_two_views(ms, cuda, oracle):
    """BASELINE configs[0] geometry (2 views 640x480, yaw -/+25 deg, hfov 90, scale 2000/2pi; SURVEY App. C known ROIs),
    composited with the multiband path on the GPU and compared with the oracle."""
    import math
    sc = float(np.float32(2000.0 / (2 * math.pi)))
    comp = ms.Compositor(2, (640, 480), ms.PROJ_SPHERICAL, sc, num_bands=5, out_size=(2000, 1000))
    cams = [synth.camera(1, 640, 480, 90.0, 0, yaw=math.radians(a)) for a in (-25.0, 25.0)]
    gains = [0.97, 1.04]
    for i, (K, R) in enumerate(cams):
        comp.set_camera(i, K, R); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    assert comp.view_geom(0).roi.tuple() == (-388, 295, 499, 410) and comp.view_geom(1).roi.tuple() == (-111, 295, 500, 410)
    frames = [synth.frame(640, 480, i, 0) for i in range(2)]
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    cfg = dict(n=2, num_bands=5)
    ref16, refmask = run_oracle(oracle, comp, cfg, gains, frames)
    assert np.array_equal(host(out16), ref16) and np.array_equal(host(comp.result_mask()), refmask)
    comp.close()


@pytest.mark.parametrize("rig", ["config1", "mini4"])
def test_feather_blender_matches_oracle(ms, cuda, oracle, rig):
    """BASELINE configs[0]: the 2-view 640x480 example composited with FeatherBlender (blenders.cpp:139-186; weights =
    createWeightMap(mask, 0.02)).  GPU: context with num_bands = 0 + ms_init_feather, one ms_stitch.  Oracle: the CPU restatement of
    prepare / feed / blend fed with the oracle's own warp of the same frames.  Bit-exact 16S result, mask and 8U canvas."""
    import math
    if rig == "config1":
        n, w, h, out = 2, 640, 480, (2000, 1000)
        sc = float(np.float32(2000.0 / (2 * math.pi)))
        cams = [synth.camera(1, w, h, 90.0, 0, yaw=math.radians(a)) for a in (-25.0, 25.0)]
        gains = [0.97, 1.04]
    else:
        cfg = synth.CONFIGS[rig]
        n, w, h, out = cfg["n"], cfg["w"], cfg["h"], (cfg["out_w"], cfg["out_h"])
        sc = synth.warp_scale(cfg["out_w"])
        cams = [synth.camera(n, w, h, cfg["hfov_deg"], i) for i in range(n)]
        gains = synth.gains(n)
    comp = ms.Compositor(n, (w, h), ms.PROJ_SPHERICAL, sc, num_bands=0, out_size=out, max_frames=2)
    for i, (K, R) in enumerate(cams):
        comp.set_camera(i, K, R); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(0)          # FeatherBlender takes the warped all-255 masks, no seams
    comp.init_feather(0.02)
    pg = comp.pano_geom()
    assert pg.num_bands == 0
    frames = [[synth.frame(w, h, i, t) for i in range(n)] for t in range(2)]
    out16 = [torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda) for _ in range(2)]
    out8 = [torch.zeros((out[1], out[0], 3), dtype=torch.uint8, device=cuda) for _ in range(2)]
    comp.stitch([[to_dev(f) for f in fr] for fr in frames], out8u=out8, out16s=out16)
    torch.cuda.synchronize()
    corners = [comp.view_geom(i).roi.tuple()[:2] for i in range(n)]
    masks = [host(comp.mask(i)) for i in range(n)]
    for t in range(2):
        warped = []
        for i in range(n):
            xm, ym = [host(m) for m in comp.maps(i)]
            warped.append(oracle.convert_scale_8u(oracle.remap_linear_8uc3(frames[t][i], xm, ym), gains[i]))
        ref16, refmask, roi = oracle.feather_blend(corners, warped, masks, 0.02)
        assert roi == pg.dst_roi_final.tuple()
        assert np.array_equal(host(out16[t]), ref16)
        assert np.array_equal(host(comp.result_mask()), refmask)
        assert np.array_equal(host(out8[t]), canvas_from(ref16, pg, out[0], out[1]))
    # feathering really happens: inside the overlap the result mixes both views
    assert refmask.any() and (host(out16[0]) != 0).any()
    comp.close()


def test_config1_cpu_flavour_remap_and_feather(ms, cuda, oracle):
    """BASELINE configs[0] as the reference's CPU pipeline computes it: spherical maps, cv::remap's fixed-point bilinear (imgwarp.cpp), gain,
    FeatherBlender(0.02).  GPU: the reference kernels with the CPU-flavoured projection remap (ms_config.cpu_flavour_remap) + ms_init_feather."""
    import math
    sc = float(np.float32(2000.0 / (2 * math.pi)))
    comp = ms.Compositor(2, (640, 480), ms.PROJ_SPHERICAL, sc, num_bands=0, out_size=(2000, 1000), simple_kernels=True, cv_remap=True)
    cams = [synth.camera(1, 640, 480, 90.0, 0, yaw=math.radians(a)) for a in (-25.0, 25.0)]
    gains = [0.97, 1.04]
    for i, (K, R) in enumerate(cams):
        comp.set_camera(i, K, R); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(0); comp.init_feather(0.02)
    pg = comp.pano_geom()
    frames = [synth.frame(640, 480, i, 1) for i in range(2)]
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    corners = [comp.view_geom(i).roi.tuple()[:2] for i in range(2)]
    masks = [host(comp.mask(i)) for i in range(2)]
    warped, warped_cuda = [], []
    for i in range(2):
        xm, ym = [host(m) for m in comp.maps(i)]
        warped.append(oracle.convert_scale_8u(oracle.cv_remap_linear(frames[i], xm, ym), gains[i]))
        warped_cuda.append(oracle.convert_scale_8u(oracle.remap_linear_8uc3(frames[i], xm, ym), gains[i]))
    ref16, refmask, roi = oracle.feather_blend(corners, warped, masks, 0.02)
    assert np.array_equal(host(out16), ref16) and np.array_equal(host(comp.result_mask()), refmask)
    other16, _, _ = oracle.feather_blend(corners, warped_cuda, masks, 0.02)
    assert not np.array_equal(other16, ref16)         # the flavour matters: the CUDA arithmetic gives a different panorama
    comp.close()
    with pytest.raises(ms.MsError):                     # only in the reference kernels
        ms.Compositor(2, (640, 480), ms.PROJ_SPHERICAL, sc, num_bands=0, out_size=(2000, 1000), cv_remap=True)


@pytest.mark.parametrize("proj", ["cylindrical", "plane"])
def test_other_projections_match_oracle(ms, cuda, oracle, proj):
    """The three *WarperGpu projections share the per-frame path (warpers_cuda.cpp:149-277): cylindrical on the 4-view rig,
    plane on two views 30 deg apart (a plane cannot hold a full circle).  ROIs equal the oracle's warpRoi, the 16S result is exact."""
    import math
    if proj == "cylindrical":
        pid, n, w, h, hfov = ms.PROJ_CYLINDRICAL, 4, 200, 150, 110.0
        cams = [synth.camera(n, w, h, hfov, i) for i in range(n)]
        sc, out = synth.warp_scale(512), (512, 256)
    else:
        pid, n, w, h = ms.PROJ_PLANE, 2, 240, 160
        cams = [synth.camera(1, w, h, 70.0, 0, yaw=math.radians(a)) for a in (-15.0, 15.0)]
        sc, out = 170.0, (0, 0)
    gains = [1.0 - 0.03 * i for i in range(n)]
    comp = ms.Compositor(n, (w, h), pid, sc, num_bands=3, out_size=out)
    for i, (K, R) in enumerate(cams):
        comp.set_camera(i, K, R); comp.set_gain(i, gains[i])
    comp.build_maps(); comp.build_masks(1); comp.init_blender()
    for i, (K, R) in enumerate(cams):
        assert comp.view_geom(i).roi.tuple() == oracle.warp_roi(pid, K, R, sc, w, h)
    frames = [synth.frame(w, h, i, 2) for i in range(n)]
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out16])
    torch.cuda.synchronize()
    ref16, refmask = run_oracle(oracle, comp, dict(n=n, num_bands=3), gains, frames)
    assert np.array_equal(host(out16), ref16) and np.array_equal(host(comp.result_mask()), refmask)
    assert refmask.mean() > 100          # most of the ROI is covered
    comp.close()


def test_feather_needs_single_band_context(ms, cuda):
    comp, cfg, _ = make_rig(ms, "mini4")
    with pytest.raises(ms.MsError, match="num_bands = 0"):
        comp.init_feather(0.02)
    comp.close()


def test_concurrent_contexts_on_separate_streams(ms, cuda):
    """bench.py's default: several contexts, each on its own HIP stream, in flight at once.  Three contexts x 4 frames issued back to back
    on three streams for several rounds must give exactly what one context gives frame by frame."""
    comps = [make_rig(ms, "mini6", max_frames=4)[0] for _ in range(3)]
    cfg = synth.CONFIGS["mini6"]
    pg = comps[0].pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, t)) for i in range(cfg["n"])] for t in range(12)]
    want = []
    for t in range(12):
        o = torch.zeros(shape, dtype=torch.int16, device=cuda)
        comps[0].stitch([frames[t]], out16s=[o])
        want.append(o)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=cuda) for _ in range(3)]
    for rnd in range(5):
        got = [torch.zeros(shape, dtype=torch.int16, device=cuda) for _ in range(12)]
        torch.cuda.synchronize()
        for k in range(3):
            with torch.cuda.stream(streams[k]):
                comps[k].stitch(frames[4 * k:4 * k + 4], out16s=got[4 * k:4 * k + 4])
        torch.cuda.synchronize()
        for t in range(12):
            assert torch.equal(got[t], want[t]), (rnd, t)
    for c in comps:
        c.close()


def test_feed_then_blend_equals_stitch(ms, cuda):
    """The reference's call shape -- stitch_online per view, then blend (timed.cpp:127-137) -- gives the ms_stitch frame; blend without all
    views fed is a state error."""
    comp, cfg, _ = make_rig(ms, "mini6")
    frames = [to_dev(synth.frame(cfg["w"], cfg["h"], i, 4)) for i in range(cfg["n"])]
    pg = comp.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    want = torch.zeros(shape, dtype=torch.int16, device=cuda); got = torch.zeros(shape, dtype=torch.int16, device=cuda)
    comp.stitch([frames], out16s=[want])
    for i in reversed(range(cfg["n"])):            # any order
        comp.feed(i, frames[i])
    comp.blend(out16s=got)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    comp.feed(0, frames[0])
    with pytest.raises(ms.MsError, match="were fed"):
        comp.blend(out16s=got)
    comp.close()


def test_state_errors(ms, cuda):
    comp = ms.Compositor(2, (64, 48), ms.PROJ_SPHERICAL, 50.0, num_bands=2, out_size=(0, 0))
    with pytest.raises(ms.MsError, match="camera 0 not set"):
        comp.build_maps()
    K, R = synth.camera(2, 64, 48, 90.0, 0)
    comp.set_camera(0, K, R)
    comp.set_camera(1, *synth.camera(2, 64, 48, 90.0, 1))
    with pytest.raises(ms.MsError, match="ms_build_maps first"):
        comp.build_masks(0)
    comp.build_maps()
    with pytest.raises(ms.MsError, match="must be built first"):
        comp.init_blender()
    comp.close()


@pytest.mark.parametrize("rig,cpw", [("mini6", False), ("cfg2", False), ("mini6", True), ("cfg5", False)])
def test_stitch_i420_equals_canvas_then_conversion(ms, cuda, rig, cpw):
    """ms_stitch_i420: the level-0 band kernel writes the encoder's planar I420 itself; must equal ms_stitch(out8u) followed by
    cvtColor(BGR2YUV_I420) (ms_bgr_to_i420) of the same canvas rows, bit for bit, for batches of frames."""
    comp, cfg, _ = make_rig(ms, rig, enable_cpw=cpw, max_frames=2)
    if cpw:
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            comp.set_mesh(i, *synth.mesh(r.width, r.height, 8, 9, phase=0.2 * i, amp=5.0))
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, t)) for i in range(cfg["n"])] for t in range(2)]
    canv = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=cuda) for _ in range(2)]
    comp.stitch(frames, out8u=canv)
    y0, rows = comp.i420_rows()
    pg = comp.pano_geom()
    assert y0 % 2 == 0 and rows % 2 == 0 and y0 <= pg.canvas_y and y0 + rows >= min(cfg["out_h"], pg.canvas_y + pg.dst_roi_final.height)
    outs = comp.new_i420(2)
    comp.stitch_i420(frames, outs)
    torch.cuda.synchronize()
    for t in range(2):
        want = host(ms.bgr_to_i420(canv[t][y0:y0 + rows]))
        got = host(outs[t])
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    # a second call into the same buffers gives the same bytes (nothing accumulates)
    comp.stitch_i420(frames, outs)
    torch.cuda.synchronize()
    assert np.array_equal(host(outs[1]), host(ms.bgr_to_i420(canv[1][y0:y0 + rows])))
    comp.close()


def test_stitch_i420_needs_the_tiled_band_path(ms, cuda):
    comp, cfg, _ = make_rig(ms, "mini6", simple_kernels=True)
    frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 0)) for i in range(cfg["n"])]]
    with pytest.raises(ms.MsError):
        comp.stitch_i420(frames, comp.new_i420(1))
    comp.close()


def test_create_destroy_cycles_release_device_memory(ms, cuda):
    """Every device buffer of a context belongs to it (RAII): create -> calibrate -> set meshes -> update_mask -> stitch -> destroy,
    repeated, must not lower the free device memory (owner maps, re-warped masks and the mesh displacement words used to leak)."""
    def cycle(k=0):
        comp, cfg, gains = make_rig(ms, "mini6", enable_cpw=True, update_mask_margin=8 * (k % 2))      # both forms of update_mask (the second set of tables too)
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            comp.set_mesh(i, *synth.mesh(r.width, r.height, 6, 6, phase=0.3 * i, amp=3.0))
        comp.update_mask(1)
        pg = comp.pano_geom()
        out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
        comp.stitch([[to_dev(synth.frame(cfg["w"], cfg["h"], i, 0)) for i in range(cfg["n"])]], out16s=[out16])
        torch.cuda.synchronize()
        comp.close()
        del out16
    cycle()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(6):
        cycle(k)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (1 << 20), "six create/destroy cycles lost %d bytes of device memory" % (free0 - free1)


def test_second_stream_is_ordered_behind_the_first(ms, cuda):
    """A context has one set of per-batch intermediates: a call on another stream than the previous one waits for it on the GPU
    (the reference creates a fresh cuda::Stream per stitch_online call), so back-to-back calls on two streams stay correct."""
    import ctypes
    comp, cfg, gains = make_rig(ms, "mini6")
    pg = comp.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    fa = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 1)) for i in range(cfg["n"])]]
    fb = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 2)) for i in range(cfg["n"])]]
    want = [torch.zeros(shape, dtype=torch.int16, device=cuda) for _ in range(2)]
    comp.stitch(fa, out16s=[want[0]]); torch.cuda.synchronize()
    comp.stitch(fb, out16s=[want[1]]); torch.cuda.synchronize()
    got = [torch.zeros(shape, dtype=torch.int16, device=cuda) for _ in range(2)]
    s1, s2 = torch.cuda.Stream(device=cuda), torch.cuda.Stream(device=cuda)
    ra, rb = comp.prepared(fa, out16s=[got[0]]), comp.prepared(fb, out16s=[got[1]])
    for _ in range(20):
        ra(ctypes.c_void_p(s1.cuda_stream)); rb(ctypes.c_void_p(s2.cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    comp.close()


@pytest.mark.parametrize("rig,proj,mask_mode,nf", [("mini6", None, 1, 3), ("mini6", "cyl", 0, 2), ("cfg2", None, 1, 2), ("mini4", "cyl", 0, 1)])
def test_nv12_direct_equals_convert_then_stitch(ms, cuda, rig, proj, mask_mode, nf):
    """ms_stitch_nv12 (APP/networking.cpp:45-47 + timed.cpp:56-152): the warp samples the cameras' NV12 planes and converts every tap itself --
    bit-identical to ms_nv12_to_bgr_batch followed by ms_stitch, for noise frames (every clamp of the conversion is exercised), batches of 1 / 2 / 3 frames,
    spherical and cylindrical rigs, and with mask mode 0 (whole warped views: samples on and beyond the image border, BORDER_CONSTANT 0)."""
    pj = {None: None, "cyl": ms.PROJ_CYLINDRICAL}[proj]
    comp, cfg, _ = make_rig(ms, rig, max_frames=nf, mask_mode=mask_mode, projection=pj)
    rng = np.random.default_rng(77)
    nv = [[to_dev(rng.integers(0, 256, (cfg["h"] * 3 // 2, cfg["w"]), dtype=np.uint8)) for _ in range(cfg["n"])] for _ in range(nf)]
    # one structured frame too: the synthetic camera pattern
    nv[0] = [to_dev(synth.nv12_frame(cfg["w"], cfg["h"], i)) for i in range(cfg["n"])]
    bgr = [[ms.nv12_to_bgr(t) for t in fr] for fr in nv]
    pg = comp.pano_geom()

    def outs():
        return ([torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda") for _ in range(nf)],
                [torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device="cuda") for _ in range(nf)])
    a8, a16 = outs()
    comp.stitch(bgr, out8u=a8, out16s=a16)
    b8, b16 = outs()
    comp.stitch_nv12(nv, out8u=b8, out16s=b16)
    torch.cuda.synchronize()
    for f in range(nf):
        assert torch.equal(a16[f], b16[f]), "frame %d: 16S panorama differs" % f
        assert torch.equal(a8[f], b8[f])
    assert int(a16[0].abs().max()) > 0
    # an ROI view of a larger buffer (row step != width) works too; frames of a view with different steps are refused
    big = [torch.zeros((cfg["h"] * 3 // 2, cfg["w"] + 64), dtype=torch.uint8, device="cuda") for _ in range(cfg["n"])]
    roi = []
    for i in range(cfg["n"]):
        big[i][:, 32:32 + cfg["w"]] = nv[0][i]
        roi.append(big[i][:, 32:32 + cfg["w"]])
    c8, c16 = outs()
    comp.stitch_nv12([roi], out8u=c8[:1], out16s=c16[:1])
    torch.cuda.synchronize()
    assert torch.equal(c16[0], a16[0])
    # ... and at an odd byte offset: the planes are then not 4-byte aligned and the kernel's unaligned-read form runs (the aligned 8-byte windows are the default)
    for i in range(cfg["n"]):
        big[i][:, 33:33 + cfg["w"]] = nv[0][i]
    c8, c16 = outs()
    comp.stitch_nv12([[big[i][:, 33:33 + cfg["w"]] for i in range(cfg["n"])]], out8u=c8[:1], out16s=c16[:1])
    torch.cuda.synchronize()
    assert torch.equal(c16[0], a16[0])
    if nf >= 2:
        with pytest.raises(ms.MsError):
            comp.stitch_nv12([roi, nv[1]], out8u=c8[:2], out16s=c16[:2])
    comp.close()


def test_nv12_direct_with_cpw_samples_the_planes_in_stage_one(ms, cuda):
    """CPW contexts: ms_stitch_nv12 runs the first remap (stage 1) on the NV12 planes (k_stage1_nv12) and the mesh remap on its result -- bit-identical to converting first;
    batches of 1 and 3 frames (one- and two-frame instantiations, a short last group), a mesh beyond the stage-1 skip bound included."""
    for amp, nf in ((6.0, 3), (40.0, 1)):
        comp, cfg, _ = make_rig(ms, "mini6", enable_cpw=True, max_frames=nf)
        for i in range(cfg["n"]):
            g = comp.view_geom(i).roi
            mx, my = synth.mesh(g.width, g.height, 9, 11, phase=0.3 * i, amp=amp)
            comp.set_mesh(i, mx, my)
        rng = np.random.default_rng(5)
        nv = [[to_dev(rng.integers(0, 256, (cfg["h"] * 3 // 2, cfg["w"]), dtype=np.uint8)) for _ in range(cfg["n"])] for _ in range(nf)]
        bgr = [[ms.nv12_to_bgr(t) for t in fr] for fr in nv]
        pg = comp.pano_geom()
        a16 = [torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device="cuda") for _ in range(nf)]
        b16 = [torch.zeros_like(a16[0]) for _ in range(nf)]
        comp.stitch(bgr, out16s=a16)
        comp.stitch_nv12(nv, out16s=b16)
        torch.cuda.synchronize()
        for f in range(nf):
            assert torch.equal(a16[f], b16[f]), (amp, f)
        assert int(a16[0].abs().max()) > 0
        comp.close()


def test_nv12_direct_is_refused_where_it_does_not_apply(ms, cuda):
    comp, cfg, _ = make_rig(ms, "mini6", simple_kernels=True)
    nv = [[to_dev(synth.nv12_frame(cfg["w"], cfg["h"], i)) for i in range(cfg["n"])]]
    with pytest.raises(ms.MsError):
        comp.stitch_nv12(nv, out8u=[torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda")])
    comp.close()


# ---- frames of one view that do NOT share row step / alignment: the per-frame forms of the remap kernels ---------------------------------
# ms_stitch picks the shared-offset kernels (k_warp_s / k_stage1_s) when every frame of a view has the same row step and the same address modulo 4 -- what every
# other test and the bench hand over.  A caller with per-frame ROI views of buffers of different pitch (cv::cuda::GpuMat ROIs: `data + y * step`, any step,
# cuda_types.hpp:95-107) or byte-offset buffers lands in k_warp_t<., aligned>, k_warp_t<., unaligned> and the unshared k_stage1_t: same pixels
# (cudawarping/src/cuda/remap.cu:56-86, border_interpolate.hpp:698-717), other address arithmetic.
def _laid_out(frame_np, step_extra, offset):
    """the frame in a flat device buffer with row step = w * 3 + step_extra, starting `offset` bytes in (torch allocations are 512-byte aligned: address mod 4 = offset mod 4)"""
    h, w, _ = frame_np.shape
    step = w * 3 + step_extra
    flat = torch.full((offset + h * step + 64,), 201, dtype=torch.uint8, device="cuda")        # (201: a wrong read beyond a row shows)
    v = flat[offset:offset + h * step].as_strided((h, w, 3), (step, 3, 1))
    v.copy_(torch.from_numpy(np.ascontiguousarray(frame_np)).cuda())
    assert v.data_ptr() % 4 == offset % 4 and v.stride(0) == step
    return v


def _layouts(kind, t):
    if kind == "uniform":
        return 0, 0
    if kind == "steps":         # another row step in every frame, all 4-byte aligned
        return 4 * (1 + t % 3) + (12 if t % 2 else 0), 4 * (t % 5)
    if kind == "steps_odd":     # odd row steps: every ROW of a frame starts at another address modulo 4
        return 1 + 2 * (t % 4), t % 4
    assert kind == "offsets"    # one row step, the start addresses 0 / 1 / 2 / 3 modulo 4
    return 8, (t % 4) if t else 0


@pytest.mark.parametrize("rig,cpw,nf,expect", [
    ("mini6", False, 2, "aligned"), ("mini6", False, 3, "aligned"), ("mini6", False, 32, "aligned"), ("mini6", True, 3, "aligned"), ("mini6", True, 32, "aligned"),
    ("mini4", False, 3, None), ("mini4", True, 2, None), ("cfg2", False, 3, "aligned"), ("cfg2", True, 2, "aligned"), ("cfg5", False, 2, "unaligned")])
def test_frames_with_unequal_row_step_or_alignment_take_the_per_frame_kernels(ms, cuda, oracle, rig, cpw, nf, expect):
    comp, cfg, gains = make_rig(ms, rig, enable_cpw=cpw, max_frames=nf)
    meshes = None
    if cpw:
        meshes = []
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            mx, my = synth.mesh(r.width, r.height, 10, 10, phase=0.2 * i, amp=4.0 if rig.startswith("mini") else 8.0)
            comp.set_mesh(i, mx, my)
            meshes.append(tuple(host(m) for m in comp.mesh_maps(i)))       # the dense maps of convertMeshesToMap (checked against the oracle's in test_tables_gpu.py)
    pg = comp.pano_geom()
    shape = (pg.dst_roi_final.height, pg.dst_roi_final.width, 3)
    n_sets = min(nf, 4)
    sets = [[synth.frame(cfg["w"], cfg["h"], i, t) for i in range(cfg["n"])] for t in range(n_sets)]
    results, kernels = {}, {}
    for kind in ("uniform", "steps", "steps_odd", "offsets"):
        frames = [[_laid_out(sets[t % n_sets][i], *_layouts(kind, t)) for i in range(cfg["n"])] for t in range(nf)]
        outs = [torch.full(shape, -9, dtype=torch.int16, device=cuda) for _ in range(nf)]
        comp.stitch(frames, out16s=outs)
        torch.cuda.synchronize()
        results[kind] = outs
        kernels[kind] = comp.stitch_kernels()
        del frames
    # which kernels ran: the shared-offset forms for uniform frames, the per-frame forms otherwise (with CPW: the first remap; the mesh remap reads the context's own
    # stage images and stays shared)
    which = 1 if cpw else 0
    assert kernels["uniform"][which].startswith("shared_"), kernels
    if cpw:
        assert all(k[0] == "shared_aligned" for k in kernels.values()), kernels
    else:
        assert all(k[1] == "none" for k in kernels.values()), kernels
    # (uniform frames take the aligned shared form at every minification; which per-frame form a rig falls back to follows its minification: config 5 the unaligned one)
    fb = kernels["steps"][which]
    assert fb.startswith("per_frame_") and kernels["steps_odd"][which] == fb, kernels
    al = fb[len("per_frame_"):]
    assert expect is None or al == expect, kernels
    # one row step, other start alignment: the aligned form needs the per-frame kernel; the unaligned form only needs equal steps and stays shared (the first CPW
    # remap has no shared unaligned form)
    assert kernels["offsets"][which] == ("per_frame_aligned" if al == "aligned" else ("per_frame_unaligned" if cpw else "shared_unaligned")), kernels
    for kind in ("steps", "steps_odd", "offsets"):
        for t in range(nf):
            assert torch.equal(results[kind][t], results["uniform"][t]), (kind, t)
    assert n_sets == 1 or not torch.equal(results["uniform"][0], results["uniform"][1])
    # ... and against the oracle (full-size rigs: first and last frame; cfg5's oracle parity at full size is test_full_size_config5_and_config3_match_oracle)
    if rig != "cfg5":
        for t in (sorted({0, nf - 1}) if rig == "cfg2" else range(min(nf, n_sets))):
            ref16, _ = run_oracle(oracle, comp, cfg, gains, sets[t % n_sets], meshes)
            for kind in ("steps_odd", "offsets"):
                assert np.array_equal(host(results[kind][t]), ref16), (kind, t)
    comp.close()
