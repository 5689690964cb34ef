"""ms_create_mesh / ms_mesh_saliency (the CPW mesh optimiser, MeshWarper::createMesh after feature matching,
360_stitcher/meshwarper.cpp:279-301) against oracle/mesh_oracle.py.

Tolerances: the triangle statistics are exact integers and the salience formula is evaluated in the same order, so saliences must be
bit-equal; the solve is fp64 CG with a different summation order than the oracle's, vertices must agree to 1e-3 px (the
observed gap is ~1e-7 px)."""
import numpy as np
import pytest
import torch

import mesh_oracle as mo
import synth
from helpers import host, make_rig, to_dev
from test_mesh_oracle import rig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,M,N", [(90, 60, 6, 5), (960, 627, 10, 10), (333, 207, 40, 40), (64, 64, 2, 2)])
def test_saliency_is_bit_equal_to_oracle(ms, cuda, w, h, M, N):
    rng = np.random.default_rng(w + M)
    yy, xx = np.mgrid[0:h, 0:w]
    im = np.clip(128 + 80 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 13.0)[..., None] + rng.integers(-40, 40, (h, w, 3)), 0, 255).astype(np.uint8)
    if M == 10:
        im[:, : w // 3] = 0                       # flat cells: variance 0, salience sqrt(0.5)
    got = ms.mesh_saliency(to_dev(im), M, N)
    want = mo.saliency(im, M, N)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert ok.sum() == sum(all(0 <= j + dx < M and 0 <= i + dy < N for dx, dy in tri) for i in range(N) for j in range(M) for tri in mo.TRIANGLES)
    assert np.array_equal(got[ok], want[ok])
    if M == 10:
        assert got[5, 1, 0] == np.float32(np.sqrt(0.5))


def test_triangle_masks_equal_fill_convex_poly(ms, cuda):
    """The device kernel's closed-form row spans against the oracle's general cv::fillConvexPoly restatement (drawing.cpp:1109-1271): every
    cell size up to 24 x 24 (incl. the degenerate 1-pixel-wide / 1-pixel-high cells whose diagonal clipLine rejects), the cell sizes of the
    10 x 10 and 40 x 40 meshes on 1080p-class views, and very flat / very tall cells."""
    sizes = [(w, h) for w in range(1, 25) for h in range(1, 25)] + [(213, 120), (106, 69), (49, 27), (24, 16), (640, 37), (37, 640), (1, 300), (300, 1), (333, 332)]
    for (w, h) in sizes:
        got, cnt = ms.mesh_triangle_masks(w, h)
        for t in range(8):
            ref = mo.triangle_mask(t, np.float32(w) + np.float32(0.4), np.float32(h) + np.float32(0.7))      # float cell sizes truncate to (w, h)
            assert ref.shape == (h, w)
            assert np.array_equal(got[t], ref), (t, w, h)
            assert cnt[t] == int(np.count_nonzero(ref))


def test_saliency_accepts_pitched_views(ms, cuda):
    rng = np.random.default_rng(1)
    big = torch.from_numpy(rng.integers(0, 256, (80, 140, 3), dtype=np.uint8)).cuda()
    view = big[7:67, 11:101]
    assert np.array_equal(ms.mesh_saliency(view, 6, 5), mo.saliency(host(view), 6, 5), equal_nan=True)


@pytest.mark.parametrize("n,M,N,seed", [(3, 6, 5, 0), (2, 4, 7, 1), (6, 10, 10, 2)])
def test_create_mesh_matches_oracle(ms, cuda, n, M, N, seed):
    images, matches = rig(n=n, seed=seed)
    # GLOBAL_DIST scaled to the 90 x 60 test views (30 px would unpin half of every view, see the ill-conditioned test below)
    prm = ms.mesh_default_params(mesh_cols=M, mesh_rows=N, focal_length=60.0, theta_rule=1, global_dist=8)
    mx, my, info = ms.create_mesh([to_dev(im) for im in images], matches, prm)
    rx, ry, rinfo = mo.create_mesh(images, matches, M, N, focal=60.0, global_dist=8, theta_fn=lambda s, d: mo.generic_theta(s, d, n))
    assert (info["rows"], info["cols"], info["nnz"]) == (rinfo["rows"], rinfo["cols"], rinfo["nnz"])
    assert np.abs(mx - rx).max() < 1e-3 and np.abs(my - ry).max() < 1e-3, (np.abs(mx - rx).max(), np.abs(my - ry).max())
    # and against a solver-independent answer: the dense least-squares solution of the oracle's system
    A, b = mo.assemble(images, matches, M, N, focal=60.0, global_dist=8, theta_fn=lambda s, d: mo.generic_theta(s, d, n)).csr()
    dx, dy = mo.vector_to_mesh(np.linalg.lstsq(A.toarray(), b, rcond=None)[0], n, M, N)
    assert np.abs(mx - dx).max() < 1e-3 and np.abs(my - dy).max() < 1e-3
    assert 0 < info["iterations"] < 2 * info["cols"] and info["error"] < 2.3e-16
    assert abs(info["iterations"] - rinfo["iterations"]) <= max(3, rinfo["iterations"] // 50)
    # the matches do move the mesh
    gx = np.array([j * images[0].shape[1] // (M - 1) for j in range(M)], np.float32)
    assert np.abs(mx - gx[None, None, :]).max() > 1.0


def test_ill_conditioned_system_stops_where_eigen_stops(ms, cuda):
    """GLOBAL_DIST = 30 on 90 x 60 views unpins every vertex near a feature; what holds them is the 5e-5 smoothness weight (condition
    number ~4e6, coordinates in the 1e4 range).  CG does not converge in Eigen's 2 * cols iterations: both sides must run exactly that many
    and still agree to 1e-3 relative to the solution's scale."""
    images, matches = rig(n=6, seed=2)
    prm = ms.mesh_default_params(focal_length=60.0, theta_rule=1)
    mx, my, info = ms.create_mesh([to_dev(im) for im in images], matches, prm)
    rx, ry, rinfo = mo.create_mesh(images, matches, 10, 10, focal=60.0, theta_fn=lambda s, d: mo.generic_theta(s, d, 6))
    assert info["iterations"] == rinfo["iterations"] == 2 * info["cols"]
    scale = max(np.abs(rx).max(), np.abs(ry).max())
    assert scale > 1e3
    assert np.abs(mx - rx).max() < 1e-3 * scale and np.abs(my - ry).max() < 1e-3 * scale
    assert 0.1 * rinfo["error"] < info["error"] < 10 * rinfo["error"] and info["error"] > 1e-12      # not converged, on either side


def test_reference_rig_rule_and_temporal_term(ms, cuda):
    images, matches = rig(n=6, seed=4)
    rng = np.random.default_rng(9)
    temporal = [[(x, y, x + rng.normal(0, 2), y + rng.normal(0, 2)) for (x, y, _, _, _) in m[:5]] for m in matches]
    alphas = (1.0, 0.01, 0.00005, 0.25)
    prm = ms.mesh_default_params(mesh_cols=7, mesh_rows=6, focal_length=40.0, compose_scale=0.5, work_scale=0.25, alphas=alphas, global_dist=8)
    mx, my, info = ms.create_mesh([to_dev(im) for im in images], matches, prm, temporal=temporal)
    rx, ry, rinfo = mo.create_mesh(images, matches, 7, 6, alphas=alphas, focal=40.0, compose_scale=0.5, work_scale=0.25, temporal=temporal, global_dist=8)
    assert info["rows"] == rinfo["rows"] and info["nnz"] == rinfo["nnz"]
    assert np.abs(mx - rx).max() < 1e-3 and np.abs(my - ry).max() < 1e-3
    # without the temporal lists the temporal weight is inert, as USE_TEMPORAL && prev_features.empty() in the reference
    mx0, _, info0 = ms.create_mesh([to_dev(im) for im in images], matches, prm)
    assert info0["rows"] == info["rows"] - 2 * 5 * 6 and np.abs(mx0 - mx).max() > 1e-3


def test_iteration_cap_and_reproducibility(ms, cuda):
    images, matches = rig(n=3, seed=7)
    views = [to_dev(im) for im in images]
    prm = ms.mesh_default_params(mesh_cols=6, mesh_rows=5, focal_length=60.0, theta_rule=1, max_iterations=9, global_dist=8)
    mx, my, info = ms.create_mesh(views, matches, prm)
    A, b = mo.assemble(images, matches, 6, 5, focal=60.0, global_dist=8, theta_fn=lambda s, d: mo.generic_theta(s, d, 3)).csr()
    x, it, err = mo.lscg(A, b, max_iterations=9)
    rx, ry = mo.vector_to_mesh(x, 3, 6, 5)
    assert info["iterations"] == it == 9
    assert np.abs(mx - rx).max() < 1e-3 and abs(info["error"] - err) < 1e-6 * err + 1e-12
    # fixed-order reductions: the full solve is reproducible bit for bit
    prm.max_iterations = 0
    a = ms.create_mesh(views, matches, prm)
    b2 = ms.create_mesh(views, matches, prm)
    assert np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1]) and a[2] == b2[2]
    # a loose tolerance stops early, exactly where the oracle's restatement of Eigen's loop stops
    prm.tolerance = 1e-4
    _, _, info_t = ms.create_mesh(views, matches, prm)
    _, it_t, _ = mo.lscg(A, b, tolerance=1e-4)
    assert abs(info_t["iterations"] - it_t) <= 1 and info_t["iterations"] < a[2]["iterations"]


def test_zero_right_hand_side_and_bad_arguments(ms, cuda):
    images, _ = rig(n=2, seed=2)
    views = [to_dev(im) for im in images]
    prm = ms.mesh_default_params(mesh_cols=4, mesh_rows=4, alphas=(1.0, 0.0, 0.00005, 0.0))
    mx, my, info = ms.create_mesh(views, [[], []], prm)       # no matches, no global term: b = 0 -> x = 0 (Eigen returns at rhsNorm2 == 0)
    assert info["iterations"] == 0 and not mx.any() and not my.any()
    with pytest.raises(ms.MsError):
        ms.create_mesh(views, [[], []], ms.mesh_default_params(mesh_cols=1))
    with pytest.raises(ms.MsError):
        ms.create_mesh(views, [[], []], ms.mesh_default_params(theta_rule=5))
    with pytest.raises(ms.MsError):
        ms.create_mesh([views[0][:3, :3].contiguous(), views[1]], [[], []], ms.mesh_default_params())


def test_solved_mesh_drives_the_compositor(ms, cuda, oracle):
    """recalibrateMesh without the feature front-end: warp the frames, solve the mesh from (synthetic) matches, hand it to the compositor."""
    comp, cfg, gains = make_rig(ms, "mini6", enable_cpw=True)
    n = cfg["n"]
    frames_np = [synth.frame(cfg["w"], cfg["h"], i, 2) for i in range(n)]
    rois = [comp.view_geom(i).roi for i in range(n)]
    warped = [ms.remap(to_dev(frames_np[i]), *comp.maps(i)) for i in range(n)]       # images[idx] of createMesh (meshwarper.cpp:66-72)
    rng = np.random.default_rng(3)
    matches = []
    for i in range(n):
        d = (i + 1) % n
        off = (rois[d].x - rois[i].x) % cfg["out_w"]
        lst = []
        narrow = rois[i].width < cfg["out_w"] // 2 and rois[d].width < cfg["out_w"] // 2      # the view straddling +-pi spans the panorama
        for _ in range(20 if narrow and off + 4 < rois[i].width else 0):
            x1, y1 = rng.uniform(off + 2, rois[i].width - 2), rng.uniform(4, rois[i].height - 4)
            if x1 - off + 3 < rois[d].width:
                lst.append((x1, y1, x1 - off + 3.0, y1 + 1.5, d))           # a 3 px / 1.5 px parallax the mesh has to absorb
        matches.append(lst)
    M = N = 8
    prm = ms.mesh_default_params(mesh_cols=M, mesh_rows=N, focal_length=synth.warp_scale(cfg["out_w"]), theta_rule=1)
    assert sum(len(m) for m in matches) >= 40
    mx, my, info = ms.create_mesh(warped, matches, prm)
    rx, ry, _ = mo.create_mesh([host(w) for w in warped], matches, M, N, focal=synth.warp_scale(cfg["out_w"]),
                               theta_fn=lambda s, d: mo.generic_theta(s, d, n))
    assert np.abs(mx - rx).max() < 1e-3 and np.abs(my - ry).max() < 1e-3
    for i in range(n):
        comp.set_mesh(i, mx[i], my[i])
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames_np]], out16s=[out16])
    torch.cuda.synchronize()
    b, _ = __import__("helpers").oracle_blender_from(oracle, comp, cfg)
    for i in range(n):
        xm, ym = [host(t) for t in comp.maps(i)]
        dmx, dmy = [host(t) for t in comp.mesh_maps(i)]
        b.stitch_online(i, frames_np[i], xm, ym, gains[i], dmx, dmy)
    ref16, _ = b.blend()
    b.close()
    assert np.array_equal(host(out16), ref16)
    comp.close()


@pytest.mark.parametrize("case", ["single_view_temporal", "mesh_2x2", "border_matches", "mesh_40x40"])
def test_create_mesh_edge_cases(ms, cuda, case):
    rng = np.random.default_rng(11)
    if case == "single_view_temporal":
        images, _ = rig(n=1, seed=8)
        matches = [[]]
        temporal = [[(x, y, x + 1.5, y - 0.5) for x, y in zip(rng.uniform(5, 85, 15), rng.uniform(5, 55, 15))]]
        kw = dict(M=6, N=5, alphas=(1.0, 0.01, 0.00005, 0.5))
    elif case == "mesh_2x2":
        images, matches = rig(n=2, seed=9)
        temporal, kw = None, dict(M=2, N=2, alphas=mo.DEFAULT_ALPHAS)
    elif case == "border_matches":
        images, matches = rig(n=3, seed=10)
        h, w = images[0].shape[:2]
        # points on the image border, outside it (ignored, meshwarper.cpp:641-644) and on the last mesh line, where float rounding can put the cell index at M - 1
        matches[0] += [(0.0, 0.0, 0.0, 0.0, 1), (w - 1e-3, h - 1e-3, w - 1e-3, h - 1e-3, 1), (-1.0, 5.0, 3.0, 5.0, 1), (3.0, 5.0, float(w), 5.0, 1),
                       (np.nextafter(np.float32(w), np.float32(0)), 10.0, 20.0, 10.0, 1), (float("nan"), 5.0, 3.0, 5.0, 1), (4.0, 5.0, 3.0, float("inf"), 1)]
        temporal, kw = None, dict(M=6, N=5, alphas=mo.DEFAULT_ALPHAS)
    else:
        images, matches = rig(n=2, seed=12, w=333, h=207)
        temporal, kw = None, dict(M=40, N=40, alphas=mo.DEFAULT_ALPHAS)
    n = len(images)
    prm = ms.mesh_default_params(mesh_cols=kw["M"], mesh_rows=kw["N"], focal_length=60.0, theta_rule=1, global_dist=8, alphas=kw["alphas"])
    mx, my, info = ms.create_mesh([to_dev(im) for im in images], matches, prm, temporal=temporal)
    rx, ry, rinfo = mo.create_mesh(images, matches, kw["M"], kw["N"], alphas=kw["alphas"], focal=60.0, global_dist=8, temporal=temporal,
                                   theta_fn=lambda s, d: mo.generic_theta(s, d, n))
    assert (info["rows"], info["cols"], info["nnz"]) == (rinfo["rows"], rinfo["cols"], rinfo["nnz"])
    assert np.abs(mx - rx).max() < 1e-3 and np.abs(my - ry).max() < 1e-3, (np.abs(mx - rx).max(), np.abs(my - ry).max())
    assert abs(info["iterations"] - rinfo["iterations"]) <= max(3, rinfo["iterations"] // 50)


from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402
import os  # noqa: E402


@settings(max_examples=int(os.environ.get("MS_TEST_EXAMPLES", 12)), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(n=st.integers(1, 4), M=st.integers(2, 9), N=st.integers(2, 9), w=st.integers(40, 120), h=st.integers(36, 90), nm=st.integers(0, 25),
       gd=st.integers(2, 12), temporal=st.booleans(), rule=st.sampled_from([0, 1]), seed=st.integers(0, 10 ** 6))
def test_create_mesh_random_systems(ms, cuda, n, M, N, w, h, nm, gd, temporal, rule, seed):
    """Random views, mesh sizes, match lists (some out of range / on borders), weights and both theta rules: system structure identical to the
    oracle's, vertices within 1e-3 px (relative to the solution's scale when the draw is ill-conditioned), same stopping behaviour."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    images = [np.clip(128 + 70 * np.sin(xx / (5.0 + v) + yy / 9.0)[..., None] + rng.integers(-40, 40, (h, w, 3)), 0, 255).astype(np.uint8) for v in range(n)]
    matches = []
    for v in range(n):
        lst = []
        for _ in range(nm if n > 1 else 0):
            d = int(rng.integers(0, n))
            if d == v:
                d = (v + 1) % n
            lst.append((float(rng.uniform(-3, w + 3)), float(rng.uniform(-3, h + 3)), float(rng.uniform(-3, w + 3)), float(rng.uniform(-3, h + 3)), d))
        matches.append(lst)
    temp = [[(float(rng.uniform(0, w - 1)), float(rng.uniform(0, h - 1)), float(rng.uniform(0, w - 1)), float(rng.uniform(0, h - 1))) for _ in range(4)] for _ in range(n)] if temporal else None
    alphas = (float(rng.uniform(0.2, 2.0)), float(rng.uniform(0.005, 0.05)), float(rng.uniform(1e-5, 1e-3)), float(rng.uniform(0.05, 0.5)) if temporal else 0.0)
    prm = ms.mesh_default_params(mesh_cols=M, mesh_rows=N, focal_length=float(w), theta_rule=rule, global_dist=gd, alphas=alphas, compose_scale=1.0, work_scale=2.0)
    mx, my, info = ms.create_mesh([to_dev(im) for im in images], matches, prm, temporal=temp)
    theta = (lambda s, d: mo.reference_theta(s, d, n)) if rule == 0 else (lambda s, d: mo.generic_theta(s, d, n))
    rx, ry, rinfo = mo.create_mesh(images, matches, M, N, alphas=alphas, focal=float(w), global_dist=gd, compose_scale=1.0, work_scale=2.0, theta_fn=theta, temporal=temp)
    assert (info["rows"], info["cols"], info["nnz"]) == (rinfo["rows"], rinfo["cols"], rinfo["nnz"])
    both_converged = info["error"] < 1e-12 and rinfo["error"] < 1e-12
    if both_converged:
        scale = max(1.0, float(np.abs(rx).max()), float(np.abs(ry).max()))
        tol = 1e-3 * max(1.0, scale / 1e3)
        assert np.abs(mx - rx).max() < tol and np.abs(my - ry).max() < tol, (np.abs(mx - rx).max(), np.abs(my - ry).max(), scale, info, rinfo)
        # (iteration counts are not compared here: how many steps push ||A^T r|| below DBL_EPSILON * ||A^T b|| depends on the last bits of every
        # reduction, and either side may creep along just above the threshold until Eigen's cap)
    # CG in floating point may stagnate on one side and not the other (different summation orders; Eigen's own behaviour there is not pinned):
    # whatever the stopping point, a solve that reports convergence must BE the least-squares solution
    if info["error"] < 1e-12:
        A, b = mo.assemble(images, matches, M, N, alphas=alphas, focal=float(w), global_dist=gd, compose_scale=1.0, work_scale=2.0, theta_fn=theta, temporal=temp).csr()
        xg = np.stack([mx, my], -1).astype(np.float64).ravel()
        xd = np.linalg.lstsq(A.toarray(), b, rcond=None)[0]
        # (the meshes come back as float32: their rounding alone moves the residual by ~1e-6 * ||A|| * ||x||)
        assert np.linalg.norm(A @ xg - b) <= np.linalg.norm(A @ xd - b) * (1 + 1e-5) + 1e-4 * max(1.0, np.linalg.norm(b))
    else:
        assert info["iterations"] == 2 * info["cols"]          # not converged: Eigen's iteration cap, exactly
