"""The CPW mesh optimiser's oracle (oracle/mesh_oracle.py) checked on its own: the restated cv::fillConvexPoly against the geometry it
rasterises, the restated Eigen LSCG against a dense least-squares solve, and the assembled system's structure
(360_stitcher/meshwarper.cpp:389-709).  CPU only."""
import numpy as np
import pytest

import mesh_oracle as mo

F = np.float32


def tri_points(t, cw, ch):
    vi = [list(p) for p in mo.TRIANGLES[t]]
    if min(p[0] for p in vi) < 0:
        for p in vi:
            p[0] += 1
    if min(p[1] for p in vi) < 0:
        for p in vi:
            p[1] += 1
    return [(int(F(p[0]) * cw), int(F(p[1]) * ch)) for p in vi]


def signed_dist_inside(pts, x, y):
    """min over edges of the signed distance of (x, y) to the edge line, positive inside (either orientation)."""
    (x0, y0), (x1, y1), (x2, y2) = pts
    area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
    sgn = 1.0 if area > 0 else -1.0
    d = []
    for (ax, ay), (bx, by) in (((x0, y0), (x1, y1)), ((x1, y1), (x2, y2)), ((x2, y2), (x0, y0))):
        ln = np.hypot(bx - ax, by - ay)
        d.append(sgn * ((bx - ax) * (y - ay) - (by - ay) * (x - ax)) / ln)
    return np.minimum(np.minimum(d[0], d[1]), d[2])


@pytest.mark.parametrize("cw,ch", [(F(960) / F(9), F(627) / F(9)), (F(960) / F(39), F(627) / F(39)), (F(20), F(12)), (F(7.5), F(31.2))])
def test_triangle_masks_cover_the_triangle(cw, ch):
    for t in range(8):
        m = mo.triangle_mask(t, cw, ch) > 0
        assert m.shape == (int(ch), int(cw))
        yy, xx = np.mgrid[0:m.shape[0], 0:m.shape[1]]
        d = signed_dist_inside(tri_points(t, cw, ch), xx.astype(np.float64), yy.astype(np.float64))
        assert m[d > 0.75].all(), "pixels well inside the triangle are filled"
        assert not m[d < -1.5].any(), "pixels well outside are not (clipLine moves a clipped outline by up to a pixel)"
    # the two triangles of a quad (split by one diagonal) cover the cell
    for a, b in ((0, 1), (2, 3), (4, 5), (6, 7)):
        assert ((mo.triangle_mask(a, cw, ch) > 0) | (mo.triangle_mask(b, cw, ch) > 0)).all()


def test_fill_convex_poly_known_small_cases():
    # a 4 x 4 right triangle with the right angle at the bottom left; the far vertices lie one past the image and are clipped
    m = (mo.fill_convex_poly(4, 4, [(0, 4), (4, 4), (0, 0)]) > 0).astype(int)
    assert m.tolist() == [[1, 0, 0, 0], [1, 1, 0, 0], [1, 1, 1, 0], [1, 1, 1, 1]]
    # a polygon entirely outside leaves the mask empty, one covering it fills it
    assert not mo.fill_convex_poly(3, 5, [(10, 10), (12, 10), (10, 12)]).any()
    assert mo.fill_convex_poly(3, 5, [(-5, -5), (30, -5), (-5, 30)]).all()


def rig(n=3, w=90, h=60, seed=0):   # n = 1: no matches (there is no other view)
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    images = []
    for v in range(n):
        base = 128 + 60 * np.sin(xx / (7.0 + v) + yy / 11.0)
        im = np.stack([base + 20 * c for c in range(3)], -1) + rng.integers(-30, 30, (h, w, 3))
        images.append(np.clip(im, 0, 255).astype(np.uint8))
    matches = []
    for v in range(n):
        dst = (v + 1) % n
        pts = []
        for _ in range(12 if n > 1 else 0):
            x1, y1 = rng.uniform(w * 0.8, w - 1), rng.uniform(4, h - 4)
            pts.append((x1, y1, x1 - w * 0.7 + rng.normal(0, 1.5), y1 + rng.normal(0, 1.0), dst))
        matches.append(pts)
    return images, matches


def test_system_structure():
    images, matches = rig()
    M, N = 6, 5
    S = mo.assemble(images, matches, M, N, focal=60.0, theta_fn=lambda s, d: mo.generic_theta(s, d, 3))
    A, b = S.csr()
    n = len(images)
    assert S.cols == 2 * M * N * n
    tri_rows = 2 * sum(all(0 <= j + dx < M and 0 <= i + dy < N for dx, dy in tri)
                       for i in range(N) for j in range(M) for tri in mo.TRIANGLES)
    assert S.rows == n * (2 * 12 + 2 * M * N + tri_rows)
    per_row = np.diff(A.indptr)
    assert per_row.max() == 8 and per_row.min() == 1           # local rows 8, smoothness 6, global 1
    # local rows: bilinear weights of each end point sum to +-sqrt(alpha_local)
    first = A[0].toarray().ravel()
    assert np.isclose(first[first > 0].sum(), 1.0, atol=1e-5) and np.isclose(first[first < 0].sum(), -1.0, atol=1e-5)
    # smoothness rows come in identical pairs (the reference writes the same six coefficients to the x and the y row)
    r0 = 2 * 12 + 2 * M * N
    assert (A[r0].toarray() == A[r0 + 1].toarray()).all() and b[r0] == 0 and b[r0 + 1] == 0


def test_lscg_matches_dense_least_squares():
    images, matches = rig(seed=3)
    M, N = 5, 5
    S = mo.assemble(images, matches, M, N, focal=60.0, theta_fn=lambda s, d: mo.generic_theta(s, d, 3))
    A, b = S.csr()
    x, it, err = mo.lscg(A, b)
    ref = np.linalg.lstsq(A.toarray(), b, rcond=None)[0]
    assert it > 10
    assert np.abs(x - ref).max() < 1e-6, (it, err, np.abs(x - ref).max())
    # an iteration cap is honoured and reported the way Eigen reports it
    x2, it2, err2 = mo.lscg(A, b, max_iterations=7)
    assert it2 == 7 and err2 > err


def test_no_matches_keeps_vertices_on_the_global_grid_within_smoothness_pull():
    images, _ = rig(n=2, seed=5)
    M, N = 5, 4
    mx, my, info = mo.create_mesh(images, [[], []], M, N)
    h, w = images[0].shape[:2]
    gx = np.array([j * w // (M - 1) for j in range(M)], np.float32)
    gy = np.array([i * h // (N - 1) for i in range(N)], np.float32)
    # alpha_smooth (5e-5) is tiny next to alpha_global (1e-2): vertices stay within a few pixels of the global grid
    assert np.abs(mx - gx[None, None, :]).max() < 6 and np.abs(my - gy[None, :, None]).max() < 6
    assert info["cols"] == 2 * M * N * 2


def test_reference_theta_rule():
    assert mo.reference_theta(0, 1, 6) == F(float(F(1)) * (2 * mo.PI / 6))
    assert mo.reference_theta(0, 5, 6) == F(float(F(-1)) * (2 * mo.PI / 6))
    assert mo.reference_theta(3, 4, 6) == F(4.25 * (2 * mo.PI / 6)) and mo.reference_theta(4, 5, 6) == F(-0.25 * (2 * mo.PI / 6))
    assert mo.generic_theta(0, 3, 4) == F(-1 * (2 * mo.PI / 4)) or mo.generic_theta(0, 3, 4) == F(float(F(-1)) * (2 * mo.PI / 4))
