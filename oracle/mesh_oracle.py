"""CPU oracle of the CPW mesh optimiser (MeshWarper::createMesh after feature matching), numpy.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
(video-stitcher_amd) never imports this module.

What it restates (reference file:line):
  * calcLocalTerm          360_stitcher/meshwarper.cpp:596-709
  * calcGlobalTerm         360_stitcher/meshwarper.cpp:389-418
  * calcSmoothnessTerm     360_stitcher/meshwarper.cpp:421-593   (triangle masks: cv::fillConvexPoly,
                           sources/modules/imgproc/src/drawing.cpp:1109-1271 + Line/LineIterator :165-277, clipLine :97-148;
                           statistics: cv::meanStdDev, sources/modules/core/src/stat.cpp:1939-1943)
  * calcTemporalLocalTerm  360_stitcher/meshwarper.cpp:711-786
  * the solve              360_stitcher/meshwarper.cpp:294-301  Eigen::LeastSquaresConjugateGradient<SparseMatrix<double>>
  * convertVectorToMesh    360_stitcher/meshwarper.cpp:810-818

Pinning: PARITY UNPINNED.  The solver is Eigen (third party, NOT under /root/reference; the reference's README asks for the HEAD of
eigenteam/eigen-git-mirror, no pinned version) and no test of the reference holds a mesh.  `lscg` below restates the published algorithm of
Eigen/src/IterativeLinearSolvers/LeastSquareConjugateGradient.h (Eigen 3.3: CG on the normal equations without forming A^T A, Jacobi
preconditioner 1/||A_col||^2, x0 = 0, tolerance = DBL_EPSILON on ||A^T r|| / ||A^T b||, at most 2*cols iterations); tests additionally
check the solution against a dense numpy least-squares solve.  The coefficient arithmetic follows the reference's float expressions
operation by operation (np.float32 scalars; x86-64 gcc does not contract them).
"""
import math

import numpy as np

F = np.float32
XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT

DEFAULT_ALPHAS = (1.0, 0.01, 0.00005, 0.0)      # defs.h:69   local, global, smoothness, temporal
DEFAULT_GLOBAL_DIST = 30                        # defs.h:71
PI = 3.1415926535897932384626                   # defs.h:76

# meshwarper.cpp:441-486: the 8 triangles around vertex (j, i), offsets (x, y) of V1, V2 (= the vertex), V3
TRIANGLES = (
    ((-1, 0), (0, 0), (-1, -1)), ((0, -1), (0, 0), (-1, -1)), ((0, -1), (0, 0), (1, -1)), ((1, 0), (0, 0), (1, -1)),
    ((-1, 0), (0, 0), (-1, 1)), ((0, 1), (0, 0), (-1, 1)), ((0, 1), (0, 0), (1, 1)), ((1, 0), (0, 0), (1, 1)),
)


# ----------------------------------------------------------------------------- cv::fillConvexPoly (8-connected, shift 0)

def _clip_line(w, h, p1, p2):
    """clipLine(Size2l, Point2l&, Point2l&)  drawing.cpp:97-148"""
    x1, y1 = p1
    x2, y2 = p2
    right, bottom = w - 1, h - 1
    if w <= 0 or h <= 0:
        return False, p1, p2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))      # (int64)(double) truncates toward zero
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def _line8(img, pt1, pt2):
    """Line(img, pt1, pt2, color, 8) = LineIterator(img, pt1, pt2, 8, left_to_right=true)  drawing.cpp:165-277, imgproc.hpp:4810-4816"""
    h, w = img.shape
    (x1, y1), (x2, y2) = pt1, pt2
    if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
        ok, (x1, y1), (x2, y2) = _clip_line(w, h, (x1, y1), (x2, y2))
        if not ok:
            return
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:                                   # left_to_right: start from the left end point
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    step_minus, step_plus = (1, 0), (0, sy)      # minusStep = one pixel along x, plusStep = one row
    if dy > dx:
        dx, dy = dy, dx
        step_minus, step_plus = step_plus, step_minus
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    x, y = x1, y1
    for _ in range(dx + 1):
        img[y, x] = 255
        m = err < 0
        err += minus_delta + (plus_delta if m else 0)
        x += step_minus[0] + (step_plus[0] if m else 0)
        y += step_minus[1] + (step_plus[1] if m else 0)


def fill_convex_poly(h, w, pts):
    """cv::fillConvexPoly(mask(h, w) = 0, pts, 255) with the default line_type 8, shift 0  (drawing.cpp:1109-1271)."""
    img = np.zeros((h, w), np.uint8)
    n = len(pts)
    v = [(int(x), int(y)) for x, y in pts]
    p0 = v[-1]
    ymin = ymax = v[0][1]
    xmin = xmax = v[0][0]
    imin = 0
    for i, p in enumerate(v):
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax = max(ymax, p[1]); xmax = max(xmax, p[0]); xmin = min(xmin, p[0])
        _line8(img, p0, p)
        p0 = p
    if n < 3 or xmax < 0 or ymax < 0 or xmin >= w or ymin >= h:
        return img
    ymax = min(ymax, h - 1)
    e = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=n - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = n
    y = ymin
    delta1 = delta2 = XY_ONE >> 1
    while True:
        for i in range(2):
            if y >= e[i]["ye"]:
                idx0, di = e[i]["idx"], e[i]["di"]
                idx = idx0 + di
                if idx >= n:
                    idx -= n
                while True:
                    edges -= 1
                    if edges < 0:                                   # `for (; edges-- > 0; )`
                        break
                    ty = v[idx][1]
                    if ty > y:
                        xs, xe = v[idx0][0] << XY_SHIFT, v[idx][0] << XY_SHIFT
                        e[i]["ye"] = ty
                        num, den = (xe - xs) * 2 + (ty - y), 2 * (ty - y)
                        q = abs(num) // den                                     # C++ int64 division truncates toward zero
                        e[i]["dx"] = q if num >= 0 else -q
                        e[i]["x"] = xs
                        e[i]["idx"] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= n:
                        idx -= n
        if edges < 0:
            break
        if y >= 0:
            l, r = (1, 0) if e[0]["x"] > e[1]["x"] else (0, 1)
            xx1 = (e[l]["x"] + delta1) >> XY_SHIFT
            xx2 = (e[r]["x"] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < w:
                xx1 = max(xx1, 0)
                xx2 = min(xx2, w - 1)
                if xx2 >= xx1:
                    img[y, xx1:xx2 + 1] = 255
        e[0]["x"] += e[0]["dx"]
        e[1]["x"] += e[1]["dx"]
        y += 1
        if y > ymax:
            break
    return img


def triangle_mask(t, cell_w, cell_h):
    """meshwarper.cpp:527-551: mask(cell_height, cell_width) with triangle t filled; cell sizes are the FLOAT cell sizes."""
    vi = [list(p) for p in TRIANGLES[t]]
    if min(p[0] for p in vi) < 0:
        for p in vi:
            p[0] += 1
    if min(p[1] for p in vi) < 0:
        for p in vi:
            p[1] += 1
    pts = [(int(F(p[0]) * cell_w), int(F(p[1]) * cell_h)) for p in vi]     # Point(float, float) -> int truncation
    return fill_convex_poly(int(cell_h), int(cell_w), pts)


# ----------------------------------------------------------------------------- terms

def saliency(image, M, N):
    """meshwarper.cpp:497-563 for every vertex and triangle of one view: (N, M, 8) float32, NaN where the triangle leaves the mesh."""
    hgt, wid = image.shape[:2]
    width, height = F(wid), F(hgt)
    cw, ch = width / F(M - 1), height / F(N - 1)
    masks = [triangle_mask(t, cw, ch) > 0 for t in range(8)]
    sal = np.full((N, M, 8), np.nan, np.float32)
    for i in range(N):
        for j in range(M):
            for t in range(8):
                tot = [(j + dx, i + dy) for dx, dy in TRIANGLES[t]]
                if any(x < 0 or y < 0 or x >= M or y >= N for x, y in tot):
                    continue
                vx = [F(x) * cw for x, _ in tot]
                vy = [F(y) * ch for _, y in tot]
                cx, cy = int(min(vx)), int(min(vy))
                crop = image[cy:cy + int(ch), cx:cx + int(cw)].astype(np.int64)
                m = masks[t]
                assert crop.shape[:2] == m.shape, "crop leaves the image (cv::Mat::operator() would assert)"
                nz = int(m.sum())
                var = []
                for c in range(3):
                    px = crop[..., c][m]
                    s, sq = float(px.sum()), float((px * px).sum())
                    scale = 1.0 / nz if nz else 0.0
                    mean = s * scale
                    dev = math.sqrt(max(sq * scale - mean * mean, 0.0))          # stat.cpp:1939-1943
                    var.append(dev * dev)                                        # cv::pow(deviation, 2)
                nrm = math.sqrt(var[0] * var[0] + var[1] * var[1] + var[2] * var[2])   # norm(variance, NORM_L2)
                sal[i, j, t] = F(math.sqrt(nrm + 0.5))
    return sal


def reference_theta(src, dst, n_views, wrap_around=True):
    """meshwarper.cpp:617-629 (hard-coded for the reference's 6-camera rig; view 3 straddles the +-pi split)."""
    theta = F(dst - src)
    if src == 0 and dst == n_views - 1 and wrap_around:
        theta = F(-1)
    if src == 3:
        theta = F(4.25)
    if src == 4:
        theta = F(-0.25)
    return F(float(theta) * (2 * PI / 6))


def generic_theta(src, dst, n_views):
    d = dst - src
    if d > n_views / 2:
        d -= n_views
    if d < -n_views / 2:
        d += n_views
    return F(float(F(d)) * (2 * PI / n_views))


class System:
    """Rows of |A x - b|^2 in the reference's order; entries as (row, col, double(value))."""

    def __init__(self, n_views, M, N):
        self.M, self.N, self.n_views = M, N, n_views
        self.cols = 2 * N * M * n_views
        self.r, self.c, self.v, self.b = [], [], [], []
        self.rows = 0

    def put(self, row_off, col, val):
        self.r.append(self.rows + row_off); self.c.append(int(col)); self.v.append(float(val))

    def finish_rows(self, b0, b1):
        self.b += [float(b0), float(b1)]
        self.rows += 2

    def csr(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.v, (self.r, self.c)), shape=(self.rows, self.cols)), np.array(self.b, np.float64)


def _cell(x, y, w, h, M, N):
    t = int(math.floor(F(F(y) * F(N - 1)) / F(h)))
    l = int(math.floor(F(F(x) * F(M - 1)) / F(w)))
    top = F(F(t) * F(h)) / F(N - 1)
    bot = top + F(h) / F(N - 1)
    left = F(F(l) * F(w)) / F(M - 1)
    right = left + F(w) / F(M - 1)
    u = (F(x) - left) / (right - left)
    v = (F(y) - top) / (bot - top)
    return t, l, u, v


def local_term(S, matches, sizes, idx, alpha, focal, compose_scale, work_scale, theta_fn):
    """calcLocalTerm  meshwarper.cpp:596-709.  matches: (x1, y1, x2, y2, dst); sizes[v] = (w, h) of the warped views."""
    M, N = S.M, S.N
    f = F(focal)
    a = F(math.sqrt(F(alpha)))
    scale = F(compose_scale / work_scale)
    one = F(1)
    for (x1, y1, x2, y2, dst) in matches:
        src = idx
        w1, h1 = F(sizes[src][0]), F(sizes[src][1])
        w2, h2 = F(sizes[dst][0]), F(sizes[dst][1])
        x1, y1, x2, y2 = F(x1), F(y1), F(x2), F(y2)
        if x1 < 0 or x2 < 0 or y1 < 0 or y2 < 0 or x1 >= w1 or x2 >= w2 or y1 >= h1 or y2 >= h2:
            continue
        if not np.isfinite([x1, y1, x2, y2]).all():
            continue        # NaN passes the reference's range test and is undefined behaviour there; skipped (stated deviation)
        t1, l1, u1, v1 = _cell(x1, y1, w1, h1, M, N)
        t2, l2, u2, v2 = _cell(x2, y2, w2, h2, M, N)
        if l1 + 1 >= M or l2 + 1 >= M or t1 + 1 >= N or t2 + 1 >= N:
            continue        # float rounding put the point on the last mesh line: the reference would index past the row (undefined); skipped
        theta = theta_fn(src, dst)
        base1, base2 = M * N * src, M * N * dst
        for k in (0, 1):
            S.put(k, 2 * (l1 + M * t1 + base1) + k, (one - u1) * (one - v1) * a)
            S.put(k, 2 * (l1 + 1 + M * t1 + base1) + k, u1 * (one - v1) * a)
            S.put(k, 2 * (l1 + M * (t1 + 1) + base1) + k, v1 * (one - u1) * a)
            S.put(k, 2 * (l1 + 1 + M * (t1 + 1) + base1) + k, u1 * v1 * a)
            S.put(k, 2 * (l2 + M * t2 + base2) + k, -(one - u2) * (one - v2) * a)
            S.put(k, 2 * (l2 + 1 + M * t2 + base2) + k, -u2 * (one - v2) * a)
            S.put(k, 2 * (l2 + M * (t2 + 1) + base2) + k, -v2 * (one - u2) * a)
            S.put(k, 2 * (l2 + 1 + M * (t2 + 1) + base2) + k, -u2 * v2 * a)
        S.finish_rows(theta * f * scale * a, 0.0)


def temporal_term(S, matches, sizes, idx, alpha):
    """calcTemporalLocalTerm  meshwarper.cpp:711-786.  matches: (x1, y1, x2, y2): this view now / in the previous calibration."""
    M, N = S.M, S.N
    a = F(math.sqrt(F(alpha)))
    one = F(1)
    w, h = F(sizes[idx][0]), F(sizes[idx][1])
    for m in matches:
        x1, y1, x2, y2 = F(m[0]), F(m[1]), F(m[2]), F(m[3])
        if x1 < 0 or x2 < 0 or y1 < 0 or y2 < 0 or x1 >= w or x2 >= w or y1 >= h or y2 >= h:
            continue
        if not np.isfinite([x1, y1, x2, y2]).all():
            continue
        t1, l1, u1, v1 = _cell(x1, y1, w, h, M, N)
        if l1 + 1 >= M or t1 + 1 >= N:
            continue
        base = M * N * idx
        for k in (0, 1):
            S.put(k, 2 * (l1 + M * t1 + base) + k, (one - u1) * (one - v1) * a)
            S.put(k, 2 * (l1 + 1 + M * t1 + base) + k, u1 * (one - v1) * a)
            S.put(k, 2 * (l1 + M * (t1 + 1) + base) + k, v1 * (one - u1) * a)
            S.put(k, 2 * (l1 + 1 + M * (t1 + 1) + base) + k, u1 * v1 * a)
        S.finish_rows(x2 * a, y2 * a)


def cv_round(x):
    return int(np.rint(np.float64(x)))       # saturate_cast<int>(float) = cvRound: nearest, ties to even


def global_term(S, points, size, idx, alpha, global_dist):
    """calcGlobalTerm  meshwarper.cpp:389-418.  points: integer (x, y) = Point(keypoint.pt) of the selected matches."""
    M, N = S.M, S.N
    a = F(math.sqrt(F(alpha)))
    col = N * M * 2 * idx
    for i in range(N):
        for j in range(M):
            x1 = F(j * size[0] // (M - 1))
            y1 = F(i * size[1] // (N - 1))
            tau = F(1)
            for (px, py) in points:
                dx, dy = float(F(px) - x1), float(F(py) - y1)
                if math.sqrt(dx * dx + dy * dy) < global_dist:
                    tau = F(0)
                    break
            S.put(0, col, a * tau)
            S.put(1, col + 1, a * tau)
            S.finish_rows(a * tau * x1, a * tau * y1)
            col += 2


def smoothness_term(S, sal, size, idx, alpha):
    """calcSmoothnessTerm  meshwarper.cpp:421-593 (both rows of a triangle carry the same coefficients, as in the reference)."""
    M, N = S.M, S.N
    a = F(math.sqrt(F(alpha)))
    width, height = F(size[0]), F(size[1])
    cw, ch = width / F(M - 1), height / F(N - 1)
    two, one = F(2), F(1)
    for i in range(N):
        for j in range(M):
            for t in range(8):
                tot = [(j + dx, i + dy) for dx, dy in TRIANGLES[t]]
                if any(x < 0 or y < 0 or x >= M or y >= N for x, y in tot):
                    continue
                V1x, V2x, V3x = [F(x) * cw for x, _ in tot]
                V1y, V2y, V3y = [F(y) * ch for _, y in tot]
                den = two * (V2x - V3x) * (V2y - V3y)
                u = (-V1x * V2y + V1x * V3y - V2x * V1y + two * V2x * V2y - V2x * V3y + V3x * V1y - V3x * V2y) / den
                v = (V1x * V2y - V1x * V3y - V2x * V1y + V2x * V3y + V3x * V1y - V3x * V2y) / den
                s = F(sal[i, j, t])
                cols = [2 * (x + M * y + M * N * idx) for x, y in tot]
                coef = [a * s, a * s, a * (u - v - one) * s, a * (u + v - one) * s, a * (-u + v) * s, a * (-u - v) * s]
                for k in (0, 1):
                    S.put(k, cols[0], coef[0]); S.put(k, cols[0] + 1, coef[1])
                    S.put(k, cols[1], coef[2]); S.put(k, cols[1] + 1, coef[3])
                    S.put(k, cols[2], coef[4]); S.put(k, cols[2] + 1, coef[5])
                S.finish_rows(0.0, 0.0)


def assemble(images, matches, M, N, alphas=DEFAULT_ALPHAS, global_dist=DEFAULT_GLOBAL_DIST, focal=1.0, compose_scale=1.0,
             work_scale=1.0, theta_fn=None, temporal=None, sal=None):
    """The loop of createMesh  meshwarper.cpp:279-292.  images[v]: warped view (h, w, 3) uint8; matches[v]: list of (x1,y1,x2,y2,dst)."""
    n = len(images)
    sizes = [(im.shape[1], im.shape[0]) for im in images]
    if theta_fn is None:
        theta_fn = lambda s, d: reference_theta(s, d, n)
    S = System(n, M, N)
    for idx in range(n):
        local_term(S, matches[idx], sizes, idx, alphas[0], focal, compose_scale, work_scale, theta_fn)
        pts = [(cv_round(m[0]), cv_round(m[1])) for m in matches[idx] if np.isfinite(m[0]) and np.isfinite(m[1]) and abs(m[0]) < 1e9 and abs(m[1]) < 1e9]
        global_term(S, pts, sizes[idx], idx, alphas[1], global_dist)
        smoothness_term(S, saliency(images[idx], M, N) if sal is None else sal[idx], sizes[idx], idx, alphas[2])
        if alphas[3] != 0.0 and temporal is not None:
            temporal_term(S, temporal[idx], sizes, idx, alphas[3])
    return S


# ----------------------------------------------------------------------------- Eigen::LeastSquaresConjugateGradient

def lscg(A, b, max_iterations=0, tolerance=0.0):
    """Published algorithm of Eigen 3.3 LeastSquareConjugateGradient.h (least_square_conjugate_gradient + the diagonal preconditioner).
    Returns x, iterations, error exactly as the solver object reports them."""
    m, n = A.shape
    At = A.T.tocsr()
    tol = tolerance if tolerance > 0 else np.finfo(np.float64).eps
    max_it = max_iterations if max_iterations > 0 else 2 * n
    d = np.asarray(A.multiply(A).sum(axis=0)).ravel()
    invdiag = np.where(d > 0, 1.0 / np.where(d > 0, d, 1.0), 1.0)
    x = np.zeros(n)
    residual = b - A @ x
    rhs_norm2 = float(np.dot(At @ b, At @ b))
    if rhs_norm2 == 0:
        return x, 0, 0.0
    threshold = tol * tol * rhs_norm2
    nr = At @ residual
    res_norm2 = float(np.dot(nr, nr))
    if res_norm2 < threshold:
        return x, 0, math.sqrt(res_norm2 / rhs_norm2)
    p = invdiag * nr
    abs_new = float(np.dot(nr, p))
    i = 0
    while i < max_it:
        tmp = A @ p
        alpha = abs_new / float(np.dot(tmp, tmp))
        x += alpha * p
        residual -= alpha * tmp
        nr = At @ residual
        res_norm2 = float(np.dot(nr, nr))
        if res_norm2 < threshold:
            break
        z = invdiag * nr
        abs_old = abs_new
        abs_new = float(np.dot(nr, z))
        p = z + (abs_new / abs_old) * p
        i += 1
    return x, i, math.sqrt(res_norm2 / rhs_norm2)


def vector_to_mesh(x, n_views, M, N):
    """convertVectorToMesh  meshwarper.cpp:810-818 -> (mesh_x, mesh_y) float32 (n_views, N, M)."""
    v = np.asarray(x, np.float64).reshape(n_views, N, M, 2)
    return v[..., 0].astype(np.float32), v[..., 1].astype(np.float32)


def create_mesh(images, matches, M, N, max_iterations=0, tolerance=0.0, **kw):
    S = assemble(images, matches, M, N, **kw)
    A, b = S.csr()
    x, it, err = lscg(A, b, max_iterations, tolerance)
    mx, my = vector_to_mesh(x, len(images), M, N)
    return mx, my, dict(iterations=it, error=err, rows=S.rows, cols=S.cols, nnz=len(S.v))
