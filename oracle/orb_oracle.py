"""CPU oracle of the feature front-end of featurefinder::findFeatures (360_stitcher/featurefinder.cpp:13-46): cuda::ORB::create(2500, 1.2f, 8)
->detectAndCompute on a grey image, numpy.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).

Restates, stage by stage, sources/modules/cudafeatures2d/src/orb.cpp:430-865 (host) and cuda/{fast,orb}.cu (kernels):
  buildScalePyramids :655-717 | FAST 9-16 with score + 3x3 non-max suppression fast.cu:224-344 (fast.cpp:100-146) | cull by FAST score to 2n,
  HarrisResponses orb.cu:93-138, cull to n orb.cpp:719-781 | IC_Angle orb.cu:160-211 | computeOrbDescriptor<2> orb.cu:222-246,352-367 |
  mergeKeyPoints orb.cpp:823-865.
Where the reference leaves the ORDER of keypoints to atomic counters (fast.cu:294, :333) and an unstable device sort (orb.cu:61-88), this
oracle fixes it: raster order after FAST, stable descending sort in the culls -- tests compare keypoints as sets.  Transcendentals (atan2f, sincosf of
the CUDA fast-math build) are taken as correctly rounded floats of the double functions; float expressions are evaluated without contraction.
Parity unpinned by execution (no CUDA here; opencv_extra's ORB fixtures are absent)."""
import math
import os

import numpy as np

F = np.float32
CV_PI_F = F(3.14159265)
HARRIS_K = F(0.04)
PATTERN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "orb_pattern.npy")).astype(np.int32)   # (512, 2) x, y
# FAST circle, bit k of the 16-bit masks (fast.cu:224-262 packs them as C[k / 4] byte k % 4): (dy, dx)
CIRCLE = [(3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3), (0, -3), (1, -3), (2, -2), (3, -1)]


def cv_round(v):
    return int(np.rint(v))


def get_scale(scale_factor, first_level, level):
    return F(math.pow(float(F(scale_factor)), level - first_level))            # pow(float, int) -> double -> float   orb.cpp:585-588


def n_features_per_level(nfeatures=2500, scale_factor=1.2, nlevels=8):
    factor = F(1.0) / F(scale_factor)
    n_desired = F(float(F(F(nfeatures) * (F(1.0) - factor))) / (1.0 - math.pow(float(factor), nlevels)))     # int * float -> float, / double  orb.cpp:501-502
    out, s = [], 0
    for _ in range(nlevels - 1):
        out.append(cv_round(n_desired)); s += out[-1]
        n_desired = F(n_desired * factor)
    out.append(nfeatures - s)
    return out


def u_max_table(half=15):
    """orb.cpp:514-529"""
    u = [0] * (half + 2)
    vmax = int(math.floor(half * float(np.sqrt(F(2.0))) / 2 + 1))
    for v in range(0, vmax + 1):
        u[v] = cv_round(np.sqrt(F(half * half - v * v)))
    v0 = 0
    v = half
    while v >= half * float(np.sqrt(F(2.0))) / 2:
        while u[v0] == u[v0 + 1]:
            v0 += 1
        u[v] = v0
        v0 += 1
        v -= 1
    return u


def has_arc9(mask):
    """16-bit circular mask (array of ints) contains 9 contiguous set bits: what fast.cu's c_table encodes (tests check it against that table)."""
    m = mask.astype(np.uint32) & 0xffff
    acc = m.copy()
    for s in range(1, 9):
        acc &= ((m >> s) | (m << (16 - s))) & 0xffff
    return acc != 0


def fast_scores(img, mask, threshold=20):
    """calcKeypoints<true> (fast.cu:264-307): int32 score map, 0 where (i, j) is no corner / masked / within 3 px of the border."""
    h, w = img.shape
    v = img[3:h - 3, 3:w - 3].astype(np.int16)
    d = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int16) - v for dy, dx in CIRCLE], axis=-1)     # x_k - v
    wts = (1 << np.arange(16)).astype(np.uint32)
    dark = ((d < -threshold) * wts).sum(axis=-1).astype(np.uint32)
    bright = ((d > threshold) * wts).sum(axis=-1).astype(np.uint32)
    is_kp = has_arc9(dark) | has_arc9(bright)
    if mask is not None:
        is_kp &= mask[3:h - 3, 3:w - 3] != 0
    # cornerScore (fast.cu:208-222): the largest threshold at which the pixel is still a corner = max over arcs of the arc's smallest |difference|, - 1
    dd = np.concatenate([d, d[..., :8]], axis=-1)
    a = np.full(v.shape, -32768, np.int32); b = a.copy()
    for s in range(16):
        a = np.maximum(a, dd[..., s:s + 9].min(axis=-1)); b = np.maximum(b, (-dd[..., s:s + 9]).min(axis=-1))
    score = np.zeros((h, w), np.int32)
    score[3:h - 3, 3:w - 3] = np.where(is_kp, np.minimum(np.maximum(a, b), 256) - 1, 0)
    return score


def fast_detect(img, mask, threshold=20, max_points=None):
    """FAST_Impl::detectAsync with non-max suppression (fast.cpp:100-146): (loc (n, 2) int x, y; response float) in raster order."""
    score = fast_scores(img, mask, threshold)
    ys, xs = np.nonzero(score)
    if max_points is not None and len(ys) > max_points:           # `if (ind < maxKeypoints)`: which ones survive is a race in the reference; here the first
        ys, xs = ys[:max_points], xs[:max_points]
    s = score[ys, xs]
    keep = np.ones(len(ys), bool)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy or dx:
                keep &= s > score[ys + dy, xs + dx]
    return np.stack([xs[keep], ys[keep]], axis=1).astype(np.int32), s[keep].astype(np.float32)


def cull(loc, resp, n):
    """orb.cpp:719-733 + thrust::sort_by_key(greater) orb.cu:61-88, made stable."""
    if len(loc) <= n:
        return loc, resp
    order = np.argsort(-resp.astype(np.float64), kind="stable")[:n]
    return loc[order], resp[order]


def harris_responses(img, loc, block=7, k=HARRIS_K):
    im = img.astype(np.int64)
    r = block // 2
    out = np.empty(len(loc), np.float32)
    scale = F(1.0) / (F(4 * block) * F(255.0))
    s4 = F(F(F(scale * scale) * scale) * scale)
    for n, (x, y) in enumerate(loc):
        p = im[y - r - 1:y + r + 2, x - r - 1:x + r + 2]
        ix = (p[1:-1, 2:] - p[1:-1, :-2]) * 2 + (p[:-2, 2:] - p[:-2, :-2]) + (p[2:, 2:] - p[2:, :-2])
        iy = (p[2:, 1:-1] - p[:-2, 1:-1]) * 2 + (p[2:, :-2] - p[:-2, :-2]) + (p[2:, 2:] - p[:-2, 2:])
        a, b, c = int((ix * ix).sum()), int((iy * iy).sum()), int((ix * iy).sum())
        fa, fb, fc = F(a), F(b), F(c)
        s = F(fa + fb)
        out[n] = F(F(F(F(fa * fb) - F(fc * fc)) - F(F(k * s) * s)) * s4)
    return out


def ic_angles(img, loc, half=15):
    im = img.astype(np.int64)
    umax = u_max_table(half)
    out = np.empty(len(loc), np.float32)
    for n, (x, y) in enumerate(loc):
        m10 = int((np.arange(-half, half + 1) * im[y, x - half:x + half + 1]).sum())
        m01 = 0
        for v in range(1, half + 1):
            d = umax[v]
            plus, minus = im[y + v, x - d:x + d + 1], im[y - v, x - d:x + d + 1]
            m01 += v * int((plus - minus).sum())
            m10 += int((np.arange(-d, d + 1) * (plus + minus)).sum())
        a = F(math.atan2(float(F(m01)), float(F(m10))))
        if a < 0:
            a = F(a + F(F(2.0) * CV_PI_F))
        out[n] = F(a * F(F(180.0) / CV_PI_F))
    return out


def descriptors(img, loc, angles):
    out = np.zeros((len(loc), 32), np.uint8)
    px, py = PATTERN[:, 0].astype(np.float32), PATTERN[:, 1].astype(np.float32)
    for n, ((x, y), ang) in enumerate(zip(loc, angles)):
        a = F(ang * F(CV_PI_F / F(180.0)))
        sina, cosa = F(math.sin(float(a))), F(math.cos(float(a)))
        yy = y + np.rint((px * sina).astype(np.float32) + (py * cosa).astype(np.float32)).astype(np.int64)       # __float2int_rn
        xx = x + np.rint((px * cosa).astype(np.float32) - (py * sina).astype(np.float32)).astype(np.int64)
        val = img[yy, xx].astype(np.int32)
        bits = (val[0::2] < val[1::2]).astype(np.uint8)
        out[n] = np.packbits(bits.reshape(32, 8)[:, ::-1], axis=1)[:, 0]
    return out


def orb_detect_and_compute(gray, mask=None, nfeatures=2500, scale_factor=1.2, nlevels=8, edge=31, first_level=0, patch=31, fast_threshold=20,
                           resize=None, threshold_mask=None):
    """Returns (keypoints (n, 6) float32 rows x, y, response, angle, octave, size ; descriptors (n, 32) uint8), levels concatenated like mergeKeyPoints.
    resize(img, (w, h)) = cuda::resize(INTER_LINEAR) restatement (oracle.resize_linear_8u), injected to keep this module numpy-only."""
    assert resize is not None
    nper = n_features_per_level(nfeatures, scale_factor, nlevels)
    half = patch // 2
    kps, descs = [], []
    img_prev, mask_prev = None, None
    for level in range(nlevels):
        scale = F(1.0) / get_scale(scale_factor, first_level, level)
        sz = (cv_round(F(gray.shape[1]) * scale), cv_round(F(gray.shape[0]) * scale))
        if level == first_level:
            img = gray.copy(); m = None if mask is None else mask.copy()
        else:
            img = resize(img_prev, sz)
            m = None
            if mask is not None:
                m = resize(mask_prev, sz)
                m = np.where(m > 254, m, 0).astype(np.uint8)              # cuda::threshold(254, THRESH_TOZERO)  orb.cpp:693
        img_prev, mask_prev = img, m
        border = np.zeros(img.shape, np.uint8)
        if sz[0] > 2 * edge and sz[1] > 2 * edge:
            border[edge:sz[1] - edge, edge:sz[0] - edge] = 255
        lm = border if m is None else (m & border)
        loc, resp = fast_detect(img, lm, fast_threshold, int(0.05 * img.shape[0] * img.shape[1]))
        if len(loc) == 0:
            continue
        n = nper[level]
        loc, resp = cull(loc, resp, 2 * n)
        resp = harris_responses(img, loc)
        loc, resp = cull(loc, resp, n)
        if len(loc) == 0:
            continue
        ang = ic_angles(img, loc, half)
        descs.append(descriptors(img, loc, ang))
        sf = get_scale(scale_factor, first_level, level)
        loc_scale = sf if level != first_level else F(1.0)
        k = np.empty((len(loc), 6), np.float32)
        k[:, 0] = loc[:, 0].astype(np.float32) * loc_scale; k[:, 1] = loc[:, 1].astype(np.float32) * loc_scale
        k[:, 2] = resp; k[:, 3] = ang; k[:, 4] = level; k[:, 5] = F(F(patch) * sf)
        kps.append(k)
    if not kps:
        return np.zeros((0, 6), np.float32), np.zeros((0, 32), np.uint8)
    return np.concatenate(kps), np.concatenate(descs)


# ---- cv::findHomography(src, dst, mask, RANSAC)  calib3d/src/fundam.cpp:46-260, 319-402; ptsetreg.cpp:53-290; levmarq.cpp:76-214 -----------------
class CvRng:
    """cv::RNG: multiply-with-carry, CV_RNG_COEFF = 4164903690 (core/include/opencv2/core/operations.hpp)"""

    def __init__(self, state):
        self.state = (state & 0xffffffffffffffff) or 0xffffffff

    def next(self):
        self.state = ((self.state & 0xffffffff) * 4164903690 + (self.state >> 32)) & 0xffffffffffffffff
        return self.state & 0xffffffff

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def _collinear_last(p, count):
    i = count - 1
    for j in range(i):
        dx1, dy1 = float(p[j][0]) - float(p[i][0]), float(p[j][1]) - float(p[i][1])
        for k in range(j):
            dx2, dy2 = float(p[k][0]) - float(p[i][0]), float(p[k][1]) - float(p[i][1])
            if abs(dx2 * dy1 - dy2 * dx1) <= float(np.finfo(np.float32).eps) * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def _check_subset(s, d):
    if _collinear_last(s, 4) or _collinear_last(d, 4):
        return False
    neg = 0
    for t in ((0, 1, 2), (1, 2, 3), (0, 2, 3), (0, 1, 3)):
        A = np.array([[s[k][0], s[k][1], 1.0] for k in t], np.float64); B = np.array([[d[k][0], d[k][1], 1.0] for k in t], np.float64)
        neg += np.linalg.det(A) * np.linalg.det(B) < 0
    return neg in (0, 4)


def homography_dlt(M, m):
    """HomographyEstimatorCallback::runKernel (fundam.cpp:80-142); None where the reference returns 0 models."""
    M = M.astype(np.float64); m = m.astype(np.float64)
    n = len(M)
    cM, cm = M.sum(0) / n, m.sum(0) / n
    sM, sm = np.abs(M - cM).sum(0), np.abs(m - cm).sum(0)
    eps = np.finfo(np.float64).eps
    if (np.abs(sM) < eps).any() or (np.abs(sm) < eps).any():
        return None
    sM, sm = n / sM, n / sm
    x, y = (m[:, 0] - cm[0]) * sm[0], (m[:, 1] - cm[1]) * sm[1]
    X, Y = (M[:, 0] - cM[0]) * sM[0], (M[:, 1] - cM[1]) * sM[1]
    o, z = np.ones(n), np.zeros(n)
    Lx = np.stack([X, Y, o, z, z, z, -x * X, -x * Y, -x], 1); Ly = np.stack([z, z, z, X, Y, o, -y * X, -y * Y, -y], 1)
    LtL = Lx.T @ Lx + Ly.T @ Ly
    w, V = np.linalg.eigh(LtL)
    h0 = V[:, 0].reshape(3, 3)
    inv = np.array([[1 / sm[0], 0, cm[0]], [0, 1 / sm[1], cm[1]], [0, 0, 1]]); nrm = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    H = inv @ h0 @ nrm
    return H / H[2, 2]


def homography_errors(H, M, m):
    f = np.float32
    h = H.reshape(-1).astype(f)
    Mx, My, mx, my = M[:, 0].astype(f), M[:, 1].astype(f), m[:, 0].astype(f), m[:, 1].astype(f)
    ww = (f(1.0) / ((h[6] * Mx + h[7] * My).astype(f) + f(1.0)).astype(f)).astype(f)
    dx = ((((h[0] * Mx).astype(f) + (h[1] * My).astype(f)).astype(f) + h[2]).astype(f) * ww).astype(f) - mx
    dy = ((((h[3] * Mx).astype(f) + (h[4] * My).astype(f)).astype(f) + h[5]).astype(f) * ww).astype(f) - my
    return ((dx * dx).astype(f) + (dy * dy).astype(f)).astype(f)


def _update_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0); ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny); denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(np.rint(num / denom))


def _lm_refine(M, m, h):
    """HomographyRefineCallback + LMSolverImpl::run with 10 iterations (fundam.cpp:168-214, levmarq.cpp:88-199)"""
    M = M.astype(np.float64); m = m.astype(np.float64)
    eps = np.finfo(np.float64).eps

    def compute(p, jac):
        ww = p[6] * M[:, 0] + p[7] * M[:, 1] + 1.0
        ww = np.where(np.abs(ww) > eps, 1.0 / ww, 0.0)
        xi = (p[0] * M[:, 0] + p[1] * M[:, 1] + p[2]) * ww; yi = (p[3] * M[:, 0] + p[4] * M[:, 1] + p[5]) * ww
        err = np.stack([xi - m[:, 0], yi - m[:, 1]], 1).reshape(-1)
        if not jac:
            return err, None
        z = np.zeros(len(M))
        J0 = np.stack([M[:, 0] * ww, M[:, 1] * ww, ww, z, z, z, -M[:, 0] * ww * xi, -M[:, 1] * ww * xi], 1)
        J1 = np.stack([z, z, z, M[:, 0] * ww, M[:, 1] * ww, ww, -M[:, 0] * ww * yi, -M[:, 1] * ww * yi], 1)
        return err, np.stack([J0, J1], 1).reshape(-1, 8)
    x = h.copy()
    r, J = compute(x, True)
    S = float(r @ r); A = J.T @ J; v = J.T @ r; D = np.diag(A).copy()
    lam, lc, it = 1.0, 0.75, 0
    while True:
        d = np.linalg.solve(A + lam * np.diag(D), v)
        xd = x - d
        rd, _ = compute(xd, False)
        Sd = float(rd @ rd)
        dS = float(d @ (2 * v - A @ d))
        R = (S - Sd) / (dS if abs(dS) > eps else 1.0)
        if R > 0.75:
            lam *= 0.5
            if lam < lc:
                lam = 0.0
        elif R < 0.25:
            t = float(d @ v)
            nu = min(max((Sd - S) / (t if abs(t) > eps else 1.0) + 2.0, 2.0), 10.0)
            if lam == 0:
                lam = lc = 1.0 / max(eps, float(np.abs(np.diag(np.linalg.inv(A))).max()))
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S = Sd; x = xd
            r, J = compute(x, True)
            A = J.T @ J; v = J.T @ r
        it += 1
        if not (it < 10 and np.abs(d).max() >= np.finfo(np.float32).eps and np.abs(r).max() >= np.finfo(np.float32).eps):
            break
    return x


def find_homography_ransac(src, dst, thresh=3.0, max_iters=2000, confidence=0.995):
    src = np.asarray(src, np.float32); dst = np.asarray(dst, np.float32)
    n = len(src)
    if n < 4:
        return None, np.zeros(n, np.uint8)
    t = np.float32(thresh * thresh)
    if n == 4:
        H = homography_dlt(src, dst)
        return H, np.ones(4, np.uint8)
    rng = CvRng(0xffffffffffffffff)
    niters, max_good, best, best_mask = max(max_iters, 1), 0, None, None
    it = 0
    while it < niters:
        found = False
        for _ in range(10000):
            idx = []
            for i in range(4):
                while True:
                    c = rng.uniform(0, n)
                    if c not in idx:
                        idx.append(c); break
            if _check_subset(src[idx], dst[idx]):
                found = True; break
        if not found:
            if it == 0:
                return None, np.zeros(n, np.uint8)
            break
        H = homography_dlt(src[idx], dst[idx])
        if H is not None:
            mask = homography_errors(H, src, dst) <= t
            good = int(mask.sum())
            if good > max(max_good, 3):
                best, best_mask, max_good = H, mask, good
                niters = _update_iters(confidence, (n - good) / n, 4, niters)
        it += 1
    if best is None:
        return None, np.zeros(n, np.uint8)
    H = best
    M, m = src[best_mask], dst[best_mask]
    H2 = homography_dlt(M, m)
    if H2 is not None:
        H = H2
    h = _lm_refine(M, m, H.reshape(-1)[:8].copy())
    return np.append(h, 1.0).reshape(3, 3), best_mask.astype(np.uint8)
