"""the feature front-end of the recalibration path on the device -- cuda::ORB::detectAndCompute and findHomography(RANSAC) of
360_stitcher/featurefinder.cpp -- against oracle/orb_oracle.py (numpy restatement of cudafeatures2d's kernels / calib3d's host code)."""
import numpy as np
import pytest
import torch

import orb_oracle as oo
import synth

pytestmark = pytest.mark.gpu


def gray_of(bgr):
    return ((bgr[..., 0].astype(np.int32) * 1868 + bgr[..., 1].astype(np.int32) * 9617 + bgr[..., 2].astype(np.int32) * 4899 + 8192) >> 14).astype(np.uint8)


def textured(w, h, seed):
    """corners to find: blocks of random grey levels over the synthetic camera pattern"""
    rng = np.random.default_rng(seed)
    g = gray_of(synth.frame(w, h, seed % 6, seed)).astype(np.int32)
    blocks = rng.integers(-90, 90, size=((h + 15) // 16, (w + 15) // 16))
    g += np.kron(blocks, np.ones((16, 16), np.int64))[:h, :w]
    for _ in range(60):
        y, x, r = rng.integers(0, h), rng.integers(0, w), rng.integers(3, 12)
        g[max(0, y - r):y + r, max(0, x - r):x + r] += rng.integers(-120, 120)
    return np.clip(g, 0, 255).astype(np.uint8)


def as_set(kp, desc):
    return {(float(k[0]), float(k[1]), int(k[4])): (float(k[2]), float(k[3]), float(k[5]), bytes(d)) for k, d in zip(kp, desc)}


@pytest.mark.parametrize("size,nfeatures,use_mask", [((640, 360), 500, False), ((517, 389), 2500, False), ((640, 360), 300, True), ((1920, 1080), 2500, False)])      # last: the size and budget featurefinder.cpp uses
def test_orb_matches_oracle(ms, cuda, oracle, size, nfeatures, use_mask):
    """Keypoints as a set (the reference's own order is left to atomics and an unstable sort): same locations per level, bit-equal Harris responses
    (integer sums, float formula in the same order), angles, sizes and 256-bit descriptors."""
    w, h = size
    g = textured(w, h, 3)
    mask = None
    if use_mask:
        mask = np.zeros((h, w), np.uint8); mask[:, : w // 2 + 40] = 255
    kp_ref, d_ref = oo.orb_detect_and_compute(g, mask, nfeatures=nfeatures, resize=lambda im, sz: oracle.resize_linear_8u(im, dsize=sz))
    kp, d = ms.orb_detect_and_compute(torch.from_numpy(g).to(cuda), None if mask is None else torch.from_numpy(mask).to(cuda), nfeatures=nfeatures)
    d = d.cpu().numpy()
    assert len(kp_ref) > 100 and len(set(kp_ref[:, 4])) >= 4, "the test image must give corners on several levels"
    a, b = as_set(kp, d), as_set(kp_ref, d_ref)
    assert len(kp) == len(a) and len(kp_ref) == len(b)
    # ties at a cull boundary are resolved by the (shared) stable order, so the sets are equal, not just overlapping
    assert a.keys() == b.keys(), (len(a), len(b), len(a.keys() & b.keys()))
    bad = [k for k in a if a[k] != b[k]]
    assert not bad, (len(bad), bad[:3], [a[k][:3] for k in bad[:3]], [b[k][:3] for k in bad[:3]])
    if use_mask:
        assert (kp[:, 0] <= w // 2 + 40).all()


def test_orb_per_level_budgets_and_empty_image(ms, cuda):
    kp, d = ms.orb_detect_and_compute(torch.full((200, 300), 128, dtype=torch.uint8, device=cuda))
    assert len(kp) == 0 and d.shape == (0, 32)
    assert oo.n_features_per_level() == [543, 452, 377, 314, 262, 218, 182, 152] and sum(oo.n_features_per_level()) == 2500
    kp, _ = ms.orb_detect_and_compute(torch.from_numpy(textured(1280, 720, 5)).to(cuda))
    per = [int((kp[:, 4] == l).sum()) for l in range(8)]
    assert all(p <= q for p, q in zip(per, oo.n_features_per_level())) and sum(per) > 1000


def _scene(n, outliers, seed, noise=0.4):
    rng = np.random.default_rng(seed)
    H = np.array([[1.02, 0.03, 14.0], [-0.02, 0.98, -9.0], [2e-5, -1e-5, 1.0]])
    src = rng.uniform(-400, 400, size=(n, 2)).astype(np.float32)
    p = np.c_[src, np.ones(n)] @ H.T
    dst = (p[:, :2] / p[:, 2:3] + rng.normal(0, noise, size=(n, 2))).astype(np.float32)
    bad = rng.choice(n, outliers, replace=False)
    dst[bad] += rng.uniform(20, 200, size=(outliers, 2)).astype(np.float32) * rng.choice([-1, 1], size=(outliers, 2))
    return src, dst, H, bad


@pytest.mark.parametrize("n,outliers,seed", [(120, 40, 1), (400, 250, 2), (30, 5, 3), (9, 2, 4)])
def test_find_homography_ransac_matches_oracle(ms, cuda, n, outliers, seed):
    """Same cv::RNG subset sequence, acceptance rule and adaptive iteration count as the reference => the same winning hypothesis and inlier mask;
    H agrees with the oracle to 1e-6 (relative, Frobenius: two different symmetric eigen-solvers and linear solvers) and recovers the true map."""
    src, dst, Htrue, bad = _scene(n, outliers, seed)
    H, mask = ms.find_homography_ransac(src, dst)
    Hr, mr = oo.find_homography_ransac(src, dst)
    assert H is not None and Hr is not None
    assert np.array_equal(mask, mr)
    assert np.linalg.norm(H - Hr) / np.linalg.norm(Hr) < 1e-6
    assert mask[bad].sum() <= 1 and mask.sum() >= (n - outliers) * 0.9
    p = np.c_[src, np.ones(n)] @ H.T
    good = mask.astype(bool)
    assert np.abs(p[good, :2] / p[good, 2:3] - dst[good]).max() < 3.0
    if n >= 100:                                   # (a handful of noisy points does not pin the map that closely)
        assert np.linalg.norm(H / H[2, 2] - Htrue) / np.linalg.norm(Htrue) < 2e-2


def test_orb_rejects_thresholds_the_kernels_cannot_honour(ms, cuda):
    """edge_threshold below the reach of the descriptor / angle / Harris windows would read outside the level image; fast_threshold 0 would make
    the score map's "no corner" value a corner: both are argument errors, not device faults."""
    g = torch.zeros((120, 160), dtype=torch.uint8, device="cuda")
    for kw in (dict(edge_threshold=18), dict(edge_threshold=0), dict(fast_threshold=0), dict(fast_threshold=255)):
        with pytest.raises(ms.MsError):
            ms.orb_detect_and_compute(g, nfeatures=100, **kw)
    kp, _ = ms.orb_detect_and_compute(g, nfeatures=100, edge_threshold=19, fast_threshold=1)      # the smallest legal values run (and find nothing in a flat image)
    assert len(kp) == 0


def test_find_homography_degenerate_inputs(ms, cuda):
    src = np.array([[0, 0], [1, 0], [0, 1]], np.float32)
    H, m = ms.find_homography_ransac(src, src)
    assert H is None and len(m) == 3
    sq = np.array([[0, 0], [100, 0], [100, 100], [0, 100]], np.float32)
    H, m = ms.find_homography_ransac(sq, sq * 2 + 5)                      # exactly 4 points: runKernel only, all inliers (fundam.cpp:356-360)
    Hr, _ = oo.find_homography_ransac(sq, sq * 2 + 5)
    assert m.tolist() == [1, 1, 1, 1] and np.allclose(H, Hr, atol=1e-9) and np.allclose(H, [[2, 0, 5], [0, 2, 5], [0, 0, 1]], atol=1e-9)
    line = np.c_[np.arange(20, dtype=np.float32), np.arange(20, dtype=np.float32) * 2]          # collinear: no valid subset
    H, m = ms.find_homography_ransac(line, line)
    assert H is None and m.sum() == 0


def test_front_end_recovers_the_offset_between_two_views_of_one_scene(ms, cuda):
    """findFeatures + matchFeatures in miniature: two overlapping crops of one texture -> ORB on both, Hamming 2-NN + 0.7 ratio test on the device,
    RANSAC homography on the centred points (featurefinder.cpp:68-90).  The homography must be the translation between the crops."""
    scene = textured(1100, 420, 11)
    off = 310
    a, b = scene[:, :760].copy(), scene[:, off:off + 760].copy()
    ka, da = ms.orb_detect_and_compute(torch.from_numpy(a).to(cuda), nfeatures=1500)
    kb, db = ms.orb_detect_and_compute(torch.from_numpy(b).to(cuda), nfeatures=1500)
    idx, dist = ms.knn_match_hamming2(da, db)
    keep = (idx[:, 1] >= 0) & (dist[:, 0].astype(np.float32) < 0.7 * dist[:, 1].astype(np.float32))
    q = np.nonzero(keep)[0]
    assert len(q) > 80
    src = ka[q, :2] - np.float32([760 * 0.5, 420 * 0.5]); dst = kb[idx[q, 0], :2] - np.float32([760 * 0.5, 420 * 0.5])
    H, mask = ms.find_homography_ransac(src, dst)
    assert H is not None and mask.sum() > 0.8 * len(q)
    assert np.allclose(H, [[1, 0, -off], [0, 1, 0], [0, 0, 1]], atol=0.05), H


def _homography_test_case(n, rng, sigma=0.0):
    """The data of the reference's own Calib3d_Homography test (calib3d/test/test_homography.cpp:250-300, 395-410): n random points in a 100 x 100 image,
    a random rotation + translation, optional N(0, sigma) noise on the destination points; mask = noise within the reprojection threshold."""
    image_size, thr = 100, 3.0
    src = (rng.random((n, 2)) * image_size).astype(np.float32)
    fi = rng.random() * 2 * np.pi
    tx, ty = rng.random(2) * np.sqrt(image_size)
    H = np.array([[np.cos(fi), -np.sin(fi), tx], [np.sin(fi), np.cos(fi), ty], [0, 0, 1]], np.float64)
    d = (H.astype(np.float32) @ np.c_[src, np.ones(n, np.float32)].T).T
    dst = (d[:, :2] / d[:, 2:3]).astype(np.float32)
    noise = (rng.normal(0, sigma, (n, 2)) if sigma else np.zeros((n, 2))).astype(np.float32)
    return src, dst + noise, dst, noise, H, np.hypot(noise[:, 0], noise[:, 1]) <= thr


def _check_reference_homography_criteria(find, n, seed):
    """CV_HomographyTest::run for method RANSAC (test_homography.cpp:344-440 noise-free, :472-560 noisy): max_diff 1e-2 / max_2diff 2e-2, threshold 3.0."""
    rng = np.random.default_rng(seed)
    thr, max_diff, max_2diff = 3.0, 1e-2, 2e-2
    norms = (lambda a: np.abs(a).sum(), lambda a: np.sqrt((a * a).sum()), lambda a: np.abs(a).max())      # NORM_L1, NORM_L2, NORM_INF
    # noise-free: every point an inlier, H within max_diff of the true one in all three norms
    src, dst, _, _, H, _ = _homography_test_case(n, rng)
    Hr, m = find(src, dst)
    assert Hr is not None and m.shape == (n,) and m.min() == 1 and m.max() == 1
    for nrm in norms:
        assert nrm(Hr / Hr[2, 2] - H) <= max_diff
    # noisy destination points (sigma 0.01): mask == (reprojection error <= threshold), no true inlier lost, inliers reproject within max_2diff of the noise
    src, dst_n, dst, noise, H, mask0 = _homography_test_case(n, rng, sigma=0.01)
    Hr, m = find(src, dst_n)
    assert Hr is not None and set(np.unique(m)) <= {0, 1}
    p = (Hr.astype(np.float32) @ np.c_[src, np.ones(n, np.float32)].T).T
    p = (p[:, :2] / p[:, 2:3]).astype(np.float32)
    err = np.hypot(*(p - dst_n).T)
    assert np.array_equal(m.astype(bool), err <= thr), "mask must be the reprojection test"
    assert not (mask0 & ~m.astype(bool)).any(), "an inlier of the original mask is an outlier of the found one"
    for k in np.nonzero(m)[0]:
        for nrm in norms:
            assert nrm(p[k] - dst[k]) - nrm(noise[k]) <= max_2diff


@pytest.mark.parametrize("n", [4, 5, 7, 16, 50, 151, 303])
def test_find_homography_passes_the_references_own_homography_test(ms, cuda, n):
    """ms_find_homography_ransac under the criteria of the reference's Calib3d_Homography accuracy test (the RANSAC branch), and the oracle under the same."""
    _check_reference_homography_criteria(lambda s, d: ms.find_homography_ransac(s, d, reproj_threshold=3.0), n, 100 + n)
    _check_reference_homography_criteria(lambda s, d: oo.find_homography_ransac(s, d, 3.0), n, 100 + n)
