"""Run ON THE GPU BOX: wall-clock of the recalibration front-end (featurefinder.cpp) at the reference's sizes ->
gpurun_out/<tag>_features.txt (copy into profiles/).      python tools/bench_features.py r02
ORB(2500, 1.2, 8) on a textured 1080p grey image (with and without an overlap mask), Hamming 2-NN of 2500 x 2500 descriptors, RANSAC homography."""
import sys, time
sys.path.insert(0, "video-stitcher_amd")
import numpy as np, torch, msstitch as ms

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def textured(w, h, seed):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float32)
    for s in (4, 8, 16, 32):
        g = rng.random((h // s + 2, w // s + 2)).astype(np.float32)
        img += np.kron(g, np.ones((s, s), np.float32))[:h, :w] * s
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.astype(np.uint8)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], r


out = []
scene = textured(2400, 1080, 5)
a = torch.from_numpy(scene[:, :1920].copy()).cuda(); b = torch.from_numpy(scene[:, 300:2220].copy()).cuda()
mask = torch.zeros((1080, 1920), dtype=torch.uint8, device="cuda"); mask[:, :400] = 255; mask[:, -400:] = 255
t, (ka, da) = timed(lambda: ms.orb_detect_and_compute(a))
out.append("ORB(2500, 1.2, 8) detectAndCompute, 1920x1080 grey, no mask : %.2f ms  (%d keypoints)" % (t, len(ka)))
t, (km, dm) = timed(lambda: ms.orb_detect_and_compute(a, mask))
out.append("ORB(2500, 1.2, 8) detectAndCompute, 1920x1080 grey, 2 x 400 px overlap mask : %.2f ms  (%d keypoints)" % (t, len(km)))
_, (kb, db) = timed(lambda: ms.orb_detect_and_compute(b), 3)
t, (idx, dist) = timed(lambda: ms.knn_match_hamming2(da, db))
out.append("knnMatch(k = 2), %d x %d descriptors (incl. download of the matches) : %.3f ms" % (len(ka), len(kb), t))
keep = (idx[:, 1] >= 0) & (dist[:, 0].astype(np.float32) < 0.7 * dist[:, 1].astype(np.float32))
q = np.nonzero(keep)[0]
src = ka[q, :2] - np.float32([960, 540]); dst = kb[idx[q, 0], :2] - np.float32([960, 540])
t, (H, m) = timed(lambda: ms.find_homography_ransac(src, dst))
out.append("findHomography(RANSAC, 3 px, 2000 iterations max), %d ratio-test matches : %.2f ms  (%d inliers, H[0][2] = %.2f for a 300 px offset)" % (len(q), t, int(m.sum()), H[0, 2]))
rng = np.random.default_rng(1)
src2 = rng.uniform(-900, 900, (2000, 2)).astype(np.float32); dst2 = src2 + np.float32([25, -7]); bad = rng.random(2000) < 0.5
dst2[bad] = rng.uniform(-900, 900, (int(bad.sum()), 2)).astype(np.float32)
t, (H, m) = timed(lambda: ms.find_homography_ransac(src2, dst2))
out.append("findHomography(RANSAC), 2000 pairs with 50 %% outliers : %.2f ms  (%d inliers)" % (t, int(m.sum())))
open("gpurun_out/%s_features.txt" % tag, "w").write("# tools/bench_features.py, MI355X, median of 20 wall-clock runs each (host call -> results on the host)\n" + "\n".join(out) + "\n")
print("\n".join(out))
