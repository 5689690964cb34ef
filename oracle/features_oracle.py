"""CPU oracle of the descriptor matching step of featurefinder::matchFeatures (360_stitcher/featurefinder.cpp:50-66), numpy.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
(video-stitcher_amd) never imports this module.

Restates BFMatcher(NORM_HAMMING)::knnMatch(k = 2) = cv::batchDistance with K = 2 (sources/modules/core/src/stat.cpp:3946-4008: distances
by popcount, then the K-insertion loop of BatchDistInvoker) and the 0.7 ratio test (featurefinder.cpp:61-66).  Integer arithmetic, so
parity is exact; pinned by construction only (no matcher fixture ships with the reference: opencv_extra is absent)."""
import numpy as np

INT_MAX = 2 ** 31 - 1
_POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def hamming_matrix(query, train):
    pop8 = _POP.astype(np.uint8)
    out = np.empty((len(query), len(train)), np.int32)
    for a in range(0, len(query), 64):          # chunked: the (nq, nt, bytes) cube of a 2500 x 2500 match would not fit
        out[a:a + 64] = pop8[query[a:a + 64, None, :] ^ train[None, :, :]].sum(axis=2, dtype=np.int32)
    return out


def knn2_insertion(query, train):
    """BatchDistInvoker::operator() (stat.cpp:3963-3997), literally: slow, for small inputs."""
    K = min(2, len(train))
    nq = len(query)
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), INT_MAX, np.int32)
    D = hamming_matrix(query, train) if len(train) else np.zeros((nq, 0), np.int32)
    for i in range(nq):
        for j in range(len(train)):
            d = int(D[i, j])
            if K > 0 and d < dist[i, K - 1]:
                k = K - 2
                while k >= 0 and dist[i, k] > d:
                    idx[i, k + 1] = idx[i, k]
                    dist[i, k + 1] = dist[i, k]
                    k -= 1
                idx[i, k + 1] = j
                dist[i, k + 1] = d
    return idx, dist


def knn2(query, train):
    """Same result, vectorised: the first two train rows in (distance, index) order."""
    nq, nt = len(query), len(train)
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), INT_MAX, np.int32)
    if nt == 0 or nq == 0:
        return idx, dist
    D = hamming_matrix(query, train)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    k = order.shape[1]
    idx[:, :k] = order
    dist[:, :k] = np.take_along_axis(D, order, axis=1)
    return idx, dist


def ratio_matches(idx, dist, ratio=0.7):
    """featurefinder.cpp:61-66: keep (queryIdx, trainIdx, distance) where float(d0) < 0.7 * float(d1)."""
    keep = dist[:, 0].astype(np.float32).astype(np.float64) < ratio * dist[:, 1].astype(np.float32).astype(np.float64)
    q = np.nonzero(keep)[0]
    return [(int(i), int(idx[i, 0]), float(dist[i, 0])) for i in q]
