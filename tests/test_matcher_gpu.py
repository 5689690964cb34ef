"""ms_knn_match_hamming2 against the restated BFMatcher(NORM_HAMMING).knnMatch(k = 2) (oracle/features_oracle.py): exact, ties included."""
import numpy as np
import pytest
import torch

import features_oracle as fo
from helpers import to_dev

gpu = pytest.mark.gpu


def descriptors(rng, n, nbytes, bits=None):
    d = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
    if bits is not None:                       # few distinct bit positions -> many equal distances
        d &= np.uint8(bits)
    return d


@gpu
@pytest.mark.parametrize("nq,nt,nbytes,bits", [(2500, 2500, 32, None), (300, 257, 32, 0x11), (65, 3, 32, 0x01), (40, 130, 64, 0x03), (17, 90, 4, None),
                                              (5, 1, 32, None), (5, 0, 32, None), (0, 9, 32, None)])
def test_knn2_equals_oracle(ms, cuda, nq, nt, nbytes, bits):
    rng = np.random.default_rng(nq * 7 + nt)
    q, t = descriptors(rng, nq, nbytes, bits), descriptors(rng, nt, nbytes, bits)
    if nq > 10 and nt > 10:
        t[5] = q[3]; t[9] = q[3]                # exact duplicates: distance 0 twice, the lower index first
    idx, dist = ms.knn_match_hamming2(to_dev(q) if nq else torch.empty((0, nbytes), dtype=torch.uint8, device=cuda),
                                      to_dev(t) if nt else torch.empty((0, nbytes), dtype=torch.uint8, device=cuda))
    ridx, rdist = fo.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    if nq > 10 and nt > 10:
        assert tuple(idx[3]) == (5, 9) and tuple(dist[3]) == (0, 0)


def test_vectorised_oracle_equals_the_literal_insertion_loop():
    rng = np.random.default_rng(0)
    for nq, nt, bits in ((40, 50, 0x05), (10, 2, None), (6, 1, None), (3, 0, None)):
        q, t = descriptors(rng, nq, 32, bits), descriptors(rng, nt, 32, bits)
        a, b = fo.knn2(q, t), fo.knn2_insertion(q, t)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@gpu
def test_pitched_descriptor_rows_and_ratio_test(ms, cuda):
    rng = np.random.default_rng(4)
    t = descriptors(rng, 400, 32)
    q = t[rng.permutation(400)[:200]].copy()
    flip = rng.integers(0, 256, (200, 32), dtype=np.uint8) & rng.integers(0, 256, (200, 32), dtype=np.uint8) & rng.integers(0, 256, (200, 32), dtype=np.uint8)
    q ^= flip & np.uint8(0x0f)                 # noisy copies: a clear nearest neighbour for most rows
    big_q = torch.zeros((200, 48), dtype=torch.uint8, device=cuda)
    big_q[:, :32] = to_dev(q)
    idx, dist = ms.knn_match_hamming2(big_q[:, :32], to_dev(t))
    ridx, rdist = fo.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    good = fo.ratio_matches(idx, dist)
    assert len(good) > 150 and all(np.array_equal(t[j] ^ q[i], (t[j] ^ q[i]) & 0x0f) for i, j, _ in good)


@gpu
def test_bad_arguments(ms, cuda):
    a = torch.zeros((4, 32), dtype=torch.uint8, device=cuda)
    with pytest.raises(ms.MsError):
        ms.knn_match_hamming2(a, torch.zeros((4, 16), dtype=torch.uint8, device=cuda))
    with pytest.raises(ms.MsError):
        ms.knn_match_hamming2(torch.zeros((4, 30), dtype=torch.uint8, device=cuda), torch.zeros((4, 30), dtype=torch.uint8, device=cuda))
