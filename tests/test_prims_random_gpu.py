"""Randomised shapes for the per-op kernels (hypothesis): every size from 1x1 up, odd widths, pitches that are not the row size.
Integer outputs must equal the oracle bit for bit.  Complements the fixed-size cases of test_prims_gpu.py."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import host, to_dev, to_dev_roi

pytestmark = pytest.mark.gpu
FAST = settings(max_examples=int(os.environ.get("MS_TEST_EXAMPLES", 40)), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
dims = st.tuples(st.integers(1, 70), st.integers(1, 90))


@FAST
@given(size=dims, seed=st.integers(0, 2 ** 31 - 1), roi=st.booleans())
def test_pyr_down_up_16s_any_size(ms, cuda, oracle, size, seed, roi):
    rng = np.random.default_rng(seed)
    a = rng.integers(-32768, 32768, size=(size[0], size[1], 3), dtype=np.int16)
    up = (lambda x: to_dev_roi(x, rng)) if roi else to_dev
    assert np.array_equal(host(ms.pyr_down(up(a))), oracle.pyr_down_16s(a))
    assert np.array_equal(host(ms.pyr_up(up(a))), oracle.pyr_up_16s(a))


@FAST
@given(size=dims, src_size=dims, seed=st.integers(0, 2 ** 31 - 1))
def test_remap_linear_any_size_and_wild_coordinates(ms, cuda, oracle, size, src_size, seed):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, 256, size=(src_size[0], src_size[1], 3), dtype=np.uint8)
    mx = rng.uniform(-3, src_size[1] + 3, size=size).astype(np.float32)
    my = rng.uniform(-3, src_size[0] + 3, size=size).astype(np.float32)
    wild = rng.random(size) < 0.05                       # NaN / inf / huge / exactly-on-the-border coordinates
    mx[wild] = rng.choice(np.array([np.nan, np.inf, -np.inf, 1e20, -1e20, -1.0, 0.0, src_size[1] - 1.0, float(src_size[1])], np.float32), size=int(wild.sum()))
    got = ms.remap(to_dev(src), to_dev(mx), to_dev(my), ms.INTER_LINEAR)
    assert np.array_equal(host(got), oracle.remap_linear_8uc3(src, mx, my))
    got_r = ms.remap(to_dev(src), to_dev(mx), to_dev(my), ms.INTER_LINEAR, ms.BORDER_REFLECT)
    assert np.array_equal(host(got_r), oracle.remap_linear_reflect_8uc3(src, mx, my))


@FAST
@given(size=dims, pad=st.tuples(st.integers(0, 9), st.integers(0, 9), st.integers(0, 9), st.integers(0, 9)), seed=st.integers(0, 2 ** 31 - 1))
def test_copy_make_border_reflect_any_size(ms, cuda, oracle, size, pad, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(size[0], size[1], 3), dtype=np.uint8)
    t, b, l, r = pad
    got = ms.copy_make_border(to_dev(a), t, b, l, r, ms.BORDER_REFLECT)
    assert np.array_equal(host(got), oracle.copy_make_border_reflect(a, t, b, l, r))


@FAST
@given(size=dims, seed=st.integers(0, 2 ** 31 - 1))
def test_accumulate_normalise_any_size(ms, cuda, oracle, size, seed):
    """addSrcWeightGpu32F + normalizeUsingWeightMapGpu32F (multiband_blend.cu:36-108) on random int16 / fp32 data, incl. weights of 0."""
    rng = np.random.default_rng(seed)
    src = rng.integers(-2000, 2000, size=(size[0], size[1], 3), dtype=np.int16)
    w = rng.random(size, dtype=np.float32)
    w[rng.random(size) < 0.2] = 0.0
    dst = rng.integers(-500, 500, size=src.shape, dtype=np.int16)
    dw = rng.random(size, dtype=np.float32)
    d_dev, dw_dev = to_dev(dst), to_dev(dw)
    ms.add_src_weight_32f(to_dev(src), to_dev(w), d_dev, dw_dev)
    rd, rdw = dst.copy(), dw.copy()
    oracle.add_src_weight_32f(src, w, rd, rdw)
    assert np.array_equal(host(d_dev), rd) and np.array_equal(host(dw_dev), rdw)
    ms.normalize_using_weight_32f(dw_dev, d_dev)
    oracle.normalize_32f(rdw, rd)
    assert np.array_equal(host(d_dev), rd)


@FAST
@given(size=dims, src_size=dims, cn=st.sampled_from([1, 3]), seed=st.integers(0, 2 ** 31 - 1))
def test_remap_cpu_flavour_any_size_and_wild_coordinates(ms, cuda, oracle, size, src_size, cn, seed):
    """ms_remap(MS_INTER_LINEAR_FIXPT) = cv::remap on the CPU: quantised coordinates, 15-bit weights, border taps, NaN / huge coordinates."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, 256, size=(src_size[0], src_size[1]) + ((3,) if cn == 3 else ()), dtype=np.uint8)
    mx = rng.uniform(-3, src_size[1] + 3, size=size).astype(np.float32)
    my = rng.uniform(-3, src_size[0] + 3, size=size).astype(np.float32)
    wild = rng.random(size) < 0.08
    vals = np.array([np.nan, np.inf, -np.inf, 1e20, -1e20, 3e9, -3e9, -1.0, 0.0, 0.015625, src_size[1] - 1.0, float(src_size[1]), 70000.0, -70000.0], np.float32)
    mx[wild] = rng.choice(vals, size=int(wild.sum()))
    wild2 = rng.random(size) < 0.05
    my[wild2] = rng.choice(vals, size=int(wild2.sum()))
    snap = rng.random(size) < 0.1                     # whole-pixel and 1/32-grid coordinates (the {32767, 0, 0, 1} table entry, exact ties)
    mx[snap] = np.round(mx[snap] * 32) / 32; my[snap] = np.round(my[snap])
    got = ms.remap(to_dev(src), to_dev(mx), to_dev(my), ms.INTER_LINEAR_FIXPT)
    assert np.array_equal(host(got), oracle.cv_remap_linear(src, mx, my))


@FAST
@given(nq=st.integers(0, 90), nt=st.integers(0, 150), words=st.sampled_from([1, 4, 8, 16]), bits=st.sampled_from([0xff, 0x11, 0x01]), seed=st.integers(0, 2 ** 31 - 1))
def test_knn_match_any_size(ms, cuda, nq, nt, words, bits, seed):
    import torch
    import features_oracle as fo
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, (nq, 4 * words), dtype=np.uint8) & np.uint8(bits)
    t = rng.integers(0, 256, (nt, 4 * words), dtype=np.uint8) & np.uint8(bits)
    dev = lambda a: to_dev(a) if len(a) else torch.empty((0, 4 * words), dtype=torch.uint8, device=cuda)
    idx, dist = ms.knn_match_hamming2(dev(q), dev(t))
    ridx, rdist = fo.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


@FAST
@given(h2=st.integers(1, 40), w2=st.integers(1, 50), seed=st.integers(0, 2 ** 31 - 1), roi=st.booleans())
def test_i420_and_gray_any_even_size(ms, cuda, oracle, h2, w2, seed, roi):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, 256, (2 * h2, 2 * w2, 3), dtype=np.uint8)
    up = (lambda x: to_dev_roi(x, rng)) if roi else to_dev
    assert np.array_equal(host(ms.bgr_to_i420(up(src))), oracle.bgr_to_i420(src))
    gray = ((src[..., 0].astype(np.uint32) * 1868 + src[..., 1].astype(np.uint32) * 9617 + src[..., 2].astype(np.uint32) * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    assert np.array_equal(host(ms.bgr_to_gray(up(src))), gray)
