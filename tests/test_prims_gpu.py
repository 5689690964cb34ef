"""Per-primitive parity: every cv::cuda:: image op on the hot path (include/ms_stitch.h section 1) against
the oracle's restatement of the same CUDA kernel.  Method follows the reference's own tests
(OCV/cudawarping/test/test_remap.cpp, test_pyramids.cpp, test_resize.cpp; OCV/cudaarithm/test/
test_element_operations.cpp): random data, sizes incl. odd 113 and 128 (cuda_test.hpp:212 DIFFERENT_SIZES),
ROI sub-matrices with padded steps (cuda_test.cpp:92-104).  Bar: bit-exact (integer outputs), i.e. tighter
than the reference's tolerance of 1.0."""
import math

import numpy as np
import pytest
import torch

from helpers import host, to_dev, to_dev_roi

pytestmark = pytest.mark.gpu

SIZES = [(128, 128), (113, 113), (37, 251), (1, 9), (5, 2)]   # (rows, cols)


def rng_for(*k):
    return np.random.default_rng(abs(hash(k)) % (2 ** 32))


def rot_maps(rows, cols, src_rows, src_cols):
    """45-degree rotation maps of test_remap.cpp:138-154, plus out-of-range and NaN entries."""
    M = np.array([[math.cos(math.pi / 4), -math.sin(math.pi / 4), src_cols / 2.0],
                  [math.sin(math.pi / 4), math.cos(math.pi / 4), 0.0]])
    x, y = np.meshgrid(np.arange(cols), np.arange(rows))
    mx = (M[0, 0] * x + M[0, 1] * y + M[0, 2]).astype(np.float32)
    my = (M[1, 0] * x + M[1, 1] * y + M[1, 2]).astype(np.float32)
    return mx, my


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("use_roi", [False, True])
def test_remap_linear_8uc3(ms, cuda, oracle, size, use_roi):
    rng = rng_for("remap3", size, use_roi)
    src = rng.integers(0, 256, size=(size[0] + 3, size[1] + 7, 3), dtype=np.uint8)
    mx, my = rot_maps(size[0], size[1], src.shape[0], src.shape[1])
    mx += rng.uniform(-0.5, 0.5, mx.shape).astype(np.float32)
    if mx.size > 20:
        mx.flat[3] = np.nan; my.flat[7] = np.inf; mx.flat[11] = -1.0; my.flat[11] = -1.0; mx.flat[13] = 3e9
    up = (lambda a: to_dev_roi(a, rng)) if use_roi else to_dev
    got = ms.remap(up(src), up(mx), up(my), ms.INTER_LINEAR)
    assert np.array_equal(host(got), oracle.remap_linear_8uc3(src, mx, my))


@pytest.mark.parametrize("interp", ["linear", "nearest"])
def test_remap_8uc1(ms, cuda, oracle, interp):
    rng = rng_for("remap1", interp)
    src = rng.integers(0, 256, size=(97, 131), dtype=np.uint8)
    mx, my = rot_maps(113, 120, 97, 131)
    fn = oracle.remap_linear_8uc1 if interp == "linear" else oracle.remap_nearest_8uc1
    got = ms.remap(to_dev(src), to_dev(mx), to_dev(my), ms.INTER_LINEAR if interp == "linear" else ms.INTER_NEAREST)
    assert np.array_equal(host(got), fn(src, mx, my))


@pytest.mark.parametrize("scale", [0.82, 0.5, 0.3, 1.0, 1.7])
@pytest.mark.parametrize("cn", [1, 3])
def test_resize_linear(ms, cuda, oracle, scale, cn):
    rng = rng_for("resize", scale, cn)
    src = rng.integers(0, 256, size=(113, 128) + ((3,) if cn == 3 else ()), dtype=np.uint8)
    got = ms.resize_linear(to_dev_roi(src, rng), fx=scale, fy=scale)
    assert np.array_equal(host(got), oracle.resize_linear_8u(src, fx=scale, fy=scale))


@pytest.mark.parametrize("alpha", [0.95, 1.0, 1.05, 1.37, 0.0])
def test_convert_scale_8u(ms, cuda, oracle, alpha):
    src = np.arange(256, dtype=np.uint8).repeat(3).reshape(16, 16, 3)
    assert np.array_equal(host(ms.convert_scale_8u(to_dev(src), alpha)), oracle.convert_scale_8u(src, alpha))
    d = to_dev(src)
    ms.convert_scale_8u(d, alpha, inplace=True)          # the reference converts in place (timed.cpp:94)
    assert np.array_equal(host(d), oracle.convert_scale_8u(src, alpha))


@pytest.mark.parametrize("size", SIZES)
def test_copy_make_border_reflect(ms, cuda, oracle, size):
    rng = rng_for("border", size)
    src = rng.integers(0, 256, size=size + (3,), dtype=np.uint8)
    for (t, b, l, r) in [(0, 13, 127, 97), (3, 0, 0, 1), (2 * size[0] + 1, 1, 2 * size[1] + 3, 0)]:
        got = ms.copy_make_border(to_dev_roi(src, rng), t, b, l, r, ms.BORDER_REFLECT)
        assert np.array_equal(host(got), oracle.copy_make_border_reflect(src, t, b, l, r))


def test_copy_make_border_const_32f(ms, cuda, oracle):
    rng = rng_for("borderf")
    src = rng.random((57, 33), dtype=np.float32)
    got = ms.copy_make_border(to_dev(src), 5, 9, 31, 0, ms.BORDER_CONSTANT)
    assert np.array_equal(host(got), oracle.copy_make_border_const_32f(src, 5, 9, 31, 0))


def test_convert_depths(ms, cuda, oracle):
    rng = rng_for("cvt")
    u8 = rng.integers(0, 256, size=(33, 65, 3), dtype=np.uint8)
    assert np.array_equal(host(ms.convert(to_dev(u8), torch.int16)), u8.astype(np.int16))
    s16 = rng.integers(-400, 700, size=(33, 65, 3), dtype=np.int16)
    assert np.array_equal(host(ms.convert(to_dev(s16), torch.uint8)), oracle.convert_16s_8u(s16))
    m = rng.integers(0, 256, size=(33, 65), dtype=np.uint8)
    assert np.array_equal(host(ms.convert(to_dev(m), torch.float32, 1.0 / 255.0)), oracle.convert_8u_32f_scale(m, 1.0 / 255.0))


@pytest.mark.parametrize("size", SIZES + [(640, 1184)])
@pytest.mark.parametrize("cn", [1, 3])
def test_pyr_down_16s(ms, cuda, oracle, size, cn):
    rng = rng_for("pd", size, cn)
    lo, hi = (-32768, 32768) if size[0] % 2 else (-300, 300)
    src = rng.integers(lo, hi, size=size + ((3,) if cn == 3 else ()), dtype=np.int16)
    got = ms.pyr_down(to_dev_roi(src, rng))
    assert np.array_equal(host(got), oracle.pyr_down_16s(src))


@pytest.mark.parametrize("size", SIZES)
def test_pyr_down_32f(ms, cuda, oracle, size):
    rng = rng_for("pdf", size)
    src = (rng.integers(0, 256, size=size).astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)
    got = ms.pyr_down(to_dev_roi(src, rng))
    assert np.array_equal(host(got), oracle.pyr_down_32f(src))


@pytest.mark.parametrize("size", SIZES + [(320, 592)])
@pytest.mark.parametrize("cn", [1, 3])
def test_pyr_up_16s(ms, cuda, oracle, size, cn):
    rng = rng_for("pu", size, cn)
    lo, hi = (-32768, 32768) if size[1] % 2 else (-300, 300)
    src = rng.integers(lo, hi, size=size + ((3,) if cn == 3 else ()), dtype=np.int16)
    got = ms.pyr_up(to_dev_roi(src, rng))
    assert np.array_equal(host(got), oracle.pyr_up_16s(src))


def test_add_subtract_16s_saturate(ms, cuda, oracle):
    rng = rng_for("addsub")
    a = rng.integers(-32768, 32768, size=(113, 128, 3), dtype=np.int16)
    b = rng.integers(-32768, 32768, size=(113, 128, 3), dtype=np.int16)
    assert np.array_equal(host(ms.subtract(to_dev(a), to_dev(b))), oracle.sub_16s(a, b))
    assert np.array_equal(host(ms.add(to_dev(a), to_dev(b))), oracle.add_16s(a, b))
    da = to_dev(a)
    ms.subtract(da, to_dev(b), dst=da)                   # in place, as blenders.cpp:719
    assert np.array_equal(host(da), oracle.sub_16s(a, b))


def test_add_src_weight_and_normalize(ms, cuda, oracle):
    rng = rng_for("asw")
    src = rng.integers(-600, 600, size=(64, 96, 3), dtype=np.int16)
    w = rng.random((64, 96), dtype=np.float32)
    w[rng.random(w.shape) < 0.2] = 0.0
    dst = rng.integers(-32768, 32768, size=(128, 160, 3), dtype=np.int16)   # int16 accumulation wraps
    dw = rng.random((128, 160), dtype=np.float32)
    d_dst, d_dw = to_dev(dst), to_dev(dw)
    ms.add_src_weight_32f(to_dev(src), to_dev(w), d_dst[32:96, 16:112], d_dw[32:96, 16:112])   # dst(rc)
    oracle.add_src_weight_32f(src, w, dst[32:96, 16:112], dw[32:96, 16:112])
    assert np.array_equal(host(d_dst), dst) and np.array_equal(host(d_dw), dw)
    dw[::7, ::5] = 0.0
    d_dw = to_dev(dw)
    ms.normalize_using_weight_32f(d_dw, d_dst)
    oracle.normalize_32f(dw, dst)
    assert np.array_equal(host(d_dst), dst)


def test_add_src_weight_and_normalize_16s_weights(ms, cuda, oracle):
    """MultiBandBlender(weight_type = CV_16S): the fixed-point launchers (multiband_blend.cu:10-34, 62-83) -- (v * w) >> 8 with an arithmetic shift for negative
    Laplacians, wrapping short accumulation, truncating (v << 8) / w, zero weights.  Also the blender's own use: weights 0 / 256 through a 16SC1 pyrDown."""
    rng = rng_for("asw16")
    src = rng.integers(-600, 600, size=(64, 96, 3), dtype=np.int16)
    w = rng.integers(0, 257, size=(64, 96), dtype=np.int16)
    w[rng.random(w.shape) < 0.2] = 0
    dst = rng.integers(-32768, 32768, size=(128, 160, 3), dtype=np.int16)
    dw = rng.integers(0, 2000, size=(128, 160), dtype=np.int16)
    d_dst, d_dw = to_dev(dst), to_dev(dw)
    dst0, dw0 = dst.copy(), dw.copy()
    ms.add_src_weight_16s(to_dev(src), to_dev(w), d_dst[32:96, 16:112], d_dw[32:96, 16:112])
    oracle.add_src_weight_16s(src, w, dst[32:96, 16:112], dw[32:96, 16:112])
    assert np.array_equal(host(d_dst), dst) and np.array_equal(host(d_dw), dw)
    # independent statement of the accumulate: floor division by 256 (arithmetic shift), int16 wrap-around
    term = ((np.int64(src) * np.int64(w)[..., None]) // 256).astype(np.int16)
    assert np.array_equal(dst[32:96, 16:112], (np.int64(dst0[32:96, 16:112]) + np.int64(term)).astype(np.int16))
    assert np.array_equal(dw[32:96, 16:112], (np.int64(dw0[32:96, 16:112]) + np.int64(w)).astype(np.int16))
    dw[::7, ::5] = 0
    d_dw = to_dev(dw)
    ms.normalize_using_weight_16s(d_dw, d_dst)
    before = dst.copy()
    oracle.normalize_16s(dw, dst)
    assert np.array_equal(host(d_dst), dst)
    nz = dw != 0
    q = np.trunc(np.float64(before) * 256.0 / np.where(nz, dw, 1)[..., None]).astype(np.int64).astype(np.int16)      # C division truncates toward zero
    assert np.array_equal(dst[nz], q[nz]) and not dst[~nz].any()
    # the blender's weight maps in this flavour: mask 0 / 255 -> 16S, + 1 where set (= 256), 16SC1 pyrDown (blenders.cpp:414-423)
    mask = (rng.random((40, 56)) < 0.6).astype(np.int16) * 256
    assert np.array_equal(host(ms.pyr_down(to_dev(mask))), oracle.pyr_down_16s(mask))


def test_mask_ops(ms, cuda, oracle):
    rng = rng_for("mask")
    w = rng.random((40, 77), dtype=np.float32) * 2e-5
    m = host(ms.compare_gt(to_dev(w), 1e-5))
    assert np.array_equal(m, np.where(w > np.float32(1e-5), 255, 0).astype(np.uint8))
    inv = host(ms.compare_eq(to_dev(m), 0))
    assert np.array_equal(inv, np.where(m == 0, 255, 0).astype(np.uint8))
    img = rng.integers(-500, 500, size=(40, 77, 3), dtype=np.int16)
    d = to_dev(img)
    ms.set_zero_masked(d, to_dev(inv))
    img[inv != 0] = 0
    assert np.array_equal(host(d), img)
    a = rng.integers(0, 2, size=(40, 77), dtype=np.uint8) * 255
    assert np.array_equal(host(ms.bitwise_and(to_dev(a), to_dev(m))), a & m)
    assert np.array_equal(host(ms.dilate3x3(to_dev(a))), oracle.dilate3x3_8u(a))


@pytest.mark.parametrize("proj", ["plane", "cylindrical", "spherical"])
def test_build_warp_maps(ms, cuda, oracle, proj):
    """ocl/test_warpers.cpp:85-165: K = I-ish intrinsics, 30-degree roll, scale 2 -> tolerance 1e-4 (the reference's
    own device-vs-CPU bound; device sinf/cosf are not glibc's)."""
    import synth
    pid = {"plane": ms.PROJ_PLANE, "cylindrical": ms.PROJ_CYLINDRICAL, "spherical": ms.PROJ_SPHERICAL}[proj]
    K = np.eye(3, dtype=np.float32)
    a = math.radians(30)
    R = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], np.float32)
    k_rinv = oracle.k_rinv_gpu(K, R)
    mx, my = ms.build_warp_maps(pid, -3, 2, 64, 97, k_rinv, 2.0, t=(0.1, 0.0, 0.2) if proj == "plane" else None)
    rx, ry = oracle.build_warp_maps(pid, -3, 2, 64, 97, k_rinv, 2.0, t=(0.1, 0.0, 0.2) if proj == "plane" else (0, 0, 0))
    for g, r in ((host(mx), rx), (host(my), ry)):
        ok = np.isclose(g, r, rtol=1e-4, atol=1e-4) | ((np.abs(r) > 1e3) & np.isclose(g, r, rtol=1e-2))
        assert ok.all(), float(np.abs(g - r)[~ok].max())
    # the hot-path rig (f=960, scale 611): sub-1e-3 px agreement with the glibc-evaluated oracle maps
    K, R = synth.camera(6, 1920, 1080, 90.0, 1)
    k_rinv = oracle.k_rinv_gpu(K, R)
    mx, my = ms.build_warp_maps(ms.PROJ_SPHERICAL, 160, 646, 627, 960, k_rinv, synth.warp_scale(3840))
    rx, ry = oracle.build_warp_maps(ms.PROJ_SPHERICAL, 160, 646, 627, 960, k_rinv, synth.warp_scale(3840))
    assert np.abs(host(mx) - rx).max() < 2e-3 and np.abs(host(my) - ry).max() < 2e-3


def test_custom_resize(ms, cuda, oracle):
    rng = rng_for("cres")
    src = (rng.random((10, 12), dtype=np.float32) * 900).astype(np.float32)
    src[3, 4] = np.nan
    got = host(ms.custom_resize(to_dev(src), 961, 627))
    ref = oracle.custom_resize_32f(src, 961, 627)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])


def test_argument_errors(ms, cuda):
    a = torch.zeros((8, 8, 3), dtype=torch.uint8, device=cuda)
    f = torch.zeros((8, 8), dtype=torch.float32, device=cuda)
    with pytest.raises(ms.MsError, match="size mismatch"):
        ms.remap(a, f, f[:4], ms.INTER_LINEAR)
    with pytest.raises(ms.MsError, match="not on the hot path"):
        ms.copy_make_border(f, 1, 1, 1, 1, ms.BORDER_REFLECT)
    with pytest.raises(ms.MsError, match="at least 2x2"):
        ms.custom_resize(f[:1], 8, 8)


def test_one_instruction_saturate_cast_u8_is_exact(ms, cuda):
    """sat_u8 is v_cvt_pk_u8_f32; its definition is rint (half-even) -> clamp [0,255] -> NaN to 0 (saturate_cast<uchar>(float),
    saturate_cast.hpp:96-101).  Exhaustive over all 2^32 float bit patterns, plain and insert-into-byte forms."""
    assert ms.selftest_cvt_u8() == 0


def test_shared_reciprocal_division_is_ieee(ms, cuda):
    """The band kernels divide the 3 channels of a pixel by the same w + 1e-5 with one refined reciprocal (DivBy).
    Exhaustive over all int16 numerators x denominators spanning the weight-sum range (incl. 1e-5 itself, sums of
    1..12 unit weights, random values): every quotient must equal the compiler's correctly rounded a / d bit for bit."""
    rng = np.random.default_rng(99)
    eps = np.float32(1e-5)
    dens = [eps, np.float32(1.0) + eps, np.float32(2.0) + eps, np.float32(3.0) + eps, np.float32(12.0) + eps, np.float32(0.5) + eps]
    dens += list((rng.random(1500, dtype=np.float32) * np.float32(4.0) + eps).astype(np.float32))
    dens += list((np.float32(10.0) ** rng.uniform(-5, 1.5, 500)).astype(np.float32) + eps)
    assert ms.selftest_divide(np.array(dens, np.float32)) == 0


def test_shared_reciprocal_division_is_ieee_for_every_weight_sum(ms, cuda):
    """... and by enumeration: EVERY float32 denominator in [1e-5, 64) -- a weight sum of up to 16 views + 1e-5 lies in [1e-5, 16.00001] -- times
    all 65536 int16 numerators (1.2e13 quotients, enumerated on the device).  MS_TEST_DIVIDE_FULL=0 restricts the run to the binades
    [1e-5, 2^-16) and [0.5, 4) for a quick pass; the default is the full range."""
    import os
    if os.environ.get("MS_TEST_DIVIDE_FULL", "1") == "0":
        ranges = [(1e-5, float(np.nextafter(np.float32(2.0 ** -16), np.float32(0)))), (0.5, float(np.nextafter(np.float32(4.0), np.float32(0))))]
    else:
        ranges = [(1e-5, float(np.nextafter(np.float32(64.0), np.float32(0))))]
    total = 0
    for lo, hi in ranges:
        bad, n = ms.selftest_divide_range(lo, hi)
        assert bad == 0, (lo, hi, bad)
        total += n
    assert total >= (1 << 23) * 65536


@pytest.mark.parametrize("size", [(628, 3840), (36, 50), (2, 2), (10, 6)])
def test_bgr_to_i420(ms, cuda, oracle, size):
    """consume()'s cvtColor(COLOR_BGR2YUV_I420) (timed.cpp:308-316): BT.601 fixed point, chroma from the top-left pixel."""
    rng = rng_for("i420", size)
    src = rng.integers(0, 256, size=size + (3,), dtype=np.uint8)
    src[0, 0] = (255, 255, 255); src[-1, -1] = (0, 0, 0)
    ref = oracle.bgr_to_i420(src)
    assert np.array_equal(host(ms.bgr_to_i420(to_dev(src))), ref)
    assert np.array_equal(host(ms.bgr_to_i420(to_dev_roi(src, rng))), ref)       # padded source step
    assert ref[0, 0] == 235 and ref[size[0] - 1, size[1] - 1] == 16              # studio-swing luma of white / black


@pytest.mark.parametrize("size", [(1080, 1920), (36, 50), (2, 2)])
def test_nv12_to_bgr(ms, cuda, oracle, size):
    """The capture threads' cvtColor(CV_YUV2BGR_NV12) (networking.cpp:45-47; 1920x1620 NV12 frames, defs.h:10-17)."""
    rng = rng_for("nv12", size)
    h, w = size
    src = rng.integers(0, 256, size=(h * 3 // 2, w), dtype=np.uint8)
    src[0, :2] = (16, 235); src[h, :2] = (128, 128)                      # black / white with neutral chroma
    ref = oracle.nv12_to_bgr(src)
    assert np.array_equal(host(ms.nv12_to_bgr(to_dev(src))), ref)
    assert np.array_equal(host(ms.nv12_to_bgr(to_dev_roi(src, rng))), ref)
    assert tuple(ref[0, 0]) == (0, 0, 0) and tuple(ref[0, 1]) == (255, 255, 255)


def test_nv12_to_bgr_batch_equals_single_calls(ms, cuda, oracle):
    """All cameras of a frame in one launch: equal to the oracle and to the single calls; mixed geometries are refused."""
    rng = rng_for("nv12_batch", (6,))
    srcs = [rng.integers(0, 256, size=(90 * 3 // 2, 164), dtype=np.uint8) for _ in range(6)]
    got = ms.nv12_to_bgr_batch([to_dev(x) for x in srcs])
    for x, g in zip(srcs, got):
        assert np.array_equal(host(g), oracle.nv12_to_bgr(x))
    with pytest.raises(ms.MsError, match="one geometry"):
        ms.nv12_to_bgr_batch([to_dev(srcs[0]), to_dev(srcs[1][:60 * 3 // 2])])


@pytest.mark.parametrize("cn", [1, 3])
def test_remap_cpu_flavour_fixed_point(ms, cuda, oracle, cn):
    """ms_remap(MS_INTER_LINEAR_FIXPT) = cv::remap(INTER_LINEAR) on the CPU: 1/32-px coordinates, 15-bit weight table, (v + 2^14) >> 15
    (imgwarp.cpp:211-284, :643-850, :1203-1270); bit-exact against the C oracle and the independent numpy statement."""
    import np_ref
    rng = np.random.default_rng(40 + cn)
    src = rng.integers(0, 256, size=(97, 131) + ((3,) if cn == 3 else ()), dtype=np.uint8)
    yy, xx = np.mgrid[0:120, 0:150].astype(np.float32)
    mx = (0.93 * xx - 0.36 * yy + 20.3).astype(np.float32)
    my = (0.36 * xx + 0.87 * yy - 31.7).astype(np.float32)
    mx[3, 4] = np.nan; my[5, 6] = np.inf; mx[7, 8] = -1e20; my[8, 9] = 3e9
    mx[9, 10] = 17.0; my[9, 10] = 23.0
    mx[11, :20] = np.arange(20) - 1.5; my[11, :20] = -0.5
    mx[12, :20] = 129.0 + np.arange(20) * 0.125; my[12, :20] = 95.0 + np.arange(20) * 0.125       # right / bottom border taps
    got = host(ms.remap(to_dev(src), to_dev(mx), to_dev(my), interpolation=ms.INTER_LINEAR_FIXPT))
    assert np.array_equal(got, oracle.cv_remap_linear(src, mx, my))
    assert np.array_equal(got, np_ref.cv_remap_linear_np(src, mx, my))
    d = np.abs(got.astype(int) - host(ms.remap(to_dev(src), to_dev(mx), to_dev(my))).astype(int))
    assert 0 < d.max() <= 8          # the two flavours do differ on noise (up to 6)


@pytest.mark.parametrize("src_size,out_size,keep", [((3839, 627), (4096, 2048), True), ((1279, 401), (1024, 512), True), ((640, 480), (320, 120), True),
                                                    ((333, 97), (256, 128), False), ((96, 50), (128, 66), True)])
def test_consume_i420_equals_resize_then_bars_then_conversion(ms, cuda, oracle, src_size, out_size, keep):
    """consume() (timed.cpp:251-316) as one pass: must equal the three steps it fuses, each through the oracle -- cuda::resize's INTER_LINEAR to
    out_w x image_height (timed.cpp:261-271), the result in the middle of a black frame (:287), cvtColor(BGR2YUV_I420) -- and the per-op entry points."""
    rng = rng_for("consume", src_size + out_size)
    w, h = src_size
    ow, oh = out_size
    pano = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    got, ih = ms.consume_i420(to_dev(pano), out_size, keep_aspect_ratio=keep)
    want_ih = min(int(ow / w * h + 0.5), oh) if keep else oh
    assert ih == want_ih
    frame = np.zeros((oh, ow, 3), np.uint8)
    y0 = oh // 2 - ih // 2
    frame[y0:y0 + ih] = oracle.resize_linear_8u(pano, dsize=(ow, ih))
    assert np.array_equal(host(got), oracle.bgr_to_i420(frame))
    small = ms.resize_linear(to_dev(pano), dsize=(ow, ih))
    fr = torch.zeros((oh, ow, 3), dtype=torch.uint8, device=cuda)
    fr[y0:y0 + ih] = small
    assert torch.equal(got, ms.bgr_to_i420(fr))
    got2, _ = ms.consume_i420(to_dev_roi(pano, rng), out_size, keep_aspect_ratio=keep)       # padded source step
    assert torch.equal(got, got2)


def test_resize_linear_batch_equals_single_calls(ms, cuda, oracle):
    """stitch_online's per-view cuda::resize by compose_scale (timed.cpp:75-85) for all views in one launch: equal to the single calls and to the oracle."""
    rng = rng_for("resize_batch", (6,))
    srcs = [rng.integers(0, 256, size=(270, 480, 3), dtype=np.uint8) for _ in range(6)]
    s = math.sqrt(1.4e6 / (1920 * 1080))            # compose_scale of the shipped COMPOSE_MEGAPIX for 1080p (calibration.cpp:149)
    got = ms.resize_linear_batch([to_dev(x) for x in srcs], fx=s, fy=s)
    for x, g in zip(srcs, got):
        assert np.array_equal(host(g), oracle.resize_linear_8u(x, fx=s, fy=s))
        assert torch.equal(g, ms.resize_linear(to_dev(x), fx=s, fy=s))
    with pytest.raises(ms.MsError, match="one geometry"):
        ms.resize_linear_batch([to_dev(srcs[0]), to_dev(srcs[1][:100])], fx=s, fy=s)


@pytest.mark.parametrize("shape,scale", [((1080, 1920), math.sqrt(1.4e6 / (1920 * 1080))), ((97, 131), 0.8217), ((64, 75), 0.99), ((50, 61), 0.63), ((33, 47), 0.7),
                                         ((41, 19), 0.9), ((30, 40), 0.45), ((20, 24), 1.0 / 1.6)])
def test_resize_linear_batch_four_pixel_kernel_edges(ms, cuda, oracle, shape, scale):
    """The per-frame batch kernel handles 4 output pixels per lane out of one 24-byte source window per row (k_resize_linear3_x4): ragged right edges,
    windows that would leave the source row, row pitches that are not multiples of 4, the last source row, scales at both ends of its range and beyond it
    (where the launcher falls back to one pixel per lane) -- all bit-identical to the oracle's restatement of cuda::resize (resize.cu:71-106)."""
    rng = rng_for("resize_x4", shape)
    srcs = [rng.integers(0, 256, size=shape + (3,), dtype=np.uint8) for _ in range(3)]
    t0, l0, l1 = [int(rng.integers(1, 9)) for _ in range(3)]           # embedded in a larger allocation: a row pitch that is no multiple of 4 (one geometry for the batch)
    big = np.zeros((3, shape[0] + 2 * t0, shape[1] + l0 + l1, 3), np.uint8)
    for k, x in enumerate(srcs):
        big[k, t0:t0 + shape[0], l0:l0 + shape[1]] = x
    dev = to_dev(big)
    pitched = [dev[k, t0:t0 + shape[0], l0:l0 + shape[1]] for k in range(3)]
    got = ms.resize_linear_batch(pitched, fx=scale, fy=scale)
    for x, g in zip(srcs, got):
        assert np.array_equal(host(g), oracle.resize_linear_8u(x, fx=scale, fy=scale)), (shape, scale)


def test_bgr_to_i420_batch_equals_single_calls(ms, cuda):
    rng = np.random.default_rng(77)
    frames = [to_dev(rng.integers(0, 256, (46, 64, 3), dtype=np.uint8)) for _ in range(70)]      # > one launch's table of 64
    canvases = [torch.zeros((80, 64, 3), dtype=torch.uint8, device=cuda) for _ in frames]
    for c, f in zip(canvases, frames):
        c[10:56] = f
    srcs = [c[10:56] for c in canvases]                                                            # rows of a larger canvas, as bench.py passes them
    dsts = [torch.zeros((69, 64), dtype=torch.uint8, device=cuda) for _ in frames]
    ms.bgr_to_i420_batch_prepared(srcs, dsts)()
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert np.array_equal(host(d), host(ms.bgr_to_i420(s)))
    with pytest.raises(ms.MsError):
        ms.bgr_to_i420_batch_prepared([srcs[0], srcs[1][:44]], dsts[:2])()


def test_bgr_to_gray(ms, cuda):
    """cuda::cvtColor(BGR2GRAY) (color_detail.hpp:444-447): exact integer arithmetic, every size / pitch path."""
    rng = np.random.default_rng(5)
    for h, w in ((37, 64), (20, 61), (3, 2), (1, 1)):
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = ((src[..., 0].astype(np.uint32) * 1868 + src[..., 1].astype(np.uint32) * 9617 + src[..., 2].astype(np.uint32) * 4899 + (1 << 13)) >> 14).astype(np.uint8)
        assert np.array_equal(host(ms.bgr_to_gray(to_dev(src))), want)
        assert np.array_equal(host(ms.bgr_to_gray(to_dev_roi(src, rng))), want)
    white = np.full((2, 8, 3), 255, np.uint8)
    assert (host(ms.bgr_to_gray(to_dev(white))) == 255).all()


def test_calib_shape_touches_every_line_once(ms, cuda):
    """ms_calib_shape (PMC calibration on the per-frame kernels' access shapes): shapes 0 / 1 only read; shape 2 writes every dword of the buffer exactly once in four 32-byte passes per line"""
    buf = torch.full((1 << 20,), 0xAB, dtype=torch.uint8, device=cuda)
    for shape in (0, 1):
        ms.calib_shape(buf, shape)
    torch.cuda.synchronize()
    assert int((buf != 0xAB).sum()) == 0
    ms.calib_shape(buf, 2)
    torch.cuda.synchronize()
    w = buf.view(torch.int32).cpu().numpy().reshape(-1, 4, 8)          # [line, quarter, dword]: value = lane index i = 8 * line + dword
    want = (8 * np.arange(w.shape[0])[:, None, None] + np.arange(8)[None, None, :]) * np.ones((1, 4, 1), np.int64)
    assert np.array_equal(w.astype(np.int64), want)
    with pytest.raises(ms.MsError):
        ms.calib_shape(buf[1:], 0)          # not 128-byte aligned
