"""Execution pin of the oracle's rounding / saturation helpers against the reference's OWN code: oracle/_ref/libref_pin.so is a harness over
the header-only cv::saturate_cast / cvRound / cvFloor of /root/reference/sources/modules/core/include/opencv2/core/{saturate,fast_math}.hpp,
built by `make -C oracle ref` where the reference is present (this container; the .so travels to the GPU box).  It pins these helpers and
nothing else: the pixel pipelines stay a restatement (Oracle)."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_pin.so")


@pytest.fixture(scope="module")
def pin(oracle):
    if not os.path.exists(REF):
        if os.path.isdir("/root/reference"):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libref_pin.so not built (needs /root/reference)")
    ref = C.CDLL(REF)
    ref.ref_pin_sweep_f32.restype = C.c_longlong
    ref.ref_pin_sweep_f32.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_ulonglong, C.c_float, C.c_int, C.POINTER(C.c_uint32)]
    ref.ref_pin_sweep_i32.restype = C.c_longlong
    ref.ref_pin_sweep_i32.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_ulonglong]
    return ref, oracle.lib()


def _fp(lib, name):
    return C.cast(getattr(lib, name), C.c_void_p)


@pytest.mark.parametrize("which,helper,limit,skip_nan", [
    # cv::saturate_cast<T>(float) = saturate(cvRound(v)); cvRound is INT_MIN beyond 2^31 and for NaN, where the CUDA flavour (cvt.rni.sat: keeps
    # saturating, NaN -> 0) is a different function: compared where both are the same one.  (uchar: INT_MIN saturates to 0, so NaN agrees too.)
    (0, "orc_helper_sat_u8f", 2147483648.0, 0),
    (1, "orc_helper_sat_s16f", 2147483648.0, 1),
])
def test_saturate_cast_float_all_bit_patterns(pin, which, helper, limit, skip_nan):
    ref, orc = pin
    bad_at = C.c_uint32(0)
    bad = ref.ref_pin_sweep_f32(which, _fp(orc, helper), 0, 1 << 32, limit, skip_nan, C.byref(bad_at))
    assert bad == 0, "%s differs from the reference header on %d float patterns, first 0x%08x" % (helper, bad, bad_at.value)


def test_cvround_all_bit_patterns(pin):
    """cvRound(float) (SSE cvtss2si: nearest-even, 0x80000000 for NaN / out of range) = the oracle's cv_round_f, every pattern incl. NaN."""
    ref, orc = pin
    bad_at = C.c_uint32(0)
    bad = ref.ref_pin_sweep_f32(2, _fp(orc, "orc_helper_cv_round_f"), 0, 1 << 32, float("inf"), 0, C.byref(bad_at))
    assert bad == 0, "cv_round_f differs on %d patterns, first 0x%08x" % (bad, bad_at.value)


def test_cvfloor_matches_float2int_rd_where_representable(pin):
    """cvFloor(float) against the oracle's __float2int_rd restatement for |v| < 2^31 (beyond, CUDA saturates and the CPU returns INT_MIN)."""
    ref, orc = pin
    bad_at = C.c_uint32(0)
    bad = ref.ref_pin_sweep_f32(3, _fp(orc, "orc_helper_f2i_rd"), 0, 1 << 32, 2147483648.0, 1, C.byref(bad_at))
    assert bad == 0, "f2i_rd differs on %d patterns, first 0x%08x" % (bad, bad_at.value)


def test_saturate_cast_short_from_int(pin):
    ref, orc = pin
    assert ref.ref_pin_sweep_i32(0, _fp(orc, "orc_helper_sat_s16i"), -(1 << 24), 1 << 25) == 0
    for first in (-(1 << 31), (1 << 31) - (1 << 20)):
        assert ref.ref_pin_sweep_i32(0, _fp(orc, "orc_helper_sat_s16i"), first, 1 << 20) == 0
