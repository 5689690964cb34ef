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


# ---- round 6: what else of the reference compiles from its own files alone -------------------------------------------------------------
# core/hal/interface.h + cvdef.h (type codes, element sizes), version.hpp, the double-precision rounding helpers of fast_math.hpp / saturate.hpp the HOST geometry goes
# through, and the application's own 360_stitcher/defs.h (self-contained: plain const definitions).  Everything else on the path includes a cmake-generated header
# (opencv_modules.hpp through core/base.hpp, cvconfig.h through precomp.hpp) or <cuda_runtime.h> (every core/cuda/*.hpp through common.hpp:46): DESIGN.md section 3 lists it.
@pytest.fixture(scope="module")
def pin6(pin):
    ref, _ = pin
    if not hasattr(ref, "ref_pin_const"):
        pytest.skip("oracle/_ref/libref_pin.so predates round 6 (rebuild with `make -C oracle ref` where /root/reference exists)")
    ref.ref_pin_eval_f64.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong]
    ref.ref_pin_eval_f64.restype = None
    ref.ref_pin_app_int.argtypes = [C.c_char_p, C.POINTER(C.c_longlong)]
    ref.ref_pin_app_double.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    return ref


def _app_int(ref, name):
    v = C.c_longlong(0)
    assert ref.ref_pin_app_int(name.encode(), C.byref(v)) == 1, name
    return v.value


def _app_double(ref, name):
    v = C.c_double(0)
    assert ref.ref_pin_app_double(name.encode(), C.byref(v)) == 1, name
    return v.value


def test_type_codes_and_element_sizes_equal_the_references_macros(pin6):
    """ms_image.type restates CV_MAKETYPE(depth, cn) (include/ms_stitch.h:52; core/hal/interface.h + cvdef.h evaluated by the reference's own preprocessor), and the element
    sizes the binding derives from it are CV_ELEM_SIZE; the vendored OpenCV is 3.4.0 (the version every file:line citation of DESIGN.md / oracle/ refers to)."""
    import re
    import msstitch as ms
    hdr = open(os.path.join(ROOT, "include", "ms_stitch.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"(MS_(?:8U|16S|32F)C\d)\s*=\s*(\d+)", hdr))
    names = ["MS_8UC1", "MS_8UC3", "MS_16SC1", "MS_16SC3", "MS_32FC1"]
    sizes = [1, 3, 2, 6, 4]
    for i, n in enumerate(names):
        assert enum[n] == pin6.ref_pin_const(i) == getattr(ms, n), (n, enum[n], pin6.ref_pin_const(i))
        assert pin6.ref_pin_const(10 + i) == sizes[i]
    assert (pin6.ref_pin_const(20), pin6.ref_pin_const(21), pin6.ref_pin_const(22)) == (3, 4, 0)
    assert pin6.ref_pin_const(30) == 512


def test_double_precision_rounding_helpers(pin6):
    """cvRound(double) (calibration.cpp:163-164: the compose-scale frame size -> csrc/geometry.cpp nearbyint) and cv::saturate_cast<int>(double) (cuda::resize's dsize ->
    csrc/api.cpp __builtin_rint, oracle.py np.rint) are round-half-even; cvFloor / cvCeil (double) are floor / ceil; cvIsNaN / cvIsInf classify like IEEE: 4 M doubles -- random
    bit patterns inside the int range, every tie k + 0.5, every size x scale product the geometry code can form -- through the reference's own inline functions."""
    import numpy as np
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 1 << 64, size=3_000_000, dtype=np.uint64)
    v = bits.view(np.float64)
    v = v[np.isfinite(v) & (np.abs(v) < 2147483000.0)]
    ties = np.arange(-200000, 200000, dtype=np.float64) + 0.5
    sizes = np.arange(1, 8193, dtype=np.float64)[:, None] * np.sqrt(np.array([0.6, 0.01, 1.4, 2.2]) * 1e6 / (1920.0 * 1080.0))[None, :]
    v = np.ascontiguousarray(np.concatenate([v, ties, sizes.ravel(), rng.uniform(-70000, 70000, 500000)]))
    out = np.empty(v.size, np.int32)

    def ref(which, x=v, o=out):
        pin6.ref_pin_eval_f64(which, x.ctypes.data, o.ctypes.data, x.size)
        return o.copy()
    assert np.array_equal(ref(0), np.rint(v).astype(np.int64).astype(np.int32))          # cvRound(double)
    assert np.array_equal(ref(1), np.rint(v).astype(np.int64).astype(np.int32))          # saturate_cast<int>(double)
    assert np.array_equal(ref(4), np.floor(v).astype(np.int64).astype(np.int32))         # cvFloor(double)
    assert np.array_equal(ref(5), np.ceil(v).astype(np.int64).astype(np.int32))          # cvCeil(double)
    special = np.ascontiguousarray(np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, 5e-324, 1.7976931348623157e308, np.float64(np.float32(np.nan))]))
    o = np.empty(special.size, np.int32)
    assert np.array_equal(ref(2, special, o) != 0, np.isnan(special)) and np.array_equal(ref(3, special, o) != 0, np.isinf(special))


def test_cvceil_and_cvfloor_float_all_bit_patterns(pin6):
    """cvCeil(float) / cvFloor(float) against ceilf / floorf for every float pattern with |v| < 2^31 (beyond, the SSE conversion returns INT_MIN), swept inside the harness."""
    pin6.ref_pin_sweep_ceil_floor_f32.restype = C.c_longlong
    assert pin6.ref_pin_sweep_ceil_floor_f32() == 0


def test_application_constants_equal_defs_h(pin6):
    """The drop-in surface restates constants of the reference's 360_stitcher/defs.h; here they are read from THAT file, compiled where it lies: the mesh optimiser's defaults
    (ms_mesh_default_params), the shim's recalibration constants, the shipped configuration bench.py / stitch_app run (`--config shipped`, `--reference-calib`), consume()'s
    output size, and the rig size every BASELINE 1080p config assumes."""
    import re
    import msstitch as ms
    p = ms.mesh_default_params()
    assert (p.mesh_rows, p.mesh_cols) == (_app_int(pin6, "MESH_HEIGHT"), _app_int(pin6, "MESH_WIDTH")) == (10, 10)
    for i in range(4):
        assert np_f32(p.alphas[i]) == np_f32(_app_double(pin6, "ALPHAS%d" % i)), i
    assert p.global_dist == _app_int(pin6, "GLOBAL_DIST") and p.wrap_around == _app_int(pin6, "wrapAround") == 1
    assert _app_int(pin6, "USE_TEMPORAL") == 0 and _app_int(pin6, "enable_local") == 1 and _app_int(pin6, "recalibrate") == 1
    shim = open(os.path.join(ROOT, "video-stitcher_amd", "shim", "ms_shim.hpp")).read()
    for name in ("RECALIB_THRESH", "MAX_FEATURES_PER_IMAGE"):
        m = re.search(r"static const int %s = (\d+);" % name, shim)
        assert m and int(m.group(1)) == _app_int(pin6, name), name
    app = open(os.path.join(ROOT, "video-stitcher_amd", "host", "stitch_app.cpp")).read()
    m = re.search(r"work_mp = ([\d.]+), seam_mp = ([\d.]+), compose_mp = ([\d.]+);", app)
    assert m and [float(x) for x in m.groups()] == [_app_double(pin6, "WORK_MEGAPIX"), _app_double(pin6, "SEAM_MEAGPIX"), _app_double(pin6, "COMPOSE_MEGAPIX")] == [0.6, 0.01, 1.4]
    regions = open(os.path.join(ROOT, "benchlib", "regions.py")).read()          # the shipped rig of bench.py: calibrate_cameras(..., 0.6, 0.01, 1.4), num_bands_rule(..., 5.0), 10 x 10 meshes
    assert 'cfg["hfov_deg"], 0.6, 0.01, 1.4)' in regions and "num_bands_rule(pr[2], pr[3], 5.0)" in regions and "(10, 10) if self.shipped" in regions
    assert _app_double(pin6, "BLEND_STRENGTH") == 5.0 and _app_int(pin6, "NUM_IMAGES") == 6
    import inspect
    assert "out_size=(%d, %d)" % (_app_int(pin6, "OUTPUT_WIDTH"), _app_int(pin6, "OUTPUT_HEIGHT")) in str(inspect.signature(ms.consume_i420))
    assert (_app_int(pin6, "CAPTURE_IMG_WIDTH"), _app_int(pin6, "CAPTURE_IMG_HEIGHT") * 2 // 3) == (1920, 1080)      # NV12: 1080 luma rows + 540 chroma rows
    assert _app_int(pin6, "keep_aspect_ratio") == 1 and abs(_app_double(pin6, "PI") - 3.141592653589793) < 1e-15


def np_f32(x):
    import numpy as np
    return float(np.float32(x))
