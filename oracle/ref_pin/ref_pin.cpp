// oracle/_ref/libref_pin.so -- the ONLY piece of the reference that builds here under the rules: a harness over the reference's
// header-only rounding / saturation helpers, compiled where they lie:
//     /root/reference/sources/modules/core/include/opencv2/core/{cvdef.h, hal/interface.h, fast_math.hpp, saturate.hpp, version.hpp}
// (they include no cmake-generated file; everything else of the reference's CPU path does: opencv_modules.hpp, cvconfig.h, ...).
// It pins -- by execution of the reference's own code -- exactly this and nothing more: the oracle's sat_u8f / sat_s16f / sat_s16i /
// cv_round_f helpers equal cv::saturate_cast<uchar>(float), cv::saturate_cast<short>(float), cv::saturate_cast<short>(int), cvRound(float)
// wherever the CPU and the CUDA flavours are both defined the same way.  Test infrastructure; never linked into the product.
#include <cstdint>
#include <cstring>
#include "opencv2/core/cvdef.h"
#include "opencv2/core/fast_math.hpp"
#include "opencv2/core/saturate.hpp"
#include "opencv2/core/version.hpp"

extern "C" {

typedef int (*fn_f32_i32)(float);
typedef int (*fn_i32_i32)(int);

static inline float bits_to_float(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }

// which = 0: cv::saturate_cast<uchar>(float)   1: cv::saturate_cast<short>(float)   2: cvRound(float)   3: cvFloor(float)   4: cvCeil(float)
int ref_pin_eval_f32(int which, float v)
{
    switch (which) {
    case 0: return (int)cv::saturate_cast<uchar>(v);
    case 1: return (int)cv::saturate_cast<short>(v);
    case 2: return cvRound(v);
    case 4: return cvCeil(v);
    default: return cvFloor(v);
    }
}
// round 6: the double-precision forms the HOST geometry goes through -- cvRound(double) (calibration.cpp:163-164: the compose-scale frame size), cv::saturate_cast<int>(double)
// (cuda::resize's dsize, cudawarping/src/resize.cpp), cvIsNaN / cvIsInf (fast_math.hpp) -- for n values each
void ref_pin_eval_f64(int which, const double *v, int *out, long long n)
{
    for (long long i = 0; i < n; ++i)
        out[i] = which == 0 ? cvRound(v[i]) : which == 1 ? cv::saturate_cast<int>(v[i]) : which == 2 ? cvIsNaN(v[i]) : which == 3 ? cvIsInf(v[i]) : which == 4 ? cvFloor(v[i]) : cvCeil(v[i]);
}
// The type codes and element sizes the C-ABI's ms_image.type restates (include/ms_stitch.h:52: MS_8UC1 = 0, MS_8UC3 = 16, MS_16SC1 = 3, MS_16SC3 = 19, MS_32FC1 = 5) as the
// reference's own macros evaluate them (core/hal/interface.h: CV_8U ... CV_CN_SHIFT; core/cvdef.h: CV_MAKETYPE, CV_ELEM_SIZE), and the version of the vendored OpenCV.
// which = 0..4: CV_8UC1, CV_8UC3, CV_16SC1, CV_16SC3, CV_32FC1;  10..14: CV_ELEM_SIZE of the same;  20..22: CV_VERSION_MAJOR / MINOR / REVISION;  30: CV_CN_MAX
int ref_pin_const(int which)
{
    switch (which) {
    case 0: return CV_MAKETYPE(CV_8U, 1);
    case 1: return CV_MAKETYPE(CV_8U, 3);
    case 2: return CV_MAKETYPE(CV_16S, 1);
    case 3: return CV_MAKETYPE(CV_16S, 3);
    case 4: return CV_MAKETYPE(CV_32F, 1);
    case 10: return CV_ELEM_SIZE(CV_MAKETYPE(CV_8U, 1));
    case 11: return CV_ELEM_SIZE(CV_MAKETYPE(CV_8U, 3));
    case 12: return CV_ELEM_SIZE(CV_MAKETYPE(CV_16S, 1));
    case 13: return CV_ELEM_SIZE(CV_MAKETYPE(CV_16S, 3));
    case 14: return CV_ELEM_SIZE(CV_MAKETYPE(CV_32F, 1));
    case 20: return CV_VERSION_MAJOR;
    case 21: return CV_VERSION_MINOR;
    case 22: return CV_VERSION_REVISION;
    case 30: return CV_CN_MAX;
    default: return -1;
    }
}
// which = 0: cv::saturate_cast<short>(int)   1: cv::saturate_cast<uchar>(int)
int ref_pin_eval_i32(int which, int v) { return which == 0 ? (int)cv::saturate_cast<short>(v) : (int)cv::saturate_cast<uchar>(v); }

// Sweeps the float bit patterns [first, first + count) (count <= 2^32) and counts the inputs on which `f` differs from the reference
// function `which`; only |v| < limit (and, if skip_nan, non-NaN) inputs are compared.  first_bad receives the first differing pattern.
long long ref_pin_sweep_f32(int which, fn_f32_i32 f, uint32_t first, unsigned long long count, float limit, int skip_nan, uint32_t *first_bad)
{
    long long bad = 0;
    uint32_t fb = 0;
    int have = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (long long i = 0; i < (long long)count; ++i) {
        const uint32_t b = first + (uint32_t)i;
        const float v = bits_to_float(b);
        if (v != v) { if (skip_nan) continue; }
        else if (!(v > -limit && v < limit)) continue;
        if (ref_pin_eval_f32(which, v) != f(v)) {
            ++bad;
#pragma omp critical
            if (!have || b < fb) { fb = b; have = 1; }
        }
    }
    if (first_bad) *first_bad = fb;
    return bad;
}
// every float bit pattern with |v| < 2^31 through cvCeil(float) and cvFloor(float) against libm's ceilf / floorf: the count of patterns on which either differs
long long ref_pin_sweep_ceil_floor_f32(void)
{
    long long bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (long long i = 0; i < (1ll << 32); ++i) {
        const float v = bits_to_float((uint32_t)i);
        if (!(v > -2147483000.f && v < 2147483000.f)) continue;
        bad += (cvCeil(v) != (int)__builtin_ceilf(v)) + (cvFloor(v) != (int)__builtin_floorf(v));
    }
    return bad;
}
long long ref_pin_sweep_i32(int which, fn_i32_i32 f, int first, unsigned long long count)
{
    long long bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (long long i = 0; i < (long long)count; ++i) {
        const int v = (int)((long long)first + i);
        bad += ref_pin_eval_i32(which, v) != f(v);
    }
    return bad;
}

}  // extern "C"
