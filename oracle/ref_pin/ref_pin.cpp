// oracle/_ref/libref_pin.so -- the ONLY piece of the reference that builds here under the rules: a harness over the reference's
// header-only rounding / saturation helpers, compiled where they lie:
//     /root/reference/sources/modules/core/include/opencv2/core/{cvdef.h, fast_math.hpp, saturate.hpp}
// (they include no cmake-generated file; everything else of the reference's CPU path does: opencv_modules.hpp, cvconfig.h, ...).
// It pins -- by execution of the reference's own code -- exactly this and nothing more: the oracle's sat_u8f / sat_s16f / sat_s16i /
// cv_round_f helpers equal cv::saturate_cast<uchar>(float), cv::saturate_cast<short>(float), cv::saturate_cast<short>(int), cvRound(float)
// wherever the CPU and the CUDA flavours are both defined the same way.  Test infrastructure; never linked into the product.
#include <cstdint>
#include <cstring>
#include "opencv2/core/cvdef.h"
#include "opencv2/core/fast_math.hpp"
#include "opencv2/core/saturate.hpp"

extern "C" {

typedef int (*fn_f32_i32)(float);
typedef int (*fn_i32_i32)(int);

static inline float bits_to_float(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }

// which = 0: cv::saturate_cast<uchar>(float)   1: cv::saturate_cast<short>(float)   2: cvRound(float)   3: cvFloor(float)
int ref_pin_eval_f32(int which, float v)
{
    switch (which) {
    case 0: return (int)cv::saturate_cast<uchar>(v);
    case 1: return (int)cv::saturate_cast<short>(v);
    case 2: return cvRound(v);
    default: return cvFloor(v);
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
