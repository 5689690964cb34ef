// common.hpp -- shared host/device helpers of libmsstitch (gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/ms_stitch.h"

namespace ms {

// ---- error plumbing: status code + thread-local message (never throw across the C ABI) ----------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
int require_device();

#define MS_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return ::ms::fail(MS_ERR_HIP, "%s: %s (%s:%d)", #expr,              \
                                                hipGetErrorString(e_), __FILE__, __LINE__);       \
    } while (0)

#define MS_CHECK(cond, ...)                                                                       \
    do { if (!(cond)) return ::ms::fail(MS_ERR_INVALID, __VA_ARGS__); } while (0)

#define MS_LAUNCH_CHECK() MS_HIP(hipGetLastError())

// Per-thread pinned staging memory for the entry points that move host arrays while other threads use the device (the recalibration
// thread's mesh solve): copies from pinned memory are asynchronous for real and never go through the runtime's shared staging buffers.
// Grows on demand, lives as long as the thread's library use (deliberately not freed at exit: the runtime may already be gone).
struct PinnedScratch {
    void *p = nullptr;
    size_t cap = 0;
    int dev = -1;                 // the block belongs to the device that was current when it was allocated
    void *get(size_t n)
    {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != dev) { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; dev = cur; }
        if (n > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = (n + (1u << 20)) & ~(size_t)((1u << 20) - 1);
            if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return nullptr; }
            cap = want;
        }
        return p;
    }
};
PinnedScratch &pinned_scratch();
// Per-thread device scratch for the same entry points: a grow-only hipMalloc block.  NOT the stream-ordered allocator: with hipMallocAsync /
// hipFreeAsync, while a second thread was uploading frames and synchronising on events, one mesh solve in ~30 found its freshly uploaded
// block zeroed or stale (ROCm 7.2; the pool trims on every synchronisation) -- reproduced 5 runs in 25 with stitch_app --solve-mesh, 0 in 55
// with this block (tests/test_host_app_gpu.py runs the scenario).
struct DeviceScratch {
    void *p = nullptr;
    size_t cap = 0;
    int dev = -1;                 // a thread that switches devices (hipSetDevice) must not reuse another GPU's block
    void *get(size_t n)
    {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != dev) {         // (the old block is freed on its own device)
            if (p) { if (dev >= 0) (void)hipSetDevice(dev); (void)hipFree(p); (void)hipSetDevice(cur); }
            p = nullptr; cap = 0; dev = cur;
        }
        if (n > cap) {
            if (p) (void)hipFree(p);
            p = nullptr; cap = 0;
            const size_t want = (n + (1u << 20)) & ~(size_t)((1u << 20) - 1);
            if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return nullptr; }
            cap = want;
        }
        return p;
    }
};
DeviceScratch &device_scratch();

// Developer A/B knobs (occupancy, kernel variants): read from the environment ONLY in a -DMS_DEV_KNOBS build (tools/build_ab.sh).  In the production
// library they are their defaults at compile time: which kernels a deployed libmsstitch.so launches is a function of the context, never of the
// caller's environment.  dflt is returned when the variable is unset.
#ifdef MS_DEV_KNOBS
static inline int dev_knob(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline constexpr int dev_knob(const char *, int dflt) { return dflt; }
#endif

static inline int div_up(int a, int b) { return (a + b - 1) / b; }
static inline hipStream_t as_stream(ms_stream s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device arithmetic with the reference's rounding semantics ----------------------------------
// saturate_cast<uchar>(float) == cvt.rni.sat.u8.f32 (round-half-even, clamp, NaN -> 0)
__device__ __forceinline__ uint8_t sat_u8_ref(float v)     // the definition, operation by operation
{
    float r = __builtin_rintf(v);
    r = __builtin_fminf(__builtin_fmaxf(r, 0.f), 255.f);   // fmax(NaN, 0) = 0
    return (uint8_t)(int)r;
}
// ... and the one-instruction form: v_cvt_pk_u8_f32 rounds half-even, clamps to [0,255] and maps NaN to 0 -- checked
// against sat_u8_ref over all 2^32 float bit patterns by ms_selftest_cvt_u8.  `sat_u8_into` also inserts the
// result into byte `k` of a packed word (the instruction's third operand), so packing four pixels costs nothing.
__device__ __forceinline__ unsigned sat_u8_into(float v, unsigned k, unsigned word) { return __builtin_amdgcn_cvt_pk_u8_f32(v, k, word); }
__device__ __forceinline__ uint8_t sat_u8(float v) { return (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u); }
// saturate_cast<short>(float) == cvt.rni.sat.s16.f32
// cv::remap on the CPU (INTER_LINEAR, float map pair, BORDER_CONSTANT 0, 8-bit): RemapInvoker quantises the coordinates to 1/32 px with
// cvtps2dq (nearest-even, 0x80000000 when unrepresentable), remapBilinear sums tap * BilinearTab_i over the taps inside the image and
// rounds with (v + 2^14) >> 15 (imgwarp.cpp:1203-1270, :643-850).  The table is (32-fy)(32-fx)*32, ... exactly, except entry (0, 0):
// 32768 saturates to 32767 and initInterTab2D's fix-up (imgwarp.cpp:249-265) puts the missing 1 on the last tap.
__device__ __forceinline__ int cv_round_sse(float v)
{
    const float r = __builtin_rintf(v);
    return (r >= -2147483648.f && r < 2147483648.f) ? (int)r : (int)0x80000000;
}
template <int CN>
__device__ __forceinline__ void remap_fixpt(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols, float xc, float yc, uint8_t out[CN])
{
    const int qx = cv_round_sse(xc * 32.f), qy = cv_round_sse(yc * 32.f);
    const int fx = qx & 31, fy = qy & 31;
    const int sx = max(-32768, min(32767, qx >> 5)), sy = max(-32768, min(32767, qy >> 5));
    int w[4] = {(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32};
    if ((fx | fy) == 0) { w[0] = 32767; w[3] = 1; }
    int acc[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) acc[c] = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xx = sx + (t & 1), yy = sy + (t >> 1);
        const bool inb = xx >= 0 && xx < scols && yy >= 0 && yy < srows;
        const uint8_t *p = src + (size_t)(inb ? yy : 0) * sstep + (size_t)(inb ? xx : 0) * CN;
#pragma unroll
        for (int c = 0; c < CN; ++c) acc[c] += inb ? (int)p[c] * w[t] : 0;
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) out[c] = (uint8_t)min(255, (acc[c] + (1 << 14)) >> 15);
}

__device__ __forceinline__ int16_t sat_s16(float v)
{
    float r = __builtin_rintf(v);
    r = __builtin_fminf(__builtin_fmaxf(r, -32768.f), 32767.f);
    return (int16_t)(int)r;
}
__device__ __forceinline__ int16_t sat_s16(int v)
{
    return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}
// static_cast<short>(float) == cvt.rzi.s32.f32, low 16 bits (v_cvt_i32_f32: rtz, saturating, NaN -> 0)
__device__ __forceinline__ int16_t trunc_s16(float v) { return (int16_t)(int)v; }
// __float2int_rd / __float2int_rz
__device__ __forceinline__ int f2i_rd(float v) { return (int)__builtin_floorf(v); }
__device__ __forceinline__ int f2i_rz(float v) { return (int)v; }

// BORDER_REFLECT (cudev BrdReflect) and BORDER_REFLECT_101 (BrdReflect101) index maps
__device__ __forceinline__ int reflect_idx(int i, int len)
{
    const int last = len - 1;
    const int hi = last - abs(last - i) + (i > last);
    return (abs(hi) - (hi < 0)) % len;
}
__device__ __forceinline__ int r101_low(int i, int len) { return abs(i) % len; }
__device__ __forceinline__ int r101_high(int i, int len) { const int last = len - 1; return abs(last - abs(last - i)) % len; }
__device__ __forceinline__ int r101(int i, int len) { return r101_low(r101_high(i, len), len); }
// pyrUp source index: min(n-1, |i|)
__device__ __forceinline__ int pu_idx(int i, int n) { i = abs(i); return i < n - 1 ? i : n - 1; }

// exact integer forms of the 16S pyramids (the fp32 sums in the CUDA kernels are exact dyadics,
// so round-half-even of S/2^k reproduces saturate_cast<short>(float) bit for bit)
__device__ __forceinline__ int rne_shift(int s, int k)
{
    return (s + ((1 << (k - 1)) - 1) + ((s >> k) & 1)) >> k;
}


// row * pitch for index arithmetic, both below 2^24: the full-rate 24-bit multiplier (v_mul_u32_u24 / v_mad_u32_u24) instead of the quarter-rate
// v_mul_lo_u32 / v_mad_u64_u32 the compiler emits for int * int and (size_t) * int
// per-half logical shift right of two packed 16-bit values (v_pk_lshrrev_b16): no bits move from the high half into the low one, so no mask afterwards
typedef unsigned short ms_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_lshr16(unsigned x, unsigned n)
{
    ms_u16x2 v;
    __builtin_memcpy(&v, &x, 4);
    v >>= (unsigned short)n;
    __builtin_memcpy(&x, &v, 4);
    return x;
}
__device__ __forceinline__ unsigned mul24(int a, int b) { return __umul24((unsigned)a, (unsigned)b); }

// acc + 6 x with two full-rate v_lshl_add_u32 (x + 2x, then acc + 2 (3x)) instead of the quarter-rate v_mul_lo_u32 the compiler picks for
// `6u * x` on packed 16-bit pairs (the value does not fit the 24-bit multiplier).  The empty asm keeps instcombine from re-forming the multiply.
__device__ __forceinline__ unsigned mad6(unsigned x, unsigned acc)
{
    unsigned x3 = x + (x << 1);
    asm volatile("" : "+v"(x3));
    return acc + (x3 << 1);
}

// ---- backward warp maps (build_warp_maps.cu:67-134), split into a column term, a row term and a combine -----
// so that the dense-map kernel (ms_build_warp_maps / ms_build_maps) and the fused per-frame kernel (which keeps
// only the 1-D tables and never reads dense maps) execute the SAME fp32 operations and agree bit for bit.
struct WarpParams { float k[9]; float t[3]; float scale; };

__device__ __forceinline__ float2 warp_col_term(int proj, float u, const WarpParams &P)
{
    if (proj == MS_PROJ_PLANE) return make_float2(u / P.scale - P.t[0], 0.f);
    u /= P.scale;
    return make_float2(sinf(u), cosf(u));
}
__device__ __forceinline__ float2 warp_row_term(int proj, float v, const WarpParams &P)
{
    if (proj == MS_PROJ_PLANE) return make_float2(v / P.scale - P.t[1], 0.f);
    if (proj == MS_PROJ_CYLINDRICAL) return make_float2(v / P.scale, 0.f);
    v /= P.scale;
    return make_float2(sinf(v), cosf(v));
}
__device__ __forceinline__ void warp_combine(int proj, const float2 c, const float2 r, const WarpParams &P, float &ox, float &oy)
{
    float x_, y_, z_;
    if (proj == MS_PROJ_PLANE) { x_ = c.x; y_ = r.x; z_ = 1.f - P.t[2]; }
    else if (proj == MS_PROJ_CYLINDRICAL) { x_ = c.x; y_ = r.x; z_ = c.y; }
    else { x_ = r.x * c.x; y_ = -r.y; z_ = r.x * c.y; }          // sinv*sinu, -cosv, sinv*cosu
    ox = __builtin_fmaf(P.k[2], z_, __builtin_fmaf(P.k[1], y_, P.k[0] * x_));
    oy = __builtin_fmaf(P.k[5], z_, __builtin_fmaf(P.k[4], y_, P.k[3] * x_));
    const float oz = __builtin_fmaf(P.k[8], z_, __builtin_fmaf(P.k[7], y_, P.k[6] * x_));
    // Divide unconditionally and select (no trap on the GPU): straight-line code instead of a divergent block per pixel.  x/z and y/z
    // share the denominator: one refined reciprocal (v_rcp_f32 + ONE Newton step) and, per numerator, quotient + two residual corrections.
    // This is NOT claimed to be IEEE division for every operand (no div_scale / div_fixup, one refinement step): what the results contract
    // needs is that every kernel that produces map coordinates -- the dense-map kernels (ms_build_maps, ms_build_warp_maps) and the
    // per-frame kernels -- goes through this one function, so that they agree bit for bit by construction; against the oracle's glibc
    // maps the tolerance is 1e-3 px.
    const bool ok = proj == MS_PROJ_PLANE || oz > 0;
    const float r0 = __builtin_amdgcn_rcpf(oz);
    const float e0 = __builtin_fmaf(-oz, r0, 1.f);
    const float rz = __builtin_fmaf(e0, r0, r0);
    float qx = ox * rz, qy = oy * rz;
    qx = __builtin_fmaf(__builtin_fmaf(-oz, qx, ox), rz, qx);
    qy = __builtin_fmaf(__builtin_fmaf(-oz, qy, oy), rz, qy);
    qx = __builtin_fmaf(__builtin_fmaf(-oz, qx, ox), rz, qx);
    qy = __builtin_fmaf(__builtin_fmaf(-oz, qy, oy), rz, qy);
    ox = ok ? qx : -1.f;
    oy = ok ? qy : -1.f;
}

// ---- IEEE fp32 division with the reciprocal shared between numerators -----------------------------------
// normalizeUsingWeightKernel32F divides the three colour channels of a pixel by the same (w + 1e-5).  DivBy refines the
// hardware reciprocal with ONE Newton step, forms the quotient and applies two residual corrections; there is no
// div_scale / div_fixup (|a| <= 32768 and 1e-5 <= d < 64: no extreme exponents).  That this equals the compiler's correctly
// rounded a / d bit for bit is NOT argued from the instruction sequence: it is checked -- ms_selftest_divide compares both
// over all 65536 int16 numerators for every denominator handed to it, and ms_init_blender runs that check over the distinct
// denominators of the context's own tables (MS_CHECK_DIVIDE=1, on by default in the test suite).
struct DivBy {
    float d, r;
    __device__ __forceinline__ explicit DivBy(float den) : d(den)
    {
        const float r0 = __builtin_amdgcn_rcpf(den);
        const float e0 = __builtin_fmaf(-den, r0, 1.f);
        r = __builtin_fmaf(e0, r0, r0);
    }
    __device__ __forceinline__ float operator()(float a) const
    {
        const float q0 = a * r;
        const float e1 = __builtin_fmaf(-d, q0, a);
        const float q1 = __builtin_fmaf(e1, r, q0);
        const float e2 = __builtin_fmaf(-d, q1, a);
        return __builtin_fmaf(e2, r, q1);
    }
};

template <typename T>
__device__ __forceinline__ T *row_ptr(void *base, size_t step, int y) { return (T *)((char *)base + (size_t)y * step); }
template <typename T>
__device__ __forceinline__ const T *row_ptr(const void *base, size_t step, int y) { return (const T *)((const char *)base + (size_t)y * step); }

}  // namespace ms
