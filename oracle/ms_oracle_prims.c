/*
 * ms_oracle_prims.c -- CPU ORACLE (test infrastructure only; see ms_oracle.h header).
 * Per-kernel restatement of the reference's CUDA image ops on the stitching hot path.
 * Paths: OCV = /root/reference/sources/modules, APP = /root/reference/360_stitcher.
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include "ms_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

#define ROWP(T, base, step, y) ((T *)((char *)(base) + (size_t)(y) * (step)))
#define CROWP(T, base, step, y) ((const T *)((const char *)(base) + (size_t)(y) * (step)))

/* saturate_cast<uchar>(float) = cvt.rni.sat.u8.f32  OCV/core/include/opencv2/core/cuda/saturate_cast.hpp:96-101 */
static inline uint8_t sat_u8f(float v)
{
    if (!(v == v)) return 0;
    float r = rintf(v); /* default rounding mode: nearest-even */
    if (r <= 0.f) return 0;
    if (r >= 255.f) return 255;
    return (uint8_t)(int)r;
}
/* saturate_cast<short>(float) = cvt.rni.sat.s16.f32  saturate_cast.hpp:221-226 */
static inline int16_t sat_s16f(float v)
{
    if (!(v == v)) return 0;
    float r = rintf(v);
    if (r <= -32768.f) return (int16_t)-32768;
    if (r >= 32767.f) return (int16_t)32767;
    return (int16_t)(int)r;
}
/* saturate_cast<short>(int) = cvt.sat.s16.s32  saturate_cast.hpp:209-214 */
static inline int16_t sat_s16i(int v)
{
    return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}
/* static_cast<short>(float) as written in multiband_blend.cu:46-49 (`dst += static_cast<short>(src * w)`) and :96-98 (`src / (w + 1e-5f)`).
 * CONVENTION ASSUMED: float -> int32 with round-toward-zero (cvt.rzi.s32.f32), then the low 16 bits -- what `(short)(int)v` does and what nvcc
 * emits when it converts through int.  nvcc may instead emit the direct, SATURATING cvt.rzi.s16.f32; the two agree exactly while
 * -32768 <= trunc(v) <= 32767 and differ only outside.  On this path v = L * w with |L| <= 255 * 2 and 0 <= w <= 1, or v = a / (w + 1e-5) with a
 * an int16 sum of such terms and w + 1e-5 >= the weights that produced a: both stay far inside the int16 range on every configuration the tests
 * run.  That is not proved in general, so it is COUNTED: every call outside the range increments orc_trunc_s16_range_violations(), and
 * tests/conftest.py fails the session if the counter is not 0 at the end. */
static long long g_trunc_s16_violations = 0;
long long orc_trunc_s16_range_violations(void) { return __atomic_load_n(&g_trunc_s16_violations, __ATOMIC_RELAXED); }
void orc_trunc_s16_range_reset(void) { __atomic_store_n(&g_trunc_s16_violations, 0, __ATOMIC_RELAXED); }
static inline int16_t trunc_s16f(float v)
{
    if (!(v > -32769.f && v < 32768.f)) __atomic_fetch_add(&g_trunc_s16_violations, 1, __ATOMIC_RELAXED);      /* (NaN counts too) */
    if (!(v == v)) return 0;
    if (v > 2147483520.f) return (int16_t)0x7fffffff;
    if (v <= -2147483648.f) return (int16_t)0x80000000;
    return (int16_t)(int32_t)v;
}
/* __float2int_rd with CUDA NaN->0 and saturation */
static inline int f2i_rd(float v)
{
    if (!(v == v)) return 0;
    float f = floorf(v);
    if (f > 2147483520.f) return 0x7fffffff;
    if (f <= -2147483648.f) return (int)0x80000000;
    return (int)f;
}
static inline int f2i_rz(float v)
{
    if (!(v == v)) return 0;
    if (v > 2147483520.f) return 0x7fffffff;
    if (v <= -2147483648.f) return (int)0x80000000;
    return (int)v;
}

/* ---------------------------------------------------------------------------------------------
 * K1: remap<LinearFilter<BorderReader<PtrStep<uchar3>, BrdConstant<float3>>>>
 *   kernel  OCV/cudawarping/src/cuda/remap.cu:56-68
 *   filter  OCV/core/include/opencv2/core/cuda/filters.hpp:79-117 (tap order, weights)
 *   border  OCV/core/include/opencv2/core/cuda/border_interpolate.hpp:698-717 (OOB tap -> 0)
 */
static inline void linear_tap_cn(const uint8_t *src, size_t sstep, int srows, int scols, int cn,
                                 float y, float x, float *out /* cn */)
{
    const int x1 = f2i_rd(x), y1 = f2i_rd(y);
    /* x2 = x1 + 1 in int arithmetic (wraps like the device code) */
    const int x2 = (int)((unsigned)x1 + 1u), y2 = (int)((unsigned)y1 + 1u);
    const float w11 = ((float)x2 - x) * ((float)y2 - y);
    const float w12 = (x - (float)x1) * ((float)y2 - y);
    const float w21 = ((float)x2 - x) * (y - (float)y1);
    const float w22 = (x - (float)x1) * (y - (float)y1);
    const int xs[4] = {x1, x2, x1, x2};
    const int ys[4] = {y1, y1, y2, y2};
    const float ws[4] = {w11, w12, w21, w22};
    for (int c = 0; c < cn; ++c) out[c] = 0.f;
    for (int t = 0; t < 4; ++t) {
        const int xx = xs[t], yy = ys[t];
        const int inb = (xx >= 0 && xx < scols && yy >= 0 && yy < srows);
        const uint8_t *p = inb ? CROWP(uint8_t, src, sstep, yy) + (size_t)xx * cn : NULL;
        for (int c = 0; c < cn; ++c) {
            const float s = inb ? (float)p[c] : 0.f;
            out[c] = fmaf(s, ws[t], out[c]); /* out = out + src_reg * w   (nvcc -fmad) */
        }
    }
}

static void remap_linear_8u(const uint8_t *src, size_t sstep, int srows, int scols, int cn,
                            const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                            uint8_t *dst, size_t dstep, int drows, int dcols)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        const float *mx = CROWP(float, mapx, mxstep, y);
        const float *my = CROWP(float, mapy, mystep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < dcols; ++x) {
            float acc[4];
            linear_tap_cn(src, sstep, srows, scols, cn, my[x], mx[x], acc);
            for (int c = 0; c < cn; ++c) d[(size_t)x * cn + c] = sat_u8f(acc[c]);
        }
    }
}

void orc_remap_linear_8uc3(const uint8_t *src, size_t sstep, int srows, int scols,
                           const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                           uint8_t *dst, size_t dstep, int drows, int dcols)
{
    remap_linear_8u(src, sstep, srows, scols, 3, mapx, mxstep, mapy, mystep, dst, dstep, drows, dcols);
}

void orc_remap_linear_8uc1(const uint8_t *src, size_t sstep, int srows, int scols,
                           const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                           uint8_t *dst, size_t dstep, int drows, int dcols)
{
    remap_linear_8u(src, sstep, srows, scols, 1, mapx, mxstep, mapy, mystep, dst, dstep, drows, dcols);
}

/* a19 -- cv::remap, the CPU flavour (what MeshWarper::createMesh and the config-1 CPU pipeline run): INTER_LINEAR on CV_32FC1 map pairs,
 * BORDER_CONSTANT(0), 8-bit sources.
 *   RemapInvoker (imgproc/src/imgwarp.cpp:1203-1270): sx = cvRound(x * 32) (nearest-even; INT_MIN when out of range / NaN, as cvtps2dq),
 *     XY = saturate_cast<short>(sx >> 5), table index (sy & 31) * 32 + (sx & 31)
 *   initInterTab2D (imgwarp.cpp:211-284): BilinearTab_i = saturate_cast<short>(w * 32768), then the sum fix-up -- restated literally,
 *     including its reads past the 2x2 entry (k1, k2 run over ksize/2 .. ksize/2 + 1), which land in the NEXT, still zero, entry
 *   remapBilinear<FixedPtCast<int, uchar, 15>> (imgwarp.cpp:643-850): sum of tap * weight over the taps inside the image, (v + 16384) >> 15 */
static short cv_bilinear_tab[32 * 32 * 4 + 8];
static int cv_bilinear_tab_ready;
static void cv_init_bilinear_tab(void)
{
    if (cv_bilinear_tab_ready) return;
    memset(cv_bilinear_tab, 0, sizeof cv_bilinear_tab);
    float tab1[32][2];
    const float scale = 1.f / 32;
    for (int i = 0; i < 32; ++i) { tab1[i][0] = 1.f - i * scale; tab1[i][1] = i * scale; }      /* interpolateLinear */
    short *itab = cv_bilinear_tab;
    const int ksize = 2;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j, itab += ksize * ksize) {
            int isum = 0;
            for (int k1 = 0; k1 < ksize; ++k1) {
                const float vy = tab1[i][k1];
                for (int k2 = 0; k2 < ksize; ++k2) {
                    const float v = vy * tab1[j][k2];
                    long r = lrintf(v * 32768);                                               /* saturate_cast<short>(float) */
                    if (r > 32767) r = 32767;
                    if (r < -32768) r = -32768;
                    isum += itab[k1 * ksize + k2] = (short)r;
                }
            }
            if (isum != 32768) {
                const int diff = isum - 32768;
                const int ksize2 = ksize / 2;
                int Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2 + 2; ++k1)
                    for (int k2 = ksize2; k2 < ksize2 + 2; ++k2) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    cv_bilinear_tab_ready = 1;
}

static inline int cv_round_f(float v)
{
    const float r = rintf(v);
    return (r >= -2147483648.f && r < 2147483648.f) ? (int)r : INT32_MIN;
}

/* the rounding / saturation helpers above, exported so that tests/test_ref_pin.py can compare them with the reference's own
 * header-only cv::saturate_cast / cvRound (oracle/_ref/libref_pin.so) over every float bit pattern */
int orc_helper_sat_u8f(float v) { return (int)sat_u8f(v); }
int orc_helper_sat_s16f(float v) { return (int)sat_s16f(v); }
int orc_helper_sat_s16i(int v) { return (int)sat_s16i(v); }
int orc_helper_cv_round_f(float v) { return cv_round_f(v); }
int orc_helper_f2i_rd(float v) { return f2i_rd(v); }

void orc_cv_remap_linear_8u(const uint8_t *src, size_t sstep, int srows, int scols, int cn,
                            const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                            uint8_t *dst, size_t dstep, int drows, int dcols)
{
    cv_init_bilinear_tab();
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        const float *mx = CROWP(float, mapx, mxstep, y);
        const float *my = CROWP(float, mapy, mystep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < dcols; ++x) {
            const int qx = cv_round_f(mx[x] * 32), qy = cv_round_f(my[x] * 32);
            const short *w = cv_bilinear_tab + ((qy & 31) * 32 + (qx & 31)) * 4;
            int sx = qx >> 5, sy = qy >> 5;
            sx = sx > 32767 ? 32767 : sx < -32768 ? -32768 : sx;
            sy = sy > 32767 ? 32767 : sy < -32768 ? -32768 : sy;
            for (int c = 0; c < cn; ++c) {
                int v = 0;
                for (int t = 0; t < 4; ++t) {
                    const int xx = sx + (t & 1), yy = sy + (t >> 1);
                    if (xx >= 0 && xx < scols && yy >= 0 && yy < srows) v += CROWP(uint8_t, src, sstep, yy)[xx * cn + c] * w[t];
                }
                v = (v + (1 << 14)) >> 15;
                d[x * cn + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
    }
}

/* K1': PointFilter (filters.hpp:58-77: src(__float2int_rz(y), __float2int_rz(x))) + BrdConstant(0) */
void orc_remap_nearest_8uc1(const uint8_t *src, size_t sstep, int srows, int scols,
                            const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                            uint8_t *dst, size_t dstep, int drows, int dcols)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        const float *mx = CROWP(float, mapx, mxstep, y);
        const float *my = CROWP(float, mapy, mystep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < dcols; ++x) {
            const int xx = f2i_rz(mx[x]), yy = f2i_rz(my[x]);
            d[x] = (xx >= 0 && xx < scols && yy >= 0 && yy < srows) ? CROWP(uint8_t, src, sstep, yy)[xx] : 0;
        }
    }
}

/* K2: resize_linear<T>  OCV/cudawarping/src/cuda/resize.cu:71-106; host resize.cpp:57-106 */
void orc_resize_linear_8u(const uint8_t *src, size_t sstep, int srows, int scols, int cn,
                          uint8_t *dst, size_t dstep, int drows, int dcols, float ifx, float ify)
{
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < drows; ++dy) {
        uint8_t *d = ROWP(uint8_t, dst, dstep, dy);
        const float sy = (float)dy * ify;
        const int y1 = f2i_rd(sy), y2 = y1 + 1;
        const int y2r = y2 < srows - 1 ? y2 : srows - 1;
        const uint8_t *r1 = CROWP(uint8_t, src, sstep, y1);
        const uint8_t *r2 = CROWP(uint8_t, src, sstep, y2r);
        for (int dx = 0; dx < dcols; ++dx) {
            const float sx = (float)dx * ifx;
            const int x1 = f2i_rd(sx), x2 = x1 + 1;
            const int x2r = x2 < scols - 1 ? x2 : scols - 1;
            const float w11 = ((float)x2 - sx) * ((float)y2 - sy);
            const float w12 = (sx - (float)x1) * ((float)y2 - sy);
            const float w21 = ((float)x2 - sx) * (sy - (float)y1);
            const float w22 = (sx - (float)x1) * (sy - (float)y1);
            for (int c = 0; c < cn; ++c) {
                float out = 0.f;
                out = fmaf((float)r1[(size_t)x1 * cn + c], w11, out);
                out = fmaf((float)r1[(size_t)x2r * cn + c], w12, out);
                out = fmaf((float)r2[(size_t)x1 * cn + c], w21, out);
                out = fmaf((float)r2[(size_t)x2r * cn + c], w22, out);
                d[(size_t)dx * cn + c] = sat_u8f(out);
            }
        }
    }
}

/* K3: Convertor<uchar,uchar,float>  OCV/core/src/cuda/gpu_mat.cu:488-512
 *     op.alpha = saturate_cast<float>(alpha); D = saturate_cast<uchar>(alpha * src + beta) */
void orc_convert_scale_8u(const uint8_t *src, size_t sstep, uint8_t *dst, size_t dstep,
                          int rows, int width_bytes, double alpha)
{
    const float a = (float)alpha;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint8_t *s = CROWP(uint8_t, src, sstep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < width_bytes; ++x) d[x] = sat_u8f(fmaf(a, (float)s[x], 0.f));
    }
}

/* K4: copyMakeBorder BORDER_REFLECT  OCV/cudaarithm/src/cuda/copy_make_border.cu:61-73,113-115
 *     index math  OCV/cudev/include/opencv2/cudev/ptr2d/extrapolation.hpp:100-120,171-183 */
static inline int reflect_idx(int i, int len)
{
    const int last = len - 1;
    int hi = last - abs(last - i) + (i > last);   /* BrdReflect::idx_high */
    return (abs(hi) - (hi < 0)) % len;            /* BrdReflect::idx_low  */
}

void orc_copy_make_border_reflect(const uint8_t *src, size_t sstep, int srows, int scols,
                                  int elem_size, uint8_t *dst, size_t dstep,
                                  int top, int bottom, int left, int right)
{
    const int drows = srows + top + bottom, dcols = scols + left + right;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        const uint8_t *s = CROWP(uint8_t, src, sstep, reflect_idx(y - top, srows));
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < dcols; ++x)
            memcpy(d + (size_t)x * elem_size, s + (size_t)reflect_idx(x - left, scols) * elem_size, elem_size);
    }
}

/* K4': copyMakeBorder BORDER_CONSTANT  extrapolation.hpp:60-74 */
void orc_copy_make_border_const_32f(const float *src, size_t sstep, int srows, int scols,
                                    float *dst, size_t dstep, int top, int bottom, int left, int right)
{
    const int drows = srows + top + bottom, dcols = scols + left + right;
    for (int y = 0; y < drows; ++y) {
        float *d = ROWP(float, dst, dstep, y);
        const int sy = y - top;
        for (int x = 0; x < dcols; ++x) {
            const int sx = x - left;
            d[x] = (sx >= 0 && sx < scols && sy >= 0 && sy < srows) ? CROWP(float, src, sstep, sy)[sx] : 0.f;
        }
    }
}

/* K5: convertToNoScale<uchar,short>  gpu_mat.cu:477-486 */
void orc_convert_8u_16s(const uint8_t *src, size_t sstep, int16_t *dst, size_t dstep, int rows, int width)
{
    for (int y = 0; y < rows; ++y) {
        const uint8_t *s = CROWP(uint8_t, src, sstep, y);
        int16_t *d = ROWP(int16_t, dst, dstep, y);
        for (int x = 0; x < width; ++x) d[x] = (int16_t)s[x];
    }
}

/* K17: convertToNoScale<short,uchar>: saturate_cast<uchar>(short)  (timed.cpp:251) */
void orc_convert_16s_8u(const int16_t *src, size_t sstep, uint8_t *dst, size_t dstep, int rows, int width)
{
    for (int y = 0; y < rows; ++y) {
        const int16_t *s = CROWP(int16_t, src, sstep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < width; ++x) d[x] = (uint8_t)(s[x] < 0 ? 0 : (s[x] > 255 ? 255 : s[x]));
    }
}

/* mask.convertTo(weight_map, CV_32F, 1./255.)  blenders.cpp:412 -> Convertor<uchar,float,float> */
void orc_convert_8u_32f_scale(const uint8_t *src, size_t sstep, float *dst, size_t dstep,
                              int rows, int cols, double alpha)
{
    const float a = (float)alpha;
    for (int y = 0; y < rows; ++y) {
        const uint8_t *s = CROWP(uint8_t, src, sstep, y);
        float *d = ROWP(float, dst, dstep, y);
        for (int x = 0; x < cols; ++x) d[x] = fmaf(a, (float)s[x], 0.f);
    }
}

/* BrdReflect101  OCV/core/include/opencv2/core/cuda/border_interpolate.hpp:351-381 */
static inline int r101_low(int i, int len) { return abs(i) % len; }
static inline int r101_high(int i, int len) { const int last = len - 1; return abs(last - abs(last - i)) % len; }
static inline int r101(int i, int len) { return r101_low(r101_high(i, len), len); }

/* K6: pyrDown<T, BrdReflect101>  OCV/cudawarping/src/cuda/pyr_down.cu:55-174
 *     vertical 5-tap from global (rows: idx_row_low for y-2,y-1; idx_row_high for y+1,y+2),
 *     then horizontal 5-tap on the fp32 row (cols reflected by idx_col / idx_col_high),
 *     decimation by 2, saturate_cast<T>.  Host: pyramids.cpp:66-92 (dst = (rows+1)/2 x (cols+1)/2). */
void orc_pyr_down_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn,
                      int16_t *dst, size_t dstep)
{
    const int drows = (srows + 1) / 2, dcols = (scols + 1) / 2;
    const int W = scols * cn;
#pragma omp parallel
    {
        float *row = (float *)malloc(sizeof(float) * (size_t)(scols + 4) * cn);
#pragma omp for schedule(static)
        for (int y = 0; y < drows; ++y) {
            const int sy = 2 * y;
            const int16_t *rm2 = CROWP(int16_t, src, sstep, r101_low(sy - 2, srows));
            const int16_t *rm1 = CROWP(int16_t, src, sstep, r101_low(sy - 1, srows));
            const int16_t *r0 = CROWP(int16_t, src, sstep, sy);
            const int16_t *rp1 = CROWP(int16_t, src, sstep, r101_high(sy + 1, srows));
            const int16_t *rp2 = CROWP(int16_t, src, sstep, r101_high(sy + 2, srows));
            float *v = row + 2 * cn; /* v[x*cn+c] for x in [-2, scols+2) */
            for (int i = 0; i < W; ++i) {
                float sum = 0.0625f * (float)rm2[i];
                sum = fmaf(0.25f, (float)rm1[i], sum);
                sum = fmaf(0.375f, (float)r0[i], sum);
                sum = fmaf(0.25f, (float)rp1[i], sum);
                sum = fmaf(0.0625f, (float)rp2[i], sum);
                v[i] = sum;
            }
            for (int e = 1; e <= 2; ++e)
                for (int c = 0; c < cn; ++c) {
                    v[-e * cn + c] = v[r101(-e, scols) * cn + c];
                    v[(scols - 1 + e) * cn + c] = v[r101(scols - 1 + e, scols) * cn + c];
                }
            int16_t *d = ROWP(int16_t, dst, dstep, y);
            for (int dx = 0; dx < dcols; ++dx)
                for (int c = 0; c < cn; ++c) {
                    const float *p = v + (2 * dx) * cn + c;
                    float sum = 0.0625f * p[-2 * cn];
                    sum = fmaf(0.25f, p[-cn], sum);
                    sum = fmaf(0.375f, p[0], sum);
                    sum = fmaf(0.25f, p[cn], sum);
                    sum = fmaf(0.0625f, p[2 * cn], sum);
                    d[dx * cn + c] = sat_s16f(sum);
                }
        }
        free(row);
    }
}

/* K22: the same template instantiated for float (weight pyramids; blenders.cpp:420-423) */
void orc_pyr_down_32f(const float *src, size_t sstep, int srows, int scols, float *dst, size_t dstep)
{
    const int drows = (srows + 1) / 2, dcols = (scols + 1) / 2;
    float *row = (float *)malloc(sizeof(float) * (size_t)(scols + 4));
    for (int y = 0; y < drows; ++y) {
        const int sy = 2 * y;
        const float *rm2 = CROWP(float, src, sstep, r101_low(sy - 2, srows));
        const float *rm1 = CROWP(float, src, sstep, r101_low(sy - 1, srows));
        const float *r0 = CROWP(float, src, sstep, sy);
        const float *rp1 = CROWP(float, src, sstep, r101_high(sy + 1, srows));
        const float *rp2 = CROWP(float, src, sstep, r101_high(sy + 2, srows));
        float *v = row + 2;
        for (int i = 0; i < scols; ++i) {
            float sum = 0.0625f * rm2[i];
            sum = fmaf(0.25f, rm1[i], sum);
            sum = fmaf(0.375f, r0[i], sum);
            sum = fmaf(0.25f, rp1[i], sum);
            sum = fmaf(0.0625f, rp2[i], sum);
            v[i] = sum;
        }
        for (int e = 1; e <= 2; ++e) {
            v[-e] = v[r101(-e, scols)];
            v[scols - 1 + e] = v[r101(scols - 1 + e, scols)];
        }
        float *d = ROWP(float, dst, dstep, y);
        for (int dx = 0; dx < dcols; ++dx) {
            const float *p = v + 2 * dx;
            float sum = 0.0625f * p[-2];
            sum = fmaf(0.25f, p[-1], sum);
            sum = fmaf(0.375f, p[0], sum);
            sum = fmaf(0.25f, p[1], sum);
            sum = fmaf(0.0625f, p[2], sum);
            d[dx] = sum;
        }
    }
    free(row);
}

/* K7: pyrUp<T>  OCV/cudawarping/src/cuda/pyr_up.cu:55-145; host pyramids.cpp:106-132 (dst = 2x).
 *     source index: srcx = min(cols-1, |x|)  (pyr_up.cu:70-74);
 *     horizontal pass on even dst rows only (odd rows are zero), coefficients selected by the
 *     dst column parity (pyr_up.cu:86-96); vertical 5-tap (pyr_up.cu:133-139); x4; saturate. */
static inline int pu_idx(int i, int n) { i = abs(i); return i < n - 1 ? i : n - 1; }

void orc_pyr_up_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn,
                    int16_t *dst, size_t dstep)
{
    const int drows = 2 * srows, dcols = 2 * scols;
    const int W = dcols * cn;
    /* H[r] for source rows r = -1 .. srows (clamped), horizontal pass result */
    float *H = (float *)malloc(sizeof(float) * (size_t)W * (size_t)(srows));
#pragma omp parallel for schedule(static)
    for (int r = 0; r < srows; ++r) {
        const int16_t *s = CROWP(int16_t, src, sstep, r);
        float *h = H + (size_t)r * W;
        for (int x = 0; x < dcols; ++x) {
            const int even = (x & 1) == 0;
            const float ce0 = (float)even * 0.0625f, co = (float)(!even) * 0.25f, ce1 = (float)even * 0.375f;
            /* floor((x+d)/2) via arithmetic shift, as the device code's (tidx+d)>>1 on an even base */
            const int i0 = pu_idx((x - 2) >> 1, scols), i1 = pu_idx((x - 1) >> 1, scols),
                      i2 = pu_idx(x >> 1, scols), i3 = pu_idx((x + 1) >> 1, scols),
                      i4 = pu_idx((x + 2) >> 1, scols);
            for (int c = 0; c < cn; ++c) {
                float sum = 0.f;
                sum = fmaf(ce0, (float)s[i0 * cn + c], sum);
                sum = fmaf(co, (float)s[i1 * cn + c], sum);
                sum = fmaf(ce1, (float)s[i2 * cn + c], sum);
                sum = fmaf(co, (float)s[i3 * cn + c], sum);
                sum = fmaf(ce0, (float)s[i4 * cn + c], sum);
                h[x * cn + c] = sum;
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        int16_t *d = ROWP(int16_t, dst, dstep, y);
        const float *rows5[5];
        for (int k = 0; k < 5; ++k) {
            const int r = y - 2 + k;
            rows5[k] = (r & 1) ? NULL : H + (size_t)pu_idx(r >> 1, srows) * W; /* odd dst rows hold zeros */
        }
        static const float cw[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
        for (int i = 0; i < W; ++i) {
            float sum = 0.f;
            for (int k = 0; k < 5; ++k) sum = fmaf(cw[k], rows5[k] ? rows5[k][i] : 0.f, sum);
            d[i] = sat_s16f(4.0f * sum);
        }
    }
    free(H);
}

/* K8: SubOp1<short,short>  OCV/cudaarithm/src/cuda/sub_mat.cu:59-65 (saturate_cast<short>(a - b)) */
void orc_sub_16s(const int16_t *a, size_t astep, const int16_t *b, size_t bstep,
                 int16_t *dst, size_t dstep, int rows, int width)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const int16_t *pa = CROWP(int16_t, a, astep, y), *pb = CROWP(int16_t, b, bstep, y);
        int16_t *d = ROWP(int16_t, dst, dstep, y);
        for (int x = 0; x < width; ++x) d[x] = sat_s16i((int)pa[x] - (int)pb[x]);
    }
}

/* K11: AddOp1<short,short>  OCV/cudaarithm/src/cuda/add_mat.cu:59-65 */
void orc_add_16s(const int16_t *a, size_t astep, const int16_t *b, size_t bstep,
                 int16_t *dst, size_t dstep, int rows, int width)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const int16_t *pa = CROWP(int16_t, a, astep, y), *pb = CROWP(int16_t, b, bstep, y);
        int16_t *d = ROWP(int16_t, dst, dstep, y);
        for (int x = 0; x < width; ++x) d[x] = sat_s16i((int)pa[x] + (int)pb[x]);
    }
}

/* K9: addSrcWeightKernel32F  OCV/stitching/src/cuda/multiband_blend.cu:36-51 */
void orc_add_src_weight_32f(const int16_t *src, size_t sstep, const float *w, size_t wstep,
                            int16_t *dst, size_t dstep, float *dst_w, size_t dwstep,
                            int rows, int cols)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const int16_t *s = CROWP(int16_t, src, sstep, y);
        const float *pw = CROWP(float, w, wstep, y);
        int16_t *d = ROWP(int16_t, dst, dstep, y);
        float *dw = ROWP(float, dst_w, dwstep, y);
        for (int x = 0; x < cols; ++x) {
            const float ww = pw[x];
            for (int c = 0; c < 3; ++c)
                d[3 * x + c] = (int16_t)(d[3 * x + c] + trunc_s16f((float)s[3 * x + c] * ww));
            dw[x] = dw[x] + ww;
        }
    }
}

/* K9': addSrcWeightKernel16S  OCV/stitching/src/cuda/multiband_blend.cu:10-24 (weight_type CV_16S: weights 0..256).  int promotion, arithmetic shift,
 * short(...) = low 16 bits, `+=` on short wraps. */
void orc_add_src_weight_16s(const int16_t *src, size_t sstep, const int16_t *w, size_t wstep,
                            int16_t *dst, size_t dstep, int16_t *dst_w, size_t dwstep, int rows, int cols)
{
    for (int y = 0; y < rows; ++y) {
        const int16_t *s = CROWP(int16_t, src, sstep, y), *pw = CROWP(int16_t, w, wstep, y);
        int16_t *d = ROWP(int16_t, dst, dstep, y), *dw = ROWP(int16_t, dst_w, dwstep, y);
        for (int x = 0; x < cols; ++x) {
            const int ww = pw[x];
            for (int c = 0; c < 3; ++c) {
                const int prod = (int)s[3 * x + c] * ww;
                const int sh = prod >= 0 ? prod >> 8 : -((-prod + 255) >> 8);          /* arithmetic shift = floor division by 256, spelled portably */
                d[3 * x + c] = (int16_t)(uint16_t)((unsigned)d[3 * x + c] + (unsigned)sh);
            }
            dw[x] = (int16_t)(uint16_t)((unsigned)dw[x] + (unsigned)ww);
        }
    }
}
/* K10': normalizeUsingWeightKernel16S  multiband_blend.cu:62-74: short((v << 8) / w), C division (toward zero).  w == 0 is a division by zero in the
 * reference (undefined on the device); convention here and in the product: 0. */
void orc_normalize_16s(const int16_t *w, size_t wstep, int16_t *src, size_t sstep, int rows, int cols)
{
    for (int y = 0; y < rows; ++y) {
        const int16_t *pw = CROWP(int16_t, w, wstep, y);
        int16_t *s = ROWP(int16_t, src, sstep, y);
        for (int x = 0; x < cols; ++x)
            for (int c = 0; c < 3; ++c)
                s[3 * x + c] = pw[x] ? (int16_t)(uint16_t)(unsigned)(((int)s[3 * x + c] * 256) / (int)pw[x]) : (int16_t)0;
    }
}

/* K10: normalizeUsingWeightKernel32F  multiband_blend.cu:85-100 */
void orc_normalize_32f(const float *w, size_t wstep, int16_t *src, size_t sstep, int rows, int cols)
{
    const float WEIGHT_EPS = 1e-5f;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const float *pw = CROWP(float, w, wstep, y);
        int16_t *s = ROWP(int16_t, src, sstep, y);
        for (int x = 0; x < cols; ++x) {
            const float den = pw[x] + WEIGHT_EPS;
            for (int c = 0; c < 3; ++c) s[3 * x + c] = trunc_s16f((float)s[3 * x + c] / den);
        }
    }
}

/* K12: CmpScalarOp<greater<float>>  OCV/cudaarithm/src/cuda/cmp_scalar.cu:59-82 */
void orc_compare_gt_32f(const float *src, size_t sstep, float thr, uint8_t *dst, size_t dstep, int rows, int cols)
{
    for (int y = 0; y < rows; ++y) {
        const float *s = CROWP(float, src, sstep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < cols; ++x) d[x] = (s[x] > thr) ? 255 : 0;
    }
}

void orc_compare_eq_8u(const uint8_t *src, size_t sstep, uint8_t val, uint8_t *dst, size_t dstep, int rows, int cols)
{
    for (int y = 0; y < rows; ++y) {
        const uint8_t *s = CROWP(uint8_t, src, sstep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < cols; ++x) d[x] = (s[x] == val) ? 255 : 0;
    }
}

/* K13: GpuMat::setTo(Scalar::all(0), mask)  gpu_mat.cu:369-374,431 */
void orc_set_zero_masked_16sc3(int16_t *img, size_t step, const uint8_t *mask, size_t mstep, int rows, int cols)
{
    for (int y = 0; y < rows; ++y) {
        int16_t *p = ROWP(int16_t, img, step, y);
        const uint8_t *m = CROWP(uint8_t, mask, mstep, y);
        for (int x = 0; x < cols; ++x)
            if (m[x]) p[3 * x] = p[3 * x + 1] = p[3 * x + 2] = 0;
    }
}

/* K21: cuda::bitwise_and  (APP/calibration.cpp:237) */
void orc_bitwise_and_8u(const uint8_t *a, size_t astep, const uint8_t *b, size_t bstep,
                        uint8_t *dst, size_t dstep, int rows, int cols)
{
    for (int y = 0; y < rows; ++y) {
        const uint8_t *pa = CROWP(uint8_t, a, astep, y), *pb = CROWP(uint8_t, b, bstep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < cols; ++x) d[x] = pa[x] & pb[x];
    }
}

/* K20: createMorphologyFilter(MORPH_DILATE, 3x3 default kernel) = nppiDilate_8u_C1R
 *      (OCV/cudafilters/src/filtering.cpp:519-560; APP/calibration.cpp:209,232): 3x3 max.
 *      NPP leaves a 1-px frame untouched in the OpenCV wrapper (it runs on the interior ROI
 *      of a replicated-border copy); we use replicate border, the cv::dilate default-equivalent
 *      for a max filter.  parity unpinned (closed-source NPP). */
void orc_dilate3x3_8u(const uint8_t *src, size_t sstep, uint8_t *dst, size_t dstep, int rows, int cols)
{
    for (int y = 0; y < rows; ++y) {
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < cols; ++x) {
            uint8_t m = 0;
            for (int dy = -1; dy <= 1; ++dy) {
                int yy = y + dy; yy = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
                const uint8_t *s = CROWP(uint8_t, src, sstep, yy);
                for (int dx = -1; dx <= 1; ++dx) {
                    int xx = x + dx; xx = xx < 0 ? 0 : (xx >= cols ? cols - 1 : xx);
                    if (s[xx] > m) m = s[xx];
                }
            }
            d[x] = m;
        }
    }
}

/* K18: buildWarpMapsKernel<Mapper>  OCV/stitching/src/cuda/build_warp_maps.cu:67-152
 *      dot products a*b + c*d + e*f evaluated as fmaf(e,f, fmaf(c,d, a*b)). */
void orc_build_warp_maps(int proj, int tl_u, int tl_v, int rows, int cols,
                         const float *k, const float *t, float scale,
                         float *mapx, size_t mxstep, float *mapy, size_t mystep)
{
#pragma omp parallel for schedule(static)
    for (int dv = 0; dv < rows; ++dv) {
        float *mx = ROWP(float, mapx, mxstep, dv);
        float *my = ROWP(float, mapy, mystep, dv);
        for (int du = 0; du < cols; ++du) {
            float u = (float)(tl_u + du);
            float v = (float)(tl_v + dv);
            float x_, y_, z_, x, y, z;
            if (proj == ORC_PROJ_PLANE) {
                x_ = u / scale - t[0];
                y_ = v / scale - t[1];
                z_ = 1.f - t[2];
                x = fmaf(k[2], z_, fmaf(k[1], y_, k[0] * x_));
                y = fmaf(k[5], z_, fmaf(k[4], y_, k[3] * x_));
                z = fmaf(k[8], z_, fmaf(k[7], y_, k[6] * x_));
                x /= z; y /= z;
            } else {
                if (proj == ORC_PROJ_CYLINDRICAL) {
                    u /= scale;
                    x_ = sinf(u);
                    y_ = v / scale;
                    z_ = cosf(u);
                } else {
                    v /= scale;
                    u /= scale;
                    const float sinv = sinf(v);
                    x_ = sinv * sinf(u);
                    y_ = -cosf(v);
                    z_ = sinv * cosf(u);
                }
                x = fmaf(k[2], z_, fmaf(k[1], y_, k[0] * x_));
                y = fmaf(k[5], z_, fmaf(k[4], y_, k[3] * x_));
                z = fmaf(k[8], z_, fmaf(k[7], y_, k[6] * x_));
                if (z > 0) { x /= z; y /= z; }
                else x = y = -1.f;
            }
            mx[du] = x;
            my[du] = y;
        }
    }
}

/* K19: APP/resize.cu:9-27 (custom_resize), int division for the cell index, fp32 fraction */
void orc_custom_resize_32f(const float *in, size_t istep, int rows, int cols,
                           float *out, size_t ostep, int ty, int tx)
{
#pragma omp parallel for schedule(static)
    for (int v = 0; v < ty; ++v) {
        const int top = v * (rows - 1) / ty;
        const float vv = ((float)v) * (float)(rows - 1) / (float)ty - (float)top;
        const float *r0 = CROWP(float, in, istep, top);
        const float *r1 = CROWP(float, in, istep, top + 1);
        float *o = ROWP(float, out, ostep, v);
        for (int u = 0; u < tx; ++u) {
            const int left = u * (cols - 1) / tx;
            const float uu = ((float)u) * (float)(cols - 1) / (float)tx - (float)left;
            const float a = (1.f - uu) * (1.f - vv);
            const float b = uu * (1.f - vv);
            const float c = (1.f - uu) * vv;
            const float d = uu * vv;
            float r = a * r0[left];
            r = fmaf(b, r0[left + 1], r);
            r = fmaf(c, r1[left], r);
            r = fmaf(d, r1[left + 1], r);
            o[u] = r;
        }
    }
}

/* cvtColor(COLOR_BGR2YUV_I420) = RGB888toYUV420pInvoker (bIdx 0, uIdx 1, planar)  OCV/imgproc/src/color.cpp:8745-8756,9082-9160
 * (the encoder input of consume(), APP/timed.cpp:308-316).  dst is the contiguous planar I420 image:
 * Y[h][w], then U[h/2][w/2], then V[h/2][w/2]; chroma is taken from the top-left pixel of each 2x2 block. */
void orc_bgr_to_i420(const uint8_t *src, size_t sstep, int w, int h, uint8_t *dst)
{
    const int SH = 20, half = 1 << (SH - 1);
    const int CRY = 269484, CGY = 528482, CBY = 102760, CRU = -155188, CGU = -305135, CBU = 460324, CGV = -385875, CBV = -74448;
    uint8_t *Y = dst, *U = dst + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
    for (int y = 0; y < h; ++y) {
        const uint8_t *p = CROWP(uint8_t, src, sstep, y);
        for (int x = 0; x < w; ++x) {
            const int b = p[3 * x], g = p[3 * x + 1], r = p[3 * x + 2];
            int yy = (CRY * r + CGY * g + CBY * b + half + (16 << SH)) >> SH;
            Y[(size_t)y * w + x] = (uint8_t)(yy < 0 ? 0 : (yy > 255 ? 255 : yy));
            if (((x | y) & 1) == 0) {
                int u = (CRU * r + CGU * g + CBU * b + half + (128 << SH)) >> SH;
                int v = (CBU * r + CGV * g + CBV * b + half + (128 << SH)) >> SH;
                U[(size_t)(y / 2) * (w / 2) + x / 2] = (uint8_t)(u < 0 ? 0 : (u > 255 ? 255 : u));
                V[(size_t)(y / 2) * (w / 2) + x / 2] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}

/* K1 with BORDER_REFLECT: remap<LinearFilter<BorderReader<PtrStep<uchar3>, BrdReflect<float3>>>> (the seam-scale image warp,
 * APP/calibration.cpp:118).  Tap indices go through BrdReflect::idx_row/idx_col (border_interpolate.hpp:485-525). */
void orc_remap_linear_reflect_8uc3(const uint8_t *src, size_t sstep, int srows, int scols,
                                   const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                                   uint8_t *dst, size_t dstep, int drows, int dcols)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < drows; ++y) {
        const float *mx = CROWP(float, mapx, mxstep, y);
        const float *my = CROWP(float, mapy, mystep, y);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < dcols; ++x) {
            const float xc = mx[x], yc = my[x];
            const int x1 = f2i_rd(xc), y1 = f2i_rd(yc);
            const int x2 = (int)((unsigned)x1 + 1u), y2 = (int)((unsigned)y1 + 1u);
            const float ws[4] = {((float)x2 - xc) * ((float)y2 - yc), (xc - (float)x1) * ((float)y2 - yc),
                                 ((float)x2 - xc) * (yc - (float)y1), (xc - (float)x1) * (yc - (float)y1)};
            const int xs[4] = {x1, x2, x1, x2}, ys[4] = {y1, y1, y2, y2};
            float acc[3] = {0.f, 0.f, 0.f};
            for (int t = 0; t < 4; ++t) {
                const uint8_t *p = CROWP(uint8_t, src, sstep, reflect_idx(ys[t], srows)) + (size_t)reflect_idx(xs[t], scols) * 3;
                for (int c = 0; c < 3; ++c) acc[c] = fmaf((float)p[c], ws[t], acc[c]);
            }
            for (int c = 0; c < 3; ++c) d[(size_t)x * 3 + c] = sat_u8f(acc[c]);
        }
    }
}

/* ingest: cvtColor(COLOR_YUV2BGR_NV12) = YUV420sp2RGB888Invoker<bIdx 0, uIdx 0>  OCV/imgproc/src/color.cpp:8738-8745 and the
 * invoker above them; called per received camera frame in APP/networking.cpp:45-47.  src: h rows of Y then h/2 rows of
 * interleaved UV (same stride); w, h even. */
void orc_nv12_to_bgr(const uint8_t *src, size_t sstep, int w, int h, uint8_t *dst, size_t dstep)
{
    const int SH = 20, CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;
    const uint8_t *uvp = src + (size_t)h * sstep;
    for (int y = 0; y < h; ++y) {
        const uint8_t *yr = CROWP(uint8_t, src, sstep, y), *uv = CROWP(uint8_t, uvp, sstep, y / 2);
        uint8_t *d = ROWP(uint8_t, dst, dstep, y);
        for (int x = 0; x < w; ++x) {
            const int u = (int)uv[(x & ~1)] - 128, v = (int)uv[(x & ~1) + 1] - 128;
            const int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
            int yy = (int)yr[x] - 16; if (yy < 0) yy = 0;
            yy *= CY;
            const int b = (yy + buv) >> SH, g = (yy + guv) >> SH, r = (yy + ruv) >> SH;
            d[3 * x] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
            d[3 * x + 1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
            d[3 * x + 2] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}
