/* ms_oracle_cpu.c -- TEST INFRASTRUCTURE.  The reference's CPU pipeline (the path BASELINE's CPU baseline names):
 *   cv::pyrDown / cv::pyrUp on 16S and 32F        OCV/imgproc/src/pyramids.cpp:851-1078 (FixPtCast<short,8> :52-57,1375; <short,6> :1483;
 *                                                 FltCast<float,8> :59-64; the SSE vertical pass of the float version :145-187)
 *   MultiBandBlender::feed (CPU branch)           OCV/stitching/src/blenders.cpp:585-696
 *   createLaplacePyr (16S branch), restoreImageFromLaplacePyr, normalizeUsingWeightMap      blenders.cpp:997-1008, 1040-1050, 880-941
 * restated in plain C.  Differences from the CUDA flavour of ms_oracle_prims.c: the integer pyramids round half UP ((x + 128) >> 8,
 * (x + 32) >> 6) instead of half to even, the float pyramid sums with integer weights and one final * (1/256) in the order of the
 * x86 SSE build, and the weight pyramid is rebuilt on every feed.  Parity unpinned by execution (the reference's imgproc does not
 * build here: precomp.hpp -> cmake-generated headers); its helpers (cvRound, saturate_cast) are pinned by oracle/_ref.
 * Used by tests and by bench.py's cpu_baseline only. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ms_oracle.h"

#define ROWP(T, base, step, y) ((T *)((char *)(base) + (size_t)(y) * (step)))
#define CROWP(T, base, step, y) ((const T *)((const char *)(base) + (size_t)(y) * (step)))

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101)  OCV/core/src/copy.cpp (the pyramids' border mode, BORDER_DEFAULT) */
static inline int bi101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* pyrDown_<FixPtCast<short,8>>  pyramids.cpp:851-964: horizontal 1 4 6 4 1 into int rows, vertical 1 4 6 4 1, (sum + 128) >> 8 */
void orc_cv_pyr_down_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn, int16_t *dst, size_t dstep)
{
    const int drows = (srows + 1) / 2, dcols = (scols + 1) / 2;
#pragma omp parallel
    {
        int *hrow = (int *)malloc(sizeof(int) * 5 * (size_t)dcols * cn);
#pragma omp for schedule(static)
        for (int y = 0; y < drows; ++y) {
            for (int k = 0; k < 5; ++k) {
                const int16_t *s = CROWP(int16_t, src, sstep, bi101(2 * y - 2 + k, srows));
                int *row = hrow + (size_t)k * dcols * cn;
                for (int x = 0; x < dcols; ++x) {
                    const int x0 = bi101(2 * x - 2, scols) * cn, x1 = bi101(2 * x - 1, scols) * cn, x2 = 2 * x * cn,
                              x3 = bi101(2 * x + 1, scols) * cn, x4 = bi101(2 * x + 2, scols) * cn;
                    for (int c = 0; c < cn; ++c)
                        row[x * cn + c] = s[x2 + c] * 6 + (s[x1 + c] + s[x3 + c]) * 4 + s[x0 + c] + s[x4 + c];
                }
            }
            int16_t *d = ROWP(int16_t, dst, dstep, y);
            const int *r0 = hrow, *r1 = r0 + (size_t)dcols * cn, *r2 = r1 + (size_t)dcols * cn, *r3 = r2 + (size_t)dcols * cn, *r4 = r3 + (size_t)dcols * cn;
            for (int i = 0; i < dcols * cn; ++i)
                d[i] = (int16_t)((r2[i] * 6 + (r1[i] + r3[i]) * 4 + r0[i] + r4[i] + 128) >> 8);
        }
        free(hrow);
    }
}

/* pyrDown_<FltCast<float,8>, PyrDownVec_32f>, one channel: scalar horizontal pass; vertical pass in the x86 SSE order for the
 * first (width / 8) * 8 columns ((r0 + r4) + (r2 + r2)) + ((r1 + r3) + r2) * 4, scalar tail ((r2 * 6 + (r1 + r3) * 4) + r0) + r4; * (1/256) */
void orc_cv_pyr_down_32f(const float *src, size_t sstep, int srows, int scols, float *dst, size_t dstep)
{
    const int drows = (srows + 1) / 2, dcols = (scols + 1) / 2;
    const float scale = (float)(1. / 256);
#pragma omp parallel
    {
    float *hrow = (float *)malloc(sizeof(float) * 5 * (size_t)dcols);
#pragma omp for schedule(static)
    for (int y = 0; y < drows; ++y) {
        for (int k = 0; k < 5; ++k) {
            const float *s = CROWP(float, src, sstep, bi101(2 * y - 2 + k, srows));
            float *row = hrow + (size_t)k * dcols;
            for (int x = 0; x < dcols; ++x) {
                const float a = s[bi101(2 * x - 2, scols)], b = s[bi101(2 * x - 1, scols)], c = s[2 * x], d = s[bi101(2 * x + 1, scols)],
                            e = s[bi101(2 * x + 2, scols)];
                float t = c * 6.f;
                t = t + (b + d) * 4.f;
                t = t + a;
                t = t + e;
                row[x] = t;
            }
        }
        float *d = ROWP(float, dst, dstep, y);
        const float *r0 = hrow, *r1 = r0 + dcols, *r2 = r1 + dcols, *r3 = r2 + dcols, *r4 = r3 + dcols;
        int x = 0;
        for (; x <= dcols - 8; x += 8)
            for (int i = x; i < x + 8; ++i) {
                float a = r0[i] + r4[i];
                float b = (r1[i] + r3[i]) + r2[i];
                a = a + (r2[i] + r2[i]);
                d[i] = (a + b * 4.f) * scale;
            }
        for (; x < dcols; ++x) {
            float t = r2[x] * 6.f;
            t = t + (r1[x] + r3[x]) * 4.f;
            t = t + r0[x];
            t = t + r4[x];
            d[x] = t * scale;
        }
    }
    free(hrow);
    }
}

/* pyrUp_<FixPtCast<short,6>>  pyramids.cpp:976-1078 to dst = (2 rows, 2 cols): even/odd horizontal taps (1 6 1 | 4 4), left edge mirrored
 * (6 a + 2 b), right edge replicated (b + 7 c | 8 c), rows by borderInterpolate(2 sy, 2 rows, REFLECT_101) / 2, (sum + 32) >> 6 */
void orc_cv_pyr_up_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn, int16_t *dst, size_t dstep)
{
    const int dcols = 2 * scols;
#pragma omp parallel
    {
        int *hrow = (int *)malloc(sizeof(int) * 3 * (size_t)dcols * cn);
#pragma omp for schedule(static)
        for (int y = 0; y < srows; ++y) {
            for (int k = 0; k < 3; ++k) {
                const int sy = y - 1 + k;
                const int16_t *s = CROWP(int16_t, src, sstep, bi101(2 * sy, 2 * srows) / 2);
                int *row = hrow + (size_t)k * dcols * cn;
                if (scols == 1) {
                    for (int c = 0; c < cn; ++c) row[c] = row[cn + c] = s[c] * 8;
                    continue;
                }
                for (int c = 0; c < cn; ++c) {
                    row[c] = s[c] * 6 + s[cn + c] * 2;
                    row[cn + c] = (s[c] + s[cn + c]) * 4;
                    const int sx = (scols - 1) * cn + c, dx = 2 * (scols - 1) * cn + c;
                    row[dx] = s[sx - cn] + s[sx] * 7;
                    row[dx + cn] = s[sx] * 8;
                }
                for (int x = 1; x < scols - 1; ++x)
                    for (int c = 0; c < cn; ++c) {
                        const int sx = x * cn + c, dx = 2 * x * cn + c;
                        row[dx] = s[sx - cn] + s[sx] * 6 + s[sx + cn];
                        row[dx + cn] = (s[sx] + s[sx + cn]) * 4;
                    }
            }
            int16_t *d0 = ROWP(int16_t, dst, dstep, 2 * y), *d1 = ROWP(int16_t, dst, dstep, 2 * y + 1);
            const int *r0 = hrow, *r1 = r0 + (size_t)dcols * cn, *r2 = r1 + (size_t)dcols * cn;
            for (int i = 0; i < dcols * cn; ++i) {
                d1[i] = (int16_t)(((r1[i] + r2[i]) * 4 + 32) >> 6);
                d0[i] = (int16_t)((r0[i] + r1[i] * 6 + r2[i] + 32) >> 6);
            }
        }
        free(hrow);
    }
}
