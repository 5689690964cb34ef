// features.hip -- the feature front-end of the recalibration path on the device (gfx950):
//   featurefinder::findFeatures   (360_stitcher/featurefinder.cpp:13-46)  cuda::ORB::create(2500, 1.2f, 8)->detectAndCompute
//       host logic   sources/modules/cudafeatures2d/src/orb.cpp:430-865 (pyramid, per-level budgets, culls, merge)
//       kernels      cudafeatures2d/src/cuda/fast.cu:224-344 (FAST 9-16 + score + non-max suppression),
//                    cuda/orb.cu:93-138 (Harris response), :160-211 (intensity-centroid angle), :222-367 (rBRIEF, WTA_K = 2)
//   featurefinder::matchFeatures' findHomography(src, dst, mask, RANSAC)  (featurefinder.cpp:68-90)
//       calib3d/src/fundam.cpp:46-260, 319-402; ptsetreg.cpp:53-290; levmarq.cpp:76-214
//
// What is on the device: every per-pixel and per-keypoint stage of ORB (image / mask pyramids, FAST scores, non-max suppression and
// ordered compaction, Harris responses, angles, descriptors) and RANSAC's hypothesis generation-and-scoring (one thread per 4-point
// model, all `maxIters` models at once).  What stays on the host, as in the reference: the keypoint COUNTS between stages (the reference
// reads them back too: fast.cu:338-341, :379-382), the two culls (tiny stable sorts; the reference sorts on the device with an unstable
// thrust sort, so its keypoint order is not defined -- ours is: raster order, stable culls), RANSAC's sequential subset draw and best-model
// scan, and the final N-point fit + Levenberg-Marquardt polish of ONE model.
// Conventions where CUDA leaves freedom (same as oracle/orb_oracle.py): float expressions without fma contraction; atan2f / sincosf as the
// correctly rounded float of the double function.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>
#include "common.hpp"
#include "launchers.hpp"

namespace ms {
namespace {

struct P2 { signed char x, y; };
__constant__ P2 c_pattern[512] = {
#include "orb_pattern.inc"
};
__constant__ int c_umax[32];

// FAST circle in the bit order of fast.cu's masks (bit k = C[k / 4] byte k % 4): (dy, dx)
__constant__ signed char c_circ[16][2] = {{3, 0}, {3, 1}, {2, 2}, {1, 3}, {0, 3}, {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}, {0, -3}, {1, -3}, {2, -2}, {3, -1}};

// cuda::threshold(m, m, 254, 0, THRESH_TOZERO) of the mask pyramid (orb.cpp:693)
__global__ void __launch_bounds__(256) k_tozero254(uint8_t *m, size_t step, int rows, int cols)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    uint8_t *p = m + (size_t)y * step + x;
    if (*p <= 254) *p = 0;
}

// calcKeypoints<true> (fast.cu:264-307): score map, 0 = no corner.  The mask of the level = user mask AND the inner rectangle left by edgeThreshold
// (orb.cpp:707-712), applied analytically.  A pixel is a corner when 9 contiguous circle pixels are all darker than v - th or all brighter than v + th
// (the c_table lookup of fast.cu:201-206 encodes exactly "the 16-bit mask holds 9 contiguous ones", checked against that table in tests);
// cornerScore's binary search (fast.cu:208-222) finds the largest threshold that still passes = (best arc's smallest difference) - 1.
__global__ void __launch_bounds__(256) k_fast_score(const uint8_t *__restrict__ img, size_t step, int rows, int cols, const uint8_t *__restrict__ mask, size_t mstep,
                                                    int edge, int th, uint8_t *__restrict__ score, size_t sstep)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= cols || i >= rows) return;
    int s = 0;
    const bool inner = i >= edge && j >= edge && i < rows - edge && j < cols - edge && cols > 2 * edge && rows > 2 * edge;
    if (i >= 3 && j >= 3 && i < rows - 3 && j < cols - 3 && inner && (!mask || mask[(size_t)i * mstep + j])) {
        const int v = img[(size_t)i * step + j];
        int d[16];
        unsigned dark = 0, bright = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            d[k] = (int)img[(size_t)(i + c_circ[k][0]) * step + (j + c_circ[k][1])] - v;
            dark |= (unsigned)(d[k] < -th) << k;
            bright |= (unsigned)(d[k] > th) << k;
        }
        auto arc9 = [](unsigned m) { unsigned a = m; for (int q = 1; q < 9; ++q) a &= ((m >> q) | (m << (16 - q))) & 0xffffu; return a != 0; };
        if (arc9(dark) || arc9(bright)) {
            int a = -32768, b = -32768;
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                int mn = 32767, mx = 32767;
#pragma unroll
                for (int q = 0; q < 9; ++q) { const int dv = d[(st + q) & 15]; mn = min(mn, dv); mx = min(mx, -dv); }
                a = max(a, mn); b = max(b, mx);
            }
            s = min(max(a, b), 256) - 1;
        }
    }
    score[(size_t)i * sstep + j] = (uint8_t)s;
}

__device__ __forceinline__ bool is_local_max(const uint8_t *__restrict__ score, size_t sstep, int i, int j)
{
    const int s = score[(size_t)i * sstep + j];
    if (s == 0) return false;
    const uint8_t *r0 = score + (size_t)(i - 1) * sstep + j, *r1 = r0 + sstep, *r2 = r1 + sstep;
    return s > r0[-1] && s > r0[0] && s > r0[1] && s > r1[-1] && s > r1[1] && s > r2[-1] && s > r2[0] && s > r2[1];
}
// nonmaxSuppression (fast.cu:346-371) in raster order instead of atomic order: per row, how many corners survive (and how many raw corners there are)
__global__ void __launch_bounds__(256) k_row_counts(const uint8_t *__restrict__ score, size_t sstep, int rows, int cols, int *__restrict__ rowcnt, unsigned *raw_total)
{
    __shared__ int s_cnt[4], s_raw[4];
    const int i = blockIdx.x;
    int cnt = 0, raw = 0;
    if (i >= 3 && i < rows - 3)
        for (int j = 3 + (int)threadIdx.x; j < cols - 3; j += 256) { raw += score[(size_t)i * sstep + j] != 0; cnt += is_local_max(score, sstep, i, j); }
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); raw += __shfl_xor(raw, o); }
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = cnt; s_raw[threadIdx.x >> 6] = raw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        rowcnt[i] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const int r = s_raw[0] + s_raw[1] + s_raw[2] + s_raw[3];
        if (r) atomicAdd(raw_total, (unsigned)r);
    }
}
__global__ void __launch_bounds__(256) k_row_scan(int *rowcnt, int rows, int *total)           // exclusive scan in place, one workgroup
{
    __shared__ int s_part[256];
    const int per = (rows + 255) / 256, a = threadIdx.x * per, b = min(a + per, rows);
    int s = 0;
    for (int i = a; i < b; ++i) s += rowcnt[i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    int off = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) off += s_part[t];
    for (int i = a; i < b; ++i) { const int c = rowcnt[i]; rowcnt[i] = off; off += c; }
    if (threadIdx.x == 255) *total = off;
}
__global__ void __launch_bounds__(64) k_row_write(const uint8_t *__restrict__ score, size_t sstep, int rows, int cols, const int *__restrict__ rowoff,
                                                  short2 *__restrict__ loc, float *__restrict__ resp)
{
    const int i = blockIdx.x;                       // one wave per row: ballot gives the in-row order
    if (i < 3 || i >= rows - 3) return;
    int off = rowoff[i];
    for (int j0 = 3; j0 < cols - 3; j0 += 64) {
        const int j = j0 + (int)threadIdx.x;
        const bool keep = j < cols - 3 && is_local_max(score, sstep, i, j);
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int k = off + __popcll(m & ((1ull << threadIdx.x) - 1ull));
            loc[k] = make_short2((short)j, (short)i);
            resp[k] = (float)score[(size_t)i * sstep + j];
        }
        off += __popcll(m);
    }
}

// createMesh's feature mask (meshwarper.cpp:82-115): overlap bands AND not-black
__global__ void __launch_bounds__(256) k_feature_mask(const uint8_t *__restrict__ img, size_t step, int rows, int cols, int ax0, int ax1, int bx0, int bx1,
                                                      uint8_t *__restrict__ mask, size_t mstep)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const uint8_t *p = img + (size_t)y * step + 3 * (size_t)x;
    const bool band = (x >= ax0 && x < ax1) || (x >= bx0 && x < bx1);
    mask[(size_t)y * mstep + x] = (band && (p[0] | p[1] | p[2])) ? 255 : 0;
}

// HarrisResponses (orb.cu:93-138): 7 x 7 block, integer gradient sums, float formula evaluated operation by operation
__global__ void __launch_bounds__(64) k_harris(const uint8_t *__restrict__ img, size_t step, const short2 *__restrict__ loc, float *__restrict__ resp, int n, int block, float k)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n) return;
    const int r = block / 2, x0 = loc[p].x - r, y0 = loc[p].y - r;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < block; ++i)
        for (int j = 0; j < block; ++j) {
            const uint8_t *q = img + (size_t)(y0 + i) * step + (x0 + j);
            const int ix = ((int)q[1] - (int)q[-1]) * 2 + ((int)q[1 - (ptrdiff_t)step] - (int)q[-1 - (ptrdiff_t)step]) + ((int)q[1 + step] - (int)q[step - 1]);
            const int iy = ((int)q[step] - (int)q[-(ptrdiff_t)step]) * 2 + ((int)q[step - 1] - (int)q[-(ptrdiff_t)step - 1]) + ((int)q[step + 1] - (int)q[1 - (ptrdiff_t)step]);
            a += ix * ix; b += iy * iy; c += ix * iy;
        }
    float scale = (float)(1 << 2) * (float)block * 255.0f;
    scale = 1.0f / scale;
    const float s4 = ((scale * scale) * scale) * scale;
    const float fa = (float)a, fb = (float)b, fc = (float)c, s = fa + fb;
    resp[p] = ((fa * fb - fc * fc) - (k * s) * s) * s4;
}

// IC_Angle (orb.cu:160-211)
__global__ void __launch_bounds__(64) k_ic_angle(const uint8_t *__restrict__ img, size_t step, const short2 *__restrict__ loc, float *__restrict__ angle, int n, int half)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n) return;
    const uint8_t *c = img + (size_t)loc[p].y * step + loc[p].x;
    int m01 = 0, m10 = 0;
    for (int u = -half; u <= half; ++u) m10 += u * (int)c[u];
    for (int v = 1; v <= half; ++v) {
        const int d = c_umax[v];
        int vs = 0, msum = 0;
        for (int u = -d; u <= d; ++u) {
            const int pl = c[(ptrdiff_t)v * (ptrdiff_t)step + u], mi = c[-(ptrdiff_t)v * (ptrdiff_t)step + u];
            vs += pl - mi; msum += u * (pl + mi);
        }
        m10 += msum; m01 += v * vs;
    }
    const float PI_F = 3.14159265f;
    float a = (float)atan2((double)(float)m01, (double)(float)m10);
    if (a < 0) a += 2.0f * PI_F;
    angle[p] = a * (180.0f / PI_F);
}

// computeOrbDescriptor<2> (orb.cu:222-246, :352-367): byte b of keypoint p = 8 tests on the pattern rotated by the keypoint angle
__global__ void __launch_bounds__(256) k_orb_desc(const uint8_t *__restrict__ img, size_t step, const short2 *__restrict__ loc, const float *__restrict__ angle, int n,
                                                  uint8_t *__restrict__ desc, size_t dstep)
{
    const int b = threadIdx.x & 31, p = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (p >= n) return;
    const float PI_F = 3.14159265f;
    const float a = angle[p] * (float)(PI_F / 180.f);
    const float sina = (float)sin((double)a), cosa = (float)cos((double)a);
    const int x = loc[p].x, y = loc[p].y;
    auto val = [&](int idx) {
        const float px = (float)c_pattern[16 * b + idx].x, py = (float)c_pattern[16 * b + idx].y;
        const int yy = y + (int)__builtin_rintf(px * sina + py * cosa), xx = x + (int)__builtin_rintf(px * cosa - py * sina);
        return (int)img[(size_t)yy * step + xx];
    };
    int v = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) v |= (val(2 * t) < val(2 * t + 1)) << t;
    desc[(size_t)p * dstep + b] = (uint8_t)v;
}

// ---- RANSAC: one thread per 4-point hypothesis -- HomographyEstimatorCallback::runKernel (fundam.cpp:80-142) then the inlier count of
// RANSACPointSetRegistrator::findInliers with computeError's float arithmetic (fundam.cpp:144-166, ptsetreg.cpp:85-104)
__host__ __device__ inline void jacobi_eig9(double A[9][9], double V[9][9], double w[9])
{
    for (int i = 0; i < 9; ++i) { for (int j = 0; j < 9; ++j) V[i][j] = 0; V[i][i] = 1; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 9; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 9; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 9; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
    for (int i = 0; i < 9; ++i) w[i] = A[i][i];
}
}  // namespace

// (host and device) the DLT of runKernel on `count` correspondences M[i] -> m[i]; false where the reference returns 0 models
__host__ __device__ static bool homography_dlt(const float *Mx, const float *My, const float *mx, const float *my, int count, double H[9])
{
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; ++i) { cmx += mx[i]; cmy += my[i]; cMx += Mx[i]; cMy += My[i]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; ++i) { smx += fabs(mx[i] - cmx); smy += fabs(my[i] - cmy); sMx += fabs(Mx[i] - cMx); sMy += fabs(My[i] - cMy); }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return false;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    double L[9][9], V[9][9], w[9];
    for (int j = 0; j < 9; ++j) for (int k = 0; k < 9; ++k) L[j][k] = 0;
    for (int i = 0; i < count; ++i) {
        const double x = (mx[i] - cmx) * smx, y = (my[i] - cmy) * smy, X = (Mx[i] - cMx) * sMx, Y = (My[i] - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; ++j) for (int k = j; k < 9; ++k) L[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; ++j) for (int k = 0; k < j; ++k) L[j][k] = L[k][j];
    jacobi_eig9(L, V, w);
    int best = 0;
    for (int i = 1; i < 9; ++i) if (w[i] < w[best]) best = i;               // cv::eigen sorts descending and the reference takes the last row
    double h0[9];
    for (int i = 0; i < 9; ++i) h0[i] = V[i][best];
    const double inv[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1}, nrm[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    double t[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) t[3 * r + c] = inv[3 * r] * h0[c] + inv[3 * r + 1] * h0[3 + c] + inv[3 * r + 2] * h0[6 + c];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[3 * r + c] = t[3 * r] * nrm[c] + t[3 * r + 1] * nrm[3 + c] + t[3 * r + 2] * nrm[6 + c];
    const double s = 1. / H[8];
    for (int i = 0; i < 9; ++i) H[i] *= s;
    return true;
}
__host__ __device__ static inline int homography_inliers(const double H[9], const float *Mx, const float *My, const float *mx, const float *my, int n, float t, uint8_t *mask)
{
    const float h0 = (float)H[0], h1 = (float)H[1], h2 = (float)H[2], h3 = (float)H[3], h4 = (float)H[4], h5 = (float)H[5], h6 = (float)H[6], h7 = (float)H[7];
    int good = 0;
    for (int i = 0; i < n; ++i) {
        const float ww = 1.f / (h6 * Mx[i] + h7 * My[i] + 1.f);
        const float dx = (h0 * Mx[i] + h1 * My[i] + h2) * ww - mx[i], dy = (h3 * Mx[i] + h4 * My[i] + h5) * ww - my[i];
        const int f = dx * dx + dy * dy <= t;
        if (mask) mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

namespace {
__global__ void __launch_bounds__(64) k_ransac_score(const float *__restrict__ Mx, const float *__restrict__ My, const float *__restrict__ mx, const float *__restrict__ my, int n,
                                                     const int *__restrict__ subsets, int n_hyp, float thresh_sq, double *__restrict__ models, int *__restrict__ good)
{
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= n_hyp) return;
    float sMx[4], sMy[4], smx[4], smy[4];
    for (int k = 0; k < 4; ++k) { const int i = subsets[4 * h + k]; sMx[k] = Mx[i]; sMy[k] = My[i]; smx[k] = mx[i]; smy[k] = my[i]; }
    double H[9];
    int g = -1;
    if (homography_dlt(sMx, sMy, smx, smy, 4, H)) g = homography_inliers(H, Mx, My, mx, my, n, thresh_sq, nullptr);
    else for (int i = 0; i < 9; ++i) H[i] = 0;
    good[h] = g;
    for (int i = 0; i < 9; ++i) models[9 * h + i] = H[i];
}

// cv::RNG (core/include/opencv2/core/operations.hpp: multiply-with-carry, CV_RNG_COEFF 4164903690)
struct CvRng {
    unsigned long long state;
    explicit CvRng(unsigned long long s) : state(s ? s : 0xffffffffull) {}
    unsigned next() { state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

bool collinear_last(const float *x, const float *y, int count)            // haveCollinearPoints (calib3d/src/precomp.hpp:118-139): only the LAST point is tested
{
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = x[j] - x[i], dy1 = y[j] - y[i];
        for (int k = 0; k < j; ++k) {
            const double dx2 = x[k] - x[i], dy2 = y[k] - y[i];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
        }
    }
    return false;
}
bool check_subset4(const float *sx, const float *sy, const float *dx, const float *dy)      // HomographyEstimatorCallback::checkSubset (fundam.cpp:49-78)
{
    if (collinear_last(sx, sy, 4) || collinear_last(dx, dy, 4)) return false;
    static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; ++i) {
        const int *t = tt[i];
        auto det = [&](const float *x, const float *y) {
            const double a = x[t[0]], b = y[t[0]], c = x[t[1]], d = y[t[1]], e = x[t[2]], f = y[t[2]];
            return a * (d - f) - b * (c - e) + (c * f - d * e);            // Matx33d determinant, rows (x, y, 1)
        };
        negative += det(sx, sy) * det(dx, dy) < 0;
    }
    return negative == 0 || negative == 4;
}

int ransac_update_iters(double p, double ep, int model_points, int max_iters)       // RANSACUpdateNumIters (ptsetreg.cpp:53-73)
{
    p = std::min(std::max(p, 0.), 1.); ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN), denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::nearbyint(num / denom);
}

// solve the 8 x 8 symmetric system (A + lambda diag) d = v by Gaussian elimination with partial pivoting (the reference: solve(..., DECOMP_EIG))
bool solve8(const double A[8][8], const double v[8], double d[8])
{
    double M[8][9];
    for (int i = 0; i < 8; ++i) { for (int j = 0; j < 8; ++j) M[i][j] = A[i][j]; M[i][8] = v[i]; }
    for (int i = 0; i < 8; ++i) {
        int p = i;
        for (int r = i + 1; r < 8; ++r) if (std::fabs(M[r][i]) > std::fabs(M[p][i])) p = r;
        if (std::fabs(M[p][i]) < 1e-300) return false;
        if (p != i) for (int c = 0; c < 9; ++c) std::swap(M[i][c], M[p][c]);
        for (int r = i + 1; r < 8; ++r) { const double f = M[r][i] / M[i][i]; for (int c = i; c < 9; ++c) M[r][c] -= f * M[i][c]; }
    }
    for (int i = 7; i >= 0; --i) { double s = M[i][8]; for (int c = i + 1; c < 8; ++c) s -= M[i][c] * d[c]; d[i] = s / M[i][i]; }
    return true;
}

// HomographyRefineCallback::compute (fundam.cpp:175-214) + LMSolverImpl::run, 10 iterations (levmarq.cpp:88-199)
void lm_refine(const float *Mx, const float *My, const float *mx, const float *my, int count, double h[8])
{
    auto compute = [&](const double *p, std::vector<double> &err, double (*JtJ)[8], double *Jtr) {
        err.resize(2 * (size_t)count);
        if (JtJ) { for (int a = 0; a < 8; ++a) { for (int b = 0; b < 8; ++b) JtJ[a][b] = 0; Jtr[a] = 0; } }
        for (int i = 0; i < count; ++i) {
            const double X = Mx[i], Y = My[i];
            double ww = p[6] * X + p[7] * Y + 1.;
            ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            const double xi = (p[0] * X + p[1] * Y + p[2]) * ww, yi = (p[3] * X + p[4] * Y + p[5]) * ww;
            err[2 * i] = xi - mx[i]; err[2 * i + 1] = yi - my[i];
            if (JtJ) {
                const double J0[8] = {X * ww, Y * ww, ww, 0, 0, 0, -X * ww * xi, -Y * ww * xi}, J1[8] = {0, 0, 0, X * ww, Y * ww, ww, -X * ww * yi, -Y * ww * yi};
                for (int a = 0; a < 8; ++a) {
                    for (int b = 0; b < 8; ++b) JtJ[a][b] += J0[a] * J0[b] + J1[a] * J1[b];
                    Jtr[a] += J0[a] * err[2 * i] + J1[a] * err[2 * i + 1];
                }
            }
        }
    };
    auto sq = [](const std::vector<double> &e) { double s = 0; for (double v : e) s += v * v; return s; };
    double x[8], xd[8], A[8][8], v[8], D[8], d[8];
    memcpy(x, h, sizeof(x));
    std::vector<double> r, rd;
    compute(x, r, A, v);
    double S = sq(r);
    for (int i = 0; i < 8; ++i) D[i] = A[i][i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    for (int iter = 0;;) {
        double Ap[8][8];
        memcpy(Ap, A, sizeof(Ap));
        for (int i = 0; i < 8; ++i) Ap[i][i] += lambda * D[i];
        if (!solve8(Ap, v, d)) break;
        for (int i = 0; i < 8; ++i) xd[i] = x[i] - d[i];
        compute(xd, rd, nullptr, nullptr);
        const double Sd = sq(rd);
        double dS = 0, dv = 0;
        for (int i = 0; i < 8; ++i) { double t = 2 * v[i]; for (int j = 0; j < 8; ++j) t -= A[i][j] * d[j]; dS += d[i] * t; dv += d[i] * v[i]; }
        const double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
        else if (R < Rlo) {
            double nu = (Sd - S) / (std::fabs(dv) > DBL_EPSILON ? dv : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                double maxval = DBL_EPSILON;              // largest diagonal entry of inv(A): one unit-vector solve per column
                for (int c = 0; c < 8; ++c) { double e[8] = {0}, col[8]; e[c] = 1; if (solve8(A, e, col)) maxval = std::max(maxval, std::fabs(col[c])); }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) { S = Sd; memcpy(x, xd, sizeof(x)); compute(x, r, A, v); }
        ++iter;
        double nd = 0, nr = 0;
        for (int i = 0; i < 8; ++i) nd = std::max(nd, std::fabs(d[i]));
        for (double e : r) nr = std::max(nr, std::fabs(e));
        if (!(iter < 10 && nd >= FLT_EPSILON && nr >= FLT_EPSILON)) break;
    }
    memcpy(h, x, sizeof(x));
}

}  // namespace
}  // namespace ms

using namespace ms;

extern "C" {

int ms_orb_default_params(ms_orb_params *p)
{
    if (!p) return fail(MS_ERR_INVALID, "ms_orb_default_params: null");
    p->nfeatures = 2500; p->scale_factor = 1.2f; p->nlevels = 8;          // cuda::ORB::create(2500, 1.2f, 8), featurefinder.cpp:15
    p->edge_threshold = 31; p->first_level = 0; p->patch_size = 31; p->fast_threshold = 20;      // cuda::ORB::create defaults (cudafeatures2d.hpp)
    return MS_OK;
}

int ms_orb_detect_and_compute(const ms_image *gray, const ms_image *mask, const ms_orb_params *prm, float *kp_host, int max_keypoints,
                              ms_image *desc, int *n_out, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(gray && gray->data && gray->type == MS_8UC1 && prm && kp_host && desc && desc->data && n_out, "ms_orb_detect_and_compute: bad argument");
    MS_CHECK(!mask || (mask->data && mask->type == MS_8UC1 && mask->rows == gray->rows && mask->cols == gray->cols), "ms_orb_detect_and_compute: mask must be 8UC1 of the image size");
    MS_CHECK(prm->nlevels >= 1 && prm->nlevels <= 16 && prm->first_level == 0 && prm->patch_size == 31 && prm->nfeatures >= 1 && prm->scale_factor > 1.f,
             "ms_orb_detect_and_compute: supports first_level 0, patch_size 31 (the learned pattern), 1..16 levels");
    // the device kernels read a (rotated) 31-px patch, the 15-px intensity-centroid disc and the 7 x 7 Harris block + Sobel halo around every keypoint WITHOUT bounds
    // checks: keypoints must keep the creator's border (orb.cpp:686-689, edgeThreshold 31 by default; 19 = ceil(18.4) is the reach of the rotated pattern);
    // a FAST threshold of 0 would make score 0 -- "no corner" in the score map -- a valid corner (fast.cu:224-282 has the same encoding), 255 can never fire
    MS_CHECK(prm->edge_threshold >= 19 && prm->edge_threshold >= prm->patch_size / 2 + 1 && prm->fast_threshold >= 1 && prm->fast_threshold <= 254,
             "ms_orb_detect_and_compute: edge_threshold %d must be >= 19 and fast_threshold %d in [1, 254]", prm->edge_threshold, prm->fast_threshold);
    MS_CHECK(desc->type == MS_8UC1 && desc->cols == 32 && desc->rows >= max_keypoints && max_keypoints >= prm->nfeatures, "ms_orb_detect_and_compute: descriptors must be 8UC1 max_keypoints x 32, max_keypoints >= nfeatures");
    hipStream_t st = as_stream(stream);
    const int half = prm->patch_size / 2;
    {   // u_max of the circular patch (orb.cpp:514-529)
        std::vector<int> u(half + 2, 0);
        const float r2 = std::sqrt(2.f);
        for (int v = 0; v <= half * r2 / 2 + 1; ++v) u[v] = (int)std::nearbyint(std::sqrt((float)(half * half - v * v)));
        for (int v = half, v0 = 0; v >= half * r2 / 2; --v) { while (u[v0] == u[v0 + 1]) ++v0; u[v] = v0; ++v0; }
        int tab[32] = {0};
        for (size_t i = 0; i < u.size() && i < 32; ++i) tab[i] = u[i];
        MS_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_umax), tab, sizeof(tab), 0, hipMemcpyHostToDevice, st));
    }
    // per-level budgets (orb.cpp:501-512)
    std::vector<int> nper(prm->nlevels);
    {
        const float factor = 1.0f / prm->scale_factor;
        float nd = (float)((double)(prm->nfeatures * (1.0f - factor)) / (1.0 - std::pow((double)factor, prm->nlevels)));
        int sum = 0;
        for (int l = 0; l < prm->nlevels - 1; ++l) { nper[l] = (int)std::nearbyint(nd); sum += nper[l]; nd *= factor; }
        nper[prm->nlevels - 1] = prm->nfeatures - sum;
    }
    auto level_scale = [&](int l) { return (float)std::pow((double)prm->scale_factor, l - prm->first_level); };     // getScale: pow(float, int) -> double -> float
    struct Dev { void *p = nullptr; ~Dev() { if (p) (void)hipFree(p); } int alloc(size_t n) { if (p) (void)hipFree(p); p = nullptr; MS_HIP(hipMalloc(&p, n ? n : 16)); return MS_OK; } };
    Dev img_a, img_b, msk_a, msk_b, score, rowcnt, counters, loc, resp, ang;
    const size_t full = (size_t)gray->rows * gray->cols;
    if (int e = img_a.alloc(full)) return e;
    if (int e = img_b.alloc(full)) return e;
    if (mask) { if (int e = msk_a.alloc(full)) return e; if (int e = msk_b.alloc(full)) return e; }
    if (int e = score.alloc(full)) return e;
    if (int e = rowcnt.alloc(sizeof(int) * (size_t)gray->rows)) return e;
    if (int e = counters.alloc(16)) return e;
    ms_image prev_img{}, prev_msk{};
    int total = 0;
    for (int level = 0; level < prm->nlevels; ++level) {
        const float sc = 1.0f / level_scale(level);
        const int w = (int)std::nearbyint((float)gray->cols * sc), h = (int)std::nearbyint((float)gray->rows * sc);      // cvRound(image.cols * scale)  orb.cpp:668
        if (w < 8 || h < 8) break;
        ms_image cur{(level & 1) ? img_b.p : img_a.p, (size_t)w, w, h, MS_8UC1}, curm{};
        if (level == prm->first_level) {
            MS_HIP(hipMemcpy2DAsync(cur.data, cur.step, gray->data, gray->step, (size_t)w, h, hipMemcpyDeviceToDevice, st));
            if (mask) { curm = ms_image{msk_a.p, (size_t)w, w, h, MS_8UC1}; MS_HIP(hipMemcpy2DAsync(curm.data, curm.step, mask->data, mask->step, (size_t)w, h, hipMemcpyDeviceToDevice, st)); }
        } else {
            if (int e = launch_resize_linear(prev_img, cur, 0, 0, st)) return e;                                            // orb.cpp:686
            if (mask) {
                curm = ms_image{(level & 1) ? msk_b.p : msk_a.p, (size_t)w, w, h, MS_8UC1};
                if (int e = launch_resize_linear(prev_msk, curm, 0, 0, st)) return e;                                     // :690
                k_tozero254<<<dim3(div_up(w, 64), div_up(h, 4)), dim3(64, 4), 0, st>>>((uint8_t *)curm.data, curm.step, h, w);   // :693
                MS_LAUNCH_CHECK();
            }
        }
        prev_img = cur; prev_msk = curm;
        // FAST 9-16, threshold, score, non-max suppression, raster-ordered compaction
        MS_HIP(hipMemsetAsync(counters.p, 0, 16, st));
        k_fast_score<<<dim3(div_up(w, 64), div_up(h, 4)), dim3(64, 4), 0, st>>>((const uint8_t *)cur.data, cur.step, h, w, mask ? (const uint8_t *)curm.data : nullptr, curm.step,
                                                                                  prm->edge_threshold, prm->fast_threshold, (uint8_t *)score.p, (size_t)w);
        MS_LAUNCH_CHECK();
        k_row_counts<<<h, 256, 0, st>>>((const uint8_t *)score.p, (size_t)w, h, w, (int *)rowcnt.p, (unsigned *)counters.p);
        MS_LAUNCH_CHECK();
        k_row_scan<<<1, 256, 0, st>>>((int *)rowcnt.p, h, (int *)counters.p + 1);
        MS_LAUNCH_CHECK();
        unsigned hc[2] = {0, 0};
        MS_HIP(hipMemcpyAsync(hc, counters.p, sizeof(hc), hipMemcpyDeviceToHost, st));
        MS_HIP(hipStreamSynchronize(st));                                   // (the reference synchronises here as well: fast.cu:340-341, :381-382)
        const unsigned raw = hc[0];
        int count = (int)hc[1];
        const int max_npoints = (int)(0.05 * ((double)w * h));              // fastDetector_->setMaxNumPoints(0.05 * area)  orb.cpp:753
        std::vector<short2> hloc;
        std::vector<float> hresp;
        if (count == 0) continue;
        if ((int)raw > max_npoints) {
            // more raw corners than the detector's buffer: the reference keeps whichever max_npoints win its atomic counter; here the first ones in
            // raster order do.  Rare (5 % of the pixels are corners): done on the host from the score map.
            std::vector<uint8_t> hs((size_t)w * h);
            MS_HIP(hipMemcpy(hs.data(), score.p, hs.size(), hipMemcpyDeviceToHost));
            int seen = 0;
            for (int i = 3; i < h - 3 && seen < max_npoints; ++i)
                for (int j = 3; j < w - 3 && seen < max_npoints; ++j) {
                    const int s = hs[(size_t)i * w + j];
                    if (!s) continue;
                    ++seen;
                    bool mx = true;
                    for (int dy = -1; dy <= 1 && mx; ++dy) for (int dx = -1; dx <= 1; ++dx) if ((dy || dx) && s <= hs[(size_t)(i + dy) * w + j + dx]) { mx = false; break; }
                    if (mx) { hloc.push_back(make_short2((short)j, (short)i)); hresp.push_back((float)s); }
                }
            count = (int)hloc.size();
            if (count == 0) continue;
        }
        if (int e = loc.alloc(sizeof(short2) * (size_t)count)) return e;
        if (int e = resp.alloc(sizeof(float) * (size_t)count)) return e;
        const int n = nper[level];
        auto cull = [&](int keep) -> int {                                  // orb.cpp:719-733; stable instead of thrust's unstable device sort
            if (count <= keep) return MS_OK;
            if (hloc.empty()) {
                hloc.resize(count); hresp.resize(count);
                MS_HIP(hipMemcpyAsync(hloc.data(), loc.p, sizeof(short2) * (size_t)count, hipMemcpyDeviceToHost, st));
                MS_HIP(hipMemcpyAsync(hresp.data(), resp.p, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, st));
                MS_HIP(hipStreamSynchronize(st));
            }
            std::vector<int> order(count);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return hresp[a] > hresp[b]; });
            std::vector<short2> l2(keep);
            std::vector<float> r2(keep);
            for (int i = 0; i < keep; ++i) { l2[i] = hloc[order[i]]; r2[i] = hresp[order[i]]; }
            hloc.swap(l2); hresp.swap(r2);
            count = keep;
            MS_HIP(hipMemcpyAsync(loc.p, hloc.data(), sizeof(short2) * (size_t)count, hipMemcpyHostToDevice, st));
            MS_HIP(hipMemcpyAsync(resp.p, hresp.data(), sizeof(float) * (size_t)count, hipMemcpyHostToDevice, st));
            MS_HIP(hipStreamSynchronize(st));          // (the host vectors are reused)
            return MS_OK;
        };
        if (hloc.empty()) {
            k_row_write<<<h, 64, 0, st>>>((const uint8_t *)score.p, (size_t)w, h, w, (const int *)rowcnt.p, (short2 *)loc.p, (float *)resp.p);
            MS_LAUNCH_CHECK();
        } else {
            MS_HIP(hipMemcpyAsync(loc.p, hloc.data(), sizeof(short2) * (size_t)count, hipMemcpyHostToDevice, st));
            MS_HIP(hipMemcpyAsync(resp.p, hresp.data(), sizeof(float) * (size_t)count, hipMemcpyHostToDevice, st));
            MS_HIP(hipStreamSynchronize(st));
        }
        if (int e = cull(2 * n)) return e;                                                                                  // orb.cpp:772
        k_harris<<<div_up(count, 64), 64, 0, st>>>((const uint8_t *)cur.data, cur.step, (const short2 *)loc.p, (float *)resp.p, count, 7, 0.04f);   // :774
        MS_LAUNCH_CHECK();
        hloc.clear(); hresp.clear();
        if (int e = cull(n)) return e;                                                                                      // :778
        if (count == 0) continue;
        if (int e = ang.alloc(sizeof(float) * (size_t)count)) return e;
        k_ic_angle<<<div_up(count, 64), 64, 0, st>>>((const uint8_t *)cur.data, cur.step, (const short2 *)loc.p, (float *)ang.p, count, half);          // :780
        MS_LAUNCH_CHECK();
        if (total + count > max_keypoints) return fail(MS_ERR_INVALID, "ms_orb_detect_and_compute: more than max_keypoints (%d) keypoints", max_keypoints);
        k_orb_desc<<<div_up(count, 8), 256, 0, st>>>((const uint8_t *)cur.data, cur.step, (const short2 *)loc.p, (const float *)ang.p, count,
                                                     (uint8_t *)desc->data + (size_t)total * desc->step, desc->step);                                  // :783-821
        MS_LAUNCH_CHECK();
        // mergeKeyPoints (orb.cpp:823-865): x, y, response, angle, octave, size
        if (hloc.empty()) { hloc.resize(count); MS_HIP(hipMemcpyAsync(hloc.data(), loc.p, sizeof(short2) * (size_t)count, hipMemcpyDeviceToHost, st)); }
        hresp.resize(count);
        std::vector<float> hang(count);
        MS_HIP(hipMemcpyAsync(hresp.data(), resp.p, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, st));
        MS_HIP(hipMemcpyAsync(hang.data(), ang.p, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, st));
        MS_HIP(hipStreamSynchronize(st));
        const float sf = level_scale(level), loc_scale = level != prm->first_level ? sf : 1.0f;
        for (int i = 0; i < count; ++i) {
            float *k = kp_host + 6 * (size_t)(total + i);
            k[0] = (float)hloc[i].x * loc_scale; k[1] = (float)hloc[i].y * loc_scale; k[2] = hresp[i]; k[3] = hang[i]; k[4] = (float)level; k[5] = (float)prm->patch_size * sf;
        }
        total += count;
    }
    MS_HIP(hipStreamSynchronize(st));
    *n_out = total;
    return MS_OK;
}

int ms_feature_mask(const ms_image *img, int a_x0, int a_w, int b_x0, int b_w, ms_image *mask, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(img && mask && img->data && mask->data && img->type == MS_8UC3 && mask->type == MS_8UC1 && img->rows == mask->rows && img->cols == mask->cols,
             "ms_feature_mask: need an 8UC3 image and an 8UC1 mask of the same size");
    k_feature_mask<<<dim3(div_up(img->cols, 64), div_up(img->rows, 4)), dim3(64, 4), 0, as_stream(stream)>>>(
        (const uint8_t *)img->data, img->step, img->rows, img->cols, a_x0, a_x0 + a_w, b_x0, b_x0 + b_w, (uint8_t *)mask->data, mask->step);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

int ms_find_homography_ransac(const float *src_xy, const float *dst_xy, int n, double reproj_threshold, int max_iters, double confidence,
                              double *H_out, uint8_t *inlier_mask, int *n_inliers, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(src_xy && dst_xy && H_out && n >= 0, "ms_find_homography_ransac: bad argument");
    if (reproj_threshold <= 0) reproj_threshold = 3;                       // defaultRANSACReprojThreshold (fundam.cpp:325, :351)
    if (max_iters <= 0) max_iters = 2000;
    if (!(confidence > 0 && confidence < 1)) confidence = 0.995;
    if (n_inliers) *n_inliers = 0;
    if (inlier_mask) memset(inlier_mask, 0, (size_t)n);
    for (int i = 0; i < 9; ++i) H_out[i] = 0;
    if (n < 4) return 1;                                                   // (the reference returns an empty Mat)
    std::vector<float> Mx(n), My(n), mx(n), my(n);
    for (int i = 0; i < n; ++i) { Mx[i] = src_xy[2 * i]; My[i] = src_xy[2 * i + 1]; mx[i] = dst_xy[2 * i]; my[i] = dst_xy[2 * i + 1]; }
    std::vector<uint8_t> best_mask(n, 0);
    double H[9];
    bool ok = false;
    if (n == 4) {                                                          // method == 0 || npoints == 4 (fundam.cpp:356-360)
        ok = homography_dlt(Mx.data(), My.data(), mx.data(), my.data(), 4, H);
        std::fill(best_mask.begin(), best_mask.end(), (uint8_t)1);
    } else {
        // every subset the sequential loop COULD draw, in its order (ptsetreg.cpp:106-173): the draw does not depend on the models found
        CvRng rng((unsigned long long)-1);
        std::vector<int> subsets;
        subsets.reserve(4 * (size_t)max_iters);
        int drawn = 0;
        for (; drawn < max_iters; ++drawn) {
            int idx[4];
            bool found = false;
            for (int iters = 0; iters < 10000; ++iters) {
                for (int i = 0; i < 4; ++i) {
                    for (;;) { idx[i] = rng.uniform(0, n); int j = 0; for (; j < i; ++j) if (idx[i] == idx[j]) break; if (j == i) break; }
                }
                float sx[4], sy[4], dx[4], dy[4];
                for (int i = 0; i < 4; ++i) { sx[i] = Mx[idx[i]]; sy[i] = My[idx[i]]; dx[i] = mx[idx[i]]; dy[i] = my[idx[i]]; }
                if (check_subset4(sx, sy, dx, dy)) { found = true; break; }
            }
            if (!found) break;
            subsets.insert(subsets.end(), idx, idx + 4);
        }
        if (drawn == 0) return 1;
        // all hypotheses scored at once on the device
        hipStream_t st = as_stream(stream);
        struct Dev { void *p = nullptr; ~Dev() { if (p) (void)hipFree(p); } };
        Dev d_pts, d_sub, d_models, d_good;
        MS_HIP(hipMalloc(&d_pts.p, sizeof(float) * 4 * (size_t)n));
        MS_HIP(hipMalloc(&d_sub.p, sizeof(int) * subsets.size()));
        MS_HIP(hipMalloc(&d_models.p, sizeof(double) * 9 * (size_t)drawn));
        MS_HIP(hipMalloc(&d_good.p, sizeof(int) * (size_t)drawn));
        float *dp = (float *)d_pts.p;
        MS_HIP(hipMemcpyAsync(dp, Mx.data(), sizeof(float) * n, hipMemcpyHostToDevice, st));
        MS_HIP(hipMemcpyAsync(dp + n, My.data(), sizeof(float) * n, hipMemcpyHostToDevice, st));
        MS_HIP(hipMemcpyAsync(dp + 2 * (size_t)n, mx.data(), sizeof(float) * n, hipMemcpyHostToDevice, st));
        MS_HIP(hipMemcpyAsync(dp + 3 * (size_t)n, my.data(), sizeof(float) * n, hipMemcpyHostToDevice, st));
        MS_HIP(hipMemcpyAsync(d_sub.p, subsets.data(), sizeof(int) * subsets.size(), hipMemcpyHostToDevice, st));
        const float t = (float)(reproj_threshold * reproj_threshold);
        k_ransac_score<<<div_up(drawn, 64), 64, 0, st>>>(dp, dp + n, dp + 2 * (size_t)n, dp + 3 * (size_t)n, n, (const int *)d_sub.p, drawn, t, (double *)d_models.p, (int *)d_good.p);
        MS_LAUNCH_CHECK();
        std::vector<int> good(drawn);
        std::vector<double> models(9 * (size_t)drawn);
        MS_HIP(hipMemcpyAsync(good.data(), d_good.p, sizeof(int) * (size_t)drawn, hipMemcpyDeviceToHost, st));
        MS_HIP(hipMemcpyAsync(models.data(), d_models.p, sizeof(double) * 9 * (size_t)drawn, hipMemcpyDeviceToHost, st));
        MS_HIP(hipStreamSynchronize(st));
        // the sequential scan of RANSACPointSetRegistrator::run (ptsetreg.cpp:208-240): first strictly better model wins, the iteration budget shrinks as it goes
        int niters = std::max(max_iters, 1), max_good = 0, best = -1;
        for (int iter = 0; iter < niters && iter < drawn; ++iter) {
            if (good[iter] < 0) continue;                                  // runKernel returned no model
            if (good[iter] > std::max(max_good, 3)) {
                best = iter; max_good = good[iter];
                niters = ransac_update_iters(confidence, (double)(n - max_good) / n, 4, niters);
            }
        }
        if (best >= 0) {
            memcpy(H, &models[9 * (size_t)best], sizeof(H));
            homography_inliers(H, Mx.data(), My.data(), mx.data(), my.data(), n, t, best_mask.data());
            ok = true;
        }
    }
    if (!ok) return 1;
    // compressElems + runKernel on the inliers + Levenberg-Marquardt polish (fundam.cpp:370-385)
    if (n > 4) {
        std::vector<float> iMx, iMy, imx, imy;
        for (int i = 0; i < n; ++i) if (best_mask[i]) { iMx.push_back(Mx[i]); iMy.push_back(My[i]); imx.push_back(mx[i]); imy.push_back(my[i]); }
        const int m = (int)iMx.size();
        if (m > 0) {
            double H2[9];
            if (homography_dlt(iMx.data(), iMy.data(), imx.data(), imy.data(), m, H2)) memcpy(H, H2, sizeof(H));
            lm_refine(iMx.data(), iMy.data(), imx.data(), imy.data(), m, H);      // the 8 free parameters; H[8] stays 1
        }
    }
    memcpy(H_out, H, sizeof(H));
    int cnt = 0;
    for (int i = 0; i < n; ++i) { if (inlier_mask) inlier_mask[i] = best_mask[i]; cnt += best_mask[i]; }
    if (n_inliers) *n_inliers = cnt;
    return MS_OK;
}

}  // extern "C"
