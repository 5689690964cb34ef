// calib.hip -- calibration-time pixel work that the reference runs on the CPU, on the device:
//   VoronoiSeamFinder::findInPair       sources/modules/stitching/src/seam_finders.cpp:111-160
//       distanceTransform(DIST_L1, 3)   sources/modules/imgproc/src/distransform.cpp:70-137
//   GainCompensator::feed               sources/modules/stitching/src/exposure_compensate.cpp:71-145 (overlap sums, normal equations, cv::solve)
// so that ms_build_masks / ms_calibrate_seam move no pixels to the host: the only thing that comes back is the N gains.
//
// Exactness.  The reference's two-pass 3 x 3 chamfer with costs (1, 2) IS the city-block distance to the nearest zero pixel of the window
// (a diagonal step costs two axis steps), in 16.16 fixed point; here the same integers come from two separable 1-D passes (columns, then
// rows), which parallelise.  Where a window has no zero pixel at all the chamfer leaves INIT + 1 everywhere; the separable form leaves a
// large constant too, and the only use of the distances is `dist1 < dist2`, so the seam is the same.
// The gain sums are DOUBLE sums of square roots in raster order (exposure_compensate.cpp:103-117): order matters for the last bit, so one
// thread per image pair walks its overlap in that order (the overlaps are seam-scale images: a few thousand pixels); the n x n solve follows
// cv::solve's closed forms / LU (as the host version did) in a single thread.
#include <vector>
#include "common.hpp"
#include "launchers.hpp"

namespace ms {
namespace {

constexpr int VGAP = 10;                 // findInPair's `gap`
constexpr int DINF = 1 << 28;

struct PairGeom { int rx, ry, rw, rh; int x1, y1, w1, h1; int x2, y2, w2, h2; };        // overlap roi, the two views' rois

__device__ __forceinline__ uint8_t at_mask(const uint8_t *m, int w, int h, int y, int x) { return (y >= 0 && x >= 0 && y < h && x < w) ? m[(size_t)y * w + x] : 0; }

// columns of the (rh + 2 gap) x (rw + 2 gap) window: distance, along the column, to the nearest pixel that is set in exactly one of the two masks
__global__ void __launch_bounds__(64) k_vor_cols(const uint8_t *__restrict__ m1, const uint8_t *__restrict__ m2, PairGeom g, int *__restrict__ d1, int *__restrict__ d2)
{
    const int C = g.rw + 2 * VGAP, R = g.rh + 2 * VGAP;
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= C) return;
    int a1 = DINF, a2 = DINF;
    for (int y = 0; y < R; ++y) {
        const uint8_t a = at_mask(m1, g.w1, g.h1, g.ry - g.y1 + y - VGAP, g.rx - g.x1 + x - VGAP);
        const uint8_t b = at_mask(m2, g.w2, g.h2, g.ry - g.y2 + y - VGAP, g.rx - g.x2 + x - VGAP);
        const bool both = a && b;
        a1 = (!both && a) ? 0 : min(a1 + 1, DINF);
        a2 = (!both && b) ? 0 : min(a2 + 1, DINF);
        d1[(size_t)y * C + x] = a1; d2[(size_t)y * C + x] = a2;
    }
    a1 = a2 = DINF;
    for (int y = R - 1; y >= 0; --y) {
        const size_t i = (size_t)y * C + x;
        a1 = min(min(a1 + 1, DINF), d1[i]); a2 = min(min(a2 + 1, DINF), d2[i]);
        d1[i] = a1; d2[i] = a2;
    }
}
// rows: the second 1-D pass, then the seam decision of findInPair (:148-159) on the overlap itself
__global__ void __launch_bounds__(64) k_vor_rows(uint8_t *__restrict__ m1, uint8_t *__restrict__ m2, PairGeom g, int *__restrict__ d1, int *__restrict__ d2)
{
    const int C = g.rw + 2 * VGAP, R = g.rh + 2 * VGAP;
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= R) return;
    int *r1 = d1 + (size_t)y * C, *r2 = d2 + (size_t)y * C;
    int a1 = DINF, a2 = DINF;
    for (int x = 0; x < C; ++x) { a1 = min(min(a1 + 1, DINF), r1[x]); a2 = min(min(a2 + 1, DINF), r2[x]); r1[x] = a1; r2[x] = a2; }
    a1 = a2 = DINF;
    const bool inner_row = y >= VGAP && y < VGAP + g.rh;
    for (int x = C - 1; x >= 0; --x) {
        a1 = min(min(a1 + 1, DINF), r1[x]); a2 = min(min(a2 + 1, DINF), r2[x]);
        if (inner_row && x >= VGAP && x < VGAP + g.rw) {
            const int yy = y - VGAP, xx = x - VGAP;
            if (a1 < a2) m2[(size_t)(g.ry - g.y2 + yy) * g.w2 + (g.rx - g.x2 + xx)] = 0;
            else m1[(size_t)(g.ry - g.y1 + yy) * g.w1 + (g.rx - g.x1 + xx)] = 0;
        }
    }
}

// GainCompensator::feed's overlap statistics (exposure_compensate.cpp:90-121), one thread per pair i <= j
struct GainViews { const uint8_t *img[MS_MAX_VIEWS]; const uint8_t *mask[MS_MAX_VIEWS]; ms_rect roi[MS_MAX_VIEWS]; int n; };
__global__ void __launch_bounds__(64) k_gain_pairs(GainViews V, int *__restrict__ Nm, double *__restrict__ Im)
{
    const int p = blockIdx.x * 64 + threadIdx.x, n = V.n;
    if (p >= n * n) return;
    const int i = p / n, j = p % n;
    if (j < i) return;
    const ms_rect a = V.roi[i], b = V.roi[j];
    const int x0 = max(a.x, b.x), y0 = max(a.y, b.y), x1 = min(a.x + a.width, b.x + b.width), y1 = min(a.y + a.height, b.y + b.height);
    if (!(x0 < x1 && y0 < y1)) return;                                    // (the matrices are zero-initialised)
    int cnt = 0;
    double s1 = 0, s2 = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const size_t p1 = (size_t)(y - a.y) * a.width + (x - a.x), p2 = (size_t)(y - b.y) * b.width + (x - b.x);
            if (V.mask[i][p1] != 255 || V.mask[j][p2] != 255) continue;
            ++cnt;
            const uint8_t *u = V.img[i] + 3 * p1, *w = V.img[j] + 3 * p2;
            s1 += sqrt(static_cast<double>(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
            s2 += sqrt(static_cast<double>(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
        }
    const int N = max(1, cnt);
    Nm[i * n + j] = Nm[j * n + i] = N;
    Im[i * n + j] = s1 / N;
    Im[j * n + i] = s2 / N;
}
// the normal equations (:123-139) and cv::solve(A, b, gains) with DECOMP_LU semantics (closed forms up to 3 x 3: lapack.cpp:1107-1237; LU with partial
// pivoting otherwise: matrix_decomp.cpp:52-112) in one thread; ok = 0 if the system is singular
__global__ void k_gain_solve(int n, const int *__restrict__ Nm, const double *__restrict__ Im, double *__restrict__ gains, int *__restrict__ ok)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double A[MS_MAX_VIEWS * MS_MAX_VIEWS], b[MS_MAX_VIEWS];
    const double alpha = 0.01, beta = 100;
    for (int i = 0; i < n; ++i) { b[i] = 0; for (int j = 0; j < n; ++j) A[i * n + j] = 0; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            b[i] += beta * Nm[i * n + j];
            A[i * n + i] += beta * Nm[i * n + j];
            if (j == i) continue;
            A[i * n + i] += 2 * alpha * Im[i * n + j] * Im[i * n + j] * Nm[i * n + j];
            A[i * n + j] -= 2 * alpha * Im[i * n + j] * Im[j * n + i] * Nm[i * n + j];
        }
#define AT(i, j) A[(i) * n + (j)]
    *ok = 1;
    if (n == 1) { if (AT(0, 0) == 0.) *ok = 0; else b[0] = b[0] / AT(0, 0); }
    else if (n == 2) {
        double d = AT(0, 0) * AT(1, 1) - AT(0, 1) * AT(1, 0);
        if (d == 0.) *ok = 0;
        else { d = 1. / d; const double t = (b[0] * AT(1, 1) - b[1] * AT(0, 1)) * d; b[1] = (b[1] * AT(0, 0) - b[0] * AT(1, 0)) * d; b[0] = t; }
    } else if (n == 3) {
        double d = AT(0, 0) * (AT(1, 1) * AT(2, 2) - AT(1, 2) * AT(2, 1)) - AT(0, 1) * (AT(1, 0) * AT(2, 2) - AT(1, 2) * AT(2, 0)) +
                   AT(0, 2) * (AT(1, 0) * AT(2, 1) - AT(1, 1) * AT(2, 0));
        if (d == 0.) *ok = 0;
        else {
            d = 1. / d;
            const double t0 = ((AT(1, 1) * AT(2, 2) - AT(1, 2) * AT(2, 1)) * b[0] + (AT(0, 2) * AT(2, 1) - AT(0, 1) * AT(2, 2)) * b[1] + (AT(0, 1) * AT(1, 2) - AT(0, 2) * AT(1, 1)) * b[2]) * d;
            const double t1 = ((AT(1, 2) * AT(2, 0) - AT(1, 0) * AT(2, 2)) * b[0] + (AT(0, 0) * AT(2, 2) - AT(0, 2) * AT(2, 0)) * b[1] + (AT(0, 2) * AT(1, 0) - AT(0, 0) * AT(1, 2)) * b[2]) * d;
            const double t2 = ((AT(1, 0) * AT(2, 1) - AT(1, 1) * AT(2, 0)) * b[0] + (AT(0, 1) * AT(2, 0) - AT(0, 0) * AT(2, 1)) * b[1] + (AT(0, 0) * AT(1, 1) - AT(0, 1) * AT(1, 0)) * b[2]) * d;
            b[0] = t0; b[1] = t1; b[2] = t2;
        }
    } else {
        const double eps = 2.220446049250313e-16 * 100;
        for (int i = 0; i < n && *ok; ++i) {
            int k = i;
            for (int j = i + 1; j < n; ++j) if (fabs(AT(j, i)) > fabs(AT(k, i))) k = j;
            if (fabs(AT(k, i)) < eps) { *ok = 0; break; }
            if (k != i) { for (int j = i; j < n; ++j) { const double t = AT(i, j); AT(i, j) = AT(k, j); AT(k, j) = t; } const double t = b[i]; b[i] = b[k]; b[k] = t; }
            const double d = -1 / AT(i, i);
            for (int j = i + 1; j < n; ++j) {
                const double al = AT(j, i) * d;
                for (int q = i + 1; q < n; ++q) AT(j, q) += al * AT(i, q);
                b[j] += al * b[i];
            }
        }
        for (int i = n - 1; i >= 0 && *ok; --i) {
            double sv = b[i];
            for (int q = i + 1; q < n; ++q) sv -= AT(i, q) * b[q];
            b[i] = sv / AT(i, i);
        }
    }
#undef AT
    for (int i = 0; i < n; ++i) gains[i] = *ok ? b[i] : 1.0;
}

}  // namespace

// VoronoiSeamFinder over DEVICE masks (contiguous, roi-sized), in place, pairs in the reference's order (PairwiseSeamFinder::run: i < j, overlapping rois)
int voronoi_seams_device(int n, const ms_rect *rois, uint8_t *const *masks, hipStream_t st)
{
    size_t cells = 0;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const int x0 = std::max(rois[i].x, rois[j].x), y0 = std::max(rois[i].y, rois[j].y);
            const int x1 = std::min(rois[i].x + rois[i].width, rois[j].x + rois[j].width), y1 = std::min(rois[i].y + rois[i].height, rois[j].y + rois[j].height);
            if (x0 < x1 && y0 < y1) cells = std::max(cells, (size_t)(x1 - x0 + 2 * VGAP) * (y1 - y0 + 2 * VGAP));
        }
    if (!cells) return MS_OK;
    int *d = nullptr;
    MS_HIP(hipMalloc((void **)&d, 2 * cells * sizeof(int)));
    int rc = MS_OK;
    for (int i = 0; i + 1 < n && rc == MS_OK; ++i)
        for (int j = i + 1; j < n && rc == MS_OK; ++j) {
            const int x0 = std::max(rois[i].x, rois[j].x), y0 = std::max(rois[i].y, rois[j].y);
            const int x1 = std::min(rois[i].x + rois[i].width, rois[j].x + rois[j].width), y1 = std::min(rois[i].y + rois[i].height, rois[j].y + rois[j].height);
            if (!(x0 < x1 && y0 < y1)) continue;
            const PairGeom g{x0, y0, x1 - x0, y1 - y0, rois[i].x, rois[i].y, rois[i].width, rois[i].height, rois[j].x, rois[j].y, rois[j].width, rois[j].height};
            const int C = g.rw + 2 * VGAP, R = g.rh + 2 * VGAP;
            k_vor_cols<<<div_up(C, 64), 64, 0, st>>>(masks[i], masks[j], g, d, d + cells);
            k_vor_rows<<<div_up(R, 64), 64, 0, st>>>(masks[i], masks[j], g, d, d + cells);
            if (hipGetLastError() != hipSuccess) rc = fail(MS_ERR_HIP, "voronoi_seams_device: launch failed");
        }
    if (hipStreamSynchronize(st) != hipSuccess && rc == MS_OK) rc = fail(MS_ERR_HIP, "voronoi_seams_device: sync failed");
    (void)hipFree(d);
    return rc;
}

// GainCompensator::feed over DEVICE images (8UC3, contiguous) and masks (8UC1, contiguous); gains come back to the host (n doubles)
int estimate_gains_device(int n, const ms_rect *rois, const uint8_t *const *images, const uint8_t *const *masks, double *gains_host, hipStream_t st)
{
    if (n > MS_MAX_VIEWS) return fail(MS_ERR_INVALID, "estimate_gains_device: too many views");
    GainViews V{};
    V.n = n;
    for (int i = 0; i < n; ++i) { V.img[i] = images[i]; V.mask[i] = masks[i]; V.roi[i] = rois[i]; }
    char *buf = nullptr;
    const size_t nn = (size_t)n * n, bytes = nn * sizeof(double) + nn * sizeof(int) + n * sizeof(double) + 16;
    MS_HIP(hipMalloc((void **)&buf, bytes));
    double *Im = (double *)buf, *g = Im + nn;
    int *Nm = (int *)(g + n), *ok = Nm + nn;
    int rc = MS_OK;
    if (hipMemsetAsync(buf, 0, bytes, st) != hipSuccess) rc = fail(MS_ERR_HIP, "estimate_gains_device: memset failed");
    if (rc == MS_OK) {
        k_gain_pairs<<<div_up((int)nn, 64), 64, 0, st>>>(V, Nm, Im);
        k_gain_solve<<<1, 1, 0, st>>>(n, Nm, Im, g, ok);
        int hok = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(gains_host, g, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(&hok, ok, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            rc = fail(MS_ERR_HIP, "estimate_gains_device: launch / copy failed");
        else if (!hok) rc = fail(MS_ERR_INVALID, "singular gain system");
    }
    (void)hipFree(buf);
    return rc;
}

}  // namespace ms
