// mesh_solver.hip -- the CPW mesh optimiser on the device (gfx950): MeshWarper::createMesh after feature matching
// (360_stitcher/meshwarper.cpp:279-301): local / global / smoothness / temporal terms of |A x - b|^2, then the solve the reference hands
// to Eigen::LeastSquaresConjugateGradient<SparseMatrix<double>> (third party; algorithm of Eigen 3.3 LeastSquareConjugateGradient.h:
// CG on the normal equations, Jacobi preconditioner, x0 = 0, tolerance DBL_EPSILON, at most 2*cols iterations).
//
// Split of the work:
//   device  k_tri_stats     masked sum / sum of squares of every (vertex, triangle) cell crop -- the only pass over pixels
//           k_lscg_step     tail of iteration i - 1 (beta, x += alpha p, p = z + beta p) + head of iteration i (tmp = A p, ELL, one thread per row)
//           k_lscg_cols     A^T residual, z     (CSC, 16 lanes per column)
//           all fp64; every reduction is a fixed-order tree over per-block partials (<= LSCG_PARTS blocks), re-summed by each consumer
//           block, so the solve is run-to-run reproducible and needs no in-launch hand-off between workgroups.
//           k_tri_masks     the 8 triangle masks of a cell (what cv::fillConvexPoly produces for them), closed form per row
//   host    coefficient arithmetic in the reference's float expressions, CSR/CSC layout.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <vector>
#include "common.hpp"

namespace ms {
namespace {

// ------------------------------------------------------------------------------------------------ triangle masks of a mesh cell
// meshwarper.cpp:441-486: offsets (x, y) of V1, V2 (the vertex itself), V3 of the 8 triangles around a vertex
const int TRI[8][3][2] = {
    {{-1, 0}, {0, 0}, {-1, -1}}, {{0, -1}, {0, 0}, {-1, -1}}, {{0, -1}, {0, 0}, {1, -1}}, {{1, 0}, {0, 0}, {1, -1}},
    {{-1, 0}, {0, 0}, {-1, 1}},  {{0, 1}, {0, 0}, {-1, 1}},   {{0, 1}, {0, 0}, {1, 1}},   {{1, 0}, {0, 0}, {1, 1}},
};

// meshwarper.cpp:527-551 fills `Mat mask(cell_height, cell_width)` with cv::fillConvexPoly of the triangle's three vertices in
// cell-relative pixels.  Those vertices are always three CORNERS of the W x H cell (W = (int)cell_width, H = (int)cell_height;
// the corners at x = W / y = H lie one pixel outside the mask), so each mask is one of the four halves of the cell cut by a diagonal,
// and what fillConvexPoly (drawing.cpp:1109-1271) sets in row y has a closed form -- evaluated here by one lane per row, no polygon
// walker and no line iterator:
//   * scan conversion: between the triangle's vertical leg (x = 0 or W) and its diagonal, both in 16.16 fixed point with the rounded slope
//     s = trunc((2 dX + H) / 2H), dX = +-W << 16: x_d(y) = x_top + y s; columns round(min) .. round(max) (+ 0x8000 >> 16), clipped to the mask;
//   * outline: a leg on row 0 sets the whole row, a leg on column 0 the whole column (the legs on row H / column W are outside and
//     rejected by clipLine); the diagonal is an 8-connected Bresenham line from its LEFT end between the end points clipLine (:97-148)
//     leaves inside the mask.  After k steps along its major axis the minor coordinate of that line is round-half-down(k b / a)
//     = floor((2 b k + a - 1) / 2a) (a >= b the axis extents; LineIterator's error term starts at a - 2b and steps when negative), so a
//     y-major line sets one pixel per row and an x-major line the run of k with that minor coordinate.
// The mask is the union.  oracle/mesh_oracle.py keeps the general fillConvexPoly restatement; tests compare the two bit for bit.
struct TriShape {
    int row0, col0;       // the leg on row 0 / on column 0 belongs to the triangle
    int main_diag;        // hypotenuse (0,0)-(W,H) rather than (W,0)-(0,H)
    int leg_right;        // the vertical leg is at x = W rather than x = 0
};
struct TriShapes { TriShape s[8]; };

TriShapes triangle_shapes()
{
    TriShapes out;
    for (int t = 0; t < 8; ++t) {
        int minx = 0, miny = 0;
        for (int k = 0; k < 3; ++k) { minx = std::min(minx, TRI[t][k][0]); miny = std::min(miny, TRI[t][k][1]); }
        bool corner[2][2] = {{false, false}, {false, false}};      // [x][y] in cell units after the shift into the cell
        for (int k = 0; k < 3; ++k) corner[TRI[t][k][0] - minx][TRI[t][k][1] - miny] = true;
        TriShape &S = out.s[t];
        S.row0 = corner[0][0] && corner[1][0];
        S.col0 = corner[0][0] && corner[0][1];
        S.main_diag = corner[0][0] && corner[1][1];
        // the third corner (the right angle) carries the vertical leg
        S.leg_right = S.main_diag ? !corner[0][1] : !corner[0][0];
    }
    return out;
}

__device__ __forceinline__ long long floor_div(long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }      // b > 0

// block t = triangle t; one lane per mask row; nz[t] = number of set pixels
__global__ void __launch_bounds__(256) k_tri_masks(TriShapes shapes, int W, int H, uint8_t *__restrict__ masks, unsigned *__restrict__ nz)
{
    const int t = blockIdx.x;
    const TriShape S = shapes.s[t];
    uint8_t *m = masks + (size_t)t * W * H;
    // diagonal of the scan conversion: from its top end down, rounded 16.16 slope (C division truncates toward zero)
    const long long top = (S.main_diag ? 0ll : (long long)W) << 16, bot = (S.main_diag ? (long long)W : 0ll) << 16;
    const long long num = (bot - top) * 2 + H;
    const long long slope = num >= 0 ? num / (2ll * H) : -((-num) / (2ll * H));
    const long long leg = (S.leg_right ? (long long)W : 0ll) << 16;
    // diagonal of the outline: left end (lx, ly), extents (ax, ay) >= 0, direction of y from the left end
    int lx = 0, ly = 0, ax = 0, ay = 0, ydir = 1;
    bool line = true;
    if (S.main_diag) {                    // (0,0) - (W,H): the far end is pulled back onto the last row, then (narrow cells) onto the last column
        if (W >= H) { ax = W - W / H; ay = H - 1; }
        else { ax = W - 1; ay = H - 1 - (H - 1) / W; }
    } else if (H == 1) {
        line = false;                     // (W,0) - (0,1): both ends stay right of the mask after the first clip: rejected
    } else {                              // (W,0) - (0,H): left end pulled up onto the last row, right end pulled in onto the last column
        lx = W / H; ly = H - 1;
        ax = W - 1 - lx; ay = ly - (H - 1) / (W - lx);
        ydir = -1;
    }
    unsigned count = 0;
    for (int y = (int)threadIdx.x; y < H; y += (int)blockDim.x) {
        // scan-conversion span
        const long long xd = top + (long long)y * slope;
        long long f0 = ((xd < leg ? xd : leg) + 0x8000) >> 16, f1 = ((xd < leg ? leg : xd) + 0x8000) >> 16;
        if (f1 < 0 || f0 >= W) { f0 = 1; f1 = 0; }
        else { if (f0 < 0) f0 = 0; if (f1 > W - 1) f1 = W - 1; }
        // diagonal-line span
        long long d0 = 1, d1 = 0;
        const int k = (y - ly) * ydir;
        if (line && k >= 0 && k <= ay) {
            if (ay > ax) d0 = d1 = lx + floor_div(2ll * ax * k + ay - 1, 2ll * ay);
            else if (ay == 0) { d0 = lx; d1 = lx + ax; }
            else {
                long long i0 = -floor_div(-(2ll * ax * k - ax + 1), 2ll * ay), i1 = floor_div(2ll * ax * k + ax, 2ll * ay);
                if (i0 < 0) i0 = 0;
                if (i1 > ax) i1 = ax;
                d0 = lx + i0; d1 = lx + i1;
            }
        }
        const bool full = S.row0 && y == 0;
        for (int x = 0; x < W; ++x) {
            const bool in = full || (S.col0 && x == 0) || (x >= f0 && x <= f1) || (x >= d0 && x <= d1);
            m[(size_t)y * W + x] = in ? 255 : 0;
            count += in;
        }
    }
    for (int off = 32; off; off >>= 1) count += __shfl_xor(count, off);
    if ((threadIdx.x & 63) == 0 && count) atomicAdd(&nz[t], count);
}

// moves between the per-thread pinned staging block and the per-thread device scratch (see PinnedScratch / DeviceScratch in common.hpp)
int copy_async(const void *src, void *dst, size_t bytes, hipStream_t st)
{
    if (bytes) MS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------ device: triangle statistics
struct TriJob { int x, y, t; };

// one wave per (vertex, triangle): sum and sum of squares of the three channels over the masked cell crop (exact integers)
__global__ void __launch_bounds__(64) k_tri_stats(const uint8_t *__restrict__ img, size_t step, const uint8_t *__restrict__ masks, int mw, int mh,
                                                 const TriJob *__restrict__ jobs, unsigned long long *__restrict__ out)
{
    const TriJob jb = jobs[blockIdx.x];
    const uint8_t *m = masks + (size_t)jb.t * mw * mh;
    unsigned long long s[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < mw * mh; i += 64) {
        if (!m[i]) continue;
        const int yy = i / mw, xx = i - yy * mw;
        const uint8_t *px = img + (size_t)(jb.y + yy) * step + (size_t)(jb.x + xx) * 3;
        for (int c = 0; c < 3; ++c) { const unsigned v = px[c]; s[c] += v; q[c] += v * v; }
    }
    for (int c = 0; c < 3; ++c)
        for (int off = 32; off; off >>= 1) {
            s[c] += __shfl_xor(s[c], off);
            q[c] += __shfl_xor(q[c], off);
        }
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; ++c) { out[(size_t)blockIdx.x * 6 + c] = s[c]; out[(size_t)blockIdx.x * 6 + 3 + c] = q[c]; }
}

// saliency of every in-mesh (vertex, triangle) of one view (meshwarper.cpp:497-563); NaN marks triangles that leave the mesh
int view_saliency(const ms_image &im, int M, int N, std::vector<float> &sal, hipStream_t st)
{
    const float width = (float)im.cols, height = (float)im.rows;
    const float cw = width / (M - 1), ch = height / (N - 1);
    const int mw = (int)cw, mh = (int)ch;                                               // Mat mask(cell_height, cell_width): float -> int truncation
    MS_CHECK(mw >= 1 && mh >= 1, "ms_create_mesh: a %dx%d mesh on a %dx%d view has empty cells", M, N, im.cols, im.rows);
    const size_t mask_bytes = (size_t)8 * mw * mh;
    std::vector<TriJob> jobs;
    std::vector<int> slot((size_t)N * M * 8, -1);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j)
            for (int t = 0; t < 8; ++t) {
                bool in = true;
                float vx[3], vy[3];
                for (int k = 0; k < 3; ++k) {
                    const int x = j + TRI[t][k][0], y = i + TRI[t][k][1];
                    in &= x >= 0 && y >= 0 && x < M && y < N;
                    vx[k] = x * cw; vy[k] = y * ch;
                }
                if (!in) continue;
                TriJob jb{(int)std::min(std::min(vx[0], vx[1]), vx[2]), (int)std::min(std::min(vy[0], vy[1]), vy[2]), t};
                MS_CHECK(jb.x >= 0 && jb.y >= 0 && jb.x + mw <= im.cols && jb.y + mh <= im.rows, "ms_create_mesh: cell crop leaves the view (cv::Mat::operator() asserts)");
                slot[((size_t)i * M + j) * 8 + t] = (int)jobs.size();
                jobs.push_back(jb);
            }
    // device block: masks (built on the device) | jobs (up) | sums, mask pixel counts (down)
    const size_t off_jobs = (mask_bytes + 15) & ~(size_t)15, off_sums = off_jobs + ((jobs.size() * sizeof(TriJob) + 15) & ~(size_t)15);
    const size_t off_nz = off_sums + ((jobs.size() * 6 * sizeof(unsigned long long) + 15) & ~(size_t)15), total = off_nz + 64;
    uint8_t *host = (uint8_t *)pinned_scratch().get(total);                             // (same offsets as the device block; the mask area stays unused)
    if (!host) return fail(MS_ERR_NOMEM, "ms_create_mesh: cannot allocate %zu bytes of pinned staging memory", total);
    memcpy(host + off_jobs, jobs.data(), jobs.size() * sizeof(TriJob));
    uint8_t *dev = (uint8_t *)device_scratch().get(total);
    if (!dev) return fail(MS_ERR_NOMEM, "ms_create_mesh: cannot allocate %zu bytes of device scratch", total);
    if (int e = copy_async(host + off_jobs, dev + off_jobs, off_sums - off_jobs, st)) return e;
    MS_HIP(hipMemsetAsync(dev + off_nz, 0, 64, st));
    k_tri_masks<<<8, 256, 0, st>>>(triangle_shapes(), mw, mh, dev, (unsigned *)(dev + off_nz));
    MS_LAUNCH_CHECK();
    k_tri_stats<<<(unsigned)jobs.size(), 64, 0, st>>>((const uint8_t *)im.data, im.step, dev, mw, mh, (const TriJob *)(dev + off_jobs), (unsigned long long *)(dev + off_sums));
    MS_LAUNCH_CHECK();
    if (int e = copy_async(dev + off_sums, host + off_sums, total - off_sums, st)) return e;
    MS_HIP(hipStreamSynchronize(st));
    const unsigned *nz = (const unsigned *)(host + off_nz);
    const unsigned long long *sums = (const unsigned long long *)(host + off_sums);
    sal.assign((size_t)N * M * 8, NAN);
    for (size_t k = 0; k < slot.size(); ++k) {
        if (slot[k] < 0) continue;
        const unsigned long long *o = sums + (size_t)slot[k] * 6;
        const int cnt = nz[jobs[slot[k]].t];
        const double scale = cnt ? 1. / cnt : 0.;                                      // cv::meanStdDev, stat.cpp:1939-1943
        double nrm = 0;
        for (int c = 0; c < 3; ++c) {
            const double mean = (double)o[c] * scale;
            const double dev = std::sqrt(std::max((double)o[3 + c] * scale - mean * mean, 0.));
            const double var = dev * dev;                                             // cv::pow(deviation, 2)
            nrm += var * var;
        }
        sal[k] = (float)std::sqrt(std::sqrt(nrm) + 0.5f);                               // sqrt(norm(variance, NORM_L2) + 0.5f)
    }
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------ host: the linear system
struct Entry { int row, col; double val; };
struct LinSys {
    int M, N, cols, rows = 0;
    std::vector<Entry> e;
    std::vector<double> b;
    void put(int row_off, int col, float v) { e.push_back({rows + row_off, col, (double)v}); }
    void end_rows(float b0, float b1) { b.push_back(b0); b.push_back(b1); rows += 2; }
};

float theta_of(int rule, int src, int dst, int n, bool wrap)
{
    float theta;
    if (rule == 0) {                            // the reference's 6-camera rig, meshwarper.cpp:617-629
        theta = (float)(dst - src);
        if (src == 0 && dst == n - 1 && wrap) theta = -1;
        if (src == 3) theta = 4.25f;
        if (src == 4) theta = -0.25f;
        theta *= 2 * 3.1415926535897932384626 / 6;
    } else {                                    // evenly spaced ring of n cameras
        int d = dst - src;
        if (d > n / 2.0) d -= n;
        if (d < -n / 2.0) d += n;
        theta = (float)d;
        theta *= 2 * 3.1415926535897932384626 / n;
    }
    return theta;
}

struct Cell { int t, l; float u, v; };
Cell locate(float x, float y, float w, float h, int M, int N)       // meshwarper.cpp:646-668
{
    Cell c;
    c.t = (int)std::floor(y * (N - 1) / h);
    c.l = (int)std::floor(x * (M - 1) / w);
    const float top = c.t * h / (N - 1), bot = top + h / (N - 1);
    const float left = c.l * w / (M - 1), right = left + w / (M - 1);
    c.u = (x - left) / (right - left);
    c.v = (y - top) / (bot - top);
    return c;
}

void put_bilinear(LinSys &S, int k, int base, const Cell &c, float a, bool negate)
{
    const int M = S.M;
    const float u = c.u, v = c.v;
    if (!negate) {
        S.put(k, 2 * (c.l + M * c.t + base) + k, (1 - u) * (1 - v) * a);
        S.put(k, 2 * (c.l + 1 + M * c.t + base) + k, u * (1 - v) * a);
        S.put(k, 2 * (c.l + M * (c.t + 1) + base) + k, v * (1 - u) * a);
        S.put(k, 2 * (c.l + 1 + M * (c.t + 1) + base) + k, u * v * a);
    } else {
        S.put(k, 2 * (c.l + M * c.t + base) + k, -(1 - u) * (1 - v) * a);
        S.put(k, 2 * (c.l + 1 + M * c.t + base) + k, -u * (1 - v) * a);
        S.put(k, 2 * (c.l + M * (c.t + 1) + base) + k, -v * (1 - u) * a);
        S.put(k, 2 * (c.l + 1 + M * (c.t + 1) + base) + k, -u * v * a);
    }
}

// calcLocalTerm, meshwarper.cpp:596-709
void local_term(LinSys &S, const ms_mesh_match *m, int count, const ms_image *views, int n, int idx, const ms_mesh_params &P)
{
    const int M = S.M, N = S.N;
    const float f = P.focal_length, a = std::sqrt(P.alphas[0]);
    const float scale = (float)(P.compose_scale / P.work_scale);
    for (int k = 0; k < count; ++k) {
        const int dst = m[k].dst;
        if (dst < 0 || dst >= n) continue;
        const float w1 = (float)views[idx].cols, h1 = (float)views[idx].rows, w2 = (float)views[dst].cols, h2 = (float)views[dst].rows;
        const float x1 = m[k].x1, y1 = m[k].y1, x2 = m[k].x2, y2 = m[k].y2;
        if (x1 < 0 || x2 < 0 || y1 < 0 || y2 < 0 || x1 >= w1 || x2 >= w2 || y1 >= h1 || y2 >= h2) continue;
        if (!(std::isfinite(x1) && std::isfinite(y1) && std::isfinite(x2) && std::isfinite(y2))) continue;      // NaN passes the reference's range test (undefined there)
        const Cell c1 = locate(x1, y1, w1, h1, M, N), c2 = locate(x2, y2, w2, h2, M, N);
        // float rounding can put a point on the last mesh line; the reference would index past the mesh row there (undefined) -- skipped
        if (c1.l + 1 >= M || c2.l + 1 >= M || c1.t + 1 >= N || c2.t + 1 >= N) continue;
        const float theta = theta_of(P.theta_rule, idx, dst, n, P.wrap_around != 0);
        for (int r = 0; r < 2; ++r) {
            put_bilinear(S, r, M * N * idx, c1, a, false);
            put_bilinear(S, r, M * N * dst, c2, a, true);
        }
        S.end_rows(theta * f * scale * a, 0.f);
    }
}

// calcTemporalLocalTerm, meshwarper.cpp:711-786 ((x2, y2) = the same feature in the previous calibration)
void temporal_term(LinSys &S, const ms_mesh_match *m, int count, const ms_image &view, int idx, const ms_mesh_params &P)
{
    const int M = S.M, N = S.N;
    const float a = std::sqrt(P.alphas[3]);
    const float w = (float)view.cols, h = (float)view.rows;
    for (int k = 0; k < count; ++k) {
        const float x1 = m[k].x1, y1 = m[k].y1, x2 = m[k].x2, y2 = m[k].y2;
        if (x1 < 0 || x2 < 0 || y1 < 0 || y2 < 0 || x1 >= w || x2 >= w || y1 >= h || y2 >= h) continue;
        if (!(std::isfinite(x1) && std::isfinite(y1) && std::isfinite(x2) && std::isfinite(y2))) continue;
        const Cell c = locate(x1, y1, w, h, M, N);
        if (c.l + 1 >= M || c.t + 1 >= N) continue;
        for (int r = 0; r < 2; ++r) put_bilinear(S, r, M * N * idx, c, a, false);
        S.end_rows(x2 * a, y2 * a);
    }
}

// calcGlobalTerm, meshwarper.cpp:389-418; feature points are Point(keypoint.pt) = cvRound of the match positions (meshwarper.cpp:190-196)
void global_term(LinSys &S, const ms_mesh_match *m, int count, const ms_image &view, int idx, const ms_mesh_params &P)
{
    const int M = S.M, N = S.N;
    const float a = std::sqrt(P.alphas[1]);
    std::vector<int> px(count), py(count);
    for (int k = 0; k < count; ++k) {
        const bool ok = std::isfinite(m[k].x1) && std::isfinite(m[k].y1) && std::fabs(m[k].x1) < 1e9f && std::fabs(m[k].y1) < 1e9f;
        px[k] = ok ? (int)std::nearbyint((double)m[k].x1) : INT_MIN / 2;        // (a point that far away never zeroes a tau)
        py[k] = ok ? (int)std::nearbyint((double)m[k].y1) : INT_MIN / 2;
    }
    int col = N * M * 2 * idx;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j) {
            const float x1 = (float)(j * view.cols / (M - 1)), y1 = (float)(i * view.rows / (N - 1));
            float tau = 1;
            for (int k = 0; k < count; ++k) {
                const double dx = px[k] - x1, dy = py[k] - y1;
                if (std::sqrt(dx * dx + dy * dy) < P.global_dist) { tau = 0; break; }
            }
            S.put(0, col, a * tau);
            S.put(1, col + 1, a * tau);
            S.end_rows(a * tau * x1, a * tau * y1);
            col += 2;
        }
}

// calcSmoothnessTerm, meshwarper.cpp:421-593 (the x row and the y row of a triangle carry the same six coefficients, as in the reference)
void smoothness_term(LinSys &S, const std::vector<float> &sal, const ms_image &view, int idx, const ms_mesh_params &P)
{
    const int M = S.M, N = S.N;
    const float a = std::sqrt(P.alphas[2]);
    const float width = (float)view.cols, height = (float)view.rows;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < M; ++j)
            for (int t = 0; t < 8; ++t) {
                const float s = sal[((size_t)i * M + j) * 8 + t];
                if (s != s) continue;                                    // triangle leaves the mesh
                int X[3], Y[3];
                for (int k = 0; k < 3; ++k) { X[k] = j + TRI[t][k][0]; Y[k] = i + TRI[t][k][1]; }
                const float V1x = X[0] * (width / (M - 1)), V2x = X[1] * (width / (M - 1)), V3x = X[2] * (width / (M - 1));
                const float V1y = Y[0] * (height / (N - 1)), V2y = Y[1] * (height / (N - 1)), V3y = Y[2] * (height / (N - 1));
                const float u = (-V1x * V2y + V1x * V3y - V2x * V1y + 2 * V2x * V2y - V2x * V3y + V3x * V1y - V3x * V2y) / (2 * (V2x - V3x) * (V2y - V3y));
                const float v = (V1x * V2y - V1x * V3y - V2x * V1y + V2x * V3y + V3x * V1y - V3x * V2y) / (2 * (V2x - V3x) * (V2y - V3y));
                const int c0 = 2 * (X[0] + M * Y[0] + M * N * idx), c1 = 2 * (X[1] + M * Y[1] + M * N * idx), c2 = 2 * (X[2] + M * Y[2] + M * N * idx);
                for (int r = 0; r < 2; ++r) {
                    S.put(r, c0, a * s);
                    S.put(r, c0 + 1, a * s);
                    S.put(r, c1, a * (u - v - 1) * s);
                    S.put(r, c1 + 1, a * (u + v - 1) * s);
                    S.put(r, c2, a * (-u + v) * s);
                    S.put(r, c2 + 1, a * (-u - v) * s);
                }
                S.end_rows(0.f, 0.f);
            }
}

// ------------------------------------------------------------------------------------------------ device: least-squares CG, fp64
constexpr int LSCG_PARTS = 512;                 // at most this many workgroups produce partial sums (2 x the CU count)
constexpr int ELL_W = 8;                        // no row of the system has more than 8 coefficients
constexpr int LSCG_BLOCK = 64;                  // iterations between two looks at the convergence flag

struct LscgState {                              // written only by workgroup 0 of the kernel named, read by later kernels
    double alpha;                               // k_lscg_cols
    double abs_new[2];                          // k_lscg_step, by iteration parity (the other slot is abs_old)
    double rhs_norm2, res_norm2, threshold;     // k_lscg_step
    int pending;                                // k_lscg_step: -1 running, >= 0 converged in that iteration, -2 zero right-hand side
    int done;                                   // k_lscg_cols copies `pending` here: the flag k_lscg_step itself may read while it writes `pending`
    int iter_step, iter_cols;                   // the iteration each kernel is at: k_lscg_cols advances iter_step, k_lscg_step hands its value to iter_cols
};                                              // (kept on the device: the launches of every iteration are identical)

__device__ double block_sum_impl(double v, double *lds)     // fixed-order tree over the 256 threads
{
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int w = 128; w; w >>= 1) {
        if (t < w) lds[t] += lds[t + w];
        __syncthreads();
    }
    const double r = lds[0];
    __syncthreads();
    return r;
}

// fixed-order sum of <= LSCG_PARTS partials; every thread of every workgroup gets the same value
__device__ double sum_parts(const double *__restrict__ part, int stride, int n, double *lds)
{
    const int t = threadIdx.x;                  // 256 threads
    double v = 0;
    for (int k = t; k < LSCG_PARTS; k += 256) v += k < n ? part[(size_t)k * stride] : 0.0;
    return block_sum_impl(v, lds);
}

__device__ double block_sum(double v, double *lds) { return block_sum_impl(v, lds); }

// One launch = the tail of iteration iter - 1 and the head of iteration iter (two launches per iteration instead of three):
//   tail: residualNorm2 = sum nr.nr -> converged?  beta = abs_new / abs_old;  x += alpha p;  p' = z + beta p   (p' into the other p buffer)
//   head: residual -= alpha tmp (deferred);  tmp = A p'  with p' formed on the fly from z and p (the same expression, so the same bits);
//         part1[block] = sum tmp^2
// The iteration index lives in the state: 0 runs the prologue's tail (p' = z, rhsNorm2, threshold), max_it a tail only, beyond that nothing.
__global__ void __launch_bounds__(256) k_lscg_step(int R, int n, int max_it, int nparts3, double tol, const int *__restrict__ ecol, const double *__restrict__ eval,
                                                   const double *__restrict__ z, double *__restrict__ p0, double *__restrict__ p1, double *__restrict__ x,
                                                   double *__restrict__ residual, double *__restrict__ tmp, LscgState *__restrict__ S,
                                                   const double *__restrict__ part3, double *__restrict__ part1)
{
    __shared__ double lds[256];
    const int iter = S->iter_step;
    if (blockIdx.x == 0 && threadIdx.x == 0) S->iter_cols = iter;
    if (S->done != -1 || iter > max_it) return;
    const bool final = iter == max_it;                      // after the last iteration: its tail only
    const double *__restrict__ p_in = (iter & 1) ? p1 : p0;
    double *__restrict__ p_out = (iter & 1) ? p0 : p1;
    const int u = iter - 1;                                 // the iteration whose tail this is; -1 = prologue
    const bool init = iter == 0;
    const double res_norm2 = sum_parts(part3, 2, nparts3, lds), abs_new = sum_parts(part3 + 1, 2, nparts3, lds);
    const double alpha = S->alpha;                          // of iteration u (0 in the prologue)
    double threshold, beta;
    bool converged;
    if (init) {
        threshold = tol * tol * res_norm2;                  // rhsNorm2 == residualNorm2: x0 = 0
        converged = res_norm2 == 0 || res_norm2 < threshold;
        beta = 0;
    } else {
        threshold = S->threshold;
        converged = res_norm2 < threshold;
        beta = abs_new / S->abs_new[(u + 1) & 1];
    }
    const int gid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
    for (int j = gid; j < n; j += nthreads) {
        const double pj = p_in[j];
        x[j] = x[j] + alpha * pj;
        if (!converged) p_out[j] = z[j] + beta * pj;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S->abs_new[u & 1] = abs_new;
        S->res_norm2 = res_norm2;
        if (init) { S->rhs_norm2 = res_norm2; S->threshold = threshold; }
        if (converged) S->pending = init ? (res_norm2 == 0 ? -2 : 0) : u;
    }
    if (converged || final) return;
    double acc = 0;
    for (int r = gid; r < R; r += nthreads) {
        residual[r] = residual[r] - alpha * tmp[r];
        double t = 0;
        for (int k = 0; k < ELL_W; ++k) {
            const int c = ecol[(size_t)k * R + r];
            t += eval[(size_t)k * R + r] * (z[c] + beta * p_in[c]);
        }
        tmp[r] = t;
        acc += t * t;
    }
    const double sacc = block_sum(acc, lds);
    if (threadIdx.x == 0) part1[blockIdx.x] = sacc;
}

// alpha = abs_new / sum(part1); normal residual nr = A^T (residual - alpha tmp); z = invdiag nr; partials of nr.nr and nr.z
// init = 1: the prologue (alpha = 0, tmp = 0): nr = A^T b
__global__ void __launch_bounds__(256) k_lscg_cols(int n, int max_it, int init, int nparts1, const int *__restrict__ cptr, const int *__restrict__ crow,
                                                   const double *__restrict__ cval, const double *__restrict__ invdiag, const double *__restrict__ residual,
                                                   const double *__restrict__ tmp, double *__restrict__ z, LscgState *__restrict__ S,
                                                   const double *__restrict__ part1, double *__restrict__ part3)
{
    __shared__ double lds[256];
    const int pend = S->pending, iter = init ? -1 : S->iter_cols;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S->done = pend;                                       // the flag k_lscg_step reads (it writes `pending` itself)
        if (!init) S->iter_step = iter + 1;
    }
    if (pend != -1 || iter >= max_it) return;
    double alpha = 0;
    if (!init) alpha = S->abs_new[(iter + 1) & 1] / sum_parts(part1, 1, nparts1, lds);
    if (blockIdx.x == 0 && threadIdx.x == 0) S->alpha = alpha;
    const int lane = threadIdx.x & 15;
    double a_nn = 0, a_nz = 0;
    for (int j = (blockIdx.x * 256 + threadIdx.x) >> 4; j < n; j += gridDim.x * 16) {      // the 16 lanes of a group share j
        double acc = 0;
        for (int k = cptr[j] + lane; k < cptr[j + 1]; k += 16) {
            const int r = crow[k];
            acc += cval[k] * (residual[r] - alpha * tmp[r]);
        }
        for (int off = 8; off; off >>= 1) acc += __shfl_xor(acc, off, 16);
        if (lane == 0) {
            const double zj = invdiag[j] * acc;
            z[j] = zj;
            a_nn += acc * acc;
            a_nz += acc * zj;
        }
    }
    const double snn = block_sum(a_nn, lds), snz = block_sum(a_nz, lds);
    if (threadIdx.x == 0) { part3[blockIdx.x * 2] = snn; part3[blockIdx.x * 2 + 1] = snz; }
}

int solve_lscg(const LinSys &S, int max_iterations, double tolerance, std::vector<double> &x, ms_mesh_info *info, hipStream_t st)
{
    const int R = S.rows, n = S.cols;
    const size_t nnz = S.e.size();
    // Everything the device needs is laid out in ONE pinned block (PinnedScratch) and moved with one copy into the thread's device scratch:
    //   doubles: eval[8R] | cval[nnz] | invdiag[n] | b[R] | state      ints: ecol[8R] | crow[nnz] | cptr[n+1]      then the landing zone: state | x[n]
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_eval = 0, o_cval = o_eval + al((size_t)ELL_W * R * 8), o_inv = o_cval + al(nnz * 8), o_b = o_inv + al((size_t)n * 8), o_state = o_b + al((size_t)R * 8);
    const size_t o_ecol = o_state + al(sizeof(LscgState)), o_crow = o_ecol + al((size_t)ELL_W * R * 4), o_cptr = o_crow + al(nnz * 4), up_bytes = o_cptr + al((size_t)(n + 1) * 4);
    const size_t o_hstate = up_bytes, o_hx = o_hstate + al(sizeof(LscgState)), host_bytes = o_hx + al((size_t)n * 8);
    uint8_t *host = (uint8_t *)pinned_scratch().get(host_bytes);
    if (!host) return fail(MS_ERR_NOMEM, "ms_create_mesh: cannot allocate %zu bytes of pinned staging memory", host_bytes);
    memset(host, 0, up_bytes);
    double *eval = (double *)(host + o_eval), *cval = (double *)(host + o_cval), *invdiag = (double *)(host + o_inv);
    int *ecol = (int *)(host + o_ecol), *crow = (int *)(host + o_crow), *cptr = (int *)(host + o_cptr);
    // rows: ELL (column-major slabs, padded with 0 * p[0]); entries of a row ordered by column as a column-major SpMV visits them
    std::vector<int> fill(R, 0);
    std::vector<Entry> byrow(S.e);
    std::stable_sort(byrow.begin(), byrow.end(), [](const Entry &a, const Entry &b) { return a.row != b.row ? a.row < b.row : a.col < b.col; });
    for (const Entry &e : byrow) {
        MS_CHECK(e.col >= 0 && e.col < n && fill[e.row] < ELL_W, "ms_create_mesh: malformed system row %d", e.row);
        ecol[(size_t)fill[e.row] * R + e.row] = e.col;
        eval[(size_t)fill[e.row] * R + e.row] = e.val;
        fill[e.row]++;
    }
    // columns: CSC with rows ascending; Jacobi preconditioner 1 / ||A_col||^2 (1 for an empty column)
    for (const Entry &e : byrow) cptr[e.col + 1]++;
    for (int j = 0; j < n; ++j) cptr[j + 1] += cptr[j];
    {
        std::vector<int> pos(cptr, cptr + n);
        for (const Entry &e : byrow) { crow[pos[e.col]] = e.row; cval[pos[e.col]] = e.val; pos[e.col]++; }
    }
    for (int j = 0; j < n; ++j) {
        double sq = 0;
        for (int k = cptr[j]; k < cptr[j + 1]; ++k) sq += cval[k] * cval[k];
        invdiag[j] = sq > 0 ? 1.0 / sq : 1.0;
    }
    memcpy(host + o_b, S.b.data(), (size_t)R * 8);                       // residual = b - A 0
    LscgState *init_state = (LscgState *)(host + o_state);
    init_state->pending = init_state->done = -1;
    // device: the uploaded block, then the work vectors tmp[R] | x[n] | p[2][n] (zero) | z[n] | part1 | part3
    const size_t w_tmp = 0, w_x = w_tmp + al((size_t)R * 8), w_p = w_x + al((size_t)n * 8), w_z = w_p + 2 * al((size_t)n * 8), w_p1 = w_z + al((size_t)n * 8),
                 w_p3 = w_p1 + al(LSCG_PARTS * 8), work_bytes = w_p3 + al(LSCG_PARTS * 16);
    uint8_t *U = (uint8_t *)device_scratch().get(up_bytes + work_bytes);
    if (!U) return fail(MS_ERR_NOMEM, "ms_create_mesh: cannot allocate %zu bytes of device scratch", up_bytes + work_bytes);
    uint8_t *W = U + up_bytes;
    if (int e = copy_async(host, U, up_bytes, st)) return e;
    MS_HIP(hipMemsetAsync(W, 0, w_z, st));
    const int *d_ecol = (const int *)(U + o_ecol), *d_crow = (const int *)(U + o_crow), *d_cptr = (const int *)(U + o_cptr);
    const double *d_eval = (const double *)(U + o_eval), *d_cval = (const double *)(U + o_cval), *d_inv = (const double *)(U + o_inv);
    double *d_res = (double *)(U + o_b), *d_tmp = (double *)(W + w_tmp), *d_x = (double *)(W + w_x), *d_z = (double *)(W + w_z);
    double *d_pb[2] = {(double *)(W + w_p), (double *)(W + w_p + al((size_t)n * 8))};
    double *d_p1 = (double *)(W + w_p1), *d_p3 = (double *)(W + w_p3);
    LscgState *ds = (LscgState *)(U + o_state);

    const double tol = tolerance > 0 ? tolerance : DBL_EPSILON;
    const int max_it = max_iterations > 0 ? max_iterations : 2 * n;
    const int g_step = std::min(LSCG_PARTS, div_up(std::max(R, n), 256)), g_cols = std::min(LSCG_PARTS, div_up(n * 16, 256));
    LscgState *hs = (LscgState *)(host + o_hstate);
    // prologue: normal residual of x0 = 0; its tail (rhsNorm2, threshold, p = z) is the first k_lscg_step
    k_lscg_cols<<<g_cols, 256, 0, st>>>(n, max_it, 1, 0, d_cptr, d_crow, d_cval, d_inv, d_res, d_tmp, d_z, ds, d_p1, d_p3);
    MS_LAUNCH_CHECK();
    auto launch_block = [&](hipStream_t s2) {              // LSCG_BLOCK iterations; past the end (or after convergence) the launches return at once
        for (int k = 0; k < LSCG_BLOCK; ++k) {
            k_lscg_step<<<g_step, 256, 0, s2>>>(R, n, max_it, g_cols, tol, d_ecol, d_eval, d_z, d_pb[0], d_pb[1], d_x, d_res, d_tmp, ds, d_p3, d_p1);
            k_lscg_cols<<<g_cols, 256, 0, s2>>>(n, max_it, 0, g_step, d_cptr, d_crow, d_cval, d_inv, d_res, d_tmp, d_z, ds, d_p1, d_p3);
        }
    };
    // (Replaying one block as a captured HIP graph was measured: no faster than the direct launches -- the cost is the dependency between
    // consecutive kernels, not their submission.)
    int rc = MS_OK;
    for (int done_its = 0; done_its <= max_it; done_its += LSCG_BLOCK) {      // <= : the tail of the last iteration is one more step
        launch_block(st);
        if (copy_async(ds, hs, al(sizeof(LscgState)), st) != MS_OK || hipStreamSynchronize(st) != hipSuccess) { rc = fail(MS_ERR_HIP, "ms_create_mesh: solver synchronisation failed"); break; }
        if (hs->pending != -1) break;
    }
    if (rc != MS_OK) return rc;
    MS_LAUNCH_CHECK();
    if (int e = copy_async(ds, hs, al(sizeof(LscgState)), st)) return e;
    if (int e = copy_async(d_x, host + o_hx, al((size_t)n * 8), st)) return e;
    MS_HIP(hipStreamSynchronize(st));
    x.assign((const double *)(host + o_hx), (const double *)(host + o_hx) + n);
    if (info) {
        info->rows = R; info->cols = n; info->nnz = (int)nnz;
        info->iterations = hs->pending >= 0 ? hs->pending : hs->pending == -2 ? 0 : max_it;
        info->error = hs->rhs_norm2 > 0 ? std::sqrt(hs->res_norm2 / hs->rhs_norm2) : 0.0;
    }
    return MS_OK;
}

int check_params(const ms_mesh_params *P)
{
    MS_CHECK(P != nullptr, "ms_create_mesh: null parameters");
    MS_CHECK(P->mesh_cols >= 2 && P->mesh_rows >= 2 && P->mesh_cols <= 512 && P->mesh_rows <= 512, "ms_create_mesh: mesh %dx%d out of range", P->mesh_cols, P->mesh_rows);
    for (int k = 0; k < 4; ++k) MS_CHECK(P->alphas[k] >= 0.f, "ms_create_mesh: negative term weight");
    MS_CHECK(P->work_scale != 0.0, "ms_create_mesh: work_scale is zero");
    MS_CHECK(P->theta_rule == 0 || P->theta_rule == 1, "ms_create_mesh: theta_rule must be 0 or 1");
    return MS_OK;
}

}  // namespace
}  // namespace ms

using namespace ms;

extern "C" {

int ms_mesh_default_params(ms_mesh_params *P)
{
    if (!P) return fail(MS_ERR_INVALID, "ms_mesh_default_params: null");
    *P = ms_mesh_params{};
    P->mesh_cols = 10; P->mesh_rows = 10;                                   // defs.h:65-66
    P->alphas[0] = 1.0f; P->alphas[1] = 0.01f; P->alphas[2] = 0.00005f; P->alphas[3] = 0.0f;   // defs.h:69
    P->global_dist = 30;                                                    // defs.h:71
    P->focal_length = 1.f; P->compose_scale = 1.0; P->work_scale = 1.0;
    P->wrap_around = 1;                                                     // defs.h:25
    P->theta_rule = 0; P->max_iterations = 0; P->tolerance = 0.0;
    return MS_OK;
}

int ms_mesh_saliency(const ms_image *view, int mesh_cols, int mesh_rows, float *sal_host, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(view && sal_host && view->data && view->type == MS_8UC3, "ms_mesh_saliency: needs an 8UC3 device image and a host output");
    MS_CHECK(mesh_cols >= 2 && mesh_rows >= 2, "ms_mesh_saliency: mesh %dx%d out of range", mesh_cols, mesh_rows);
    std::vector<float> sal;
    if (int e = view_saliency(*view, mesh_cols, mesh_rows, sal, as_stream(stream))) return e;
    memcpy(sal_host, sal.data(), sal.size() * sizeof(float));
    return MS_OK;
}

int ms_mesh_triangle_masks(int cell_w, int cell_h, uint8_t *masks_host, unsigned *counts_host, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(cell_w >= 1 && cell_h >= 1 && cell_w <= 8192 && cell_h <= 8192 && masks_host, "ms_mesh_triangle_masks: cell %dx%d out of range", cell_w, cell_h);
    const size_t mask_bytes = (size_t)8 * cell_w * cell_h, off_nz = (mask_bytes + 15) & ~(size_t)15;
    uint8_t *dev = (uint8_t *)device_scratch().get(off_nz + 64);
    if (!dev) return fail(MS_ERR_NOMEM, "ms_mesh_triangle_masks: cannot allocate %zu bytes of device scratch", off_nz + 64);
    hipStream_t st = as_stream(stream);
    MS_HIP(hipMemsetAsync(dev + off_nz, 0, 64, st));
    k_tri_masks<<<8, 256, 0, st>>>(triangle_shapes(), cell_w, cell_h, dev, (unsigned *)(dev + off_nz));
    MS_LAUNCH_CHECK();
    MS_HIP(hipMemcpyAsync(masks_host, dev, mask_bytes, hipMemcpyDeviceToHost, st));
    unsigned nz[8];
    MS_HIP(hipMemcpyAsync(nz, dev + off_nz, sizeof(nz), hipMemcpyDeviceToHost, st));
    MS_HIP(hipStreamSynchronize(st));
    if (counts_host) memcpy(counts_host, nz, sizeof(nz));
    return MS_OK;
}

int ms_create_mesh(int n_views, const ms_image *views, const ms_mesh_match *matches, const int *match_count, const ms_mesh_match *temporal,
                   const int *temporal_count, const ms_mesh_params *P, float *mesh_x, float *mesh_y, ms_mesh_info *info, ms_stream stream)
{
    if (int e = require_device()) return e;
    if (int e = check_params(P)) return e;
    MS_CHECK(n_views >= 1 && views && match_count && mesh_x && mesh_y, "ms_create_mesh: null argument");
    hipStream_t st = as_stream(stream);
    const int M = P->mesh_cols, N = P->mesh_rows;
    LinSys S;
    S.M = M; S.N = N; S.cols = 2 * N * M * n_views;
    const bool use_temporal = P->alphas[3] != 0.0f && temporal && temporal_count;     // defs.h:70
    int moff = 0, toff = 0;
    std::vector<float> sal;
    for (int idx = 0; idx < n_views; ++idx) {
        MS_CHECK(views[idx].data && views[idx].type == MS_8UC3 && views[idx].cols >= M && views[idx].rows >= N, "ms_create_mesh: view %d must be an 8UC3 device image", idx);
        MS_CHECK(match_count[idx] >= 0 && (match_count[idx] == 0 || matches), "ms_create_mesh: bad match list of view %d", idx);
        local_term(S, matches + moff, match_count[idx], views, n_views, idx, *P);
        global_term(S, matches + moff, match_count[idx], views[idx], idx, *P);
        if (int e = view_saliency(views[idx], M, N, sal, st)) return e;
        smoothness_term(S, sal, views[idx], idx, *P);
        if (use_temporal) { temporal_term(S, temporal + toff, temporal_count[idx], views[idx], idx, *P); toff += temporal_count[idx]; }
        moff += match_count[idx];
    }
    std::vector<double> x;
    if (int e = solve_lscg(S, P->max_iterations, P->tolerance, x, info, st)) return e;
    for (int idx = 0; idx < n_views; ++idx)                                  // convertVectorToMesh, meshwarper.cpp:810-818
        for (int k = 0; k < N * M; ++k) {
            mesh_x[(size_t)idx * N * M + k] = (float)x[2 * ((size_t)k + (size_t)idx * M * N)];
            mesh_y[(size_t)idx * N * M + k] = (float)x[2 * ((size_t)k + (size_t)idx * M * N) + 1];
        }
    return MS_OK;
}

}  // extern "C"
