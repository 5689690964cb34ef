// prims.hip -- one HIP kernel per cv::cuda:: image op on the stitching hot path, on GpuMat-layout
// (row-pitched, interleaved) images.  These are the drop-in replacements of the reference's
// individual calls (include/ms_stitch.h section 1); the per-frame fast path is compositor.hip.
// Written for gfx950 wave64: 64 lanes along x (coalesced rows), 4 rows per 256-thread workgroup.
#include <algorithm>
#include "common.hpp"
#include "launchers.hpp"

namespace ms {

static constexpr int BX = 64, BY = 4;
static inline dim3 grid2d(int cols, int rows) { return dim3(div_up(cols, BX), div_up(rows, BY)); }
#define XY_GUARD(cols, rows)                                   \
    const int x = blockIdx.x * BX + threadIdx.x;               \
    const int y = blockIdx.y * BY + threadIdx.y;               \
    if (x >= (cols) || y >= (rows)) return;

// ------------------------------------------------------------------------------------------------
// remap, INTER_LINEAR, BORDER_CONSTANT(0): 4 taps in the order (y1,x1),(y1,x2),(y2,x1),(y2,x2),
// weights (x2-x)(y2-y)..., out = fma(tap, w, out)   [reference: remap.cu:56-68, filters.hpp:90-114]
template <int CN>
__device__ __forceinline__ void bilinear_taps(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols,
                                              float xc, float yc, float out[CN])
{
    const int x1 = f2i_rd(xc), y1 = f2i_rd(yc);
    const int x2 = (int)((unsigned)x1 + 1u), y2 = (int)((unsigned)y1 + 1u);
    const float wx2 = (float)x2 - xc, wx1 = xc - (float)x1;
    const float wy2 = (float)y2 - yc, wy1 = yc - (float)y1;
    const float w[4] = {wx2 * wy2, wx1 * wy2, wx2 * wy1, wx1 * wy1};
    const int xs[4] = {x1, x2, x1, x2};
    const int ys[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int c = 0; c < CN; ++c) out[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool inb = xs[t] >= 0 && xs[t] < scols && ys[t] >= 0 && ys[t] < srows;
        const uint8_t *p = src + (size_t)(inb ? ys[t] : 0) * sstep + (size_t)(inb ? xs[t] : 0) * CN;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            const float s = inb ? (float)p[c] : 0.f;
            out[c] = __builtin_fmaf(s, w[t], out[c]);
        }
    }
}

template <int CN>
__global__ void __launch_bounds__(256) k_remap_linear(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                      const float *__restrict__ mx, size_t mxstep,
                                                      const float *__restrict__ my, size_t mystep,
                                                      uint8_t *__restrict__ dst, size_t dstep, int drows, int dcols)
{
    XY_GUARD(dcols, drows)
    const float xc = row_ptr<float>(mx, mxstep, y)[x];
    const float yc = row_ptr<float>(my, mystep, y)[x];
    float out[CN];
    bilinear_taps<CN>(src, sstep, srows, scols, xc, yc, out);
    uint8_t *d = row_ptr<uint8_t>(dst, dstep, y) + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) d[c] = sat_u8(out[c]);
}

// cv::remap's CPU arithmetic (MS_INTER_LINEAR_FIXPT): see remap_fixpt in common.hpp
template <int CN>
__global__ void __launch_bounds__(256) k_remap_fixpt(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                     const float *__restrict__ mx, size_t mxstep, const float *__restrict__ my, size_t mystep,
                                                     uint8_t *__restrict__ dst, size_t dstep, int drows, int dcols)
{
    XY_GUARD(dcols, drows)
    uint8_t o[CN];
    remap_fixpt<CN>(src, sstep, srows, scols, row_ptr<float>(mx, mxstep, y)[x], row_ptr<float>(my, mystep, y)[x], o);
    uint8_t *d = row_ptr<uint8_t>(dst, dstep, y) + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) d[c] = o[c];
}

// remap, INTER_NEAREST (PointFilter: __float2int_rz), BORDER_CONSTANT(0), 8UC1  [filters.hpp:58-77]
__global__ void __launch_bounds__(256) k_remap_nearest_c1(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                          const float *__restrict__ mx, size_t mxstep,
                                                          const float *__restrict__ my, size_t mystep,
                                                          uint8_t *__restrict__ dst, size_t dstep, int drows, int dcols)
{
    XY_GUARD(dcols, drows)
    const int xx = f2i_rz(row_ptr<float>(mx, mxstep, y)[x]);
    const int yy = f2i_rz(row_ptr<float>(my, mystep, y)[x]);
    const bool inb = xx >= 0 && xx < scols && yy >= 0 && yy < srows;
    row_ptr<uint8_t>(dst, dstep, y)[x] = inb ? row_ptr<uint8_t>(src, sstep, yy)[xx] : (uint8_t)0;
}

// remap, INTER_LINEAR, BORDER_REFLECT, 8UC3: the seam-scale image warp of calibration (APP/calibration.cpp:118);
// every tap index goes through BrdReflect (border_interpolate.hpp:485-525)
__global__ void __launch_bounds__(256) k_remap_linear_reflect3(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                               const float *__restrict__ mx, size_t mxstep, const float *__restrict__ my, size_t mystep,
                                                               uint8_t *__restrict__ dst, size_t dstep, int drows, int dcols)
{
    XY_GUARD(dcols, drows)
    const float xc = row_ptr<float>(mx, mxstep, y)[x], yc = row_ptr<float>(my, mystep, y)[x];
    const int x1 = f2i_rd(xc), y1 = f2i_rd(yc);
    const int x2 = (int)((unsigned)x1 + 1u), y2 = (int)((unsigned)y1 + 1u);
    const float w[4] = {((float)x2 - xc) * ((float)y2 - yc), (xc - (float)x1) * ((float)y2 - yc),
                        ((float)x2 - xc) * (yc - (float)y1), (xc - (float)x1) * (yc - (float)y1)};
    const int xs[4] = {x1, x2, x1, x2}, ys[4] = {y1, y1, y2, y2};
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint8_t *p = row_ptr<uint8_t>(src, sstep, reflect_idx(ys[t], srows)) + (size_t)reflect_idx(xs[t], scols) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_fmaf((float)p[c], w[t], acc[c]);
    }
    uint8_t *d = row_ptr<uint8_t>(dst, dstep, y) + (size_t)x * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = sat_u8(acc[c]);
}

int launch_remap(const ms_image &src, const ms_image &xm, const ms_image &ym, ms_image &dst, int interp, int border, hipStream_t st)
{
    if (border == MS_BORDER_REFLECT) {
        if (!(src.type == MS_8UC3 && interp == MS_INTER_LINEAR)) return fail(MS_ERR_UNSUPPORTED, "ms_remap: BORDER_REFLECT is only used for 8UC3 / INTER_LINEAR");
        k_remap_linear_reflect3<<<grid2d(dst.cols, dst.rows), dim3(BX, BY), 0, st>>>((const uint8_t *)src.data, src.step, src.rows, src.cols,
            (const float *)xm.data, xm.step, (const float *)ym.data, ym.step, (uint8_t *)dst.data, dst.step, dst.rows, dst.cols);
        MS_LAUNCH_CHECK();
        return MS_OK;
    }
    if (border != MS_BORDER_CONSTANT) return fail(MS_ERR_UNSUPPORTED, "ms_remap: border type %d not on the path", border);
    const dim3 g = grid2d(dst.cols, dst.rows), b(BX, BY);
    auto S = (const uint8_t *)src.data; auto D = (uint8_t *)dst.data;
    auto MX = (const float *)xm.data; auto MY = (const float *)ym.data;
    if (src.type == MS_8UC3 && interp == MS_INTER_LINEAR)
        k_remap_linear<3><<<g, b, 0, st>>>(S, src.step, src.rows, src.cols, MX, xm.step, MY, ym.step, D, dst.step, dst.rows, dst.cols);
    else if (src.type == MS_8UC1 && interp == MS_INTER_LINEAR)
        k_remap_linear<1><<<g, b, 0, st>>>(S, src.step, src.rows, src.cols, MX, xm.step, MY, ym.step, D, dst.step, dst.rows, dst.cols);
    else if (src.type == MS_8UC3 && interp == MS_INTER_LINEAR_FIXPT)
        k_remap_fixpt<3><<<g, b, 0, st>>>(S, src.step, src.rows, src.cols, MX, xm.step, MY, ym.step, D, dst.step, dst.rows, dst.cols);
    else if (src.type == MS_8UC1 && interp == MS_INTER_LINEAR_FIXPT)
        k_remap_fixpt<1><<<g, b, 0, st>>>(S, src.step, src.rows, src.cols, MX, xm.step, MY, ym.step, D, dst.step, dst.rows, dst.cols);
    else if (src.type == MS_8UC1 && interp == MS_INTER_NEAREST)
        k_remap_nearest_c1<<<g, b, 0, st>>>(S, src.step, src.rows, src.cols, MX, xm.step, MY, ym.step, D, dst.step, dst.rows, dst.cols);
    else
        return fail(MS_ERR_UNSUPPORTED, "ms_remap: type %d / interpolation %d not on the hot path", src.type, interp);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// resize_linear: src = dst * (1/scale), no half-pixel centre, x2/y2 reads clamped  [resize.cu:71-106]
template <int CN>
__global__ void __launch_bounds__(256) k_resize_linear(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                       uint8_t *__restrict__ dst, size_t dstep, int drows, int dcols,
                                                       float ify, float ifx)
{
    XY_GUARD(dcols, drows)
    const float sx = (float)x * ifx, sy = (float)y * ify;
    const int x1 = f2i_rd(sx), y1 = f2i_rd(sy);
    const int x2 = x1 + 1, y2 = y1 + 1;
    const int x2r = min(x2, scols - 1), y2r = min(y2, srows - 1);
    const float w11 = ((float)x2 - sx) * ((float)y2 - sy), w12 = (sx - (float)x1) * ((float)y2 - sy);
    const float w21 = ((float)x2 - sx) * (sy - (float)y1), w22 = (sx - (float)x1) * (sy - (float)y1);
    const uint8_t *r1 = row_ptr<uint8_t>(src, sstep, y1), *r2 = row_ptr<uint8_t>(src, sstep, y2r);
    uint8_t *d = row_ptr<uint8_t>(dst, dstep, y) + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) {
        float out = 0.f;
        out = __builtin_fmaf((float)r1[(size_t)x1 * CN + c], w11, out);
        out = __builtin_fmaf((float)r1[(size_t)x2r * CN + c], w12, out);
        out = __builtin_fmaf((float)r2[(size_t)x1 * CN + c], w21, out);
        out = __builtin_fmaf((float)r2[(size_t)x2r * CN + c], w22, out);
        d[c] = sat_u8(out);
    }
}

// the same for up to RESIZE_BATCH images of one geometry in one launch: stitch_online's per-view cuda::resize by compose_scale (timed.cpp:75-85) for all views of a frame
constexpr int RESIZE_BATCH = 64;
struct ResizeBatch { const uint8_t *src[RESIZE_BATCH]; uint8_t *dst[RESIZE_BATCH]; };
__global__ void __launch_bounds__(256) k_resize_linear3_batch(ResizeBatch T, size_t sstep, int srows, int scols, size_t dstep, int drows, int dcols, float ify, float ifx)
{
    XY_GUARD(dcols, drows)
    const uint8_t *src = T.src[blockIdx.z];
    const float sx = (float)x * ifx, sy = (float)y * ify;
    const int x1 = f2i_rd(sx), y1 = f2i_rd(sy);
    const int x2 = x1 + 1, y2 = y1 + 1;
    const int x2r = min(x2, scols - 1), y2r = min(y2, srows - 1);
    const float w11 = ((float)x2 - sx) * ((float)y2 - sy), w12 = (sx - (float)x1) * ((float)y2 - sy);
    const float w21 = ((float)x2 - sx) * (sy - (float)y1), w22 = (sx - (float)x1) * (sy - (float)y1);
    const uint8_t *r1 = row_ptr<uint8_t>(src, sstep, y1), *r2 = row_ptr<uint8_t>(src, sstep, y2r);
    uint8_t *d = row_ptr<uint8_t>(T.dst[blockIdx.z], dstep, y) + (size_t)x * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float out = 0.f;
        out = __builtin_fmaf((float)r1[(size_t)x1 * 3 + c], w11, out);
        out = __builtin_fmaf((float)r1[(size_t)x2r * 3 + c], w12, out);
        out = __builtin_fmaf((float)r2[(size_t)x1 * 3 + c], w21, out);
        out = __builtin_fmaf((float)r2[(size_t)x2r * 3 + c], w22, out);
        d[c] = sat_u8(out);
    }
}
// The per-frame form of the same resize (the shipped COMPOSE_MEGAPIX puts cuda::resize of every view on the per-frame path, timed.cpp:75-85): 4 output
// pixels x RS_ROWS rows per lane.  A lane's 4 pixels sample at most 8 consecutive source pixels for downscales up to 1.6x, so each source row is ONE
// 24-byte window (read as the 7 aligned dwords around it: 3.5 lane-dwords per output pixel and row instead of 12 byte loads), the taps of a pixel are cut out of
// the window in registers (a two- or three-way dword select + v_alignbyte_b32), a source row shared by the lane's two output rows is read once, and the 12
// output bytes leave as one store.  Same fp32 expressions in the same order as k_resize_linear -> bit-identical
// (tests/test_prims_gpu.py::test_resize_linear_batch_equals_single_calls).  Lanes whose window would leave the source row, ragged right edges and stronger
// downscales take the per-pixel path.  192 images 1080p -> 1578 x 887 (32 frames of the shipped rig): 3.1 ms with the per-pixel kernel, 0.755 ms with a generic
// 4-way select (which the compiler turned into divergent branches), 0.57-0.62 ms now; with the source reads removed 0.43-0.46 (the VALU floor: 24 byte->float
// conversions and 24 fmas per pixel pair): profiles/r03_resize_ab.txt.  v_pk_fma_f32 for pixel pairs was 6 % SLOWER (packed fp32 is no faster per flop here).
#ifndef MS_RS_ROWS
#define MS_RS_ROWS 2
#endif
constexpr int RS_ROWS = MS_RS_ROWS;
__device__ __forceinline__ float rs_byte(unsigned lo, unsigned hi, int b) { return (float)(((b < 4 ? lo : hi) >> (8 * (b & 3))) & 0xffu); }
// lanes are numbered row-major over (row group, 4-pixel column group) of an image and cut into 256-lane workgroups regardless of the row length (no idle
// lanes at the right edge: 1578 columns are 395 lanes = 6.2 waves); lpr = lanes per row, lpr_magic = floor(2^32 / lpr) for the division
__global__ void __launch_bounds__(256) k_resize_linear3_x4(ResizeBatch T, size_t sstep, int srows, int scols, size_t dstep, int drows, int dcols, float ify, float ifx,
                                                           unsigned lpr, unsigned lpr_magic, unsigned n_lanes)
{
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_lanes) return;
    unsigned rg = __umulhi(idx, lpr_magic), lx = idx - rg * lpr;
    if (lx >= lpr) { ++rg; lx -= lpr; }
    const int x0 = 4 * (int)lx, yb = (int)rg * RS_ROWS;
    if (x0 >= dcols || yb >= drows) return;
    const uint8_t *src = T.src[blockIdx.z];
    uint8_t *dst = T.dst[blockIdx.z];
    float sx[4];
    int x1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sx[k] = (float)(x0 + k) * ifx; x1[k] = f2i_rd(sx[k]); }
    auto step12 = [](int d) { return d == 1 || d == 2; };
    const bool fast = x0 + 3 < dcols && x1[0] >= 0 && x1[0] + 10 <= scols && step12(x1[1] - x1[0]) && step12(x1[2] - x1[1]) && step12(x1[3] - x1[2]);
    if (!fast) {                      // the per-pixel kernel's code for this lane's pixels
        for (int r = 0; r < RS_ROWS && yb + r < drows; ++r) {
            const int y = yb + r;
            const float sy = (float)y * ify;
            const int y1 = f2i_rd(sy), y2 = y1 + 1, y2r = min(y2, srows - 1);
            const uint8_t *r1 = row_ptr<uint8_t>(src, sstep, y1), *r2 = row_ptr<uint8_t>(src, sstep, y2r);
            for (int k = 0; k < 4 && x0 + k < dcols; ++k) {
                const int x2 = x1[k] + 1, x2r = min(x2, scols - 1);
                const float w11 = ((float)x2 - sx[k]) * ((float)y2 - sy), w12 = (sx[k] - (float)x1[k]) * ((float)y2 - sy);
                const float w21 = ((float)x2 - sx[k]) * (sy - (float)y1), w22 = (sx[k] - (float)x1[k]) * (sy - (float)y1);
                uint8_t *d = row_ptr<uint8_t>(dst, dstep, y) + (size_t)(x0 + k) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float out = 0.f;
                    out = __builtin_fmaf((float)r1[(size_t)x1[k] * 3 + c], w11, out);
                    out = __builtin_fmaf((float)r1[(size_t)x2r * 3 + c], w12, out);
                    out = __builtin_fmaf((float)r2[(size_t)x1[k] * 3 + c], w21, out);
                    out = __builtin_fmaf((float)r2[(size_t)x2r * 3 + c], w22, out);
                    d[c] = sat_u8(out);
                }
            }
        }
        return;
    }
    // Pixel k's 6 tap bytes start at byte o_k = 3 (x1[k] - x1[0]) of the window, and consecutive pixels advance by one or two source pixels (checked above), so
    // o_1 is 3 or 6, o_2 is 6, 9 or 12, o_3 is 9, 12 or 15: the dwords around pixel k are one of two or three candidates (8 v_cndmask per window instead of a
    // generic 4-way select of every dword), pixel 0 needs none, and the third dword matters only for a byte shift of 3 -- where it is a fixed one.
    const int e1 = x1[1] - x1[0], e2 = x1[2] - x1[0], e3 = x1[3] - x1[0];
    const bool s1 = e1 == 2, s2a = e2 == 2, s2b = e2 == 3, s3 = e3 == 3;
    const unsigned sh1 = s1 ? 2u : 3u, sh2 = (unsigned)(4 - e2), sh3 = (unsigned)(3 * e3) & 3u;
    float wx1[4], wx2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { wx2[k] = (float)(x1[k] + 1) - sx[k]; wx1[k] = sx[k] - (float)x1[k]; }
    const unsigned col0 = (unsigned)x1[0] * 3u;
    auto load_win = [&](int yy, unsigned (&w)[6]) {
        const uint8_t *p = src + ((unsigned)yy * (unsigned)sstep + col0);      // (32-bit offsets: the launcher checks rows * step < 2^32)
        // 7 aligned dwords around the window + a byte shift (the 4 bytes past the window stay inside the row: see the fast-path condition); unaligned 16 + 8 byte
        // reads of the window itself cost 12 % more (the texture-address path works per aligned dword)
        const unsigned sh0 = (unsigned)(uintptr_t)p & 3u;
        const unsigned *q = reinterpret_cast<const unsigned *>(p - sh0);
        uint4 a; unsigned b0, b1, b2;
        __builtin_memcpy(&a, __builtin_assume_aligned(q, 4), 16);
        b0 = q[4]; b1 = q[5]; b2 = q[6];
        w[0] = __builtin_amdgcn_alignbyte(a.y, a.x, sh0); w[1] = __builtin_amdgcn_alignbyte(a.z, a.y, sh0); w[2] = __builtin_amdgcn_alignbyte(a.w, a.z, sh0);
        w[3] = __builtin_amdgcn_alignbyte(b0, a.w, sh0); w[4] = __builtin_amdgcn_alignbyte(b1, b0, sh0); w[5] = __builtin_amdgcn_alignbyte(b2, b1, sh0);
    };
    // (the window's dwords by value: selects between array elements would be turned into indexed reads of a scratch copy)
    auto taps = [&](unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned w4, unsigned w5, int k, unsigned &lo, unsigned &hi) {
        if (k == 0) { lo = w0; hi = w1; }
        else if (k == 1) {
            const unsigned d0 = s1 ? w1 : w0, d1 = s1 ? w2 : w1;
            lo = __builtin_amdgcn_alignbyte(d1, d0, sh1); hi = __builtin_amdgcn_alignbyte(w2, d1, sh1);
        } else if (k == 2) {
            const unsigned d0 = s2a ? w1 : (s2b ? w2 : w3), d1 = s2a ? w2 : (s2b ? w3 : w4);
            lo = __builtin_amdgcn_alignbyte(d1, d0, sh2); hi = d1 >> (8u * sh2);
        } else {
            const unsigned d0 = s3 ? w2 : w3, d1 = s3 ? w3 : w4;
            lo = __builtin_amdgcn_alignbyte(d1, d0, sh3); hi = __builtin_amdgcn_alignbyte(w5, d1, sh3);
        }
    };
    // all source rows of the lane's output rows are read first (the second output row usually starts on the row the first one ends on: read once)
    int y1[RS_ROWS], y2r[RS_ROWS];
    float wy1[RS_ROWS], wy2[RS_ROWS];
    bool live[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        const int y = min(yb + r, drows - 1);
        live[r] = yb + r < drows;
        const float sy = (float)y * ify;
        y1[r] = f2i_rd(sy);
        y2r[r] = min(y1[r] + 1, srows - 1);
        wy2[r] = (float)(y1[r] + 1) - sy; wy1[r] = sy - (float)y1[r];
    }
    unsigned W[RS_ROWS][2][6];
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        if (r > 0 && y1[r] == y2r[r - 1]) {
#pragma unroll
            for (int i = 0; i < 6; ++i) W[r][0][i] = W[r - 1][1][i];
        } else load_win(y1[r], W[r][0]);
        load_win(y2r[r], W[r][1]);
    }
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        if (!live[r]) break;
        unsigned o3[3] = {0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned lo[2], hi[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) taps(W[r][q][0], W[r][q][1], W[r][q][2], W[r][q][3], W[r][q][4], W[r][q][5], k, lo[q], hi[q]);
            const float w11 = wx2[k] * wy2[r], w12 = wx1[k] * wy2[r], w21 = wx2[k] * wy1[r], w22 = wx1[k] * wy1[r];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float out = 0.f;
                out = __builtin_fmaf(rs_byte(lo[0], hi[0], c), w11, out);
                out = __builtin_fmaf(rs_byte(lo[0], hi[0], 3 + c), w12, out);
                out = __builtin_fmaf(rs_byte(lo[1], hi[1], c), w21, out);
                out = __builtin_fmaf(rs_byte(lo[1], hi[1], 3 + c), w22, out);
                const int i = 3 * k + c;
                o3[i >> 2] = sat_u8_into(out, (unsigned)(i & 3), o3[i >> 2]);
            }
        }
        __builtin_memcpy(dst + ((unsigned)(yb + r) * (unsigned)dstep + (unsigned)x0 * 3u), o3, 12);
    }
}
int launch_resize_linear_batch(const ms_image *src, ms_image *dst, int n, double fx, double fy, hipStream_t st)
{
    if (!(fx > 0 && fy > 0)) { fx = (double)dst[0].cols / src[0].cols; fy = (double)dst[0].rows / src[0].rows; }
    const float ifx = (float)(1.0 / fx), ify = (float)(1.0 / fy);
    const unsigned lpr = (unsigned)div_up(dst[0].cols, 4);
    const unsigned long long n_lanes = (unsigned long long)lpr * (unsigned)div_up(dst[0].rows, RS_ROWS);
    // downscales whose 4-pixel windows fit 24 bytes (and images whose lane index times the row length fits 32 bits: the kernel's division); else one pixel per lane
    const bool x4 = ifx >= 1.f && ifx <= 1.6f && src[0].cols >= 16 && n_lanes * lpr < 0x100000000ull &&
                    (unsigned long long)src[0].rows * src[0].step < 0x100000000ull && (unsigned long long)dst[0].rows * dst[0].step < 0x100000000ull && dev_knob("MS_RESIZE_SIMPLE", 0) == 0;
    for (int i0 = 0; i0 < n; i0 += RESIZE_BATCH) {
        const int m = std::min(RESIZE_BATCH, n - i0);
        ResizeBatch T{};
        for (int i = 0; i < m; ++i) { T.src[i] = (const uint8_t *)src[i0 + i].data; T.dst[i] = (uint8_t *)dst[i0 + i].data; }
        if (x4) {
            k_resize_linear3_x4<<<dim3(div_up((int)n_lanes, 256), 1, m), 256, 0, st>>>(T, src[0].step, src[0].rows, src[0].cols, dst[0].step, dst[0].rows, dst[0].cols, ify, ifx,
                                                                                    lpr, (unsigned)(0x100000000ull / lpr), (unsigned)n_lanes);
        } else {
            const dim3 g2 = grid2d(dst[0].cols, dst[0].rows);
            k_resize_linear3_batch<<<dim3(g2.x, g2.y, m), dim3(BX, BY), 0, st>>>(T, src[0].step, src[0].rows, src[0].cols, dst[0].step, dst[0].rows, dst[0].cols, ify, ifx);
        }
        MS_LAUNCH_CHECK();
    }
    return MS_OK;
}

int launch_resize_linear(const ms_image &src, ms_image &dst, double fx, double fy, hipStream_t st)
{
    if (dst.rows == src.rows && dst.cols == src.cols) {   // resize.cpp:86-90: plain copy
        const size_t wb = (size_t)src.cols * (src.type == MS_8UC3 ? 3 : 1);
        MS_HIP(hipMemcpy2DAsync(dst.data, dst.step, src.data, src.step, wb, src.rows, hipMemcpyDeviceToDevice, st));
        return MS_OK;
    }
    // resize.cpp:72-81,105: explicit dsize -> fx = dsize.width / src.cols (double); kernel gets static_cast<float>(1.0 / fx)
    if (!(fx > 0 && fy > 0)) { fx = (double)dst.cols / src.cols; fy = (double)dst.rows / src.rows; }
    const float ifx = (float)(1.0 / fx), ify = (float)(1.0 / fy);
    const dim3 g = grid2d(dst.cols, dst.rows), b(BX, BY);
    if (src.type == MS_8UC3)
        k_resize_linear<3><<<g, b, 0, st>>>((const uint8_t *)src.data, src.step, src.rows, src.cols, (uint8_t *)dst.data, dst.step, dst.rows, dst.cols, ify, ifx);
    else if (src.type == MS_8UC1)
        k_resize_linear<1><<<g, b, 0, st>>>((const uint8_t *)src.data, src.step, src.rows, src.cols, (uint8_t *)dst.data, dst.step, dst.rows, dst.cols, ify, ifx);
    else
        return fail(MS_ERR_UNSUPPORTED, "ms_resize_linear: type %d", src.type);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// element-wise transforms over a rows x width (elements) extent
template <typename S, typename D, typename F>
__global__ void __launch_bounds__(256) k_unary(const S *__restrict__ src, size_t sstep, D *__restrict__ dst, size_t dstep,
                                               int rows, int width, F f)
{
    XY_GUARD(width, rows)
    row_ptr<D>(dst, dstep, y)[x] = f(row_ptr<S>(src, sstep, y)[x]);
}
template <typename A, typename B, typename D, typename F>
__global__ void __launch_bounds__(256) k_binary(const A *a, size_t astep, const B *b, size_t bstep, D *dst, size_t dstep,
                                                int rows, int width, F f)
{
    XY_GUARD(width, rows)
    row_ptr<D>(dst, dstep, y)[x] = f(row_ptr<A>(a, astep, y)[x], row_ptr<B>(b, bstep, y)[x]);
}

struct OpScaleU8 { float a; __device__ uint8_t operator()(uint8_t v) const { return sat_u8(__builtin_fmaf(a, (float)v, 0.f)); } };
struct OpWiden { __device__ int16_t operator()(uint8_t v) const { return (int16_t)v; } };
struct OpNarrow { __device__ uint8_t operator()(int16_t v) const { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); } };
struct OpU8F32 { float a; __device__ float operator()(uint8_t v) const { return __builtin_fmaf(a, (float)v, 0.f); } };
struct OpGt { float t; __device__ uint8_t operator()(float v) const { return v > t ? 255 : 0; } };
struct OpEq { uint8_t t; __device__ uint8_t operator()(uint8_t v) const { return v == t ? 255 : 0; } };
struct OpSub { __device__ int16_t operator()(int16_t a, int16_t b) const { return sat_s16((int)a - (int)b); } };
struct OpAdd { __device__ int16_t operator()(int16_t a, int16_t b) const { return sat_s16((int)a + (int)b); } };
struct OpAnd { __device__ uint8_t operator()(uint8_t a, uint8_t b) const { return a & b; } };

static inline int cn_of(int type) { return ((type >> 3) & 7) + 1; }

template <typename S, typename D, typename F>
static int run_unary(const ms_image &src, ms_image &dst, int width, F f, hipStream_t st)
{
    k_unary<S, D, F><<<grid2d(width, src.rows), dim3(BX, BY), 0, st>>>((const S *)src.data, src.step, (D *)dst.data, dst.step, src.rows, width, f);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
template <typename A, typename B, typename D, typename F>
static int run_binary(const ms_image &a, const ms_image &b, ms_image &dst, int width, F f, hipStream_t st)
{
    k_binary<A, B, D, F><<<grid2d(width, a.rows), dim3(BX, BY), 0, st>>>((const A *)a.data, a.step, (const B *)b.data, b.step, (D *)dst.data, dst.step, a.rows, width, f);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

int launch_convert_scale_8u(const ms_image &src, ms_image &dst, double alpha, hipStream_t st)
{
    return run_unary<uint8_t, uint8_t>(src, dst, src.cols * cn_of(src.type), OpScaleU8{(float)alpha}, st);
}

int launch_convert(const ms_image &src, ms_image &dst, double alpha, hipStream_t st)
{
    const int sd = src.type & 7, dd = dst.type & 7, w = src.cols * cn_of(src.type);
    if (sd == 0 && dd == 3) return run_unary<uint8_t, int16_t>(src, dst, w, OpWiden{}, st);
    if (sd == 3 && dd == 0) return run_unary<int16_t, uint8_t>(src, dst, w, OpNarrow{}, st);
    if (sd == 0 && dd == 5) return run_unary<uint8_t, float>(src, dst, w, OpU8F32{(float)alpha}, st);
    if (sd == dd) {  // convertTo same depth == copyTo (gpu_mat.cu:535-543)
        static const int esz[8] = {1, 1, 2, 2, 4, 4, 8, 2};
        MS_HIP(hipMemcpy2DAsync(dst.data, dst.step, src.data, src.step, (size_t)w * esz[sd], src.rows, hipMemcpyDeviceToDevice, st));
        return MS_OK;
    }
    return fail(MS_ERR_UNSUPPORTED, "ms_convert: depth %d -> %d not on the hot path", sd, dd);
}

int launch_sub_16s(const ms_image &a, const ms_image &b, ms_image &dst, hipStream_t st)
{ return run_binary<int16_t, int16_t, int16_t>(a, b, dst, a.cols * cn_of(a.type), OpSub{}, st); }
int launch_add_16s(const ms_image &a, const ms_image &b, ms_image &dst, hipStream_t st)
{ return run_binary<int16_t, int16_t, int16_t>(a, b, dst, a.cols * cn_of(a.type), OpAdd{}, st); }
int launch_and_8u(const ms_image &a, const ms_image &b, ms_image &dst, hipStream_t st)
{ return run_binary<uint8_t, uint8_t, uint8_t>(a, b, dst, a.cols, OpAnd{}, st); }
int launch_compare_gt_32f(const ms_image &src, float thr, ms_image &dst, hipStream_t st)
{ return run_unary<float, uint8_t>(src, dst, src.cols, OpGt{thr}, st); }
int launch_compare_eq_8u(const ms_image &src, int val, ms_image &dst, hipStream_t st)
{ return run_unary<uint8_t, uint8_t>(src, dst, src.cols, OpEq{(uint8_t)val}, st); }

// ------------------------------------------------------------------------------------------------
// copyMakeBorder: REFLECT for byte pixels of ES bytes, CONSTANT(0) for float  [copy_make_border.cu:61-115]
template <int ES>
__global__ void __launch_bounds__(256) k_border_reflect(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                        uint8_t *__restrict__ dst, size_t dstep, int drows, int dcols,
                                                        int top, int left)
{
    XY_GUARD(dcols, drows)
    const uint8_t *s = row_ptr<uint8_t>(src, sstep, reflect_idx(y - top, srows)) + (size_t)reflect_idx(x - left, scols) * ES;
    uint8_t *d = row_ptr<uint8_t>(dst, dstep, y) + (size_t)x * ES;
#pragma unroll
    for (int i = 0; i < ES; ++i) d[i] = s[i];
}
__global__ void __launch_bounds__(256) k_border_const_f32(const float *__restrict__ src, size_t sstep, int srows, int scols,
                                                          float *__restrict__ dst, size_t dstep, int drows, int dcols,
                                                          int top, int left)
{
    XY_GUARD(dcols, drows)
    const int sx = x - left, sy = y - top;
    const bool inb = sx >= 0 && sx < scols && sy >= 0 && sy < srows;
    row_ptr<float>(dst, dstep, y)[x] = inb ? row_ptr<float>(src, sstep, sy)[sx] : 0.f;
}

int launch_copy_make_border(const ms_image &src, ms_image &dst, int top, int left, int border_type, hipStream_t st)
{
    const dim3 g = grid2d(dst.cols, dst.rows), b(BX, BY);
    if (border_type == MS_BORDER_REFLECT && src.type == MS_8UC3)
        k_border_reflect<3><<<g, b, 0, st>>>((const uint8_t *)src.data, src.step, src.rows, src.cols, (uint8_t *)dst.data, dst.step, dst.rows, dst.cols, top, left);
    else if (border_type == MS_BORDER_REFLECT && src.type == MS_8UC1)
        k_border_reflect<1><<<g, b, 0, st>>>((const uint8_t *)src.data, src.step, src.rows, src.cols, (uint8_t *)dst.data, dst.step, dst.rows, dst.cols, top, left);
    else if (border_type == MS_BORDER_CONSTANT && src.type == MS_32FC1)
        k_border_const_f32<<<g, b, 0, st>>>((const float *)src.data, src.step, src.rows, src.cols, (float *)dst.data, dst.step, dst.rows, dst.cols, top, left);
    else
        return fail(MS_ERR_UNSUPPORTED, "ms_copy_make_border: type %d with border %d not on the hot path", src.type, border_type);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// pyrDown: 5x5 [1 4 6 4 1]^2/256, BORDER_REFLECT_101, decimate by 2  [pyr_down.cu:55-174]
// 16S: exact integer form.  32F: vertical-then-horizontal fma chains in the kernel's order.
template <int CN>
__global__ void __launch_bounds__(256) k_pyr_down_16s(const int16_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                      int16_t *__restrict__ dst, size_t dstep, int drows, int dcols)
{
    XY_GUARD(dcols, drows)
    const int sy = 2 * y, sx = 2 * x;
    const int16_t *r[5] = {row_ptr<int16_t>(src, sstep, r101_low(sy - 2, srows)), row_ptr<int16_t>(src, sstep, r101_low(sy - 1, srows)),
                           row_ptr<int16_t>(src, sstep, sy), row_ptr<int16_t>(src, sstep, r101_high(sy + 1, srows)),
                           row_ptr<int16_t>(src, sstep, r101_high(sy + 2, srows))};
    int cx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) cx[k] = r101(sx - 2 + k, scols) * CN;
    const int wv[5] = {1, 4, 6, 4, 1};
    int acc[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) acc[c] = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int c = 0; c < CN; ++c) acc[c] += wv[j] * wv[k] * (int)r[j][cx[k] + c];
    int16_t *d = row_ptr<int16_t>(dst, dstep, y) + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) d[c] = sat_s16(rne_shift(acc[c], 8));
}

__global__ void __launch_bounds__(256) k_pyr_down_32f(const float *__restrict__ src, size_t sstep, int srows, int scols,
                                                      float *__restrict__ dst, size_t dstep, int drows, int dcols)
{
    XY_GUARD(dcols, drows)
    const int sy = 2 * y, sx = 2 * x;
    const float *r[5] = {row_ptr<float>(src, sstep, r101_low(sy - 2, srows)), row_ptr<float>(src, sstep, r101_low(sy - 1, srows)),
                         row_ptr<float>(src, sstep, sy), row_ptr<float>(src, sstep, r101_high(sy + 1, srows)),
                         row_ptr<float>(src, sstep, r101_high(sy + 2, srows))};
    float v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int cx = r101(sx - 2 + k, scols);
        float s = 0.0625f * r[0][cx];
        s = __builtin_fmaf(0.25f, r[1][cx], s);
        s = __builtin_fmaf(0.375f, r[2][cx], s);
        s = __builtin_fmaf(0.25f, r[3][cx], s);
        s = __builtin_fmaf(0.0625f, r[4][cx], s);
        v[k] = s;
    }
    float s = 0.0625f * v[0];
    s = __builtin_fmaf(0.25f, v[1], s);
    s = __builtin_fmaf(0.375f, v[2], s);
    s = __builtin_fmaf(0.25f, v[3], s);
    s = __builtin_fmaf(0.0625f, v[4], s);
    row_ptr<float>(dst, dstep, y)[x] = s;
}

int launch_pyr_down(const ms_image &src, ms_image &dst, hipStream_t st)
{
    const dim3 g = grid2d(dst.cols, dst.rows), b(BX, BY);
    if (src.type == MS_16SC3)
        k_pyr_down_16s<3><<<g, b, 0, st>>>((const int16_t *)src.data, src.step, src.rows, src.cols, (int16_t *)dst.data, dst.step, dst.rows, dst.cols);
    else if (src.type == MS_16SC1)
        k_pyr_down_16s<1><<<g, b, 0, st>>>((const int16_t *)src.data, src.step, src.rows, src.cols, (int16_t *)dst.data, dst.step, dst.rows, dst.cols);
    else if (src.type == MS_32FC1)
        k_pyr_down_32f<<<g, b, 0, st>>>((const float *)src.data, src.step, src.rows, src.cols, (float *)dst.data, dst.step, dst.rows, dst.cols);
    else
        return fail(MS_ERR_UNSUPPORTED, "ms_pyr_down: type %d", src.type);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// pyrUp: zero-insert x2, same 5-tap, x4; source index min(n-1, |i|)  [pyr_up.cu:55-145]
// exact integer form: S = sum over the 3x3 (even/odd phase) taps, result = rne(S / 64).
template <int CN>
__device__ __forceinline__ void pyr_up_px(const int16_t *__restrict__ src, size_t sstep, int srows, int scols,
                                          int x, int y, int out[CN])
{
    const int i = y >> 1, j = x >> 1;
    // vertical phase: even row -> (1,6,1) over i-1,i,i+1 ; odd row -> (4,4) over i,i+1
    int ry[3], wy[3], rx[3], wx[3];
    if ((y & 1) == 0) { ry[0] = i - 1; ry[1] = i; ry[2] = i + 1; wy[0] = 1; wy[1] = 6; wy[2] = 1; }
    else              { ry[0] = i;     ry[1] = i + 1; ry[2] = i; wy[0] = 4; wy[1] = 4; wy[2] = 0; }
    if ((x & 1) == 0) { rx[0] = j - 1; rx[1] = j; rx[2] = j + 1; wx[0] = 1; wx[1] = 6; wx[2] = 1; }
    else              { rx[0] = j;     rx[1] = j + 1; rx[2] = j; wx[0] = 4; wx[1] = 4; wx[2] = 0; }
#pragma unroll
    for (int c = 0; c < CN; ++c) out[c] = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int16_t *r = row_ptr<int16_t>(src, sstep, pu_idx(ry[a], srows));
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int cx = pu_idx(rx[b], scols) * CN;
            const int w = wy[a] * wx[b];
#pragma unroll
            for (int c = 0; c < CN; ++c) out[c] += w * (int)r[cx + c];
        }
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) out[c] = rne_shift(out[c], 6);
}

template <int CN>
__global__ void __launch_bounds__(256) k_pyr_up_16s(const int16_t *__restrict__ src, size_t sstep, int srows, int scols,
                                                    int16_t *__restrict__ dst, size_t dstep, int drows, int dcols)
{
    XY_GUARD(dcols, drows)
    int out[CN];
    pyr_up_px<CN>(src, sstep, srows, scols, x, y, out);
    int16_t *d = row_ptr<int16_t>(dst, dstep, y) + (size_t)x * CN;
#pragma unroll
    for (int c = 0; c < CN; ++c) d[c] = sat_s16(out[c]);
}

int launch_pyr_up(const ms_image &src, ms_image &dst, hipStream_t st)
{
    const dim3 g = grid2d(dst.cols, dst.rows), b(BX, BY);
    if (src.type == MS_16SC3)
        k_pyr_up_16s<3><<<g, b, 0, st>>>((const int16_t *)src.data, src.step, src.rows, src.cols, (int16_t *)dst.data, dst.step, dst.rows, dst.cols);
    else if (src.type == MS_16SC1)
        k_pyr_up_16s<1><<<g, b, 0, st>>>((const int16_t *)src.data, src.step, src.rows, src.cols, (int16_t *)dst.data, dst.step, dst.rows, dst.cols);
    else
        return fail(MS_ERR_UNSUPPORTED, "ms_pyr_up: type %d", src.type);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// addSrcWeightKernel32F / normalizeUsingWeightKernel32F  [multiband_blend.cu:36-51, 85-100]
__global__ void __launch_bounds__(256) k_add_src_weight(const int16_t *__restrict__ src, size_t sstep, const float *__restrict__ w, size_t wstep,
                                                        int16_t *dst, size_t dstep, float *dstw, size_t dwstep, int rows, int cols)
{
    XY_GUARD(cols, rows)
    const int16_t *s = row_ptr<int16_t>(src, sstep, y) + 3 * x;
    const float ww = row_ptr<float>(w, wstep, y)[x];
    int16_t *d = row_ptr<int16_t>(dst, dstep, y) + 3 * x;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = (int16_t)(d[c] + trunc_s16((float)s[c] * ww));
    float *dw = row_ptr<float>(dstw, dwstep, y) + x;
    *dw = *dw + ww;
}

__global__ void __launch_bounds__(256) k_normalize(const float *__restrict__ w, size_t wstep, int16_t *src, size_t sstep, int rows, int cols)
{
    XY_GUARD(cols, rows)
    const float den = row_ptr<float>(w, wstep, y)[x] + 1e-5f;
    int16_t *s = row_ptr<int16_t>(src, sstep, y) + 3 * x;
#pragma unroll
    for (int c = 0; c < 3; ++c) s[c] = trunc_s16((float)s[c] / den);
}

// addSrcWeightKernel16S / normalizeUsingWeightKernel16S  [multiband_blend.cu:10-34, 62-83]: the fixed-point flavour of the blender (weight_type CV_16S:
// weights are 0..256 = 8 fractional bits, blenders.cpp:414-418).  int arithmetic exactly as written there: (v * w) >> 8 is an arithmetic shift,
// short(...) keeps the low 16 bits, `+=` on short wraps.  (v << 8) / w truncates toward zero; w == 0 (a pixel no view covers) is integer division by
// zero, undefined in the reference -- defined here as 0 and stated in ms_stitch.h.
__global__ void __launch_bounds__(256) k_add_src_weight_16s(const int16_t *__restrict__ src, size_t sstep, const int16_t *__restrict__ w, size_t wstep,
                                                            int16_t *dst, size_t dstep, int16_t *dstw, size_t dwstep, int rows, int cols)
{
    XY_GUARD(cols, rows)
    const int16_t *s = row_ptr<int16_t>(src, sstep, y) + 3 * x;
    const int ww = row_ptr<int16_t>(w, wstep, y)[x];
    int16_t *d = row_ptr<int16_t>(dst, dstep, y) + 3 * x;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = (int16_t)(d[c] + (int16_t)(((int)s[c] * ww) >> 8));
    int16_t *dw = row_ptr<int16_t>(dstw, dwstep, y) + x;
    *dw = (int16_t)(*dw + ww);
}
__global__ void __launch_bounds__(256) k_normalize_16s(const int16_t *__restrict__ w, size_t wstep, int16_t *src, size_t sstep, int rows, int cols)
{
    XY_GUARD(cols, rows)
    const int ww = row_ptr<int16_t>(w, wstep, y)[x];
    int16_t *s = row_ptr<int16_t>(src, sstep, y) + 3 * x;
#pragma unroll
    for (int c = 0; c < 3; ++c) s[c] = ww ? (int16_t)(((int)s[c] * 256) / ww) : (int16_t)0;
}
int launch_add_src_weight_16s(const ms_image &src, const ms_image &w, ms_image &dst, ms_image &dstw, int rcw, int rch, hipStream_t st)
{
    k_add_src_weight_16s<<<grid2d(rcw, rch), dim3(BX, BY), 0, st>>>((const int16_t *)src.data, src.step, (const int16_t *)w.data, w.step,
                                                                    (int16_t *)dst.data, dst.step, (int16_t *)dstw.data, dstw.step, rch, rcw);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
int launch_normalize_16s(const ms_image &w, ms_image &src, int width, int height, hipStream_t st)
{
    k_normalize_16s<<<grid2d(width, height), dim3(BX, BY), 0, st>>>((const int16_t *)w.data, w.step, (int16_t *)src.data, src.step, height, width);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

__global__ void __launch_bounds__(256) k_zero_masked(int16_t *img, size_t step, const uint8_t *__restrict__ mask, size_t mstep, int rows, int cols)
{
    XY_GUARD(cols, rows)
    if (row_ptr<uint8_t>(mask, mstep, y)[x]) {
        int16_t *p = row_ptr<int16_t>(img, step, y) + 3 * x;
        p[0] = p[1] = p[2] = 0;
    }
}

int launch_add_src_weight(const ms_image &src, const ms_image &w, ms_image &dst, ms_image &dstw, int rcw, int rch, hipStream_t st)
{
    k_add_src_weight<<<grid2d(rcw, rch), dim3(BX, BY), 0, st>>>((const int16_t *)src.data, src.step, (const float *)w.data, w.step,
                                                                (int16_t *)dst.data, dst.step, (float *)dstw.data, dstw.step, rch, rcw);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
int launch_normalize(const ms_image &w, ms_image &src, int width, int height, hipStream_t st)
{
    k_normalize<<<grid2d(width, height), dim3(BX, BY), 0, st>>>((const float *)w.data, w.step, (int16_t *)src.data, src.step, height, width);
    MS_LAUNCH_CHECK();
    return MS_OK;
}
int launch_zero_masked(ms_image &img, const ms_image &mask, hipStream_t st)
{
    k_zero_masked<<<grid2d(img.cols, img.rows), dim3(BX, BY), 0, st>>>((int16_t *)img.data, img.step, (const uint8_t *)mask.data, mask.step, img.rows, img.cols);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// 3x3 max (replicate border): the dilation of APP/calibration.cpp:209,232
__global__ void __launch_bounds__(256) k_dilate3(const uint8_t *__restrict__ src, size_t sstep, uint8_t *__restrict__ dst, size_t dstep, int rows, int cols)
{
    XY_GUARD(cols, rows)
    uint8_t m = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const uint8_t *s = row_ptr<uint8_t>(src, sstep, min(max(y + dy, 0), rows - 1));
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) m = max(m, s[min(max(x + dx, 0), cols - 1)]);
    }
    row_ptr<uint8_t>(dst, dstep, y)[x] = m;
}
int launch_dilate3(const ms_image &src, ms_image &dst, hipStream_t st)
{
    k_dilate3<<<grid2d(src.cols, src.rows), dim3(BX, BY), 0, st>>>((const uint8_t *)src.data, src.step, (uint8_t *)dst.data, dst.step, src.rows, src.cols);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// buildWarpMapsKernel<Mapper>  [build_warp_maps.cu:67-152]; k_rinv/t/scale by value (no __constant__ upload)
template <int PROJ>
__global__ void __launch_bounds__(256) k_build_warp_maps(int tl_u, int tl_v, int cols, int rows, float *__restrict__ mapx, size_t mxstep,
                                                         float *__restrict__ mapy, size_t mystep, WarpParams P)
{
    XY_GUARD(cols, rows)
    float ox, oy;
    warp_combine(PROJ, warp_col_term(PROJ, (float)(tl_u + x), P), warp_row_term(PROJ, (float)(tl_v + y), P), P, ox, oy);
    row_ptr<float>(mapx, mxstep, y)[x] = ox;
    row_ptr<float>(mapy, mystep, y)[x] = oy;
}

int launch_build_warp_maps(int proj, int tl_u, int tl_v, ms_image &mx, ms_image &my, const float *k_rinv, const float *t, float scale, hipStream_t st)
{
    WarpParams P;
    memcpy(P.k, k_rinv, sizeof(P.k));
    if (t) memcpy(P.t, t, sizeof(P.t)); else P.t[0] = P.t[1] = P.t[2] = 0.f;
    P.scale = scale;
    const dim3 g = grid2d(mx.cols, mx.rows), b(BX, BY);
    if (proj == MS_PROJ_PLANE)
        k_build_warp_maps<MS_PROJ_PLANE><<<g, b, 0, st>>>(tl_u, tl_v, mx.cols, mx.rows, (float *)mx.data, mx.step, (float *)my.data, my.step, P);
    else if (proj == MS_PROJ_CYLINDRICAL)
        k_build_warp_maps<MS_PROJ_CYLINDRICAL><<<g, b, 0, st>>>(tl_u, tl_v, mx.cols, mx.rows, (float *)mx.data, mx.step, (float *)my.data, my.step, P);
    else if (proj == MS_PROJ_SPHERICAL)
        k_build_warp_maps<MS_PROJ_SPHERICAL><<<g, b, 0, st>>>(tl_u, tl_v, mx.cols, mx.rows, (float *)mx.data, mx.step, (float *)my.data, my.step, P);
    else
        return fail(MS_ERR_INVALID, "ms_build_warp_maps: unknown projection %d", proj);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// cvtColor(COLOR_YUV2BGR_NV12)  [imgproc/src/color.cpp:8738-8745 + YUV420sp2RGB888Invoker<0,0>]: the per-camera ingest the
// reference does on the CPU (APP/networking.cpp:45-47).  nv12_to_bgr_cell: 2 rows x 4 pixels, byte accesses (remainders, unaligned buffers); the hot form is nv12_to_bgr_cell8 below.
__device__ __forceinline__ void nv12_to_bgr_cell(const uint8_t *__restrict__ src, size_t sstep, int w, int h, uint8_t *__restrict__ dst, size_t dstep, int x, int y)
{
    constexpr int SH = 20, CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;
    const int n = min(4, w - x);
    const uint8_t *uvrow = src + (size_t)(h + y / 2) * sstep + x;
    uint8_t uv[4] = {128, 128, 128, 128};
    for (int i = 0; i < n; ++i) uv[i] = uvrow[i];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint8_t *yr = src + (size_t)(y + r) * sstep + x;
        uint8_t *d = dst + (size_t)(y + r) * dstep + (size_t)x * 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= n) break;
            const int u = (int)uv[k & ~1] - 128, v = (int)uv[(k & ~1) + 1] - 128;
            const int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
            const int yy = max(0, (int)yr[k] - 16) * CY;
            d[3 * k] = (uint8_t)min(max((yy + buv) >> SH, 0), 255);
            d[3 * k + 1] = (uint8_t)min(max((yy + guv) >> SH, 0), 255);
            d[3 * k + 2] = (uint8_t)min(max((yy + ruv) >> SH, 0), 255);
        }
    }
}
// Round 5: one lane = 2 rows x 8 pixels (the 2 x 4 form with twelve single-byte stores per lane and row ran at 1.2 TB/s).  Two 8-byte Y windows and one
// 8-byte UV window in, 2 x 24 bytes out as three dword pairs per row: a wave reads 512 contiguous bytes per plane row and writes 1536 contiguous bytes per image row.
// The chroma terms of a 2 x 2 block are built once.  Same integer formula, evaluated per pixel exactly as nv12_to_bgr_cell does (which stays for the right-hand
// remainder of widths that are not multiples of 8 and for buffers that are not 4-byte aligned).
// (the empty asm keeps clang from fusing two neighbouring clamp((a >> 20), 0, 255) into v_ashr_pk_u8_i32: as emitted by ROCm 7.2's clang for gfx950 the result's upper 16 bits
//  are assumed zero by the v_or3_b32 that packs the dword, and on MI355X they are not -- stray bits in every third byte, found by tests/test_prims_gpu.py::test_nv12_to_bgr;
//  tests/test_abi.py::test_no_v_ashr_pk_u8_i32_in_the_device_code guards the whole library)
__device__ __forceinline__ unsigned nv12_px(int yy, int c) { constexpr int SH = 20; int v = min(max((yy + c) >> SH, 0), 255); asm("" : "+v"(v)); return (unsigned)v; }
__device__ __forceinline__ void nv12_to_bgr_cell8(const uint8_t *__restrict__ src, size_t sstep, int h, uint8_t *__restrict__ dst, size_t dstep, int x, int y)
{
    constexpr int SH = 20, CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;
    uint2 yw[2], uvw;
    __builtin_memcpy(&yw[0], __builtin_assume_aligned(src + (size_t)y * sstep + x, 4), 8);
    __builtin_memcpy(&yw[1], __builtin_assume_aligned(src + (size_t)(y + 1) * sstep + x, 4), 8);
    __builtin_memcpy(&uvw, __builtin_assume_aligned(src + (size_t)(h + y / 2) * sstep + x, 4), 8);
    int ruv[4], guv[4], buv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned w2 = p < 2 ? uvw.x : uvw.y;
        const int u = (int)((w2 >> (16 * (p & 1))) & 255u) - 128, v = (int)((w2 >> (16 * (p & 1) + 8)) & 255u) - 128;
        ruv[p] = (1 << (SH - 1)) + CVR * v; guv[p] = (1 << (SH - 1)) + CVG * v + CUG * u; buv[p] = (1 << (SH - 1)) + CUB * u;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        unsigned px[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned w2 = k < 4 ? yw[r].x : yw[r].y;
            const int yy = max(0, (int)((w2 >> (8 * (k & 3))) & 255u) - 16) * CY;
            px[k][0] = nv12_px(yy, buv[k >> 1]); px[k][1] = nv12_px(yy, guv[k >> 1]); px[k][2] = nv12_px(yy, ruv[k >> 1]);
        }
        unsigned o[6];
#pragma unroll
        for (int q = 0; q < 2; ++q) {      // 4 pixels = 12 bytes = 3 dwords: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
            const int k = 4 * q;
            o[3 * q] = px[k][0] | (px[k][1] << 8) | (px[k][2] << 16) | (px[k + 1][0] << 24);
            o[3 * q + 1] = px[k + 1][1] | (px[k + 1][2] << 8) | (px[k + 2][0] << 16) | (px[k + 2][1] << 24);
            o[3 * q + 2] = px[k + 2][2] | (px[k + 3][0] << 8) | (px[k + 3][1] << 16) | (px[k + 3][2] << 24);
        }
        __builtin_memcpy(__builtin_assume_aligned(dst + (size_t)(y + r) * dstep + (size_t)x * 3, 4), o, 24);
    }
}
__device__ __forceinline__ void nv12_to_bgr_lane(const uint8_t *__restrict__ src, size_t sstep, int w, int h, uint8_t *__restrict__ dst, size_t dstep, bool aligned)
{
    const unsigned cells_x = (unsigned)(w + 7) >> 3, id = blockIdx.x * 256u + threadIdx.x, yy = id / cells_x, cx = id - yy * cells_x;
    const int x = 8 * (int)cx, y = 2 * (int)yy;
    if (y >= h) return;
    if (aligned && x + 8 <= w) { nv12_to_bgr_cell8(src, sstep, h, dst, dstep, x, y); return; }
    nv12_to_bgr_cell(src, sstep, w, h, dst, dstep, x, y);
    if (x + 4 < w) nv12_to_bgr_cell(src, sstep, w, h, dst, dstep, x + 4, y);
}
__global__ void __launch_bounds__(256) k_nv12_to_bgr(const uint8_t *__restrict__ src, size_t sstep, int w, int h, uint8_t *__restrict__ dst, size_t dstep, bool aligned)
{
    nv12_to_bgr_lane(src, sstep, w, h, dst, dstep, aligned);
}
// every camera of a frame in one launch (the capture threads' per-camera cvtColor, networking.cpp:45-47)
constexpr int NV12_BATCH = 64;
struct Nv12Batch { const uint8_t *src[NV12_BATCH]; uint8_t *dst[NV12_BATCH]; };
__global__ void __launch_bounds__(256) k_nv12_to_bgr_batch(Nv12Batch T, size_t sstep, int w, int h, size_t dstep, bool aligned)
{
    nv12_to_bgr_lane(T.src[blockIdx.y], sstep, w, h, T.dst[blockIdx.y], dstep, aligned);
}
static unsigned nv12_grid(int w, int h) { return (unsigned)div_up((long long)((w + 7) >> 3) * (h / 2), 256); }
int launch_nv12_to_bgr_batch(const ms_image *src, ms_image *dst, int n, hipStream_t st)
{
    for (int i0 = 0; i0 < n; i0 += NV12_BATCH) {
        const int m = std::min(NV12_BATCH, n - i0);
        Nv12Batch T{};
        size_t bits = src[0].step | dst[0].step;
        for (int i = 0; i < m; ++i) { T.src[i] = (const uint8_t *)src[i0 + i].data; T.dst[i] = (uint8_t *)dst[i0 + i].data; bits |= (size_t)T.src[i] | (size_t)T.dst[i]; }
        k_nv12_to_bgr_batch<<<dim3(nv12_grid(dst[0].cols, dst[0].rows), m), dim3(256), 0, st>>>(T, src[0].step, dst[0].cols, dst[0].rows, dst[0].step, (bits & 3) == 0);
        MS_LAUNCH_CHECK();
    }
    return MS_OK;
}
int launch_nv12_to_bgr(const ms_image &src, ms_image &dst, hipStream_t st)
{
    const size_t bits = src.step | dst.step | (size_t)src.data | (size_t)dst.data;
    k_nv12_to_bgr<<<dim3(nv12_grid(dst.cols, dst.rows)), dim3(256), 0, st>>>((const uint8_t *)src.data, src.step, dst.cols, dst.rows, (uint8_t *)dst.data, dst.step, (bits & 3) == 0);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// cuda::cvtColor(BGR2GRAY) of featurefinder::findFeatures (APP/featurefinder.cpp:34) -> RGB2GrayConvert<bidx = 0>
// (core/include/opencv2/core/cuda/detail/color_detail.hpp:97-101, :444-447): CV_DESCALE(b * 1868 + g * 9617 + r * 4899, 14).
// One lane = 4 pixels (12-byte load, one dword store).
__global__ void __launch_bounds__(256) k_bgr_to_gray(const uint8_t *__restrict__ src, size_t sstep, int w, int h, uint8_t *__restrict__ dst, size_t dstep)
{
    const int x = 4 * (blockIdx.x * BX + threadIdx.x), y = blockIdx.y * BY + threadIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *p = row_ptr<uint8_t>(src, sstep, y) + (size_t)x * 3;
    uint8_t *d = row_ptr<uint8_t>(dst, dstep, y) + x;
    const int n = min(4, w - x);
    uint8_t px[12];
    if (n == 4 && ((sstep | (size_t)src) & 3) == 0) __builtin_memcpy(px, __builtin_assume_aligned(p, 4), 12);
    else for (int i = 0; i < 3 * n; ++i) px[i] = p[i];
    unsigned q = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < n) q |= (((unsigned)px[3 * k] * 1868u + (unsigned)px[3 * k + 1] * 9617u + (unsigned)px[3 * k + 2] * 4899u + (1u << 13)) >> 14) << (8 * k);
    if (n == 4 && ((dstep | (size_t)dst) & 3) == 0) *reinterpret_cast<unsigned *>(d) = q;
    else for (int k = 0; k < n; ++k) d[k] = (uint8_t)(q >> (8 * k));
}
int launch_bgr_to_gray(const ms_image &src, ms_image &dst, hipStream_t st)
{
    k_bgr_to_gray<<<dim3(div_up(div_up(src.cols, 4), BX), div_up(src.rows, BY)), dim3(BX, BY), 0, st>>>(
        (const uint8_t *)src.data, src.step, src.cols, src.rows, (uint8_t *)dst.data, dst.step);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// cvtColor(COLOR_BGR2YUV_I420)  [imgproc/src/color.cpp:8745-8756, 9082-9160]: BT.601 fixed point (shift 20), chroma from
// the top-left pixel of each 2x2 block, planar I420 output.  One lane = 2 rows x 4 pixels (12-byte row loads).
__device__ __forceinline__ uint8_t clamp_u8(int v) { return (uint8_t)min(max(v, 0), 255); }
// the same behind an empty asm, for code that packs neighbouring clamp((a >> 20), 0, 255) into one word: they must not be fused into gfx950's v_ashr_pk_u8_i32 (see nv12_px;
// tests/test_abi.py disassembles the library for it).  Not used where the compiler does not form the pattern: the barrier costs bgr_to_i420_cell8 its byte-insert packing (2 x slower)
__device__ __forceinline__ unsigned clamp_u8_opaque(int v) { int r = min(max(v, 0), 255); asm("" : "+v"(r)); return (unsigned)r; }
constexpr int I420_BATCH = 64;
__device__ __forceinline__ void bgr_to_i420_cell(const uint8_t *__restrict__ src, size_t sstep, int w, int h, uint8_t *__restrict__ dst, int x, int y)
{
    constexpr int SH = 20, HALF = 1 << (SH - 1);
    constexpr int CRY = 269484, CGY = 528482, CBY = 102760, CRU = -155188, CGU = -305135, CBU = 460324, CGV = -385875, CBV = -74448;
    uint8_t *Y = dst, *U = dst + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
    const int n = min(4, w - x);
    unsigned yq[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint8_t *p = row_ptr<uint8_t>(src, sstep, y + r) + (size_t)x * 3;
        uint8_t px[12];
        if (n == 4 && ((sstep | (size_t)src) & 3) == 0) __builtin_memcpy(px, __builtin_assume_aligned(p, 4), 12);
        else for (int i = 0; i < 3 * n; ++i) px[i] = p[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= n) break;
            const int b = px[3 * k], g = px[3 * k + 1], rr = px[3 * k + 2];
            yq[r] |= (unsigned)clamp_u8((CRY * rr + CGY * g + CBY * b + HALF + (16 << SH)) >> SH) << (8 * k);
            if (r == 0 && (k & 1) == 0) {
                U[(size_t)(y / 2) * (w / 2) + (x + k) / 2] = clamp_u8((CRU * rr + CGU * g + CBU * b + HALF + (128 << SH)) >> SH);
                V[(size_t)(y / 2) * (w / 2) + (x + k) / 2] = clamp_u8((CBU * rr + CGV * g + CBV * b + HALF + (128 << SH)) >> SH);
            }
        }
        uint8_t *yd = Y + (size_t)(y + r) * w + x;
        if (n == 4 && (w & 3) == 0) *reinterpret_cast<unsigned *>(yd) = yq[r];
        else for (int k = 0; k < n; ++k) yd[k] = (uint8_t)(yq[r] >> (8 * k));
    }
}
// 2 rows x 8 pixels per lane when the geometry allows (width a multiple of 8, 4-byte aligned rows): two 24-byte row loads, the luma rows as
// two 8-byte stores and the four chroma samples of each plane as one dword store -- no byte stores (the 4-pixel cell writes chroma bytewise)
__device__ __forceinline__ void bgr_to_i420_cell8(const uint8_t *__restrict__ src, size_t sstep, int w, int h, uint8_t *__restrict__ dst, int x, int y)
{
    constexpr int SH = 20, HALF = 1 << (SH - 1);
    constexpr int CRY = 269484, CGY = 528482, CBY = 102760, CRU = -155188, CGU = -305135, CBU = 460324, CGV = -385875, CBV = -74448;
    uint8_t *Y = dst, *U = dst + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
    unsigned uq = 0, vq = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        uint8_t px[24];
        __builtin_memcpy(px, __builtin_assume_aligned(row_ptr<uint8_t>(src, sstep, y + r) + (size_t)x * 3, 4), 24);
        unsigned yq[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = px[3 * k], g = px[3 * k + 1], rr = px[3 * k + 2];
            yq[k >> 2] |= (unsigned)clamp_u8((CRY * rr + CGY * g + CBY * b + HALF + (16 << SH)) >> SH) << (8 * (k & 3));
            if (r == 0 && (k & 1) == 0) {
                uq |= (unsigned)clamp_u8((CRU * rr + CGU * g + CBU * b + HALF + (128 << SH)) >> SH) << (8 * (k >> 1));
                vq |= (unsigned)clamp_u8((CBU * rr + CGV * g + CBV * b + HALF + (128 << SH)) >> SH) << (8 * (k >> 1));
            }
        }
        *reinterpret_cast<uint2 *>(Y + (size_t)(y + r) * w + x) = make_uint2(yq[0], yq[1]);
    }
    *reinterpret_cast<unsigned *>(U + (size_t)(y / 2) * (w / 2) + x / 2) = uq;
    *reinterpret_cast<unsigned *>(V + (size_t)(y / 2) * (w / 2) + x / 2) = vq;
}
__global__ void __launch_bounds__(256) k_bgr_to_i420(const uint8_t *__restrict__ src, size_t sstep, int w, int h, uint8_t *__restrict__ dst, int wide)
{
    const int x = (wide ? 8 : 4) * (blockIdx.x * BX + threadIdx.x), y = 2 * (blockIdx.y * BY + threadIdx.y);
    if (x >= w || y >= h) return;
    if (wide) bgr_to_i420_cell8(src, sstep, w, h, dst, x, y);
    else bgr_to_i420_cell(src, sstep, w, h, dst, x, y);
}
// consume()'s pixel work in one pass (APP/timed.cpp:251-316): cv::resize(INTER_LINEAR) of the 8U panorama to out_w x ih -- in cuda::resize's arithmetic, the
// same operations as k_resize_linear --, the result in the middle of a black out_w x out_h frame, BGR -> I420.  One lane = one 2 x 2 block of the frame.
__global__ void __launch_bounds__(256) k_consume_i420(const uint8_t *__restrict__ src, size_t sstep, int srows, int scols, uint8_t *__restrict__ dst,
                                                      int out_w, int out_h, int ih, int y_off, float ify, float ifx)
{
    const int x = 2 * (blockIdx.x * BX + threadIdx.x), y = 2 * (blockIdx.y * BY + threadIdx.y);
    if (x >= out_w || y >= out_h) return;
    constexpr int SH = 20, HALF = 1 << (SH - 1);
    constexpr int CRY = 269484, CGY = 528482, CBY = 102760, CRU = -155188, CGU = -305135, CBU = 460324, CGV = -385875, CBV = -74448;
    uint8_t *Y = dst, *U = dst + (size_t)out_w * out_h, *V = U + (size_t)(out_w / 2) * (out_h / 2);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            int b = 0, g = 0, rr = 0;                      // the black bars
            const int yy = y + dy - y_off, xx = x + dx;
            if (yy >= 0 && yy < ih) {
                const float sx = (float)xx * ifx, sy = (float)yy * ify;
                const int x1 = f2i_rd(sx), y1 = f2i_rd(sy);
                const int x2 = x1 + 1, y2 = y1 + 1;
                const int x2r = min(x2, scols - 1), y2r = min(y2, srows - 1);
                const float w11 = ((float)x2 - sx) * ((float)y2 - sy), w12 = (sx - (float)x1) * ((float)y2 - sy);
                const float w21 = ((float)x2 - sx) * (sy - (float)y1), w22 = (sx - (float)x1) * (sy - (float)y1);
                const uint8_t *r1 = row_ptr<uint8_t>(src, sstep, y1), *r2 = row_ptr<uint8_t>(src, sstep, y2r);
                int px[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float out = 0.f;
                    out = __builtin_fmaf((float)r1[(size_t)x1 * 3 + c], w11, out);
                    out = __builtin_fmaf((float)r1[(size_t)x2r * 3 + c], w12, out);
                    out = __builtin_fmaf((float)r2[(size_t)x1 * 3 + c], w21, out);
                    out = __builtin_fmaf((float)r2[(size_t)x2r * 3 + c], w22, out);
                    px[c] = sat_u8(out);
                }
                b = px[0]; g = px[1]; rr = px[2];
            }
            Y[(size_t)(y + dy) * out_w + x + dx] = clamp_u8((CRY * rr + CGY * g + CBY * b + HALF + (16 << SH)) >> SH);
            if (dy == 0 && dx == 0) {
                U[(size_t)(y / 2) * (out_w / 2) + x / 2] = clamp_u8((CRU * rr + CGU * g + CBU * b + HALF + (128 << SH)) >> SH);
                V[(size_t)(y / 2) * (out_w / 2) + x / 2] = clamp_u8((CBU * rr + CGV * g + CBV * b + HALF + (128 << SH)) >> SH);
            }
        }
}
// Round 5: one lane = 2 rows x 4 pixels of the frame (two 2 x 2 chroma blocks).  The two BGR pixels of a tap row are ONE unaligned 8-byte read (as in the remap kernels:
// 16 reads per lane instead of 96 byte reads), the four luma bytes of a row leave as one dword, the two chroma samples of each plane as one 16-bit store.  Same fp32
// expressions in the same order as k_consume_i420 above (which stays for frame widths that are not multiples of 4); a pixel whose 8-byte window would leave its source
// row (the last two source columns) takes the per-byte taps with the clamped x2.  The rig's 3840 x 1920 canvas -> 4096 x 2048: 35 us at 1.2 TB/s before (profiles/r05_f3_report.md).
typedef unsigned pr_u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
__device__ __forceinline__ uint2 gload8_at(const uint8_t *base, unsigned off)      // uniform base + 32-bit lane offset, global address space (one VGPR of address, no 64-bit multiply-add)
{
    const pr_u32x2_a1 v = *(const __attribute__((address_space(1))) pr_u32x2_a1 *)(uintptr_t)(base + off);
    return make_uint2(v.x, v.y);
}
__global__ void __launch_bounds__(256) k_consume_i420_x4(const uint8_t *__restrict__ src, unsigned sstep, int srows, int scols, uint8_t *__restrict__ dst,
                                                         int out_w, int out_h, int ih, int y_off, float ify, float ifx)
{
    const int x = 4 * (blockIdx.x * BX + threadIdx.x), y = 2 * (blockIdx.y * BY + threadIdx.y);
    if (x >= out_w || y >= out_h) return;
    constexpr int SH = 20, HALF = 1 << (SH - 1);
    constexpr int CRY = 269484, CGY = 528482, CBY = 102760, CRU = -155188, CGU = -305135, CBU = 460324, CGV = -385875, CBV = -74448;
    uint8_t *Y = dst, *U = dst + (size_t)out_w * out_h, *V = U + (size_t)(out_w / 2) * (out_h / 2);
    float sx[4], wx1[4], wx2[4];
    int x1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sx[k] = (float)(x + k) * ifx; x1[k] = f2i_rd(sx[k]); wx2[k] = (float)(x1[k] + 1) - sx[k]; wx1[k] = sx[k] - (float)x1[k]; }
    // the lane's four pixels sample non-decreasing source columns: if the last one's 8-byte window stays inside its row, all do (one branch per lane and row, not per pixel)
    const bool lane_fast = x1[3] <= scols - 3 && x1[0] >= 0;
    unsigned uq = 0, vq = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int yy = y + r - y_off;
        unsigned yq = 0;
        if (yy >= 0 && yy < ih) {
            const float sy = (float)yy * ify;
            const int y1 = f2i_rd(sy), y2 = y1 + 1, y2r = min(y2, srows - 1);
            const float wy2 = (float)y2 - sy, wy1 = sy - (float)y1;
            int px[4][3];
            if (lane_fast) {
                const unsigned o1 = (unsigned)y1 * sstep, o2 = (unsigned)y2r * sstep;
                uint2 a[4], b[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { a[k] = gload8_at(src, o1 + 3u * (unsigned)x1[k]); b[k] = gload8_at(src, o2 + 3u * (unsigned)x1[k]); }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float w11 = wx2[k] * wy2, w12 = wx1[k] * wy2, w21 = wx2[k] * wy1, w22 = wx1[k] * wy1;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float out = 0.f;
                        out = __builtin_fmaf(rs_byte(a[k].x, a[k].y, c), w11, out);
                        out = __builtin_fmaf(rs_byte(a[k].x, a[k].y, 3 + c), w12, out);
                        out = __builtin_fmaf(rs_byte(b[k].x, b[k].y, c), w21, out);
                        out = __builtin_fmaf(rs_byte(b[k].x, b[k].y, 3 + c), w22, out);
                        px[k][c] = sat_u8(out);
                    }
                }
            } else {
                const uint8_t *r1 = src + (size_t)y1 * sstep, *r2 = src + (size_t)y2r * sstep;
                for (int k = 0; k < 4; ++k) {
                    const int x2r = min(x1[k] + 1, scols - 1);
                    const float w11 = wx2[k] * wy2, w12 = wx1[k] * wy2, w21 = wx2[k] * wy1, w22 = wx1[k] * wy1;
                    for (int c = 0; c < 3; ++c) {
                        float out = 0.f;
                        out = __builtin_fmaf((float)r1[(size_t)x1[k] * 3 + c], w11, out);
                        out = __builtin_fmaf((float)r1[(size_t)x2r * 3 + c], w12, out);
                        out = __builtin_fmaf((float)r2[(size_t)x1[k] * 3 + c], w21, out);
                        out = __builtin_fmaf((float)r2[(size_t)x2r * 3 + c], w22, out);
                        px[k][c] = sat_u8(out);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bb = px[k][0], g = px[k][1], rr = px[k][2];
                yq |= clamp_u8_opaque((CRY * rr + CGY * g + CBY * bb + HALF + (16 << SH)) >> SH) << (8 * k);
                if (r == 0 && (k & 1) == 0) {
                    uq |= clamp_u8_opaque((CRU * rr + CGU * g + CBU * bb + HALF + (128 << SH)) >> SH) << (4 * k);
                    vq |= clamp_u8_opaque((CBU * rr + CGV * g + CBV * bb + HALF + (128 << SH)) >> SH) << (4 * k);
                }
            }
        } else {          // the black bars: BGR = 0
            yq = (unsigned)((HALF + (16 << SH)) >> SH) * 0x01010101u;
            if (r == 0) uq = vq = (unsigned)((HALF + (128 << SH)) >> SH) * 0x0101u;
        }
        *reinterpret_cast<unsigned *>(Y + (size_t)(y + r) * out_w + x) = yq;
    }
    *reinterpret_cast<unsigned short *>(U + (size_t)(y / 2) * (out_w / 2) + x / 2) = (unsigned short)uq;
    *reinterpret_cast<unsigned short *>(V + (size_t)(y / 2) * (out_w / 2) + x / 2) = (unsigned short)vq;
}
int launch_consume_i420(const ms_image &src, ms_image &dst, int out_w, int out_h, int ih, int y_off, hipStream_t st)
{
    // cv::resize with an explicit dsize: fx = dsize.width / src.cols (double), the kernel gets (float)(1 / fx)  (resize.cpp:72-81,105)
    const double fx = (double)out_w / src.cols, fy = (double)ih / src.rows;
    // (the 4-pixel form: dword luma stores need out_w % 4 == 0 and a 4-byte aligned frame, its 8-byte windows at least 3 source columns)
    if ((out_w & 3) == 0 && ((size_t)dst.data & 3) == 0 && ((size_t)out_w * out_h & 3) == 0 && ((size_t)(out_w / 2) * (out_h / 2) & 1) == 0 && src.cols >= 3 &&
        (unsigned long long)src.rows * src.step < 0x100000000ull)
        k_consume_i420_x4<<<dim3(div_up(out_w / 4, BX), div_up(out_h / 2, BY)), dim3(BX, BY), 0, st>>>(
            (const uint8_t *)src.data, (unsigned)src.step, src.rows, src.cols, (uint8_t *)dst.data, out_w, out_h, ih, y_off, (float)(1.0 / fy), (float)(1.0 / fx));
    else
        k_consume_i420<<<dim3(div_up(out_w / 2, BX), div_up(out_h / 2, BY)), dim3(BX, BY), 0, st>>>(
            (const uint8_t *)src.data, src.step, src.rows, src.cols, (uint8_t *)dst.data, out_w, out_h, ih, y_off, (float)(1.0 / fy), (float)(1.0 / fx));
    MS_LAUNCH_CHECK();
    return MS_OK;
}
// the same for up to I420_BATCH frames of one geometry in one launch (the egress of a batch of panoramas: one launch instead of one per frame)
struct I420Batch { const uint8_t *src[I420_BATCH]; uint8_t *dst[I420_BATCH]; };
__global__ void __launch_bounds__(256) k_bgr_to_i420_batch(I420Batch T, size_t sstep, int w, int h, int wide)
{
    const int x = (wide ? 8 : 4) * (blockIdx.x * BX + threadIdx.x), y = 2 * (blockIdx.y * BY + threadIdx.y);
    if (x >= w || y >= h) return;
    if (wide) bgr_to_i420_cell8(T.src[blockIdx.z], sstep, w, h, T.dst[blockIdx.z], x, y);
    else bgr_to_i420_cell(T.src[blockIdx.z], sstep, w, h, T.dst[blockIdx.z], x, y);
}
int launch_bgr_to_i420_batch(const ms_image *src, ms_image *dst, int n, hipStream_t st)
{
    for (int i0 = 0; i0 < n; i0 += I420_BATCH) {
        const int m = std::min(I420_BATCH, n - i0);
        I420Batch T{};
        bool wide = (src[0].cols & 7) == 0 && (src[0].step & 3) == 0 && (((size_t)src[0].cols * src[0].rows) & 7) == 0 && (((size_t)(src[0].cols / 2) * (src[0].rows / 2)) & 3) == 0;
        for (int i = 0; i < m; ++i) {
            T.src[i] = (const uint8_t *)src[i0 + i].data; T.dst[i] = (uint8_t *)dst[i0 + i].data;
            wide = wide && ((size_t)T.src[i] & 3) == 0 && ((size_t)T.dst[i] & 7) == 0;
        }
        k_bgr_to_i420_batch<<<dim3(div_up(div_up(src[0].cols, wide ? 8 : 4), BX), div_up(src[0].rows / 2, BY), m), dim3(BX, BY), 0, st>>>(T, src[0].step, src[0].cols, src[0].rows, wide ? 1 : 0);
        MS_LAUNCH_CHECK();
    }
    return MS_OK;
}
int launch_bgr_to_i420(const ms_image &src, ms_image &dst, hipStream_t st)
{
    const bool wide = (src.cols & 7) == 0 && ((src.step | (size_t)src.data) & 3) == 0 && ((size_t)dst.data & 7) == 0 &&
                      (((size_t)src.cols * src.rows) & 7) == 0 && (((size_t)(src.cols / 2) * (src.rows / 2)) & 3) == 0;      // plane offsets keep the vector stores aligned
    k_bgr_to_i420<<<dim3(div_up(div_up(src.cols, wide ? 8 : 4), BX), div_up(src.rows / 2, BY)), dim3(BX, BY), 0, st>>>(
        (const uint8_t *)src.data, src.step, src.cols, src.rows, (uint8_t *)dst.data, wide ? 1 : 0);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

// ------------------------------------------------------------------------------------------------
// custom_resize  [APP/resize.cu:9-27]
__global__ void __launch_bounds__(256) k_custom_resize(int tx, int ty, int cols, int rows, const float *__restrict__ in, size_t istep,
                                                       float *__restrict__ out, size_t ostep)
{
    XY_GUARD(tx, ty)
    const int left = x * (cols - 1) / tx, top = y * (rows - 1) / ty;
    const float uu = ((float)x) * (float)(cols - 1) / (float)tx - (float)left;
    const float vv = ((float)y) * (float)(rows - 1) / (float)ty - (float)top;
    const float *r0 = row_ptr<float>(in, istep, top), *r1 = row_ptr<float>(in, istep, top + 1);
    float r = ((1.f - uu) * (1.f - vv)) * r0[left];
    r = __builtin_fmaf(uu * (1.f - vv), r0[left + 1], r);
    r = __builtin_fmaf((1.f - uu) * vv, r1[left], r);
    r = __builtin_fmaf(uu * vv, r1[left + 1], r);
    row_ptr<float>(out, ostep, y)[x] = r;
}
int launch_custom_resize(const ms_image &in, ms_image &out, hipStream_t st)
{
    if (in.rows < 2 || in.cols < 2) return fail(MS_ERR_INVALID, "ms_custom_resize_32f: input must be at least 2x2");
    k_custom_resize<<<grid2d(out.cols, out.rows), dim3(BX, BY), 0, st>>>(out.cols, out.rows, in.cols, in.rows, (const float *)in.data, in.step, (float *)out.data, out.step);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

}  // namespace ms
