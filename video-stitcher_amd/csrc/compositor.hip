// compositor.hip -- the per-frame fast path: stitch_one (APP/timed.cpp:123-152) as a short sequence of
// batched HIP launches over ALL views (and all frames of a batch) at once.
//
//   reference, per view                                   here, per frame batch
//   -----------------------------------------------       ------------------------------------------
//   remap -> convertTo(gain) -> [remap(mesh)]             k_warp: remap+gain(+CPW stage)+reflect pad,
//   copyMakeBorder(REFLECT) -> convertTo(16S)                     writes Gaussian level 0 (planar u8)
//   nb x pyrDown                                          k_down x nb (all views x 3 planes)
//   nb x (pyrUp, subtract), (nb+1) x addSrcWeight         k_blend x (nb+1), coarse -> fine: Laplacian on
//   (nb+1) x normalize, nb x (pyrUp, add)                         the fly, weighted sum over the views
//   compare, compare, setTo, convertTo(out), 12 memsets           covering the pixel (gather, no RMW), divide
//                                                                 by the frame-invariant weight sum, add the
//                                                                 expanded coarser level, final mask + 8U/16S
//
// HBM layout (all planar, per frame): level 0 of every view as u8 (it IS the widened 8U image),
// levels 1..nb as int16; collapsed pano levels 1..nb as int16; weights/denominators fp32 (static).
// Results are bit-identical to the reference's kernel sequence (oracle: oracle/ms_oracle_blend.c):
// the 16S pyramids are evaluated in exact integer arithmetic, accumulate/normalise keep the fp32
// multiply/divide + truncation, and int16 accumulation wraps exactly like `short +=`.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>
#include "common.hpp"
#include "launchers.hpp"
#include "descs.hpp"

namespace ms {

// ------------------------------------------------------------------------------------------------
// bilinear sample of an interleaved 8UC3 image, constant-0 border (remap.cu:56-68 + filters.hpp:90-114)
__device__ __forceinline__ void sample3(const uint8_t *__restrict__ src, unsigned sstep, int srows, int scols,
                                        float xc, float yc, float out[3])
{
    const int x1 = f2i_rd(xc), y1 = f2i_rd(yc);
    const int x2 = (int)((unsigned)x1 + 1u), y2 = (int)((unsigned)y1 + 1u);
    const float wx2 = (float)x2 - xc, wx1 = xc - (float)x1;
    const float wy2 = (float)y2 - yc, wy1 = yc - (float)y1;
    const float w[4] = {wx2 * wy2, wx1 * wy2, wx2 * wy1, wx1 * wy1};
    const int xs[4] = {x1, x2, x1, x2};
    const int ys[4] = {y1, y1, y2, y2};
    out[0] = out[1] = out[2] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool inb = xs[t] >= 0 && xs[t] < scols && ys[t] >= 0 && ys[t] < srows;
        const uint8_t *p = src + (size_t)(inb ? ys[t] : 0) * sstep + (size_t)(inb ? xs[t] : 0) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = __builtin_fmaf(inb ? (float)p[c] : 0.f, w[t], out[c]);
    }
}

// CPW stage 1 (only when enable_cpw): images[i] = gain(remap(full_img, x_map, y_map))  timed.cpp:90-94
__global__ void __launch_bounds__(256) k_remap_gain(const ViewDesc *__restrict__ views, int n_views, SrcTable src,
                                                    int src_rows, int src_cols, uint8_t *__restrict__ stage, long long stage_stride)
{
    const int v = blockIdx.z % n_views, f = blockIdx.z / n_views;
    const ViewDesc &V = views[v];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= V.aw || y >= V.ah) return;
    const float xc = V.xmap[(size_t)y * V.map_pitch + x], yc = V.ymap[(size_t)y * V.map_pitch + x];
    float o[3];
    sample3(src.p[blockIdx.z], src.step[blockIdx.z], src_rows, src_cols, xc, yc, o);
    uint8_t *d = stage + (size_t)f * stage_stride + V.s1_off + (size_t)y * V.s1_pitch + (size_t)x * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = sat_u8(__builtin_fmaf(V.gain, (float)sat_u8(o[c]), 0.f));
}

// Gaussian level 0 of every view: (remap -> gain) or (remap through the mesh of the stage-1 image),
// BORDER_REFLECT pad folded into the source index, planar u8 output.
// FIX: cv::remap's CPU arithmetic for the projection remap (ms_config.cpu_flavour_remap, the reference's CPU pipeline of BASELINE configs[0])
template <bool CPW, bool FIX = false>
__global__ void __launch_bounds__(256) k_warp(const ViewDesc *__restrict__ views, int n_views, SrcTable src, int src_rows, int src_cols,
                                              MeshTable mesh, const uint8_t *__restrict__ stage, long long stage_stride,
                                              uint8_t *__restrict__ g0, long long g0_stride)
{
    const int v = blockIdx.z % n_views, f = blockIdx.z / n_views;
    const ViewDesc &V = views[v];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= V.pw || y >= V.ph) return;
    const int ax = reflect_idx(x - V.left, V.aw), ay = reflect_idx(y - V.top, V.ah);
    float o[3];
    uint8_t r[3];
    if (CPW) {
        const float xc = mesh.x[v][(size_t)ay * mesh.pitch[v] + ax], yc = mesh.y[v][(size_t)ay * mesh.pitch[v] + ax];
        sample3(stage + (size_t)f * stage_stride + V.s1_off, (unsigned)V.s1_pitch, V.ah, V.aw, xc, yc, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = sat_u8(o[c]);
    } else {
        const float xc = V.xmap[(size_t)ay * V.map_pitch + ax], yc = V.ymap[(size_t)ay * V.map_pitch + ax];
        if (FIX) {
            uint8_t q[3];
            remap_fixpt<3>(src.p[blockIdx.z], src.step[blockIdx.z], src_rows, src_cols, xc, yc, q);
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = sat_u8(__builtin_fmaf(V.gain, (float)q[c], 0.f));
        } else {
            sample3(src.p[blockIdx.z], src.step[blockIdx.z], src_rows, src_cols, xc, yc, o);
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = sat_u8(__builtin_fmaf(V.gain, (float)sat_u8(o[c]), 0.f));
        }
    }
    const LevelDesc &L = V.lv[0];
    uint8_t *d = g0 + (size_t)f * g0_stride + L.off + (size_t)y * L.pitch + x;
    const size_t plane = (size_t)L.h * L.pitch;
    d[0] = r[0]; d[plane] = r[1]; d[2 * plane] = r[2];
}

// pyrDown of one plane of one view: level l -> l+1 (pyr_down.cu:55-174 in exact integer form).  Every Gaussian level of a view is a convex
// combination of 8-bit pixels (all values in [0, 255]): the view pyramids are stored as bytes, half the traffic of the reference's 16S.
__global__ void __launch_bounds__(256) k_down(const ViewDesc *__restrict__ views, int n_views, int l,
                                              const uint8_t *__restrict__ gin, long long in_stride,
                                              uint8_t *__restrict__ gout, long long out_stride)
{
    const int z = blockIdx.z, c = z % 3, v = (z / 3) % n_views, f = z / (3 * n_views);
    const LevelDesc &Li = views[v].lv[l], &Lo = views[v].lv[l + 1];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= Lo.w || y >= Lo.h) return;
    const uint8_t *in = gin + (size_t)f * in_stride + Li.off + (size_t)c * Li.h * Li.pitch;
    const int sy = 2 * y, sx = 2 * x;
    int ry[5], cx[5];
    if (Li.h >= 3 && Li.w >= 3) {     // |overshoot| <= 2 < len: BORDER_REFLECT_101 without the integer modulo
        const int lr = Li.h - 1, lc = Li.w - 1;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int r = abs(sy - 2 + k), q = abs(sx - 2 + k);
            ry[k] = r > lr ? 2 * lr - r : r;
            cx[k] = q > lc ? 2 * lc - q : q;
        }
    } else {
        ry[0] = r101_low(sy - 2, Li.h); ry[1] = r101_low(sy - 1, Li.h); ry[2] = sy;
        ry[3] = r101_high(sy + 1, Li.h); ry[4] = r101_high(sy + 2, Li.h);
#pragma unroll
        for (int k = 0; k < 5; ++k) cx[k] = r101(sx - 2 + k, Li.w);
    }
    const int wv[5] = {1, 4, 6, 4, 1};
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint8_t *r = in + (size_t)ry[j] * Li.pitch;
#pragma unroll
        for (int k = 0; k < 5; ++k) acc += wv[j] * wv[k] * (int)r[cx[k]];
    }
    gout[(size_t)f * out_stride + Lo.off + (size_t)c * Lo.h * Lo.pitch + (size_t)y * Lo.pitch + x] = (uint8_t)rne_shift(acc, 8);       // <= 255
}

// 16 bytes from a 4-byte aligned address as one global_load_dwordx4
__device__ __forceinline__ uint4 load16_a4(const void *p)
{
    uint4 r;
    __builtin_memcpy(&r, __builtin_assume_aligned(p, 4), 16);
    return r;
}

// ---- fused tail of the Gaussian pyramids: down steps l0 .. nb-1 of one (frame, view, colour plane) in one workgroup --------
// The coarse levels are tiny (config 2: 148x80 and smaller): a launch per level is pure latency.  Level l0 is copied into LDS as
// bytes (every Gaussian value is in [0,255]), each further level is computed LDS -> LDS and also stored to the pyramid
// buffer for the band kernels.  Same exact integer arithmetic as k_down.
// Wide views are cut into strips of `sw` columns of the coarsest level; a strip recomputes the few halo columns it needs
// at the intermediate levels and stores only the columns it owns.  sw = TAIL_STRIP (16) for one or two frames per call, where the launch has to be spread over
// the chip; TAIL_STRIP_BATCH (64: a whole ordinary view per workgroup) for batches (round 5): with 16-column strips a 32-frame launch was 2 208 workgroups of
// 1.5 KB each, every one of them a chain of cold round trips and barriers -- 45 us at 1 TB/s; a third of the workgroups with three times the work per step
// finish in one round of the chip.
constexpr int TAIL_STRIP = 16, TAIL_STRIP_BATCH = 64;
__host__ __device__ inline void tail_range(const int *w, int l0, int nb, int strip, int sw, int *a, int *b)
{
    a[nb] = strip * sw; b[nb] = min(a[nb] + sw, w[nb]);
    for (int l = nb - 1; l >= l0; --l) { a[l] = max(2 * a[l + 1] - 2, 0); b[l] = min(2 * b[l + 1] + 2, w[l]); }
}
__global__ void __launch_bounds__(1024) k_down_tail(const ViewDesc *__restrict__ views, int n_views, int l0, int nb, int max_strips, int sw,
                                                   uint8_t *__restrict__ gl, long long gl_stride, unsigned own_mask)
{
    extern __shared__ uint8_t s_lv[];
    const int strip = blockIdx.x % max_strips, z = blockIdx.x / max_strips;
    const int c = z % 3, v = (z / 3) % n_views, f = z / (3 * n_views);
    if (!((own_mask >> v) & 1u)) return;
    const ViewDesc &V = views[v];
    int w[MAX_LEVELS + 1], a[MAX_LEVELS + 1], b[MAX_LEVELS + 1];
    for (int l = l0; l <= nb; ++l) w[l] = V.lv[l].w;
    if (strip * sw >= w[nb]) return;
    tail_range(w, l0, nb, strip, sw, a, b);
    uint8_t *base = gl + (size_t)f * gl_stride;
    const int tx = threadIdx.x, ty = threadIdx.y;           // block 64 x 4 (64 x 16 in live mode: the same strip on four times the lanes)
    const int nthr = (int)(blockDim.x * blockDim.y);
    uint8_t *cur = s_lv;
    {
        const LevelDesc &La = V.lv[l0];
        const uint8_t *in = base + La.off + (size_t)c * La.h * La.pitch + a[l0];
        const int wl = b[l0] - a[l0];
        // 8 columns per 8-byte read (a[l0] is even: the address is 2-byte aligned, the hardware takes it), four reads in flight per lane before the first LDS write
        const int nchunk = (wl + 7) >> 3, total = La.h * nchunk, tid = ty * 64 + tx;
        for (int i0 = tid; i0 < total; i0 += 4 * nthr) {
            uint2 q[4];
            int yy[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + nthr * u, total - 1);
                yy[u] = i / nchunk; cc[u] = i - yy[u] * nchunk;
                __builtin_memcpy(&q[u], in + (size_t)yy[u] * La.pitch + 8 * cc[u], 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + nthr * u >= total) continue;
                const unsigned d[2] = {q[u].x, q[u].y};
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (8 * cc[u] + k < wl) cur[yy[u] * wl + 8 * cc[u] + k] = (uint8_t)((d[k >> 2] >> (8 * (k & 3))) & 0xffu);
            }
        }
    }
    __syncthreads();
    for (int l = l0; l < nb; ++l) {
        const LevelDesc &Li = V.lv[l], &Lo = V.lv[l + 1];
        const int wi = b[l] - a[l], wo = b[l + 1] - a[l + 1];
        uint8_t *nxt = cur + ((wi * Li.h + 15) & ~15);
        uint8_t *out = base + Lo.off + (size_t)c * Lo.h * Lo.pitch;
        const bool big = Li.h >= 3 && Li.w >= 3;           // |overshoot| <= 2 < len: BORDER_REFLECT_101 without the integer modulo
        const int lr = Li.h - 1, lc = Li.w - 1;
        const int own_a = min((strip * sw) << (nb - l - 1), Lo.w);
        const int own_b = (strip * sw + sw >= w[nb]) ? Lo.w : min(((strip + 1) * sw) << (nb - l - 1), Lo.w);
        // flat index over the strip's outputs: every lane works (strips are only 16..40 columns wide); i -> (y, xo) with one fp32
        // multiply: (i + 0.5) / wo is at least 0.5 / 64 away from an integer, far more than the fp32 error for i < 2^16
        const float rwo = 1.0f / (float)wo;
        for (int i = ty * 64 + tx; i < wo * Lo.h; i += nthr) {
            const int y = (int)(((float)i + 0.5f) * rwo), xo = i - y * wo;
            const int x = a[l + 1] + xo;
            int ry[5], cx[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int r = 2 * y - 2 + k, q = 2 * x - 2 + k;
                ry[k] = (big ? (abs(r) > lr ? 2 * lr - abs(r) : abs(r)) : r101(r, Li.h)) * wi;
                cx[k] = (big ? (abs(q) > lc ? 2 * lc - abs(q) : abs(q)) : r101(q, Li.w)) - a[l];
            }
            // separable: 5 horizontal sums (1 4 6 4 1), then the vertical one -- same integers as the 25-term sum
            int hs[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const uint8_t *r = cur + ry[j];
                hs[j] = (int)r[cx[0]] + (int)r[cx[4]] + 4 * ((int)r[cx[1]] + (int)r[cx[3]]) + 6 * (int)r[cx[2]];
            }
            const int o = rne_shift(hs[0] + hs[4] + 4 * (hs[1] + hs[3]) + 6 * hs[2], 8);
            nxt[i] = (uint8_t)o;
            if (x >= own_a && x < own_b) out[(size_t)y * Lo.pitch + x] = (uint8_t)o;
        }
        __syncthreads();
        cur = nxt;
    }
}

// ---- vectorised pyrDown: 2 output rows x 4 output columns per thread --------------------------------
// Needs input width % 8 == 0 (true for every level l <= nb-3 because padded views are multiples of 2^nb).
// One input row contributes columns [8t-2, 8t+8]: a 16-byte (int16) / 8-byte (u8) aligned body plus a
// 2-pixel left and 1-pixel right halo; BORDER_REFLECT_101 only touches the first/last thread of a row,
// where the mirrored columns are already inside the body (col -2 -> 2, -1 -> 1, w -> w-2).
// Raw row fetch for columns [8t-2, 8t+8] of one input row, issued unconditionally so that the
// rows' loads are in flight together; the reflect selection happens after, in registers.
//   ONE 16-byte load at byte 8t-4 (dword aligned) covers bytes 8t-4 .. 8t+11
// (wave-wide gathers cost per instruction, so fewer, wider loads win).  Reads may run a few bytes past the end of
// the last row of a plane: planes are contiguous and the buffers carry 64 bytes of slack.
struct Row11u8 { uint4 b; };
__device__ __forceinline__ Row11u8 fetch_row11(const uint8_t *__restrict__ row, int t, int w)
{
    Row11u8 o;      // at t == 0 the first dword lies before the row (the previous row's padding, the previous plane, or the buffer's lead): down_row replaces it
    __builtin_memcpy(&o.b, __builtin_assume_aligned(row + (8 * t - 4), 4), 16);
    return o;
}

}  // namespace ms
#include <type_traits>
#include "tile_kernels.hpp"
namespace ms {

// pyrUp of a 2x2 quad whose top-left fine pixel is (2i, 2j), from a planar coarse level (T = uint8_t: a view's Gaussian level,
// int16_t: a collapsed band) (pyr_up.cu:55-145 in exact integer form).  out[0..3] = (2i,2j) (2i,2j+1) (2i+1,2j) (2i+1,2j+1).
template <typename T>
__device__ __forceinline__ void up_quad(const T *__restrict__ cs, int cpitch, int ch, int cw, int i, int j, int out[4])
{
    const int r0 = pu_idx(i - 1, ch), r1 = pu_idx(i, ch), r2 = pu_idx(i + 1, ch);
    const int c0 = pu_idx(j - 1, cw), c1 = pu_idx(j, cw), c2 = pu_idx(j + 1, cw);
    const T *p0 = cs + (size_t)r0 * cpitch, *p1 = cs + (size_t)r1 * cpitch, *p2 = cs + (size_t)r2 * cpitch;
    const int a00 = p0[c0], a01 = p0[c1], a02 = p0[c2];
    const int a10 = p1[c0], a11 = p1[c1], a12 = p1[c2];
    const int a20 = p2[c0], a21 = p2[c1], a22 = p2[c2];
    // horizontal: even column (1,6,1), odd column (4,4)
    const int he0 = a00 + 6 * a01 + a02, ho0 = 4 * (a01 + a02);
    const int he1 = a10 + 6 * a11 + a12, ho1 = 4 * (a11 + a12);
    const int he2 = a20 + 6 * a21 + a22, ho2 = 4 * (a21 + a22);
    out[0] = rne_shift(he0 + 6 * he1 + he2, 6);
    out[1] = rne_shift(ho0 + 6 * ho1 + ho2, 6);
    out[2] = rne_shift(4 * (he1 + he2), 6);
    out[3] = rne_shift(4 * (ho1 + ho2), 6);
}

// coarsest band: C_nb = trunc( sum_v trunc(G_{v,nb} * w_{v,nb}) / den_nb )   (L_nb = G_nb)
template <int MODE>      // 0 = whole frame, 1 = partial sums of the owned views, 2 = finish from partial sums (view sharding)
__global__ void __launch_bounds__(256) k_blend_top(const ViewDesc *__restrict__ views, PanoDesc P,
                                                   const uint8_t *__restrict__ g0, long long g0_stride,
                                                   const uint8_t *__restrict__ gl, long long gl_stride,
                                                   int16_t *__restrict__ cl, long long cl_stride, ShardArgs S)
{
    const int l = P.nb, f = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= P.qw[l] || y >= P.qh[l]) return;
    int16_t acc[3] = {0, 0, 0};
    const size_t pplane = (size_t)P.qh[l] * P.qpitch[l], po = (size_t)f * S.pstride + P.poff[l] + (size_t)y * P.qpitch[l] + x;
    if (MODE == 2) {
        for (int s = 0; s < S.n_parts; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = (int16_t)(acc[c] + S.part[s][po + c * pplane]);
    }
    for (int v = 0; v < P.n_views && MODE != 2; ++v) {
        if (MODE == 1 && !((S.own_mask >> v) & 1u)) continue;
        const LevelDesc &L = views[v].lv[l];
        const int lx = x - L.x_tl, ly = y - L.y_tl;
        if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) continue;
        const float w = L.wgt[(size_t)ly * L.wpitch + lx];
        const size_t plane = (size_t)L.h * L.pitch, o = (size_t)ly * L.pitch + lx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int g = (l == 0) ? (int)g0[(size_t)f * g0_stride + L.off + c * plane + o]
                                   : (int)gl[(size_t)f * gl_stride + L.off + c * plane + o];
            acc[c] = (int16_t)(acc[c] + trunc_s16((float)g * w));
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) S.pout[po + c * pplane] = acc[c];
        return;
    }
    const float den = P.den[l][(size_t)y * P.dpitch[l] + x];
    const size_t plane = (size_t)P.qh[l] * P.qpitch[l];
    int16_t *d = cl + (size_t)f * cl_stride + P.coff[l] + (size_t)y * P.qpitch[l] + x;
    const DivBy div(den);
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c * plane] = trunc_s16(div((float)acc[c]));
}

// band l < nb, one 2x2 quad per thread:
//   N_l = trunc( sum_v trunc( sat(G_{v,l} - up(G_{v,l+1})) * w_{v,l} ) / den_l ),  C_l = sat(up(C_{l+1}) + N_l)
// L0 (l == 0): fine level is the u8 level-0 buffer and the result goes to the outputs
//   (mask = gpu_dst_mask_, setTo(0, !mask), convertTo 16S out, convertTo 8U canvas).
template <bool L0, int MODE>
__global__ void __launch_bounds__(256) k_blend(const ViewDesc *__restrict__ views, PanoDesc P, int l,
                                               const uint8_t *__restrict__ g0, long long g0_stride,
                                               const uint8_t *__restrict__ gl, long long gl_stride,
                                               int16_t *__restrict__ cl, long long cl_stride, OutTable out, ShardArgs S)
{
    const int f = blockIdx.z;
    const int qx = blockIdx.x * 64 + threadIdx.x, qy = blockIdx.y * 4 + threadIdx.y;
    if (2 * qx >= P.qw[l] || 2 * qy >= P.qh[l]) return;
    const int x0 = 2 * qx, y0 = 2 * qy;
    int16_t acc[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0;
    const size_t pplane = (size_t)P.qh[l] * P.qpitch[l], po = (size_t)f * S.pstride + P.poff[l] + (size_t)y0 * P.qpitch[l] + x0;
    const int pq[4] = {0, 1, P.qpitch[l], P.qpitch[l] + 1};
    if (MODE == 2) {
        for (int s = 0; s < S.n_parts; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c][k] = (int16_t)(acc[c][k] + S.part[s][po + c * pplane + pq[k]]);
    }

    for (int v = 0; v < P.n_views && MODE != 2; ++v) {
        if (MODE == 1 && !((S.own_mask >> v) & 1u)) continue;
        const LevelDesc &L = views[v].lv[l];
        const int lx = x0 - L.x_tl, ly = y0 - L.y_tl;
        if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) continue;   // rects are even-aligned below level nb
        const LevelDesc &C = views[v].lv[l + 1];
        const float *wp = L.wgt + (size_t)ly * L.wpitch + lx;
        const float w[4] = {wp[0], wp[1], wp[L.wpitch], wp[L.wpitch + 1]};
        const size_t fplane = (size_t)L.h * L.pitch, cplane = (size_t)C.h * C.pitch;
        const size_t fo = (size_t)ly * L.pitch + lx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int up[4];
            up_quad(gl + (size_t)f * gl_stride + C.off + c * cplane, C.pitch, C.h, C.w, ly >> 1, lx >> 1, up);
            const uint8_t *p = (L0 ? g0 + (size_t)f * g0_stride : gl + (size_t)f * gl_stride) + L.off + c * fplane + fo;
            const int g[4] = {p[0], p[1], p[L.pitch], p[L.pitch + 1]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lap = sat_s16(g[k] - (int)sat_s16(up[k]));
                acc[c][k] = (int16_t)(acc[c][k] + trunc_s16((float)lap * w[k]));
            }
        }
    }

    if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) S.pout[po + c * pplane + pq[k]] = acc[c][k];
        return;
    }
    const float *dp = P.den[l] + (size_t)y0 * P.dpitch[l] + x0;
    const float den[4] = {dp[0], dp[1], dp[P.dpitch[l]], dp[P.dpitch[l] + 1]};
    const size_t cplane = (size_t)P.qh[l + 1] * P.qpitch[l + 1];
    const int16_t *cc = cl + (size_t)f * cl_stride + P.coff[l + 1];
    int res[3][4];
    const DivBy div[4] = {DivBy(den[0]), DivBy(den[1]), DivBy(den[2]), DivBy(den[3])};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int up[4];
        up_quad(cc + c * cplane, P.qpitch[l + 1], P.qh[l + 1], P.qw[l + 1], qy, qx, up);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            res[c][k] = sat_s16((int)sat_s16(up[k]) + (int)trunc_s16(div[k]((float)acc[c][k])));
    }

    if (!L0) {
        const size_t plane = (size_t)P.qh[l] * P.qpitch[l];
        int16_t *d = cl + (size_t)f * cl_stride + P.coff[l] + (size_t)y0 * P.qpitch[l] + x0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d[c * plane] = (int16_t)res[c][0]; d[c * plane + 1] = (int16_t)res[c][1];
            d[c * plane + P.qpitch[l]] = (int16_t)res[c][2]; d[c * plane + P.qpitch[l] + 1] = (int16_t)res[c][3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + (k & 1), y = y0 + (k >> 1);
            if (x >= P.fw || y >= P.fh) continue;               // dst_rc = unpadded ROI (blenders.cpp:762)
            const bool m = P.mask[(size_t)y * P.mask_pitch + x] != 0;
            const int b = m ? res[0][k] : 0, g = m ? res[1][k] : 0, r = m ? res[2][k] : 0;
            if (out.p16[f]) {
                int16_t *d = (int16_t *)((char *)out.p16[f] + (size_t)y * out.step16[f]) + 3 * x;
                d[0] = (int16_t)b; d[1] = (int16_t)g; d[2] = (int16_t)r;
            }
            if (out.p8[f]) {
                const int cx = x + P.canvas_x, cy = y + P.canvas_y;
                if (cx >= 0 && cx < P.out_w && cy >= 0 && cy < P.out_h) {
                    uint8_t *d = out.p8[f] + (size_t)cy * out.step8[f] + 3 * cx;
                    d[0] = (uint8_t)min(max(b, 0), 255); d[1] = (uint8_t)min(max(g, 0), 255); d[2] = (uint8_t)min(max(r, 0), 255);
                }
            }
        }
    }
}

// ---- fused tail of the band chain: the coarse bands nb .. t of one (frame, column strip) in one workgroup -------------------
// The coarsest bands are tiny (config 2: 120x20, 240x40, 480x80): a launch per band is pure latency (3 x 10-17 us), and at one frame per
// call these launches are a third of the frame time.  Each band is computed from its views exactly as k_blend_top / k_blend do; the
// collapsed bands above t stay in LDS (three int16 planes of the strip plus the columns pyrUp needs), only band t is stored -- it is all
// the next (vectorised) band kernel reads.  Column ranges of a strip: band t owns [a_t, b_t); band l+1 needs [a_l/2 - 1, (b_l-1)/2 + 1].
constexpr int BTAIL_W = 16;             // columns of band t per strip (the plans' default)
__host__ __device__ inline void btail_range(const int *qw, int t, int nb, int strip, int bw, int *a, int *b)
{
    a[t] = strip * bw; b[t] = min(a[t] + bw, qw[t]);
    for (int l = t; l < nb; ++l) {
        a[l + 1] = max(a[l] / 2 - 1, 0) & ~1;
        b[l + 1] = min(((b[l] - 1) / 2 + 2 + 1) & ~1, qw[l + 1]);
    }
}
template <bool LIVE>      // LIVE: one or two frames per call (1024 lanes per workgroup, latency matters); otherwise the chip is full and skipped work matters
__global__ void __launch_bounds__(1024) k_blend_tail(const ViewDesc *__restrict__ views, PanoDesc P, int t,
                                                    const uint8_t *__restrict__ gl, long long gl_stride,
                                                    int16_t *__restrict__ cl, long long cl_stride, int strip0, int bw)
{
    // Two phases.  (1) The normalised Laplacian term of EVERY band nb .. t of the strip, D_l = trunc(sum_v trunc(L_v * w_v) / den) (the coarsest: trunc(sum_v trunc(G_v * w_v) / den)),
    // depends on the views only, not on the collapsed band below it: all bands' items are spread over the workgroup at once (one band per wave: item ranges are padded to 64, so the
    // descriptors stay scalar), into LDS.  (2) The collapse C_l = sat(pyrUp(C_{l+1}) + D_l) runs band by band over LDS alone.  The serial chain of global round trips is that of ONE
    // band instead of nb - t + 1 (one frame per call: 16.3 -> 11.0 us; the arithmetic and its order per pixel are unchanged).
    extern __shared__ int16_t s_c[];
    __shared__ int s_a[MAX_LEVELS + 1], s_w[MAX_LEVELS + 1], s_off[MAX_LEVELS + 1], s_first[MAX_LEVELS + 2];
    const int nb = P.nb, f = blockIdx.z, strip = blockIdx.x + strip0, c = blockIdx.y;      // one colour plane per workgroup (strip0: first strip of a column window)
    const int tid = (int)threadIdx.y * 64 + (int)threadIdx.x, nthr = (int)(blockDim.x * blockDim.y);      // 256 lanes per workgroup, 1024 in live mode
    if (tid == 0) {
        int a[MAX_LEVELS + 1], b[MAX_LEVELS + 1];
        btail_range(P.qw, t, nb, strip, bw, a, b);
        int off = 0, first = 0;
        for (int l = nb; l >= t; --l) {
            const int lw = b[l] - a[l], lh = P.qh[l];
            s_a[l] = a[l]; s_w[l] = lw; s_off[l] = off; s_first[l] = first;
            off += (lw * lh + 7) & ~7;
            first += ((l == nb ? lw * lh : (lw >> 1) * (lh >> 1)) + 63) & ~63;
        }
        s_first[t - 1] = first;
    }
    __syncthreads();
    const uint8_t *glf = gl + (size_t)f * gl_stride;
    const int total = s_first[t - 1];
    for (int i = tid; i < total; i += nthr) {
        const int ib = __builtin_amdgcn_readfirstlane(i);      // (a wave's 64 items belong to one band)
        int l = nb;
        while (l > t && ib >= s_first[l - 1]) --l;
        l = __builtin_amdgcn_readfirstlane(l);
        const int j = i - __builtin_amdgcn_readfirstlane(s_first[l]);
        const int lw = __builtin_amdgcn_readfirstlane(s_w[l]), al = __builtin_amdgcn_readfirstlane(s_a[l]), lh = P.qh[l];
        int16_t *cur = s_c + __builtin_amdgcn_readfirstlane(s_off[l]);
        if (l == nb) {
            // coarsest band: C = trunc( sum_v trunc(G * w) / den ), per pixel (view rects need not be even-aligned here)
            if (j >= lw * lh) continue;
            const int y = j / lw, x = al + (j - y * lw);
            const float dn = P.den[l][(size_t)y * P.dpitch[l] + x];
            int16_t acc = 0;
            for (int v = 0; v < P.n_views; ++v) {
                const LevelDesc &L = views[v].lv[l];
                const int lx = x - L.x_tl, ly = y - L.y_tl;
                if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) continue;
                const float w = L.wgt[(size_t)ly * L.wpitch + lx];
                acc = (int16_t)(acc + trunc_s16((float)(int)glf[L.off + (size_t)c * L.h * L.pitch + (size_t)ly * L.pitch + lx] * w));
            }
            const DivBy div(dn);
            cur[j] = trunc_s16(div((float)acc));
        } else {
            // band l < nb: Laplacian of every view formed on the fly, accumulate, normalise; 2x2 quads
            const int qwl = lw >> 1, qhl = lh >> 1;
            if (j >= qwl * qhl) continue;
            const int qy = j / qwl, qxl = j - qy * qwl;
            const int x0 = al + 2 * qxl, y0 = 2 * qy;
            const float *dp = P.den[l] + (size_t)y0 * P.dpitch[l] + x0;
            const float dn[4] = {dp[0], dp[1], dp[P.dpitch[l]], dp[P.dpitch[l] + 1]};
            int16_t acc[4] = {0, 0, 0, 0};
            for (int v = 0; v < P.n_views; ++v) {
                const LevelDesc &L = views[v].lv[l];
                const int lx = x0 - L.x_tl, ly = y0 - L.y_tl;
                if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) continue;   // rects are even-aligned below level nb
                const LevelDesc &C = views[v].lv[l + 1];
                const float *wp = L.wgt + (size_t)ly * L.wpitch + lx;
                const float w[4] = {wp[0], wp[1], wp[L.wpitch], wp[L.wpitch + 1]};
                // LIVE: no early-out on four zero weights -- the view's taps are requested together with its weights, one round trip per view ((short)(L * 0) == 0 anyway)
                if (!LIVE && w[0] == 0.f && w[1] == 0.f && w[2] == 0.f && w[3] == 0.f) continue;
                int up[4];
                up_quad(glf + C.off + (size_t)c * C.h * C.pitch, C.pitch, C.h, C.w, ly >> 1, lx >> 1, up);
                const uint8_t *p = glf + L.off + (size_t)c * L.h * L.pitch + (size_t)ly * L.pitch + lx;
                const int g[4] = {p[0], p[1], p[L.pitch], p[L.pitch + 1]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lap = sat_s16(g[k] - (int)sat_s16(up[k]));
                    acc[k] = (int16_t)(acc[k] + trunc_s16((float)lap * w[k]));
                }
            }
            const DivBy div[4] = {DivBy(dn[0]), DivBy(dn[1]), DivBy(dn[2]), DivBy(dn[3])};
            int16_t *d = cur + (size_t)y0 * lw + 2 * qxl;
            d[0] = trunc_s16(div[0]((float)acc[0])); d[1] = trunc_s16(div[1]((float)acc[1]));
            d[lw] = trunc_s16(div[2]((float)acc[2])); d[lw + 1] = trunc_s16(div[3]((float)acc[3]));
        }
    }
    __syncthreads();
    for (int l = nb - 1; l >= t; --l) {
        const int lw = s_w[l], al = s_a[l], lh = P.qh[l], qwl = lw >> 1, qhl = lh >> 1;
        int16_t *cur = s_c + s_off[l];
        const int16_t *prev = s_c + s_off[l + 1];       // columns [a[l+1], b[l+1]) of the collapsed band l+1: index with the global column (clamped to the band by pu_idx) minus a[l+1]
        const int prev_w = s_w[l + 1], ap = s_a[l + 1];
        for (int i = tid; i < qwl * qhl; i += nthr) {
            const int qy = i / qwl, qxl = i - qy * qwl;
            const int x0 = al + 2 * qxl, y0 = 2 * qy;
            int up[4], res[4];
            up_quad(prev - ap, prev_w, P.qh[l + 1], P.qw[l + 1], qy, x0 >> 1, up);
            int16_t *d = cur + (size_t)y0 * lw + 2 * qxl;
            const int dd[4] = {d[0], d[1], d[lw], d[lw + 1]};
#pragma unroll
            for (int k = 0; k < 4; ++k) res[k] = sat_s16((int)sat_s16(up[k]) + dd[k]);
            if (l == t) {
                int16_t *o = cl + (size_t)f * cl_stride + P.coff[l] + (size_t)c * P.qh[l] * P.qpitch[l] + (size_t)y0 * P.qpitch[l] + x0;
                o[0] = (int16_t)res[0]; o[1] = (int16_t)res[1]; o[P.qpitch[l]] = (int16_t)res[2]; o[P.qpitch[l] + 1] = (int16_t)res[3];
            } else {
                d[0] = (int16_t)res[0]; d[1] = (int16_t)res[1]; d[lw] = (int16_t)res[2]; d[lw + 1] = (int16_t)res[3];
            }
        }
        __syncthreads();
    }
}

// ---- vectorised band kernel: 2 rows x 8 columns per thread ------------------------------------------
// Valid where every view rect and the pano level are 8-aligned in x and 2-aligned in y (levels l <= nb-3).
// Same arithmetic as k_blend; views whose 16 weights are all zero are skipped (trunc(L*0) == 0 exactly),
// which removes most of the work of the +-pi-straddling full-width view.
__device__ __forceinline__ void unpack8(const uint4 b, int v[8])
{
    v[0] = (int16_t)(b.x & 0xffff); v[1] = (int)b.x >> 16; v[2] = (int16_t)(b.y & 0xffff); v[3] = (int)b.y >> 16;
    v[4] = (int16_t)(b.z & 0xffff); v[5] = (int)b.z >> 16; v[6] = (int16_t)(b.w & 0xffff); v[7] = (int)b.w >> 16;
}
__device__ __forceinline__ void unpack8(const uint2 b, int v[8])
{
    v[0] = b.x & 0xff; v[1] = (b.x >> 8) & 0xff; v[2] = (b.x >> 16) & 0xff; v[3] = b.x >> 24;
    v[4] = b.y & 0xff; v[5] = (b.y >> 8) & 0xff; v[6] = (b.y >> 16) & 0xff; v[7] = b.y >> 24;
}
__device__ __forceinline__ uint2 load8_a1(const void *p)
{
    uint2 r;
    __builtin_memcpy(&r, p, 8);
    return r;
}
// pyrUp of the 2x8 fine block whose top-left is (2i, 2*j0) (j0 % 4 == 0, cw % 4 == 0) in plain 32-bit arithmetic, from the three coarse rows ALREADY in registers
// (up_rows_load: columns j0-2 .. j0+5 cover the 6 taps j0-1 .. j0+4; at j0 == 0 column -1 mirrors to 1, past the right edge column cw clamps to cw-1).
// The fallback of up_2x8_pkb for collapsed values outside the packed range.  It deliberately issues no load of its own: a load inside this rarely taken branch
// makes the wait-count pass assume the worst at the join, and the planes' software pipeline (next plane's rows in flight) degrades to "wait for everything".
__device__ __forceinline__ void up_2x8_raw(const uint4 raw[3], int cw, int j0, int ue[8], int uo[8])
{
    int he[3][4], ho[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int v[8], a[6];
        unpack8(raw[r], v);
        a[0] = j0 == 0 ? v[3] : v[1]; a[1] = v[2]; a[2] = v[3]; a[3] = v[4]; a[4] = v[5]; a[5] = v[6];
        if (j0 + 4 >= cw) a[5] = a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { he[r][q] = a[q] + 6 * a[q + 1] + a[q + 2]; ho[r][q] = 4 * (a[q + 1] + a[q + 2]); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ue[2 * q] = sat_s16(rne_shift(he[0][q] + 6 * he[1][q] + he[2][q], 6));
        ue[2 * q + 1] = sat_s16(rne_shift(ho[0][q] + 6 * ho[1][q] + ho[2][q], 6));
        uo[2 * q] = sat_s16(rne_shift(4 * (he[1][q] + he[2][q]), 6));
        uo[2 * q + 1] = sat_s16(rne_shift(4 * (ho[1][q] + ho[2][q]), 6));
    }
}

// ---- packed 16-bit forms for the VIEW side of a band -------------------------------------------------------------
// Every Gaussian level of a view is a convex combination of 8-bit pixels, rounded: all values are in [0,255] (the
// per-frame pyramid buffers are zero-filled at creation and only ever hold such values, so stale tiles are in range
// too).  pyrUp's partial sums are then < 64*255 + 32 < 2^15: two of them live in the 16-bit halves of one register
// and plain 32-bit adds/shifts never carry from one half into the other.  Results are identical to up_2x8.
__device__ __forceinline__ unsigned rne6_pk(unsigned s)
{
    const unsigned t = pk_lshr16(s, 6) & 0x00010001u;
    return pk_lshr16(s + t + 0x001f001fu, 6);          // (the halves stay below 2^16: the 32-bit add carries nothing across)
}
// pixel order of the four output registers of a row: (0,2) (1,3) (4,6) (5,7)  [low half, high half]
__device__ __forceinline__ void up_rows_load(const int16_t *__restrict__ cs, int cpitch, int ch, int i, int j0, uint4 raw[3])
{
    const int rr[3] = {pu_idx(i - 1, ch), pu_idx(i, ch), pu_idx(i + 1, ch)};
    // columns j0-2 .. j0+5 unconditionally: at j0 == 0 the first dword lies before the row (padding / previous plane / the buffer's lead) and is replaced below
#pragma unroll
    for (int r = 0; r < 3; ++r) raw[r] = load16_a4((cs - 2) + (mul24(rr[r], cpitch) + (unsigned)j0));
}
__device__ __forceinline__ void up_2x8_pk(const uint4 raw[3], int cw, int j0, unsigned ue[4], unsigned uo[4])
{
    unsigned he[3][2], ho[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        unsigned w0 = raw[r].x, w1 = raw[r].y, w2 = raw[r].z, w3 = raw[r].w;
        if (j0 == 0) w0 = w1;                            // tap -1 mirrors to column 1 = hi(w1)
        if (j0 + 4 >= cw) w3 = w2 >> 16;                 // tap j0+4 clamps to cw-1
        // taps t0 = hi(w0), (t1,t2) = w1, (t3,t4) = w2, t5 = lo(w3)
        const unsigned t01 = __builtin_amdgcn_alignbyte(w1, w0, 2), t23 = __builtin_amdgcn_alignbyte(w2, w1, 2),
                       t45 = __builtin_amdgcn_alignbyte(w3, w2, 2);
        he[r][0] = mad6(w1, t01 + t23);                  // (he0, he1), he_q = t_q + 6 t_{q+1} + t_{q+2}
        he[r][1] = mad6(w2, t23 + t45);                  // (he2, he3)
        ho[r][0] = 4u * (w1 + t23);                      // (ho0, ho1), ho_q = 4 (t_{q+1} + t_{q+2})
        ho[r][1] = 4u * (w2 + t45);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ue[2 * q] = rne6_pk(mad6(he[1][q], he[0][q] + he[2][q]));
        ue[2 * q + 1] = rne6_pk(mad6(ho[1][q], ho[0][q] + ho[2][q]));
        uo[2 * q] = rne6_pk(4u * (he[1][q] + he[2][q]));
        uo[2 * q + 1] = rne6_pk(4u * (ho[1][q] + ho[2][q]));
    }
}

// A view's Gaussian level is stored as bytes: the 8-column window is ONE 8-byte read per row (2-byte aligned at worst, the hardware takes it),
// widened to the 16-bit pairs of up_2x8_pk in registers.
__device__ __forceinline__ void up_rows_load(const uint8_t *__restrict__ cs, int cpitch, int ch, int i, int j0, uint2 raw[3])
{
    const int rr[3] = {pu_idx(i - 1, ch), pu_idx(i, ch), pu_idx(i + 1, ch)};
#pragma unroll
    for (int r = 0; r < 3; ++r) __builtin_memcpy(&raw[r], (cs - 2) + (mul24(rr[r], cpitch) + (unsigned)j0), 8);      // columns j0-2 .. j0+5 (see the int16 form)
}
__device__ __forceinline__ void up_2x8_pk(const uint2 raw8[3], int cw, int j0, unsigned ue[4], unsigned uo[4])
{
    uint4 raw[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        raw[r] = make_uint4(__builtin_amdgcn_perm(0u, raw8[r].x, 0x0c010c00u), __builtin_amdgcn_perm(0u, raw8[r].x, 0x0c030c02u),
                            __builtin_amdgcn_perm(0u, raw8[r].y, 0x0c010c00u), __builtin_amdgcn_perm(0u, raw8[r].y, 0x0c030c02u));
    up_2x8_pk(raw, cw, j0, ue, uo);
}

// The COLLAPSED coarser level has no such bound (it is a saturate_cast<short> of a sum), but in practice it stays near the 8-bit
// range: add a bias of 384 and use the same packed arithmetic when all 18 taps of the lane are in [-384, 639]
// (pyrUp's weights sum to 64 and 384 is even, so rne(S + 64*384, 6) == rne(S, 6) + 384 exactly); lanes with a tap outside
// that range take the 32-bit up_2x8.  Returns false if the lane must fall back.  Outputs are biased: value + 384.
#ifdef MS_BLEND_WAVES
#define MS_BLEND_OCC __attribute__((amdgpu_waves_per_eu(MS_BLEND_WAVES, 8)))
#else
#define MS_BLEND_OCC
#endif
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned add_pk_u16(unsigned a, unsigned b)      // v_pk_add_u16: halves add independently (wrap)
{
    u16x2 x, y;
    __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    x += y;
    __builtin_memcpy(&a, &x, 4);
    return a;
}
__device__ __forceinline__ unsigned sub_pk_u16(unsigned a, unsigned b)      // v_pk_sub_u16
{
    u16x2 x, y;
    __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    x -= y;
    __builtin_memcpy(&a, &x, 4);
    return a;
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(unsigned a) { s16x2 x; __builtin_memcpy(&x, &a, 4); return x; }
__device__ __forceinline__ unsigned as_u32(s16x2 x) { unsigned a; __builtin_memcpy(&a, &x, 4); return a; }
// v_pk_add_i16 clamp / v_pk_min_i16 / v_pk_max_i16: two int16 per register; the saturating add is add + saturate_cast<short> per half
__device__ __forceinline__ unsigned addsat_pk_i16(unsigned a, unsigned b) { return as_u32(__builtin_elementwise_add_sat(as_s16x2(a), as_s16x2(b))); }
__device__ __forceinline__ unsigned min_pk_i16(unsigned a, unsigned b) { return as_u32(__builtin_elementwise_min(as_s16x2(a), as_s16x2(b))); }
__device__ __forceinline__ unsigned max_pk_i16(unsigned a, unsigned b) { return as_u32(__builtin_elementwise_max(as_s16x2(a), as_s16x2(b))); }
constexpr int UP_BIAS = 384;
__device__ __forceinline__ bool up_2x8_pkb(const uint4 raw[3], int cw, int j0, unsigned ue[4], unsigned uo[4])
{
    unsigned he[3][2], ho[3][2], bad = 0u;
    const unsigned bias = (unsigned)UP_BIAS * 0x00010001u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        unsigned w0 = add_pk_u16(raw[r].x, bias), w1 = add_pk_u16(raw[r].y, bias), w2 = add_pk_u16(raw[r].z, bias), w3 = add_pk_u16(raw[r].w, bias);
        if (j0 == 0) w0 = w1;
        if (j0 + 4 >= cw) w3 = w2 >> 16;
        bad |= (w0 & 0xfc000000u) | (w1 & 0xfc00fc00u) | (w2 & 0xfc00fc00u) | (w3 & 0x0000fc00u);   // taps: hi(w0), w1, w2, lo(w3)
        const unsigned t01 = __builtin_amdgcn_alignbyte(w1, w0, 2), t23 = __builtin_amdgcn_alignbyte(w2, w1, 2),
                       t45 = __builtin_amdgcn_alignbyte(w3, w2, 2);
        he[r][0] = mad6(w1, t01 + t23);
        he[r][1] = mad6(w2, t23 + t45);
        ho[r][0] = 4u * (w1 + t23);
        ho[r][1] = 4u * (w2 + t45);
    }
    if (bad) return false;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ue[2 * q] = rne6_pk(mad6(he[1][q], he[0][q] + he[2][q]));
        ue[2 * q + 1] = rne6_pk(mad6(ho[1][q], ho[0][q] + ho[2][q]));
        uo[2 * q] = rne6_pk(4u * (he[1][q] + he[2][q]));
        uo[2 * q + 1] = rne6_pk(4u * (ho[1][q] + ho[2][q]));
    }
    return true;
}

template <bool L0, int MODE, int CLS = 0>      // CLS 1: only the integer cells (owned + exclusive) of the band, 2: only the general ones, 0: all
__global__ void __launch_bounds__(64) MS_BLEND_OCC k_blend8(const BlendTile *__restrict__ tiles, const ViewDesc *__restrict__ views, PanoDesc P, int l,
                                                const uint8_t *__restrict__ g0, long long g0_stride,
                                                const uint8_t *__restrict__ gl, long long gl_stride,
                                                int16_t *__restrict__ cl, long long cl_stride, OutTable out, ShardArgs S)
{
    MS_PRIO_LOADS(MS_PRIO_BLEND);
    const BlendTile T = tiles[blockIdx.x];
    const int f = blockIdx.z;
    // A tile is 256 x 16 px = four 64 x 16 px cells (8 lanes x 8 lane-rows of 8 x 2 px), ONE WAVE PER WORKGROUP (blockIdx.y = the cell): the waves share
    // nothing, and single-wave workgroups fill the SIMDs more evenly than four-wave ones (level 0: 134 -> 127 us).  A whole wave usually lies inside one
    // view's exclusive region and can take the single-view path below.
    const int tid = 64 * (int)blockIdx.y + (int)threadIdx.y * 32 + (int)threadIdx.x, lane = tid & 63;
    const int x0 = T.x0 + 64 * (tid >> 6) + 8 * (lane & 7), y0 = T.y0 + 2 * (lane >> 3);
    if (x0 >= P.qw[l] || y0 >= P.qh[l]) return;
    // owner of this wave's cell: the one view with non-zero weights there, all exactly 1 -- then (short)(L * 1.f) = L and the weight sum
    // is exactly 1.00001f; 255 = general case.  Wave-uniform by construction (one byte per 64 x 16 cell).
    int owner = 255;
    if (MODE == 0 && P.pure[l]) {
        // the cell index is made of scalars only (tile record, blockIdx.y): the byte comes through the scalar cache as part of its aligned dword (build_plan aligns every
        // band's map to 4 bytes) instead of a per-lane load + readfirstlane -- one vector-memory round trip less before the wave knows which path it takes
        const unsigned ci = (unsigned)(T.y0 >> 4) * (unsigned)P.ppitch[l] + (unsigned)((T.x0 >> 6) + (int)blockIdx.y);
        const unsigned word = reinterpret_cast<const unsigned *>(P.pure[l])[ci >> 2];
        owner = (int)((word >> (8u * (ci & 3u))) & 0xffu);
    }
    // 254 (level 0): a seam runs through the cell, but every pixel has exactly one contributing view with weight exactly 1 (binary masks): the owner
    // arithmetic per pixel -- the Laplacian of each view ANDed with its mask bytes, no float multiply, no division (k_owner_map)
    if (!L0 && owner == 254) owner = 255;                 // (never written for l > 0)
    const bool excl = L0 && owner == 254, pure = owner < 254, integer_cell = CLS == 1 || (CLS != 2 && owner != 255);
    if ((CLS == 1 && owner == 255) || (CLS == 2 && owner != 255)) return;
    // The accumulators are the reference's int16 `dst += (short)(v * w)` themselves: two pixels per register, added with the packed 16-bit
    // add (wraps per half exactly like `short +=`).  Pixel order of the four registers of a row: (0,2) (1,3) (4,6) (5,7), the order
    // the packed pyrUp produces.  Half as many accumulator registers = more waves in flight (the kernel waits on memory, not on VALU).
    unsigned accp[3][2][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) accp[c][0][q] = accp[c][1][q] = 0u;

    const size_t pplane = (size_t)P.qh[l] * P.qpitch[l], po = (size_t)f * S.pstride + P.poff[l] + (mul24(y0, P.qpitch[l]) + (unsigned)x0);
    if (MODE == 2) {
        for (int s = 0; s < S.n_parts; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint4 b = *reinterpret_cast<const uint4 *>(S.part[s] + po + c * pplane + (size_t)r * P.qpitch[l]);   // natural order (0,1) (2,3) ..
                    accp[c][r][0] = add_pk_u16(accp[c][r][0], __builtin_amdgcn_perm(b.y, b.x, 0x05040100u));
                    accp[c][r][1] = add_pk_u16(accp[c][r][1], __builtin_amdgcn_perm(b.y, b.x, 0x07060302u));
                    accp[c][r][2] = add_pk_u16(accp[c][r][2], __builtin_amdgcn_perm(b.w, b.z, 0x05040100u));
                    accp[c][r][3] = add_pk_u16(accp[c][r][3], __builtin_amdgcn_perm(b.w, b.z, 0x07060302u));
                }
    }
    for (unsigned vm = (MODE == 2) ? 0u : (MODE == 1 ? (T.view_mask & S.own_mask) : (pure ? (1u << owner) : T.view_mask)); vm; vm &= vm - 1) {   // views with a non-zero weight in this tile
        const int v = __builtin_ctz(vm);
        const LevelDesc &L = views[v].lv[l];
        const int lx = x0 - L.x_tl, ly = y0 - L.y_tl;
        if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) continue;
        float w[2][8];
        unsigned mq[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};       // exclusive cells: 0xffff / 0 per pixel, in the accumulators' pixel order
        if (pure) {
#pragma unroll
            for (int k = 0; k < 8; ++k) w[0][k] = w[1][k] = 1.f;       // (unused: the owner path adds L itself)
        } else if (L0) {   // level-0 weights are mask * (1/255) (blenders.cpp:412): rebuilt from the padded 8-bit mask, 1 byte/px
            const uint8_t *mp = views[v].wm0 + (mul24(ly, views[v].wm0_pitch) + (unsigned)lx);
            typedef unsigned u32x2_g __attribute__((ext_vector_type(2)));      // (global address space named: the pointer comes out of the view descriptor -- see load_px2)
            const u32x2_g ga = *(const MS_GLOBAL_AS u32x2_g *)(uintptr_t)mp, gb = *(const MS_GLOBAL_AS u32x2_g *)(uintptr_t)(mp + views[v].wm0_pitch);
            const uint2 ma = make_uint2(ga.x, ga.y), mb = make_uint2(gb.x, gb.y);
            if ((ma.x | ma.y | mb.x | mb.y) == 0u) continue;          // all 16 weights zero: (short)(L * 0) == 0
            if (excl || CLS == 1) {                                   // mask bytes are 0 / 255 here
                mq[0][0] = __builtin_amdgcn_perm(0u, ma.x, 0x02020000u); mq[0][1] = __builtin_amdgcn_perm(0u, ma.x, 0x03030101u);
                mq[0][2] = __builtin_amdgcn_perm(0u, ma.y, 0x02020000u); mq[0][3] = __builtin_amdgcn_perm(0u, ma.y, 0x03030101u);
                mq[1][0] = __builtin_amdgcn_perm(0u, mb.x, 0x02020000u); mq[1][1] = __builtin_amdgcn_perm(0u, mb.x, 0x03030101u);
                mq[1][2] = __builtin_amdgcn_perm(0u, mb.y, 0x02020000u); mq[1][3] = __builtin_amdgcn_perm(0u, mb.y, 0x03030101u);
            } else {
                int m0[8], m1[8];
                unpack8(ma, m0); unpack8(mb, m1);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    w[0][k] = __builtin_fmaf(P.alpha, (float)m0[k], 0.f);
                    w[1][k] = __builtin_fmaf(P.alpha, (float)m1[k], 0.f);
                }
            }
        } else {
            const float *wp = L.wgt + (mul24(ly, L.wpitch) + (unsigned)lx);
            typedef float f32x4_g __attribute__((ext_vector_type(4)));
            const f32x4_g wa = *(const MS_GLOBAL_AS f32x4_g *)(uintptr_t)wp, wb = *(const MS_GLOBAL_AS f32x4_g *)(uintptr_t)(wp + 4);
            const f32x4_g wc = *(const MS_GLOBAL_AS f32x4_g *)(uintptr_t)(wp + L.wpitch), wd = *(const MS_GLOBAL_AS f32x4_g *)(uintptr_t)(wp + L.wpitch + 4);
            w[0][0] = wa.x; w[0][1] = wa.y; w[0][2] = wa.z; w[0][3] = wa.w; w[0][4] = wb.x; w[0][5] = wb.y; w[0][6] = wb.z; w[0][7] = wb.w;
            w[1][0] = wc.x; w[1][1] = wc.y; w[1][2] = wc.z; w[1][3] = wc.w; w[1][4] = wd.x; w[1][5] = wd.y; w[1][6] = wd.z; w[1][7] = wd.w;
            float wsum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) wsum += w[0][k] + w[1][k];      // weights are >= 0
            if (wsum == 0.f) continue;
        }
        const LevelDesc &C = views[v].lv[l + 1];
        const size_t fplane = (size_t)L.h * L.pitch, cplane = (size_t)C.h * C.pitch;
        const size_t fo = mul24(ly, L.pitch) + (unsigned)lx;
        // the three colour planes are software-pipelined: the reads of plane c+1 (3 coarse rows + 2 fine rows) are issued before
        // plane c is computed, so a view costs about one memory round trip instead of three
        uint2 craw[2][3], fraw[2][2];         // byte planes on both sides (level 0 = the warp output, levels >= 1 = the view's Gaussian levels)
        const uint8_t *fine = (L0 ? g0 + (size_t)f * g0_stride : gl + (size_t)f * gl_stride) + L.off + fo;
        auto issue = [&](int c, int b) {
            up_rows_load(gl + (size_t)f * gl_stride + C.off + c * cplane, C.pitch, C.h, ly >> 1, lx >> 1, craw[b]);
            const uint8_t *p = fine + c * fplane;
#pragma unroll
            for (int r = 0; r < 2; ++r) __builtin_memcpy(&fraw[b][r], __builtin_assume_aligned(p + (size_t)r * L.pitch, 8), 8);
        };
        issue(0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int b = c & 1;
            if (c + 1 < 3) issue(c + 1, b ^ 1);
            if (MS_PRIO_BLEND) { __builtin_amdgcn_sched_barrier(0); MS_PRIO_MATH(MS_PRIO_BLEND); }
            unsigned up[2][4];
            up_2x8_pk(craw[b], C.w, lx >> 1, up[0], up[1]);
            unsigned g[2][4];                 // same pixel order as up: (0,2) (1,3) (4,6) (5,7)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned w0 = fraw[b][r].x, w1 = fraw[b][r].y;
                g[r][0] = w0 & 0x00ff00ffu; g[r][1] = (w0 >> 8) & 0x00ff00ffu;
                g[r][2] = w1 & 0x00ff00ffu; g[r][3] = (w1 >> 8) & 0x00ff00ffu;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // Laplacian L = g - up in [-255,255], formed as 256 + L per half (no borrow between the halves);
                    // |L*w| <= 255: neither saturate_cast of the reference chain (sub_mat.cu:59-65, multiband_blend.cu:46-49) can trigger
                    if (pure) { accp[c][r][q] = sub_pk_u16(g[r][q], up[r][q]); continue; }             // weight exactly 1: (short)(L * 1.f) == L (one packed subtract)
                    if (excl || CLS == 1) { accp[c][r][q] |= sub_pk_u16(g[r][q], up[r][q]) & mq[r][q]; continue; }  // ... or exactly 0, per pixel; one view per pixel
                    const unsigned d = (g[r][q] | 0x01000100u) - up[r][q];
                    const int k0 = (q >> 1) * 4 + (q & 1), k1 = k0 + 2;
                    const int t0 = (int)((float)((int)(d & 0xffffu) - 256) * w[r][k0]);
                    const int t1 = (int)((float)((int)(d >> 16) - 256) * w[r][k1]);
                    accp[c][r][q] = add_pk_u16(accp[c][r][q], __builtin_amdgcn_perm((unsigned)t1, (unsigned)t0, 0x05040100u));
                }
            if (MS_PRIO_BLEND && c < 2) { __builtin_amdgcn_sched_barrier(0); MS_PRIO_LOADS(MS_PRIO_BLEND); }
        }
        if (MS_PRIO_BLEND) { __builtin_amdgcn_sched_barrier(0); MS_PRIO_LOADS(MS_PRIO_BLEND); }
    }

    if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                uint4 o;                      // back to natural pixel order
                o.x = __builtin_amdgcn_perm(accp[c][r][1], accp[c][r][0], 0x05040100u);   // px 0, 1
                o.y = __builtin_amdgcn_perm(accp[c][r][1], accp[c][r][0], 0x07060302u);   // px 2, 3
                o.z = __builtin_amdgcn_perm(accp[c][r][3], accp[c][r][2], 0x05040100u);   // px 4, 5
                o.w = __builtin_amdgcn_perm(accp[c][r][3], accp[c][r][2], 0x07060302u);   // px 6, 7
                *reinterpret_cast<uint4 *>(S.pout + po + c * pplane + (size_t)r * P.qpitch[l]) = o;
            }
        return;
    }
    float den[2][8];
    if (!integer_cell) {
        const float *dp = P.den[l] + (mul24(y0, P.dpitch[l]) + (unsigned)x0);
        const float4 da = *reinterpret_cast<const float4 *>(dp), db = *reinterpret_cast<const float4 *>(dp + 4);
        const float4 dc = *reinterpret_cast<const float4 *>(dp + P.dpitch[l]), dd = *reinterpret_cast<const float4 *>(dp + P.dpitch[l] + 4);
        den[0][0] = da.x; den[0][1] = da.y; den[0][2] = da.z; den[0][3] = da.w; den[0][4] = db.x; den[0][5] = db.y; den[0][6] = db.z; den[0][7] = db.w;
        den[1][0] = dc.x; den[1][1] = dc.y; den[1][2] = dc.z; den[1][3] = dc.w; den[1][4] = dd.x; den[1][5] = dd.y; den[1][6] = dd.z; den[1][7] = dd.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) den[0][k] = den[1][k] = 1.f;
    }
    const size_t cplane = (size_t)P.qh[l + 1] * P.qpitch[l + 1];
    const int16_t *cc = cl + (size_t)f * cl_stride + P.coff[l + 1];
    float rcp[2][8];                      // refined reciprocals, shared by the three colour planes (DivBy)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) rcp[r][k] = !integer_cell ? DivBy(den[r][k]).r : 1.f;
    // Results as int16 pairs in the accumulators' pixel order (register q of a row holds px (0,2) (1,3) (4,6) (5,7)): normalise, collapse and the
    // output conversion stay packed (v_pk_*_i16), two pixels per instruction.
    unsigned resq[3][2][4];
    uint4 ccraw[2][3];                    // the three planes of the collapsed coarser level are pipelined like the view planes above
    up_rows_load(cc, P.qpitch[l + 1], P.qh[l + 1], y0 >> 1, x0 >> 1, ccraw[0]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        MS_PRIO_LOADS(MS_PRIO_BLEND);
        if (c + 1 < 3) up_rows_load(cc + (c + 1) * cplane, P.qpitch[l + 1], P.qh[l + 1], y0 >> 1, x0 >> 1, ccraw[(c + 1) & 1]);
        if (MS_PRIO_BLEND) { __builtin_amdgcn_sched_barrier(0); MS_PRIO_MATH(MS_PRIO_BLEND); }
        unsigned upq[2][4];               // pyrUp of the collapsed coarser band (its values fit int16: the reference's saturate_cast<short> is the identity)
        if (up_2x8_pkb(ccraw[c & 1], P.qw[l + 1], x0 >> 1, upq[0], upq[1])) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) upq[r][q] = sub_pk_u16(upq[r][q], (unsigned)UP_BIAS * 0x00010001u);
        } else {
            int up[2][8];
            up_2x8_raw(ccraw[c & 1], P.qw[l + 1], x0 >> 1, up[0], up[1]);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k0 = (q >> 1) * 4 + (q & 1);
                    upq[r][q] = ((unsigned)up[r][k0] & 0xffffu) | ((unsigned)up[r][k0 + 2] << 16);
                }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // int16 accumulation wraps: (short)(sum) == successive `short +=`
                const unsigned a = accp[c][r][q];
                unsigned n;
                if (integer_cell) {
                    // owner (and exclusive) cells: a / 1.00001f for an integer |a| <= 255 lies strictly between a - sign(a) and a, further than half an ulp from a,
                    // so the correctly rounded quotient truncates to a - sign(a): the division is an integer subtraction there
                    n = sub_pk_u16(a, min_pk_i16(max_pk_i16(a, 0xffffffffu), 0x00010001u));
                } else {
                    const int k0 = (q >> 1) * 4 + (q & 1), k1 = k0 + 2;
                    DivBy d0(1.f), d1(1.f);
                    d0.d = den[r][k0]; d0.r = rcp[r][k0]; d1.d = den[r][k1]; d1.r = rcp[r][k1];
                    const int n0 = trunc_s16(d0((float)(int)(int16_t)(a & 0xffffu))), n1 = trunc_s16(d1((float)((int)a >> 16)));
                    n = ((unsigned)n0 & 0xffffu) | ((unsigned)n1 << 16);
                }
                resq[c][r][q] = addsat_pk_i16(upq[r][q], n);          // add(pyrUp, band) with saturate_cast<short> (the result replaces the expanded coarser level)
            }
    }

    if (!L0) {
        const size_t plane = (size_t)P.qh[l] * P.qpitch[l];
        int16_t *d = cl + (size_t)f * cl_stride + P.coff[l] + (mul24(y0, P.qpitch[l]) + (unsigned)x0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                uint4 o;                      // back to natural pixel order
                o.x = __builtin_amdgcn_perm(resq[c][r][1], resq[c][r][0], 0x05040100u);   // px 0, 1
                o.y = __builtin_amdgcn_perm(resq[c][r][1], resq[c][r][0], 0x07060302u);   // px 2, 3
                o.z = __builtin_amdgcn_perm(resq[c][r][3], resq[c][r][2], 0x05040100u);   // px 4, 5
                o.w = __builtin_amdgcn_perm(resq[c][r][3], resq[c][r][2], 0x07060302u);   // px 6, 7
                *reinterpret_cast<uint4 *>(d + c * plane + (size_t)r * P.qpitch[l]) = o;
            }
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + r;
            if (y >= P.fh) continue;
            const uint8_t *mrow = P.mask + mul24(y, P.mask_pitch);
            const int nvalid = min(8, P.fw - x0);
            uint2 mk8 = make_uint2(0u, 0u);                 // result mask of the 8 pixels: bytes 0 / 255 (k_finish_den)
            if (nvalid == 8) __builtin_memcpy(&mk8, __builtin_assume_aligned(mrow + x0, 8), 8);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned m = (k < nvalid) ? (unsigned)mrow[x0 + k] : 0u;
                    if (k < 4) mk8.x |= m << (8 * k); else mk8.y |= m << (8 * (k - 4));
                }
            }
            // setTo(0, mask == 0) (blenders.cpp:803-810): 0xffff / 0 per half, in the registers' pixel order
            const unsigned mq[4] = {__builtin_amdgcn_perm(0u, mk8.x, 0x02020000u), __builtin_amdgcn_perm(0u, mk8.x, 0x03030101u),
                                    __builtin_amdgcn_perm(0u, mk8.y, 0x02020000u), __builtin_amdgcn_perm(0u, mk8.y, 0x03030101u)};
            unsigned v[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[c][q] = resq[c][r][q] & mq[q];
            // pixel k of the row: register (k / 4) * 2 + (k & 1), half (k >> 1) & 1
            auto val = [&](int k, int c) -> int {
                const unsigned w = v[c][(k >> 2) * 2 + (k & 1)];
                return ((k >> 1) & 1) ? ((int)w >> 16) : (int)(int16_t)(w & 0xffffu);
            };
            if (out.p16[f]) {
                int16_t *d = (int16_t *)((char *)out.p16[f] + mul24(y, (int)out.step16[f])) + 3 * x0;
                if (nvalid == 8) {
                    unsigned wds[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) {        // interleaved int16 triplets: elements 2i, 2i+1 of the 24
                        const int e0 = 2 * i, e1 = 2 * i + 1, ka = e0 / 3, kb = e1 / 3;
                        const unsigned sel = (((ka >> 1) & 1) ? 0x0302u : 0x0100u) | ((((kb >> 1) & 1) ? 0x0706u : 0x0504u) << 16);
                        wds[i] = __builtin_amdgcn_perm(v[e1 % 3][(kb >> 2) * 2 + (kb & 1)], v[e0 % 3][(ka >> 2) * 2 + (ka & 1)], sel);
                    }
                    __builtin_memcpy(d, wds, 48);
                } else {
                    for (int k = 0; k < nvalid; ++k) { d[3 * k] = (int16_t)val(k, 0); d[3 * k + 1] = (int16_t)val(k, 1); d[3 * k + 2] = (int16_t)val(k, 2); }
                }
            }
            if (!out.p8[f] && !out.pi[f]) continue;
            // convertTo(CV_8U) (timed.cpp:251): clamp to [0, 255] per half, then one dword [B, G, R, 0] per pixel
            unsigned px4[8];
            {
                unsigned bg[4], rr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned cb = min_pk_i16(max_pk_i16(v[0][q], 0u), 0x00ff00ffu), cg = min_pk_i16(max_pk_i16(v[1][q], 0u), 0x00ff00ffu);
                    rr[q] = min_pk_i16(max_pk_i16(v[2][q], 0u), 0x00ff00ffu);
                    bg[q] = cb | (cg << 8);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = (k >> 2) * 2 + (k & 1);
                    px4[k] = __builtin_amdgcn_perm(rr[q], bg[q], ((k >> 1) & 1) ? 0x0c060302u : 0x0c040100u);
                }
            }
            if (out.p8[f]) {
                const int cy = y + P.canvas_y, cx0 = x0 + P.canvas_x;
                if (cy >= 0 && cy < P.out_h) {
                    uint8_t *d = out.p8[f] + ((long long)mul24(cy, (int)out.step8[f]) + 3 * cx0);
                    if (nvalid == 8 && cx0 >= 0 && cx0 + 8 <= P.out_w) {
                        unsigned wds[6];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {       // 4 pixels -> 12 bytes
                            wds[3 * h + 0] = __builtin_amdgcn_perm(px4[4 * h + 1], px4[4 * h + 0], 0x04020100u);
                            wds[3 * h + 1] = __builtin_amdgcn_perm(px4[4 * h + 2], px4[4 * h + 1], 0x05040201u);
                            wds[3 * h + 2] = __builtin_amdgcn_perm(px4[4 * h + 3], px4[4 * h + 2], 0x06050402u);
                        }
                        __builtin_memcpy(d, wds, 24);
                    } else {
                        for (int k = 0; k < nvalid; ++k) {
                            const int cx = cx0 + k;
                            if (cx < 0 || cx >= P.out_w) continue;
                            d[3 * k] = (uint8_t)(px4[k] & 0xffu); d[3 * k + 1] = (uint8_t)((px4[k] >> 8) & 0xffu); d[3 * k + 2] = (uint8_t)((px4[k] >> 16) & 0xffu);
                        }
                    }
                }
            }
            if (out.pi[f]) {        // cvtColor(BGR2YUV_I420) of the clamped canvas pixel, written straight into the planar slab (see ms_stitch_i420)
                const int iy = y + P.canvas_y - P.i_y0, cx0 = x0 + P.canvas_x;
                if (iy >= 0 && iy < P.i_rows) {
                    constexpr int SH = 20, HALF = 1 << (SH - 1);
                    constexpr int CRY = 269484, CGY = 528482, CBY = 102760, CRU = -155188, CGU = -305135, CBU = 460324, CGV = -385875, CBV = -74448;
                    uint8_t *Yp = out.pi[f] + mul24(iy, P.out_w);
                    uint8_t *Up = out.pi[f] + (size_t)P.out_w * P.i_rows + mul24(iy >> 1, P.out_w >> 1), *Vp = Up + (size_t)(P.out_w >> 1) * (P.i_rows >> 1);
                    const bool crow = (iy & 1) == 0;                       // chroma comes from the top-left pixel of each 2 x 2 block
                    uint8_t yv[8], uv[8], vv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int b = (int)(px4[k] & 0xffu), g = (int)((px4[k] >> 8) & 0xffu), rr = (int)((px4[k] >> 16) & 0xffu);
                        yv[k] = (uint8_t)min(max((CRY * rr + CGY * g + CBY * b + HALF + (16 << SH)) >> SH, 0), 255);
                        uv[k] = (uint8_t)min(max((CRU * rr + CGU * g + CBU * b + HALF + (128 << SH)) >> SH, 0), 255);
                        vv[k] = (uint8_t)min(max((CBU * rr + CGV * g + CBV * b + HALF + (128 << SH)) >> SH, 0), 255);
                    }
                    if (nvalid == 8 && cx0 >= 0 && cx0 + 8 <= P.out_w) {
                        __builtin_memcpy(Yp + cx0, yv, 8);
                        if (crow) {
                            const int k0 = cx0 & 1;                           // first even canvas column of the cell
                            const uint8_t u4[4] = {uv[k0], uv[k0 + 2], uv[k0 + 4], uv[k0 + 6]}, v4[4] = {vv[k0], vv[k0 + 2], vv[k0 + 4], vv[k0 + 6]};
                            __builtin_memcpy(Up + ((cx0 + k0) >> 1), u4, 4);
                            __builtin_memcpy(Vp + ((cx0 + k0) >> 1), v4, 4);
                        }
                    } else {
                        for (int k = 0; k < nvalid; ++k) {
                            const int cx = cx0 + k;
                            if (cx < 0 || cx >= P.out_w) continue;
                            Yp[cx] = yv[k];
                            if (crow && (cx & 1) == 0) { Up[cx >> 1] = uv[k]; Vp[cx >> 1] = vv[k]; }
                        }
                    }
                }
            }
        }
    }
}

// single band (num_bands == 0): dst = sum_v (short)(img_v * w_v), normalised by the weight sum and masked.  With the weights of
// ms_init_feather this is FeatherBlender::feed / blend (blenders.cpp:147-186); with mask/255 weights, MultiBandBlender with no bands.
__global__ void __launch_bounds__(256) k_single_band(const ViewDesc *__restrict__ views, PanoDesc P,
                                                     const uint8_t *__restrict__ g0, long long g0_stride, OutTable out)
{
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= P.fw || y >= P.fh) return;
    int16_t acc[3] = {0, 0, 0};
    for (int v = 0; v < P.n_views; ++v) {
        const LevelDesc &L = views[v].lv[0];
        const int lx = x - L.x_tl, ly = y - L.y_tl;
        if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) continue;
        const float w = L.wgt[(size_t)ly * L.wpitch + lx];
        const uint8_t *p = g0 + (size_t)f * g0_stride + L.off + (size_t)ly * L.pitch + lx;
        const size_t plane = (size_t)L.h * L.pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = (int16_t)(acc[c] + trunc_s16((float)p[c * plane] * w));
    }
    const float den = P.den[0][(size_t)y * P.dpitch[0] + x];
    const bool m = P.mask[(size_t)y * P.mask_pitch + x] != 0;
    int r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = m ? (int)trunc_s16((float)acc[c] / den) : 0;
    if (out.p16[f]) {
        int16_t *d = (int16_t *)((char *)out.p16[f] + (size_t)y * out.step16[f]) + 3 * x;
        d[0] = (int16_t)r[0]; d[1] = (int16_t)r[1]; d[2] = (int16_t)r[2];
    }
    if (out.p8[f]) {
        const int cx = x + P.canvas_x, cy = y + P.canvas_y;
        if (cx >= 0 && cx < P.out_w && cy >= 0 && cy < P.out_h) {
            uint8_t *d = out.p8[f] + (size_t)cy * out.step8[f] + 3 * (size_t)cx;
            d[0] = (uint8_t)min(max(r[0], 0), 255); d[1] = (uint8_t)min(max(r[1], 0), 255); d[2] = (uint8_t)min(max(r[2], 0), 255);
        }
    }
}

// exhaustive check of DivBy against the compiler's IEEE division: all int16 numerators for each denominator
__global__ void __launch_bounds__(256) k_selftest_divide(const float *__restrict__ dens, int n_dens, unsigned *mismatches)
{
    const int i = blockIdx.x * 256 + threadIdx.x;      // numerator index 0..65535
    const int j = blockIdx.y;
    if (i >= 65536 || j >= n_dens) return;
    const float a = (float)(i - 32768), d = dens[j];
    const float ref = a / d;
    const float got = DivBy(d)(a);
    if (__float_as_uint(ref) != __float_as_uint(got)) atomicAdd(mismatches, 1u);
}

// ... and over a whole RANGE of denominators: every float whose bit pattern lies in [bits0, bits0 + n) (positive floats are ordered like their
// bits), one lane per denominator, all 65536 int16 numerators each.  ms_selftest_divide_range walks [1e-5, 64) with it: the proof, by
// enumeration, that DivBy is the correctly rounded quotient for every weight sum a context can hold.
__global__ void __launch_bounds__(256) k_selftest_divide_range(unsigned bits0, unsigned n, unsigned long long *mismatches)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    unsigned bad = 0;
    if (j < n) {
        const float d = __uint_as_float(bits0 + j);
        const DivBy div(d);
        for (int i = -32768; i < 32768; ++i) {
            const float a = (float)i;
            bad += __float_as_uint(a / d) != __float_as_uint(div(a));
        }
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// exhaustive check of the one-instruction saturate_cast<uchar>(float) against its definition: all 2^32 bit patterns
__global__ void __launch_bounds__(256) k_selftest_cvt_u8(unsigned long long *mismatches)
{
    unsigned bad = 0;
    const unsigned base = (blockIdx.x * 256u + threadIdx.x) << 8;        // 2^24 lanes x 256 patterns
    for (unsigned i = 0; i < 256u; ++i) {
        const float v = __uint_as_float(base | i);
        bad += (unsigned)sat_u8(v) != (unsigned)sat_u8_ref(v);
        bad += (sat_u8_into(v, 2u, 0x11223344u) != ((0x11003344u) | ((unsigned)sat_u8_ref(v) << 16)));
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// PMC calibration AND the bandwidth ceiling the per-frame kernels are compared with: a tuned 16-byte-per-lane
// streaming copy of a known size.  Four independent non-temporal loads in flight per lane, each wave-instruction one contiguous 1 KiB
// segment, 16 workgroups of 256 lanes per CU: the fastest of the 130 variants of tools/copy_probe.hip on MI355X (6.08 TB/s read + written;
// the one-load-in-flight loop this replaces reached 4.87 - 5.49 TB/s depending on the box; profiles/r03_copy_probe.txt).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_calib_copy(const u32x4_t *__restrict__ src, u32x4_t *__restrict__ dst, size_t n16)
{
    constexpr int U = 4;
    const size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n16) v[u] = __builtin_nontemporal_load(src + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n16) __builtin_nontemporal_store(v[u], dst + i);
        }
    }
}
// read-only companion (XOR-reduced, one dword written per workgroup at most): what the read side alone sustains (7.0 - 7.2 TB/s non-temporal)
__global__ void __launch_bounds__(256) k_calib_read(const u32x4_t *__restrict__ src, unsigned *__restrict__ sink, size_t n16)
{
    constexpr int U = 8;
    const size_t chunk = (size_t)U * 256;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            v[u] = i < n16 ? __builtin_nontemporal_load(src + i) : acc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) *sink = 1u;      // (keeps the loads alive)
}

// ---- static-table kernels ---------------------------------------------------------------------
// 1-D terms of the backward map: coltab[x] = f(tl_u + x), rowtab[y] = g(tl_v + y)
// Counter calibration on the access SHAPES of the per-frame kernels (FETCH_SIZE x 2.0 is calibrated on the 16 B / lane stream only).
// Every 128-byte line of the buffer is touched exactly once per launch, so the HBM bytes a launch must fetch are the buffer size, whatever the shape:
//   shape 0: k_warp_t's tap read -- one dword-aligned 12-byte read per lane at a 24-byte stride (half of the bytes of every line are used);
//   shape 1: k_blend8 / k_down_tail's row windows -- one 8-byte read per lane, contiguous across the wave;
//   shape 2: k_warp_t's plane stores -- one dword store per lane, 32 contiguous bytes per 8 lanes, rows of 32 bytes at a 128-byte pitch ... four passes
//            (blockIdx.y) fill the lines: what WRITE_SIZE reports for partial-line stores that the L2 has to merge.
typedef unsigned ms_u32x3_cal __attribute__((ext_vector_type(3), aligned(4)));
template <int shape>      // (a template so that kernel traces and PMC summaries name the shapes apart)
__global__ void __launch_bounds__(256) k_calib_shape(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, unsigned *__restrict__ sink, size_t bytes)
{
    unsigned acc = 0u;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
    if (shape == 0) {
        for (size_t i = tid; 24 * i + 12 <= bytes; i += nthr) {
            const ms_u32x3_cal v = *(const __attribute__((address_space(1))) ms_u32x3_cal *)(uintptr_t)(src + 24 * i);
            acc ^= v.x ^ v.y ^ v.z;
        }
    } else if (shape == 1) {
        for (size_t i = tid; 8 * i + 8 <= bytes; i += nthr) {
            uint2 v;
            __builtin_memcpy(&v, __builtin_assume_aligned(src + 8 * i, 8), 8);
            acc ^= v.x ^ v.y;
        }
    } else {
        const size_t q = blockIdx.y;                              // which 32-byte quarter of every 128-byte line this pass writes
        for (size_t i = tid; 128 * (i >> 3) + 128 <= bytes; i += nthr)
            *reinterpret_cast<unsigned *>(dst + 128 * (i >> 3) + 32 * q + 4 * (i & 7)) = (unsigned)i;
    }
    if (acc == 0x9e3779b9u) *sink = 1u;      // (keeps the loads alive)
}

__global__ void __launch_bounds__(256) k_warp_tabs(int proj, int tl_u, int tl_v, int cols, int rows, float2 *coltab, float2 *rowtab, WarpParams P)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < cols) coltab[i] = warp_col_term(proj, (float)(tl_u + i), P);
    if (i < rows) rowtab[i] = warp_row_term(proj, (float)(tl_v + i), P);
}
// warp(255-mask, INTER_NEAREST, BORDER_CONSTANT) == "does the truncated map coordinate hit the source"
__global__ void __launch_bounds__(256) k_valid_mask(const float *__restrict__ mx, const float *__restrict__ my, int pitch, int rows, int cols,
                                                    int src_rows, int src_cols, uint8_t *__restrict__ mask, int mpitch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const int xx = f2i_rz(mx[(size_t)y * pitch + x]), yy = f2i_rz(my[(size_t)y * pitch + x]);
    mask[(size_t)y * mpitch + x] = (xx >= 0 && xx < src_cols && yy >= 0 && yy < src_rows) ? 255 : 0;
}
// dst_w(rc) += w   (the weight half of addSrcWeightKernel32F, run once: the sums are frame-invariant)
__global__ void __launch_bounds__(256) k_acc_weight(const float *__restrict__ w, int wpitch, int rows, int cols, float *dst, int dpitch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    dst[(size_t)y * dpitch + x] = dst[(size_t)y * dpitch + x] + w[(size_t)y * wpitch + x];
}
// The same sums for one band in ONE launch (ms_update_mask): dst_w = ((0 + w_0) + w_1) + ... over the views that cover the pixel, in view order --
// the float additions k_acc_weight performs view after view --, then den = sum + WEIGHT_EPS and (level 0) mask = sum > WEIGHT_EPS.
__global__ void __launch_bounds__(256) k_den_all(const ViewDesc *__restrict__ views, int n_views, int l, float *den, int dpitch, int rows, int cols,
                                                 uint8_t *mask, int mpitch, int mrows, int mcols)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    float s = 0.f;
    for (int v = 0; v < n_views; ++v) {
        const LevelDesc &L = views[v].lv[l];
        const int lx = x - L.x_tl, ly = y - L.y_tl;
        if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) continue;
        s = s + L.wgt[(size_t)ly * L.wpitch + lx];
    }
    if (mask && x < mcols && y < mrows) mask[(size_t)y * mpitch + x] = s > 1e-5f ? 255 : 0;
    den[(size_t)y * dpitch + x] = s + 1e-5f;
}
// Owner map of one band: one wave per 64 x 16 cell of the pano level.  The cell is owned by view v when v is the only view with a non-zero weight in it,
// the cell lies inside the pano level and inside v's rect, and all of v's weights there are exactly 1.0f; 255 otherwise (see PanoDesc::pure).
__global__ void __launch_bounds__(64) k_owner_map(const ViewDesc *__restrict__ views, int n_views, int l, int qw, int qh, uint8_t *__restrict__ pure, int ppitch)
{
    const int cx = blockIdx.x, cy = blockIdx.y, lane = threadIdx.x;
    const int x0 = cx * 64, y0 = cy * 16;
    const int px = x0 + 16 * (lane & 3), py = y0 + (lane >> 2);      // 16 pixels of one row per lane
    int owner = -1, cnt = 0;
    bool ones = false, binary = true, clash = false;
    unsigned taken = 0u;                                     // pixels of this lane some view has claimed with a non-zero weight
    for (int v = 0; v < n_views; ++v) {
        const LevelDesc &L = views[v].lv[l];
        bool nz = false, one = true;
        const int ly = py - L.y_tl;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int lx = px + k - L.x_tl;
            if (lx < 0 || ly < 0 || lx >= L.w || ly >= L.h) { one = false; continue; }
            const float w = L.wgt[(size_t)ly * L.wpitch + lx];
            nz = nz || w != 0.f;
            one = one && w == 1.0f;
            if (w != 0.f) {
                binary = binary && w == 1.0f;
                clash = clash || ((taken >> k) & 1u);
                taken |= 1u << k;
            }
        }
        if (__ballot(nz) != 0ull) { owner = v; ++cnt; ones = __ballot(!one) == 0ull; }
    }
    const bool inside = x0 + 64 <= qw && y0 + 16 <= qh;
    // 254 (level 0 only, where k_blend8 has the 8-bit masks): several views meet in the cell but every PIXEL has at most one non-zero weight and it is exactly 1.0f
    // (binary seam masks: mask * (1/255) is 0 or 1) -- per pixel the same arithmetic as an owned cell, with the owner picked by the mask bytes
    const bool exclusive = l == 0 && inside && __ballot(!binary || clash) == 0ull;
    if (lane == 0) pure[(size_t)cy * ppitch + cx] = (cnt == 1 && ones && inside) ? (uint8_t)owner : (exclusive ? (uint8_t)254 : (uint8_t)255);
}
// ms_update_mask: the re-warped mask replaces the view's effective mask only while the mesh displaces by no more than the margin the work lists were
// planned for (measured on the device by ms_set_mesh); otherwise the effective mask -- and with it every table derived from it -- stays as it was.
__global__ void __launch_bounds__(256) k_mask_select(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, size_t n, const unsigned *__restrict__ disp_bits, unsigned limit_bits)
{
    if (*disp_bits > limit_bits) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// den = sum + WEIGHT_EPS ; mask = sum > WEIGHT_EPS (level 0, unpadded ROI)
__global__ void __launch_bounds__(256) k_finish_den(float *den, int dpitch, int rows, int cols, uint8_t *mask, int mpitch, int mrows, int mcols)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const float s = den[(size_t)y * dpitch + x];
    if (mask && x < mcols && y < mrows) mask[(size_t)y * mpitch + x] = s > 1e-5f ? 255 : 0;
    den[(size_t)y * dpitch + x] = s + 1e-5f;
}
// CPW mesh -> backward map: scatter-average (APP/meshwarper.cpp:859-875)
__global__ void __launch_bounds__(256) k_mesh_scatter(const float *__restrict__ bx, const float *__restrict__ by, int pitch, int rows, int cols,
                                                      float *sx, float *sy, float *cnt, int hw, int hh)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const float fx = bx[(size_t)y * pitch + x], fy = by[(size_t)y * pitch + x];
    if (!(fx > -2147483648.f && fx < 2147483648.f && fy > -2147483648.f && fy < 2147483648.f)) return;
    const int x_ = (int)fx / 2, y_ = (int)fy / 2;
    if (x_ >= 0 && y_ >= 0 && x_ < hw && y_ < hh) {
        atomicAdd(&sx[(size_t)y_ * hw + x_], (float)x);   // integer-valued partial sums: order-independent while < 2^24
        atomicAdd(&sy[(size_t)y_ * hw + x_], (float)y);
        atomicAdd(&cnt[(size_t)y_ * hw + x_], 1.f);
    }
}
__global__ void __launch_bounds__(256) k_mesh_mean(float *sx, float *sy, const float *__restrict__ cnt, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sx[i] = sx[i] / cnt[i];     // 0/0 -> NaN hole, as the reference (remap then yields 0)
    sy[i] = sy[i] / cnt[i];
}

// custom_resize (APP/resize.cu:9-27) of one output element, the taps fetched through `at(row, col)`: the same expression tree as
// k_custom_resize (prims.hip), so a fused consumer sees the bits the materialised map would hold
// one axis of custom_resize: cell index i = x * (n - 1) / t (integer division) and fraction ((float)x * (float)(n - 1) / (float)t) - (float)i.
// While t * (n - 1) < 2^23 the float quotient q is the correctly rounded value of an exact ratio whose distance to the next integer is at
// least 1/t > ulp(q)/2, so (int)q IS the integer quotient and the ~40-instruction integer division disappears (`exact` says which case).
__device__ __forceinline__ void resize_axis(int x, int n, int t, bool exact, int &i, float &frac)
{
    const float q = ((float)x) * (float)(n - 1) / (float)t;
    i = exact ? (int)q : x * (n - 1) / t;
    frac = q - (float)i;
}
__device__ __forceinline__ bool resize_axis_exact(int n, int t) { return (long long)t * (n - 1) < (1ll << 23); }
template <class At>
__device__ __forceinline__ float custom_resize_at(int tx, int ty, int cols, int rows, int x, int y, At at)
{
    int left, top;
    float uu, vv;
    resize_axis(x, cols, tx, resize_axis_exact(cols, tx), left, uu);
    resize_axis(y, rows, ty, resize_axis_exact(rows, ty), top, vv);
    float r = ((1.f - uu) * (1.f - vv)) * at(top, left);
    r = __builtin_fmaf(uu * (1.f - vv), at(top, left + 1), r);
    r = __builtin_fmaf((1.f - uu) * vv, at(top + 1, left), r);
    r = __builtin_fmaf(uu * vv, at(top + 1, left + 1), r);
    return r;
}
// convertMeshesToMap, first half in one launch (meshwarper.cpp:838-869): vertex mesh -> per-pixel forward position (custom_resize, not
// materialised) -> scatter into the half-resolution sums.  Also clears what the previous update dirtied in the other accumulator
// (ping-pong: no memset call between updates) and the displacement word of this update.
constexpr int MESH_SR = 4;                         // rows per lane: a workgroup scatters a 64 x 16 pixel block
constexpr int MESH_WW = 48, MESH_WH = 20, MESH_WMX = 6, MESH_WMY = 4;   // LDS window (half-resolution cells) around where the block's first pixel lands
// one view's update: what both launches of convertMeshesToMap need (ms_set_meshes runs all views of a context in ONE pair of launches, blockIdx.z = view)
struct MeshJob {
    const float *smx, *smy;                    // N x M vertex mesh (device)
    unsigned long long *ax, *ay;               // half-resolution accumulators of this update
    unsigned long long *clear; size_t n_clear; // what the previous update dirtied in the other accumulator
    unsigned *disp_word;
    float *dx, *dy;                            // dense maps being written (the view's inactive buffer)
    int aw, ah, hw, hh, pitch, tiles_x, n_tiles;
};
struct MeshJobs { MeshJob j[MAX_VIEWS]; };
__device__ __forceinline__ void mesh_expand_scatter_block(const MeshJob &J, int N, int M, int bx, int by, int gx, int gy)
{
    const float *__restrict__ smx = J.smx, *__restrict__ smy = J.smy;
    const int aw = J.aw, ah = J.ah, hw = J.hw, hh = J.hh;
    unsigned long long *ax = J.ax, *ay = J.ay, *__restrict__ clear = J.clear;
    const size_t n_clear = J.n_clear;
    unsigned *disp_word = J.disp_word;
    // The pixels of a block land in a compact patch of the half-resolution grid (the mesh is a smooth deformation), and four of them share a
    // cell: accumulate the patch in LDS and send one pair of global atomics per touched cell instead of one per pixel (device-scope atomics
    // were 60 % of the update's GPU time); pixels landing outside the window go to memory directly.
    __shared__ unsigned long long wx[MESH_WH * MESH_WW], wy[MESH_WH * MESH_WW];
    const int lt = threadIdx.y * 64 + threadIdx.x;
    const size_t tid = ((size_t)by * gx + bx) * 256 + lt, nthreads = (size_t)gx * gy * 256;
    for (size_t k = tid; k < n_clear; k += nthreads) clear[k] = 0ull;
    if (tid == 0) *disp_word = 0u;
    for (int k = lt; k < MESH_WH * MESH_WW; k += 256) { wx[k] = 0ull; wy[k] = 0ull; }
    const bool ex = resize_axis_exact(M, aw), ey = resize_axis_exact(N, ah);
    auto forward = [&](int x, int y, float &fx, float &fy) {          // custom_resize of both vertex maps at (x, y): the expression tree of custom_resize_at
        int left, top;
        float uu, vv;
        resize_axis(x, M, aw, ex, left, uu);
        resize_axis(y, N, ah, ey, top, vv);
        const float w00 = (1.f - uu) * (1.f - vv), w01 = uu * (1.f - vv), w10 = (1.f - uu) * vv, w11 = uu * vv;
        const int i0 = top * M + left, i1 = i0 + M;
        fx = w00 * smx[i0]; fy = w00 * smy[i0];
        fx = __builtin_fmaf(w01, smx[i0 + 1], fx); fy = __builtin_fmaf(w01, smy[i0 + 1], fy);
        fx = __builtin_fmaf(w10, smx[i1], fx);     fy = __builtin_fmaf(w10, smy[i1], fy);
        fx = __builtin_fmaf(w11, smx[i1 + 1], fx); fy = __builtin_fmaf(w11, smy[i1 + 1], fy);
    };
    auto finite_i32 = [](float v) { return v > -2147483648.f && v < 2147483648.f; };
    float ox, oy;
    forward(bx * 64, by * (4 * MESH_SR), ox, oy);          // the same for every lane: the window's anchor
    const int wx0 = (finite_i32(ox) ? (int)ox / 2 : 0) - MESH_WMX, wy0 = (finite_i32(oy) ? (int)oy / 2 : 0) - MESH_WMY;
    __syncthreads();
    const int x = bx * 64 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < MESH_SR; ++r) {
        const int y = by * (4 * MESH_SR) + r * 4 + threadIdx.y;
        if (x >= aw || y >= ah) continue;
        float fx, fy;
        forward(x, y, fx, fy);
        if (!(finite_i32(fx) && finite_i32(fy))) continue;
        const int x_ = (int)fx / 2, y_ = (int)fy / 2;
        if (x_ >= 0 && y_ >= 0 && x_ < hw && y_ < hh) {
            // sum_x += x, sum_y += y, set_values++ as two 64-bit integer adds: [count : 24 | sum : 40].  The reference's float sums are integers,
            // exact (and therefore order-independent) below 2^24, where (float)sum is the very same value
            const unsigned long long vx = (1ull << 40) | (unsigned long long)x, vy = (unsigned long long)y;
            const int cx = x_ - wx0, cy = y_ - wy0;
            if ((unsigned)cx < (unsigned)MESH_WW && (unsigned)cy < (unsigned)MESH_WH) {
                atomicAdd(&wx[cy * MESH_WW + cx], vx);
                atomicAdd(&wy[cy * MESH_WW + cx], vy);
            } else {
                atomicAdd(&ax[(size_t)y_ * hw + x_], vx);
                atomicAdd(&ay[(size_t)y_ * hw + x_], vy);
            }
        }
    }
    __syncthreads();
    for (int k = lt; k < MESH_WH * MESH_WW; k += 256) {
        const unsigned long long vx = wx[k];
        if (vx == 0ull) continue;
        const int cy = k / MESH_WW, cx = k - cy * MESH_WW;
        const size_t g = (size_t)(wy0 + cy) * hw + (wx0 + cx);
        atomicAdd(&ax[g], vx);
        atomicAdd(&ay[g], wy[k]);
    }
}
__global__ void __launch_bounds__(256) k_mesh_expand_scatter(const float *__restrict__ smx, const float *__restrict__ smy, int N, int M, int aw, int ah,
                                                             unsigned long long *ax, unsigned long long *ay, int hw, int hh,
                                                             unsigned long long *__restrict__ clear, size_t n_clear, unsigned *disp_word)
{
    MeshJob J{smx, smy, ax, ay, clear, n_clear, disp_word, nullptr, nullptr, aw, ah, hw, hh, 0, 0, 0};
    mesh_expand_scatter_block(J, N, M, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y);
}
__global__ void __launch_bounds__(256) k_mesh_expand_scatter_all(MeshJobs T, int N, int M)
{
    const MeshJob &J = T.j[blockIdx.z];
    const int gx = (J.aw + 63) / 64, gy = (J.ah + 4 * MESH_SR - 1) / (4 * MESH_SR);
    if ((int)blockIdx.x >= gx || (int)blockIdx.y >= gy) return;          // (the grid is sized for the largest view)
    mesh_expand_scatter_block(J, N, M, (int)blockIdx.x, (int)blockIdx.y, gx, gy);
}
// second half (meshwarper.cpp:870-883): mean (0/0 -> NaN hole) + custom_resize back to the view size for both maps, and the largest
// displacement of the new maps (see k_mesh_disp)
constexpr int MESH_TR = 4;                      // output rows per lane: a workgroup step is a 64 x 16 pixel tile (64 x 4 until round 5: the straddling view's 9 420 tiles were 18 serial
                                                // steps -- each a chain of cell loads, two barriers, stores -- for each of its 512 workgroups: 54 us; four times fewer, fatter steps)
constexpr int MESH_FW = 72, MESH_FH = 4 * MESH_TR / 2 + 4;       // LDS footprint (cells) of a 64 x 16 output block: <= 64 * (hw - 1) / aw + 2 columns, likewise rows
__device__ __forceinline__ void mesh_mean_resize_blocks(const MeshJob &J, int first_tile, int tile_stride)
{
    const unsigned long long *__restrict__ ax = J.ax, *__restrict__ ay = J.ay;
    const int hw = J.hw, hh = J.hh, pitch = J.pitch, aw = J.aw, ah = J.ah, tiles_x = J.tiles_x, n_tiles = J.n_tiles;
    float *__restrict__ dx = J.dx, *__restrict__ dy = J.dy;
    unsigned *disp_word = J.disp_word;
    __shared__ float mxs[MESH_FH][MESH_FW], mys[MESH_FH][MESH_FW];
    __shared__ float wmax[4];
    const bool ex = resize_axis_exact(hw, aw), ey = resize_axis_exact(hh, ah);
    constexpr unsigned long long SUM = (1ull << 40) - 1;
    float d = 0.f;
    // a workgroup walks 64 x 16 output tiles grid-stride: one atomic on the displacement word per workgroup, not per tile (thousands of
    // same-address atomics serialise: they were most of this kernel's time)
    for (int tile = first_tile; tile < n_tiles; tile += tile_stride) {
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int x = txi * 64 + threadIdx.x;
        // footprint of the tile in the half-resolution maps
        const int bx0 = txi * 64, by0 = tyi * (4 * MESH_TR), bx1 = min(bx0 + 63, aw - 1), by1 = min(by0 + 4 * MESH_TR - 1, ah - 1);
        int c0, c1, r0, r1;
        float t;
        resize_axis(bx0, hw, aw, ex, c0, t); resize_axis(bx1, hw, aw, ex, c1, t);
        resize_axis(by0, hh, ah, ey, r0, t); resize_axis(by1, hh, ah, ey, r1, t);
        const int fw = c1 - c0 + 2, fh = r1 - r0 + 2;
        const bool staged = fw <= MESH_FW && fh <= MESH_FH;             // (always for hw = aw / 2; the direct path keeps odd geometries correct)
        if (staged) {
            for (int k = threadIdx.y * 64 + threadIdx.x; k < fw * fh; k += 256) {
                const int rr = k / fw, cc = k - rr * fw, r = r0 + rr, c = c0 + cc;
                float vx = 0.f, vy = 0.f;
                if (r < hh && c < hw) {            // mean = sum / count; 0 / 0 -> NaN hole, as the reference (remap then yields 0)
                    const unsigned long long a = ax[(size_t)r * hw + c];
                    const float n = (float)(a >> 40);
                    vx = (float)(a & SUM) / n;
                    vy = (float)(ay[(size_t)r * hw + c] & SUM) / n;
                }
                mxs[rr][cc] = vx; mys[rr][cc] = vy;
            }
            __syncthreads();
        }
#pragma unroll
        for (int tr = 0; tr < MESH_TR; ++tr) {
        const int y = by0 + 4 * tr + (int)threadIdx.y;
        if (x < aw && y < ah) {
            int left, top;
            float uu, vv;
            resize_axis(x, hw, aw, ex, left, uu);
            resize_axis(y, hh, ah, ey, top, vv);
            const float w00 = (1.f - uu) * (1.f - vv), w01 = uu * (1.f - vv), w10 = (1.f - uu) * vv, w11 = uu * vv;
            float mx, my;
            if (staged) {
                const int cc = left - c0, rr = top - r0;
                mx = w00 * mxs[rr][cc];                            my = w00 * mys[rr][cc];
                mx = __builtin_fmaf(w01, mxs[rr][cc + 1], mx);     my = __builtin_fmaf(w01, mys[rr][cc + 1], my);
                mx = __builtin_fmaf(w10, mxs[rr + 1][cc], mx);     my = __builtin_fmaf(w10, mys[rr + 1][cc], my);
                mx = __builtin_fmaf(w11, mxs[rr + 1][cc + 1], mx); my = __builtin_fmaf(w11, mys[rr + 1][cc + 1], my);
            } else {
                mx = custom_resize_at(aw, ah, hw, hh, x, y, [&](int r, int c) { const unsigned long long v = ax[(size_t)r * hw + c]; return (float)(v & SUM) / (float)(v >> 40); });
                my = custom_resize_at(aw, ah, hw, hh, x, y, [&](int r, int c) { return (float)(ay[(size_t)r * hw + c] & SUM) / (float)(ax[(size_t)r * hw + c] >> 40); });
            }
            dx[(size_t)y * pitch + x] = mx;
            dy[(size_t)y * pitch + x] = my;
            const float a = fabsf(mx - (float)x), b = fabsf(my - (float)y);
            d = fmaxf(d, fmaxf(a == a ? a : 0.f, b == b ? b : 0.f));
        }
        }
        __syncthreads();                                                 // the staging arrays are rewritten by the next tile
    }
    for (int o = 32; o > 0; o >>= 1) d = fmaxf(d, __shfl_xor(d, o));
    if (threadIdx.x == 0) wmax[threadIdx.y] = d;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        d = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (d > 0.f) atomicMax(disp_word, __float_as_uint(d));
    }
}
__global__ void __launch_bounds__(256) k_mesh_mean_resize(const unsigned long long *__restrict__ ax, const unsigned long long *__restrict__ ay, int hw, int hh,
                                                          float *__restrict__ dx, float *__restrict__ dy, int pitch, int aw, int ah, int tiles_x, int n_tiles, unsigned *disp_word)
{
    MeshJob J{nullptr, nullptr, const_cast<unsigned long long *>(ax), const_cast<unsigned long long *>(ay), nullptr, 0, disp_word, dx, dy, aw, ah, hw, hh, pitch, tiles_x, n_tiles};
    mesh_mean_resize_blocks(J, (int)blockIdx.x, (int)gridDim.x);
}
__global__ void __launch_bounds__(256) k_mesh_mean_resize_all(MeshJobs T)
{
    mesh_mean_resize_blocks(T.j[blockIdx.y], (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// an owned device allocation: released with its owner (ms_ctx members, function-local scratch), never copied
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n)
    {
        release();
        if (n == 0) n = 16;
        MS_HIP(hipMalloc(&p, n));
        bytes = n;
        return MS_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace ms

using namespace ms;

struct ms_ctx {
    ms_config cfg{};
    int N = 0;
    float K[MAX_VIEWS][9], R[MAX_VIEWS][9];
    bool have_cam[MAX_VIEWS] = {};
    double gain[MAX_VIEWS];
    // stage flags
    bool maps_built = false, masks_built = false, blender_ready = false;
    // geometry
    ms_rect roi[MAX_VIEWS];
    BlendGeom bg{};
    ViewPad pad[MAX_VIEWS];
    // static device tables
    DevBuf maps;                       // per view xmap | ymap
    DevBuf tabs;                       // per view column table | row table (float2)
    size_t tab_off[MAX_VIEWS] = {};
    WarpParams wparams[MAX_VIEWS];
    size_t map_off[MAX_VIEWS] = {};    // float offset of xmap; ymap follows at + ah*pitch
    int map_pitch[MAX_VIEWS] = {};
    DevBuf masks;                      // per view 8UC1 (aw x ah, pitch = aw)
    size_t mask_off[MAX_VIEWS] = {};
    DevBuf weights;                    // per view per level fp32
    DevBuf wm0;                        // per view padded 8-bit mask (level-0 weights in 1 byte/px)
    size_t wm0_off[MAX_VIEWS] = {};
    size_t w_off[MAX_VIEWS][MAX_LEVELS] = {};
    DevBuf den;                        // per level fp32 over the padded pano
    size_t den_off[MAX_LEVELS] = {};
    DevBuf result_mask;                // 8UC1 fw x fh
    DevBuf view_tab;                   // ViewDesc[N]
    std::vector<ViewDesc> h_views;
    PanoDesc pano{};
    // per-batch device buffers
    DevBuf g0, gl, cl, stage;
    long long g0_stride = 0, gl_stride = 0, cl_stride = 0, stage_stride = 0;
    int max_pw = 0, max_ph = 0, max_aw = 0, max_ah = 0;
    bool down_vec[MAX_LEVELS] = {};    // level l -> l+1 may use the vectorised kernel
    bool blend_vec[MAX_LEVELS] = {};   // band l may use the 2x8 kernel
    // work lists (tiles that are actually needed)
    bool warp_tiled = false;
    int tail_l0 = -1, tail_lds = 0, tail_strips = 1;
    int tail_lds_b = 0, tail_strips_b = 1, tail_sw_b = 16;      // k_down_tail for batches (F > 2): wider strips
    // the fused band kernel started one band finer: used for batches of 1-2 frames (live mode), where a launch costs more than the
    // vectorised kernel saves
    int btail2_t = -1, btail2_lds = 0, btail2_strips = 0;
    int btail_t = -1, btail_lds = 0, btail_strips = 0;      // fused coarse band chain (k_blend_tail): finest band it produces, LDS bytes, strips    // fused coarse-level reduce (k_down_tail): first level it reads, LDS bytes; -1 = off
    float feather_sharpness = -1.f;    // >= 0: single-band weights are FeatherBlender weight maps (ms_init_feather)
    DevBuf warp_tiles, stage1_tiles, down_tiles[MAX_LEVELS], blend_tiles[MAX_LEVELS];
    int n_stage1_tiles = 0, n_stage1_reachable = 0;
    int last_warp_kernel = 0, last_stage1_kernel = 0;      // MS_WARP_KERNEL_* of the last ms_stitch (ms_get_stitch_kernels)
    ms_image fed[MAX_VIEWS] = {};      // ms_feed: the views of the frame being assembled (borrowed until ms_blend)
    unsigned fed_mask = 0;
    DevBuf masks_eff;                  // ms_update_mask: masks re-warped through the CPW mesh (same layout as `masks`)
    // Enqueue-only ms_update_mask (cfg.update_mask_margin > 0): a second copy of every table that depends on the masks.  `tab_active` says which
    // copy ms_stitch reads (0: the members above / below, 1: alt); an update fills the other one on its own stream and swaps under mesh_mu.
    struct AltTables { DevBuf weights, wm0, den, result_mask, pure_maps, view_tab; PanoDesc pano; std::vector<ViewDesc> h_views; } alt;
    int tab_active = 0;
    hipEvent_t tab_ready = nullptr;
    bool tab_wait = false;
    DevBuf mask_tmp, wm_scratch;       // re-warped mask / float weight map of the largest view
    size_t w_total = 0, wm0_total = 0, den_total = 0, pure_total = 0, pure_off[MAX_LEVELS] = {};
    std::atomic<bool> l0_integer_only{false};      // (atomic: launch_owner_maps clears it from the mask-update thread outside mesh_mu while ms_stitch reads it -- found by the ThreadSanitizer run of stitch_app --update-mask)
                                                   // level 0 has an owner map without a single general cell (binary, exclusive seam masks): k_blend8's integer-only build (88 VGPRs) runs it;
                                       // counted when build_plan makes the map, dropped by the first enqueue-only mask update (whose maps the host never sees)
    bool use_eff[MAX_VIEWS] = {};
    DevBuf pure_maps;                  // owner maps of the bands (PanoDesc::pure)
    DevBuf disp_dev;                   // [view][mesh buffer]: max |mesh map - identity| as float bits, written by ms_set_mesh
    int n_warp_tiles = 0, n_down_tiles[MAX_LEVELS] = {}, n_blend_tiles[MAX_LEVELS] = {};
    int warp_lds_tiles = 0;            // tiles whose source bounding box fits a staging buffer of k_warp_a
    bool warp_aligned = false;         // projection warp with the aligned 12-byte tap reads (k_warp_t<.., AL = true>): chosen from the tiles' minification
    double warp_minification = 0;      // mean source columns per output column over the warp tiles
    int n_cus = 256;
    double plan_fraction = 1.0;        // needed level-0 pixels / padded pixels
    // CPW mesh maps, double buffered
    DevBuf mesh[2];
    size_t mesh_off[MAX_VIEWS] = {};
    int mesh_active[MAX_VIEWS] = {};   // which buffer ms_stitch reads for this view
    bool mesh_set[MAX_VIEWS] = {};
    DevBuf mesh_tmp;                   // scratch for convertMeshesToMap: vertex mesh x|y, two half-resolution accumulators ([count:24|sum_x:40], [sum_y]) used in turn
    size_t mesh_small_cap = 0, mesh_half_cap = 0, mesh_dirty = 0;   // capacities (floats / cells); 64-bit words the previous update dirtied in its accumulator
    int mesh_parity = 0;
    DevBuf mesh_all;                   // scratch of ms_set_meshes (all views in one pair of launches): every view's vertex meshes, then per view two accumulator pairs used in turn
    size_t mesh_all_small = 0;         // floats per vertex map the block was sized for
    int mesh_all_parity = 0;
    bool mesh_all_dirty = false;       // the accumulators of the other parity hold the previous call's sums (cleared by the next scatter launch)
    std::mutex mesh_mu;                // guards the active indices / events shared with ms_stitch: held only across enqueues, never across a host wait
    std::mutex mesh_update_mu;         // serialises mesh updates among themselves (shared scratch, staging slots); taken BEFORE mesh_mu
    // Held by ms_stitch for the length of its enqueue and by everything that REBUILDS the static tables (ms_init_blender, and through it the synchronous
    // ms_update_mask): a rebuild on the recalibration thread reallocates weights, sums and work lists, so it must neither overlap a stitch that is
    // being enqueued (this lock) nor one that still runs on the GPU (the rebuild first waits for last_stitch under the lock).  Lock order:
    // mesh_update_mu, tables_mu, mesh_mu.  The enqueue-only update paths (ms_set_mesh, ms_update_mask with a margin) never take it.
    std::recursive_mutex tables_mu;
    hipStream_t last_stream = nullptr; bool last_stream_set = false;
    hipEvent_t last_stitch = nullptr;
    std::atomic<bool> stitch_pending{false};
    // asynchronous recalibration: a mesh update only enqueues work; `mesh_ready[v]` is recorded behind it and the next ms_stitch makes
    // its stream wait for it; `mesh_chain` orders updates among themselves (they share the scratch and the staging buffers)
    hipEvent_t mesh_ready[MAX_VIEWS] = {}, mesh_chain = nullptr;
    bool mesh_wait[MAX_VIEWS] = {}, mesh_chain_set = false;
    // ms_set_meshes updates every view behind ONE event: a view whose last update was part of such a call is ready when `mesh_chain` is (a later record of mesh_chain is a later
    // point of the same chain of updates).  Twelve event records and as many stream waits per recalibration were 50 us of idle GPU between its kernels and the next stitch.
    bool mesh_ready_via_chain[MAX_VIEWS] = {};
    float *mesh_stage = nullptr;       // pinned host staging of the vertex meshes: a ring of MESH_STAGE_GENS generations of MAX_VIEWS slots + one generation of ms_set_mesh's own behind it (
    size_t mesh_stage_floats = 0;      // ms_set_meshes walks the ring: the host waits for the COPY of the update a whole ring back -- `mesh_stage_ev` --, never for the update before this one)
    static constexpr int MESH_STAGE_GENS = 8;
    int mesh_stage_gen = 0;
    hipEvent_t mesh_stage_ev[MESH_STAGE_GENS] = {};
    bool mesh_stage_ev_set[MESH_STAGE_GENS] = {};
    int canvas_x = 0, canvas_y = 0;
    // view sharding (ms_config.view_shards = shard count S, view_shard_index = this shard's index): contiguous blocks of views per shard
    unsigned own_mask = 0xffffffffu;
    // pano-column sharding (ms_config.col_shards / col_shard_index): the window of pano-ROI columns this context composites and the views it reads for it
    int col_begin = 0, col_end = 0;    // 0, 0 = whole panorama
    unsigned needed_mask = 0xffffffffu;
    long long pacc_stride = 0;         // elements per frame of a partial-accumulator buffer
};

namespace ms {

static int ctx_check_view(const ms_ctx *c, int v)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    if (v < 0 || v >= c->N) return fail(MS_ERR_INVALID, "view index %d out of range [0,%d)", v, c->N);
    return MS_OK;
}

static ms_image view_map_image(const ms_ctx *c, int v, int which)
{
    ms_image m;
    const int aw = c->roi[v].width, ah = c->roi[v].height;
    float *base = (float *)c->maps.p + c->map_off[v] + (which ? (size_t)ah * c->map_pitch[v] : 0);
    m.data = base; m.step = (size_t)c->map_pitch[v] * sizeof(float); m.rows = ah; m.cols = aw; m.type = MS_32FC1;
    return m;
}

}  // namespace ms


// ---- the plan: which tiles of which view/level are needed ------------------------------------------------
// W_l = {w_{v,l} != 0};  N_l = pixels of Gaussian level l some consumer reads:
//   band l reads G_l on W_l and G_{l+1} on the 3x3 neighbourhood of W_l/2 (pyrUp taps);
//   pyrDown reads G_l on the 5x5 neighbourhood of 2*N_{l+1}.
// Everything outside is multiplied by a weight that is exactly 0, so never producing it is exact.
namespace ms {
namespace {

using Bits = std::vector<uint8_t>;
constexpr int CPW_DMAX = 32;      // px; meshes that displace further fall back to warping whole views in CPW stage 1

Bits dilate(const Bits &s, int w, int h, int r)
{
    Bits t((size_t)w * h, 0), o((size_t)w * h, 0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t m = 0;
            for (int k = std::max(0, x - r); k <= std::min(w - 1, x + r) && !m; ++k) m = s[(size_t)y * w + k];
            t[(size_t)y * w + x] = m;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t m = 0;
            for (int k = std::max(0, y - r); k <= std::min(h - 1, y + r) && !m; ++k) m = t[(size_t)k * w + x];
            o[(size_t)y * w + x] = m;
        }
    return o;
}
Bits half(const Bits &s, int w, int h)        // OR over 2x2 -> ((w+1)/2, (h+1)/2)
{
    const int hw = (w + 1) / 2, hh = (h + 1) / 2;
    Bits o((size_t)hw * hh, 0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            if (s[(size_t)y * w + x]) o[(size_t)(y / 2) * hw + x / 2] = 1;
    return o;
}
Bits twice(const Bits &s, int w, int h, int W, int H)   // nearest upsample to (W, H)
{
    Bits o((size_t)W * H, 0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) o[(size_t)y * W + x] = s[(size_t)std::min(y / 2, h - 1) * w + std::min(x / 2, w - 1)];
    return o;
}
bool any_in(const Bits &s, int w, int h, int x0, int y0, int tw, int th)
{
    for (int y = std::max(y0, 0); y < std::min(y0 + th, h); ++y)
        for (int x = std::max(x0, 0); x < std::min(x0 + tw, w); ++x)
            if (s[(size_t)y * w + x]) return true;
    return false;
}

}  // namespace

// Workgroup b runs on XCD b % 8 (observed dispatch order; performance only).  Reorder a tile list so that each
// XCD receives a CONTIGUOUS run of spatially adjacent tiles: neighbouring tiles share source rows / halo rows, and
// only then do those re-reads hit in that XCD's private 4 MiB L2 instead of going back to HBM.
// chunk > 0 (round 4): instead of eight long runs -- eight XCDs at eight far-apart places of the frame -- chunks of `chunk` raster-consecutive tiles are dealt round-robin to
// the XCDs, so that at any time the whole chip works inside one compact window of a few tile rows (horizontal neighbours still share an XCD; the vertical halos
// now meet in the Infinity Cache instead of an L2).  A few per cent either way depending on the list: the callers say which lists use it.
template <typename T>
static void xcd_order(std::vector<T> &tiles, int chunk = 0)
{
    const size_t n = tiles.size(), per = (n + 7) / 8;
    if (n < 16) return;
    std::vector<T> out;
    out.reserve(n);
    if (chunk > 0) {
        std::vector<std::vector<T>> q(8);
        for (size_t i = 0; i < n; ++i) q[(i / (size_t)chunk) % 8].push_back(tiles[i]);
        for (size_t i = 0; out.size() < n; ++i)
            for (int k = 0; k < 8; ++k) if (i < q[k].size()) out.push_back(q[k][i]);
        // (queues of unequal length: the tail of the list loses the b mod 8 alignment -- a few tiles)
        tiles.swap(out);
        return;
    }
    std::vector<char> used(n, 0);
    for (size_t b = 0; out.size() < n; ++b) {
        const size_t src = (b % 8) * per + b / 8;
        if (src < n && !used[src]) { used[src] = 1; out.push_back(tiles[src]); }
        if (b > 16 * n) break;
    }
    for (size_t i = 0; i < n; ++i) if (!used[i]) out.push_back(tiles[i]);
    tiles.swap(out);
}

static bool sharded_ctx(const ms_ctx *c) { return c->own_mask != ((c->N >= 32) ? 0xffffffffu : ((1u << c->N) - 1u)); }

// k_owner_map for every band that has a map, into `pure` (laid out by build_plan), from the weights `vt` points at
static int launch_owner_maps(ms_ctx *c, const ViewDesc *vt, uint8_t *pure, hipStream_t st)
{
    // whoever rewrites the owner maps invalidates what the host knows about them: the integer-only level-0 build (k_blend8<true, 0, 1>, which has NO general path and
    // leaves a general cell unwritten) may only run after the caller has looked at the new level-0 map again (build_plan does; the enqueue-only mask update never
    // sees its maps on the host and stays on the all-classes build).  Cleared HERE so that a future caller cannot forget.
    c->l0_integer_only = false;
    for (int l = 0; l < c->pano.nb; ++l) {
        if (!c->pure_off[l]) continue;
        const int pw_ = div_up(c->pano.qw[l], 64), ph_ = div_up(c->pano.qh[l], 16);
        k_owner_map<<<dim3(pw_, ph_), 64, 0, st>>>(vt, c->N, l, c->pano.qw[l], c->pano.qh[l], pure + (c->pure_off[l] - 1), pw_);
        MS_LAUNCH_CHECK();
    }
    return MS_OK;
}

static int build_plan(ms_ctx *c)
{
    const int N = c->N, nb = c->pano.nb;
    std::vector<float> hw(c->weights.bytes / sizeof(float));
    MS_HIP(hipMemcpy(hw.data(), c->weights.p, hw.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<std::vector<Bits>> W(N, std::vector<Bits>(nb + 1)), Nd(N, std::vector<Bits>(nb + 1));
    for (int v = 0; v < N; ++v)
        for (int l = 0; l <= nb; ++l) {
            const LevelDesc &L = c->h_views[v].lv[l];
            Bits &b = W[v][l];
            b.assign((size_t)L.w * L.h, 0);
            const float *p = hw.data() + c->w_off[v][l];
            for (int y = 0; y < L.h; ++y)
                for (int x = 0; x < L.w; ++x) b[(size_t)y * L.w + x] = p[(size_t)y * L.wpitch + x] != 0.f;
        }
    // Enqueue-only ms_update_mask: the lists are planned for every mask whose support stays within `margin` + 2 px (the bilinear taps) of the
    // original one -- the support of its level-l weights then stays within d_l of the original support, d_0 = margin + 2, d_{l+1} = ceil(d_l / 2) + 1 (pyrDown).
    // Zero weights inside the planned region cost work, never correctness; the owner maps are made from the true weights on the device.
    if (c->cfg.update_mask_margin > 0) {
        int d = c->cfg.update_mask_margin + 2;
        for (int l = 0; l <= nb; ++l) {
            for (int v = 0; v < N; ++v) {
                const LevelDesc &L = c->h_views[v].lv[l];
                W[v][l] = dilate(W[v][l], L.w, L.h, d);
            }
            d = (d + 1) / 2 + 1;
        }
    }
    // Pano-column window: band l is needed on the columns R_l = [ra_l, rb_l) only, R_0 = the window, R_{l+1} = the pyrUp taps of R_l.  The needed
    // regions below are derived from the weights restricted to R_l (Wn); the band kernels still composite whole tiles, and what a straddling tile
    // computes outside R_l is computed from pyramid pixels nobody produced: never read by a needed pixel, like every other unplanned pixel.
    const bool windowed = c->col_end > c->col_begin;
    int ra[MAX_LEVELS + 1], rb[MAX_LEVELS + 1];
    std::vector<std::vector<Bits>> Wn_store;
    if (windowed) {
        ra[0] = c->col_begin; rb[0] = c->col_end;
        for (int l = 0; l < nb; ++l) { ra[l + 1] = std::max(ra[l] / 2 - 1, 0); rb[l + 1] = std::min((rb[l] + 1) / 2 + 1, c->pano.qw[l + 1]); }
        Wn_store = W;
        for (int v = 0; v < N; ++v)
            for (int l = 0; l <= nb; ++l) {
                const LevelDesc &L = c->h_views[v].lv[l];
                for (int x = 0; x < L.w; ++x)
                    if (x + L.x_tl < ra[l] || x + L.x_tl >= rb[l])
                        for (int y = 0; y < L.h; ++y) Wn_store[v][l][(size_t)y * L.w + x] = 0;
            }
    }
    const std::vector<std::vector<Bits>> &Wn = windowed ? Wn_store : W;
    double need0 = 0, tot0 = 0;
    for (int v = 0; v < N; ++v) {
        const ViewDesc &V = c->h_views[v];
        for (int l = nb; l >= 0; --l) {
            const LevelDesc &L = V.lv[l];
            Bits n = Wn[v][l];
            if (l >= 1) {   // pyrUp taps of band l-1
                const LevelDesc &F = V.lv[l - 1];
                Bits u = dilate(half(Wn[v][l - 1], F.w, F.h), L.w, L.h, 1);
                for (size_t i = 0; i < n.size(); ++i) n[i] |= u[i];
            }
            if (l < nb) {   // pyrDown taps of level l+1
                const LevelDesc &C = V.lv[l + 1];
                Bits d = dilate(twice(Nd[v][l + 1], C.w, C.h, L.w, L.h), L.w, L.h, 2);
                for (size_t i = 0; i < n.size(); ++i) n[i] |= d[i];
            }
            Nd[v][l] = std::move(n);
        }
        tot0 += (double)V.pw * V.ph;
    }
    // warp tiles (level 0)
    c->warp_tiled = true;
    for (int v = 0; v < N; ++v) c->warp_tiled = c->warp_tiled && (c->h_views[v].pw % 4 == 0);
    // the tile kernels read tap rows as 8-byte pairs clamped into the image: needs >= 3 columns and >= 2 rows of source
    c->warp_tiled = c->warp_tiled && c->cfg.src_width >= 3 && c->cfg.src_height >= 2;
    for (int v = 0; v < N; ++v) c->warp_tiled = c->warp_tiled && c->h_views[v].aw >= 3 && c->h_views[v].ah >= 2;
    {
        std::vector<WarpTile> tiles;
        for (int v = 0; v < N; ++v) {
            if (!((c->own_mask >> v) & 1u)) continue;     // view sharding: another rank warps this view
            const ViewDesc &V = c->h_views[v];
            for (int y0 = 0; y0 < V.ph; y0 += WARP_TH)
                for (int x0 = 0; x0 < V.pw; x0 += WARP_TW)
                    if (any_in(Nd[v][0], V.pw, V.ph, x0, y0, WARP_TW, WARP_TH)) {
                        WarpTile t{};
                        t.view = (short)v; t.x0 = (short)x0; t.y0 = (short)y0;
                        if (x0 >= V.left && x0 + WARP_TW <= V.left + V.aw && y0 >= V.top && y0 + WARP_TH <= V.top + V.ah) {
                            t.flags |= 4;         // interior tile: its table entries are known without the view descriptor
                            t.ctab = (int)(c->tab_off[v] + (size_t)(x0 - V.left));
                            t.rtab = (int)(c->tab_off[v] + (size_t)round_up(V.aw, 4) + (size_t)(y0 - V.top));
                        }
                        tiles.push_back(t);
                        need0 += (double)std::min(WARP_TW, V.pw - x0) * std::min(WARP_TH, V.ph - y0);
                    }
        }
        c->needed_mask = 0;
        for (const WarpTile &t : tiles) c->needed_mask |= 1u << t.view;
        if (!windowed || !c->warp_tiled) c->needed_mask = 0xffffffffu;      // (the full-grid fallback kernels touch every view)
        // (chunks of 8 tiles dealt round-robin: measured, profiles/r04_warp_experiments.txt -- config 2 k_warp 402 -> 388 us per 32 frames, config 5 -2..-5 %; the CPW mesh remap
        //  of the shipped rig +1.5 %, so CPW contexts keep the eight long runs)
        // round 5: chunks of 32 (about one tile row of an ordinary view) instead of 8 for the projection warp: with three frames per lane config 2's k_warp 360-370 -> 352-356 us per
        // 32 frames and its FETCH 42.3 -> 39.5 MB per frame (fewer chunk seams inside a tile row: horizontal neighbours share 128-byte source lines), config 5's 714 -> 698 us; whole tile
        // rows per XCD (unequal runs) and chunks of 60 are slower again (profiles/r05_experiments.txt)
        if (c->cfg.raster_tile_order == 0) xcd_order(tiles, dev_knob("MS_XCD_CHUNK_WARP", c->cfg.enable_cpw ? 0 : 32));
        c->n_warp_tiles = (int)tiles.size();
        if (int e = c->warp_tiles.alloc(std::max<size_t>(1, tiles.size()) * sizeof(WarpTile))) return e;
        c->warp_lds_tiles = 0;
        if (!tiles.empty()) {
            MS_HIP(hipMemcpy(c->warp_tiles.p, tiles.data(), tiles.size() * sizeof(WarpTile), hipMemcpyHostToDevice));
            if (c->warp_tiled) {   // source bounding box of every tile (static: the projection maps do not change per frame)
                k_tile_bbox<<<c->n_warp_tiles, dim3(WARP_BX, std::min(WARP_TH, 256 / WARP_BX))>>>((WarpTile *)c->warp_tiles.p, (const ViewDesc *)c->view_tab.p, c->cfg.src_height, c->cfg.src_width, 0);
                MS_LAUNCH_CHECK();
                MS_HIP(hipMemcpy(tiles.data(), c->warp_tiles.p, tiles.size() * sizeof(WarpTile), hipMemcpyDeviceToHost));
                double sw_sum = 0; int sw_n = 0;
                for (const WarpTile &t : tiles) {
                    if (t.flags & 1) ++c->warp_lds_tiles;
                    if (t.sw > 0) { sw_sum += t.sw; ++sw_n; }
                }
                // aligned tap reads pay where neighbouring samples share dwords; at strong minification every read is isolated and the 28 extra
                // registers (4 instead of 6 waves per SIMD) cost more than the alignment saves (measured: 1.9x -6 %, 2.7x +17 %)
                c->warp_minification = sw_n ? sw_sum / sw_n / WARP_TW : 0.0;
                c->warp_aligned = c->warp_minification > 0 && c->warp_minification < 2.3;
                { const int k = dev_knob("MS_WARP_ALIGNED", -1); if (k >= 0) c->warp_aligned = k != 0; }
                if (dev_knob("MS_DEBUG_PLAN", 0)) {
                    int big = 0, last = 0, interior = 0; long long bytes = 0;
                    for (const WarpTile &t : tiles) {
                        if (t.flags & 1) bytes += 16ll * warp_lds_np(t.sw) * t.sh;
                        else if (t.sw > 0 && 16 * warp_lds_np(t.sw) * t.sh > WA_BUF_BYTES) ++big; else ++last;
                        interior += (t.flags & 4) != 0;
                    }
                    int al = 0;
                    for (const WarpTile &t : tiles) al += (t.flags & 8) != 0;
                    fprintf(stderr, "[plan] warp tiles %zu: staged %d (mean %.0f B), box too large %d, last row / empty %d; interior %d; clear of the last source row %d; "
                                    "minification %.2f -> %s tap reads\n", tiles.size(), c->warp_lds_tiles,
                            c->warp_lds_tiles ? (double)bytes / c->warp_lds_tiles : 0.0, big, last, interior, al, c->warp_minification, c->warp_aligned ? "aligned" : "unaligned");
                }
            }
        }
    }
    c->plan_fraction = tot0 > 0 ? need0 / tot0 : 1.0;
    if (c->cfg.enable_cpw) {
        // CPW stage 1 covers the whole warped view (the mesh, hence what stage 2 samples, changes at recalibration), but while a mesh
        // moves no sample further than CPW_DMAX px (measured on the device when it is set) stage 2 can only read stage-1 pixels
        // within CPW_DMAX + 2 of the level-0 pixels some consumer needs: those tiles carry flag bit 1, the others exit early.
        std::vector<WarpTile> tiles;
        for (int v = 0; v < N; ++v) {
            if (!((c->own_mask >> v) & 1u) || !((c->needed_mask >> v) & 1u)) continue;
            const ViewDesc &V = c->h_views[v];
            Bits a((size_t)V.aw * V.ah, 0);
            for (int y = 0; y < V.ph; ++y)
                for (int x = 0; x < V.pw; ++x)
                    if (Nd[v][0][(size_t)y * V.pw + x]) {
                        int ax = x - V.left, ay = y - V.top;      // BORDER_REFLECT back into the warped view
                        ax = ax < 0 ? -ax - 1 : (ax >= V.aw ? 2 * V.aw - ax - 1 : ax);
                        ay = ay < 0 ? -ay - 1 : (ay >= V.ah ? 2 * V.ah - ay - 1 : ay);
                        a[(size_t)std::min(std::max(ay, 0), V.ah - 1) * V.aw + std::min(std::max(ax, 0), V.aw - 1)] = 1;
                    }
            const int D = CPW_DMAX + 2;
            for (int y0 = 0; y0 < V.ah; y0 += WARP_TH)
                for (int x0 = 0; x0 < V.aw; x0 += WARP_TW) {
                    WarpTile t{}; t.view = (short)v; t.x0 = (short)x0; t.y0 = (short)y0;
                    t.flags = any_in(a, V.aw, V.ah, x0 - D, y0 - D, WARP_TW + 2 * D, WARP_TH + 2 * D) ? 2 : 0;
                    tiles.push_back(t);
                }
        }
        if (dev_knob("MS_DEBUG_PLAN", 0)) { int nf = 0; for (auto &t : tiles) nf += (t.flags & 2) != 0; fprintf(stderr, "[plan] stage-1 tiles %zu, reachable within %d px: %d\n", tiles.size(), CPW_DMAX, nf); }
        {   // reachable tiles first, each part in XCD order on its own: when the others exit early every XCD still gets an equal share
            std::vector<WarpTile> a, b;
            for (const WarpTile &t : tiles) ((t.flags & 2) ? a : b).push_back(t);
            c->n_stage1_reachable = (int)a.size();
            if (c->cfg.raster_tile_order == 0) { xcd_order(a, dev_knob("MS_XCD_CHUNK_S1", 32)); xcd_order(b, dev_knob("MS_XCD_CHUNK_S1", 32)); }
            tiles = a;
            tiles.insert(tiles.end(), b.begin(), b.end());
        }
        c->n_stage1_tiles = (int)tiles.size();
        if (int e = c->stage1_tiles.alloc(std::max<size_t>(1, tiles.size()) * sizeof(WarpTile))) return e;
        MS_HIP(hipMemcpy(c->stage1_tiles.p, tiles.data(), tiles.size() * sizeof(WarpTile), hipMemcpyHostToDevice));
        if (!tiles.empty() && c->warp_tiled) {     // flags bit 3 of every stage-1 tile: no sample reads the last source row (aligned tap reads allowed)
            k_tile_bbox<<<c->n_stage1_tiles, dim3(WARP_BX, std::min(WARP_TH, 256 / WARP_BX))>>>((WarpTile *)c->stage1_tiles.p, (const ViewDesc *)c->view_tab.p, c->cfg.src_height, c->cfg.src_width, 1);
            MS_LAUNCH_CHECK();
            MS_HIP(hipDeviceSynchronize());
        }
    }
    // pyrDown tiles: output tiles of level l+1
    for (int l = 0; l < nb; ++l) {
        std::vector<DownTile> tiles;
        if (c->down_vec[l])
            for (int v = 0; v < N; ++v) {
                if (!((c->own_mask >> v) & 1u)) continue;
                const LevelDesc &Lo = c->h_views[v].lv[l + 1];
                for (int y0 = 0; y0 < Lo.h; y0 += DOWN_TH)
                    for (int x0 = 0; x0 < Lo.w; x0 += DOWN_TW)
                        if (any_in(Nd[v][l + 1], Lo.w, Lo.h, x0, y0, DOWN_TW, DOWN_TH)) tiles.push_back(DownTile{(short)v, 0, (short)x0, (short)y0});
            }
        if (c->cfg.raster_tile_order == 0) xcd_order(tiles, dev_knob(l == 0 ? "MS_XCD_CHUNK_DOWN0" : "MS_XCD_CHUNK_DOWN", 0));
        c->n_down_tiles[l] = (int)tiles.size();
        if (int e = c->down_tiles[l].alloc(std::max<size_t>(1, tiles.size()) * sizeof(DownTile))) return e;
        if (!tiles.empty()) MS_HIP(hipMemcpy(c->down_tiles[l].p, tiles.data(), tiles.size() * sizeof(DownTile), hipMemcpyHostToDevice));
    }
    // band tiles: every pano tile, with the views that have a non-zero weight in it
    for (int l = 0; l < nb; ++l) {
        std::vector<BlendTile> tiles;
        if (c->blend_vec[l])
            for (int y0 = 0; y0 < c->pano.qh[l]; y0 += BLEND_TH)
                for (int x0 = 0; x0 < c->pano.qw[l]; x0 += BLEND_TW) {
                    if (windowed && (x0 + BLEND_TW <= ra[l] || x0 >= rb[l])) continue;       // no needed column of band l in this tile
                    unsigned m = 0;
                    for (int v = 0; v < N; ++v) {
                        const LevelDesc &L = c->h_views[v].lv[l];
                        if (any_in(W[v][l], L.w, L.h, x0 - L.x_tl, y0 - L.y_tl, BLEND_TW, BLEND_TH)) m |= 1u << v;
                    }
                    tiles.push_back(BlendTile{(short)x0, (short)y0, m});
                }
        if (c->cfg.raster_tile_order == 0) xcd_order(tiles, dev_knob(l == 0 ? "MS_XCD_CHUNK_BLEND0" : "MS_XCD_CHUNK_BLEND", (l == 0 && c->pano.qw[0] <= 4096) ? 8 : 0));      // (level 0 of config 2 / 3 / shipped: -2 %; config 5's 7680-wide panorama: +4 % with chunks, kept on long runs)
        c->n_blend_tiles[l] = (int)tiles.size();
        if (int e = c->blend_tiles[l].alloc(std::max<size_t>(1, tiles.size()) * sizeof(BlendTile))) return e;
        if (!tiles.empty()) MS_HIP(hipMemcpy(c->blend_tiles[l].p, tiles.data(), tiles.size() * sizeof(BlendTile), hipMemcpyHostToDevice));
    }
    // owner maps: 64 x 16 cells of a band where exactly one view has non-zero weights, every one of them exactly 1.0f (k_owner_map, from the weights on the device)
    c->pure_total = 0;
    for (int l = 0; l < nb; ++l) {
        c->pano.pure[l] = nullptr; c->pano.ppitch[l] = 0; c->pure_off[l] = 0;
        if (!c->blend_vec[l] || sharded_ctx(c) || c->cfg.debug_simple_kernels != 0) continue;
        const int pw_ = div_up(c->pano.qw[l], 64), ph_ = div_up(c->pano.qh[l], 16);
        c->pure_off[l] = c->pure_total + 1;          // (+1: 0 means "no map")
        c->pano.ppitch[l] = pw_;
        c->pure_total += ((size_t)pw_ * ph_ + 3) & ~(size_t)3;        // (k_blend8 reads a cell's byte as part of its aligned dword)
    }
    if (c->pure_total) {
        if (int e = c->pure_maps.alloc(c->pure_total)) return e;
        for (int l = 0; l < nb; ++l) if (c->pure_off[l]) c->pano.pure[l] = (const uint8_t *)c->pure_maps.p + (c->pure_off[l] - 1);
        if (int e = launch_owner_maps(c, (const ViewDesc *)c->view_tab.p, (uint8_t *)c->pure_maps.p, nullptr)) return e;
        MS_HIP(hipStreamSynchronize(nullptr));
    }
    c->l0_integer_only = false;
    if (c->pure_off[0]) {
        std::vector<uint8_t> h((size_t)div_up(c->pano.qw[0], 64) * div_up(c->pano.qh[0], 16));
        MS_HIP(hipMemcpy(h.data(), (const uint8_t *)c->pure_maps.p + (c->pure_off[0] - 1), h.size(), hipMemcpyDeviceToHost));
        c->l0_integer_only = std::find(h.begin(), h.end(), (uint8_t)255) == h.end();
    }
    return MS_OK;
}

}  // namespace ms

extern "C" {

int ms_create(const ms_config *cfg, ms_ctx **out)
{
    if (!cfg || !out) return fail(MS_ERR_INVALID, "ms_create: null argument");
    if (int e = require_device()) return e;
    MS_CHECK(cfg->struct_size == sizeof(ms_config), "ms_create: ms_config.struct_size is %u, this library expects %zu (header / library mismatch)", cfg->struct_size, sizeof(ms_config));
    MS_CHECK(cfg->self_check == 0 || cfg->self_check == 1, "ms_create: ms_config.self_check must be 0 or 1");
    MS_CHECK(cfg->update_mask_margin >= 0 && cfg->update_mask_margin <= 64 && (cfg->update_mask_margin == 0 || (cfg->enable_cpw && cfg->num_bands >= 1 && cfg->view_shards <= 1)),
             "ms_create: update_mask_margin %d needs enable_cpw, num_bands >= 1, no view shards, and must be in [0, 64]", cfg->update_mask_margin);
    MS_CHECK(cfg->num_views >= 1 && cfg->num_views <= MAX_VIEWS, "ms_create: num_views %d not in [1,%d]", cfg->num_views, MAX_VIEWS);
    MS_CHECK(cfg->src_width > 1 && cfg->src_height > 1, "ms_create: bad source size %dx%d", cfg->src_width, cfg->src_height);
    MS_CHECK(cfg->projection >= MS_PROJ_PLANE && cfg->projection <= MS_PROJ_SPHERICAL, "ms_create: bad projection %d", cfg->projection);
    MS_CHECK(cfg->warp_scale > 0.f, "ms_create: warp_scale must be positive");
    MS_CHECK(cfg->num_bands >= 0 && cfg->num_bands < MAX_LEVELS, "ms_create: num_bands %d not in [0,%d)", cfg->num_bands, MAX_LEVELS);
    const int F = cfg->max_frames > 0 ? cfg->max_frames : 1;
    MS_CHECK(F <= MAX_FRAMES && cfg->num_views <= MAX_SRC, "ms_create: max_frames %d exceeds the per-call limit %d", F, MAX_FRAMES);
    ms_ctx *c = new (std::nothrow) ms_ctx();
    if (!c) return fail(MS_ERR_NOMEM, "ms_create: out of host memory");
    c->cfg = *cfg;
    c->cfg.max_frames = F;
    c->N = cfg->num_views;
    for (int i = 0; i < MAX_VIEWS; ++i) c->gain[i] = 1.0;
    {
        const int S = cfg->view_shards > 1 ? cfg->view_shards : 1, idx = cfg->view_shard_index;
        if (S > 4 || idx < 0 || idx >= S || S > c->N) { delete c; return fail(MS_ERR_INVALID, "ms_create: bad view-shard setting %d/%d", idx, S); }
        c->own_mask = 0;
        for (int v = idx * c->N / S; v < (idx + 1) * c->N / S; ++v) c->own_mask |= 1u << v;
    }
    if (cfg->col_shards > 1 && (cfg->col_shards > 16 || cfg->col_shard_index < 0 || cfg->col_shard_index >= cfg->col_shards || cfg->view_shards > 1 ||
                                cfg->debug_simple_kernels != 0 || cfg->num_bands < 1)) {
        delete c;
        return fail(MS_ERR_INVALID, "ms_create: bad column-shard setting %d/%d (2..16 shards, not together with view shards or the reference kernels, num_bands >= 1)",
                    cfg->col_shard_index, cfg->col_shards);
    }
    if (cfg->cpu_flavour_remap != 0 && (cfg->debug_simple_kernels == 0 || cfg->enable_cpw)) {
        delete c;
        return fail(MS_ERR_INVALID, "ms_create: cpu_flavour_remap runs in the reference kernels only (debug_simple_kernels = 1) and without CPW");
    }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) c->n_cus = cus;
    }
    if (hipEventCreateWithFlags(&c->last_stitch, hipEventDisableTiming) != hipSuccess) { delete c; return fail(MS_ERR_HIP, "hipEventCreate failed"); }
    *out = c;
    return MS_OK;
}

void ms_destroy(ms_ctx *c)
{
    if (!c) return;
    (void)hipDeviceSynchronize();
    // every DevBuf member (tables, per-batch pyramids, work lists, masks_eff, pure_maps, disp_dev, mesh buffers) frees itself in ~ms_ctx
    if (c->last_stitch) (void)hipEventDestroy(c->last_stitch);
    for (int v = 0; v < MAX_VIEWS; ++v) if (c->mesh_ready[v]) (void)hipEventDestroy(c->mesh_ready[v]);
    if (c->mesh_chain) (void)hipEventDestroy(c->mesh_chain);
    if (c->tab_ready) (void)hipEventDestroy(c->tab_ready);
    if (c->mesh_stage) (void)hipHostFree(c->mesh_stage);
    for (hipEvent_t e : c->mesh_stage_ev) if (e) (void)hipEventDestroy(e);
    delete c;
}

int ms_set_camera(ms_ctx *c, int view, const float *K, const float *R)
{
    if (int e = ctx_check_view(c, view)) return e;
    MS_CHECK(K && R, "ms_set_camera: null matrix");
    memcpy(c->K[view], K, sizeof(float) * 9);
    memcpy(c->R[view], R, sizeof(float) * 9);
    c->have_cam[view] = true;
    c->maps_built = c->masks_built = c->blender_ready = false;
    return MS_OK;
}

int ms_set_gain(ms_ctx *c, int view, double gain)
{
    if (int e = ctx_check_view(c, view)) return e;
    c->gain[view] = gain;
    if (c->blender_ready) {   // keep the device table in sync (gains change at recalibration only)
        c->h_views[view].gain = (float)gain;
        MS_HIP(hipMemcpy((ViewDesc *)c->view_tab.p + view, &c->h_views[view], sizeof(ViewDesc), hipMemcpyHostToDevice));
        if (c->alt.view_tab.p && (int)c->alt.h_views.size() == c->N) {
            c->alt.h_views[view].gain = (float)gain;
            MS_HIP(hipMemcpy((ViewDesc *)c->alt.view_tab.p + view, &c->alt.h_views[view], sizeof(ViewDesc), hipMemcpyHostToDevice));
        }
    }
    return MS_OK;
}

int ms_build_maps(ms_ctx *c, ms_stream stream)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    hipStream_t st = as_stream(stream);
    for (int i = 0; i < c->N; ++i)
        if (!c->have_cam[i]) return fail(MS_ERR_STATE, "ms_build_maps: camera %d not set", i);
    const int W = c->cfg.src_width, H = c->cfg.src_height;
    size_t total = 0;
    for (int i = 0; i < c->N; ++i) {
        Projector p;
        set_camera_params(p, c->K[i], c->R[i], nullptr, c->cfg.warp_scale);
        c->roi[i] = warp_roi(c->cfg.projection, p, W, H);
        MS_CHECK(c->roi[i].width > 0 && c->roi[i].height > 0, "ms_build_maps: empty ROI for view %d", i);
        c->map_pitch[i] = round_up(c->roi[i].width, 4);
        c->map_off[i] = total;
        total += (size_t)2 * c->roi[i].height * c->map_pitch[i];
    }
    if (int e = c->maps.alloc(total * sizeof(float))) return e;
    size_t tab_total = 0;
    for (int i = 0; i < c->N; ++i) { c->tab_off[i] = tab_total; tab_total += (size_t)round_up(c->roi[i].width, 4) + round_up(c->roi[i].height, 4); }
    if (int e = c->tabs.alloc((tab_total + 8) * sizeof(float2))) return e;
    for (int i = 0; i < c->N; ++i) {
        float k_rinv[9];
        k_rinv_gemm(c->K[i], c->R[i], k_rinv);   // warpers_cuda.cpp:108
        ms_image mx = view_map_image(c, i, 0), my = view_map_image(c, i, 1);
        if (int e = launch_build_warp_maps(c->cfg.projection, c->roi[i].x, c->roi[i].y, mx, my, k_rinv, nullptr, c->cfg.warp_scale, st)) return e;
        WarpParams &W = c->wparams[i];
        memcpy(W.k, k_rinv, sizeof(W.k)); W.t[0] = W.t[1] = W.t[2] = 0.f; W.scale = c->cfg.warp_scale;
        float2 *ct = (float2 *)c->tabs.p + c->tab_off[i], *rt = ct + round_up(c->roi[i].width, 4);
        const int n = std::max(c->roi[i].width, c->roi[i].height);
        k_warp_tabs<<<div_up(n, 256), 256, 0, st>>>(c->cfg.projection, c->roi[i].x, c->roi[i].y, c->roi[i].width, c->roi[i].height, ct, rt, W);
        MS_LAUNCH_CHECK();
    }
    // blender->prepare(corners, sizes)
    c->bg = blender_prepare(result_roi(c->N, c->roi), c->cfg.num_bands);
    for (int i = 0; i < c->N; ++i)
        c->pad[i] = blender_view_pad(c->bg, c->roi[i].x, c->roi[i].y, c->roi[i].width, c->roi[i].height);
    c->canvas_x = c->bg.dst_roi.x + c->cfg.out_width / 2;
    c->canvas_y = c->cfg.projection == MS_PROJ_SPHERICAL ? c->bg.dst_roi.y : c->bg.dst_roi.y + c->cfg.out_height / 2;
    MS_HIP(hipStreamSynchronize(st));
    c->maps_built = true;
    c->masks_built = c->blender_ready = false;
    return MS_OK;
}

static int alloc_masks(ms_ctx *c)
{
    size_t total = 0;
    for (int i = 0; i < c->N; ++i) { c->mask_off[i] = total; total += (size_t)c->roi[i].width * c->roi[i].height; }
    for (int i = 0; i < MAX_VIEWS; ++i) c->use_eff[i] = false;     // fresh masks supersede any ms_update_mask result
    return c->masks.bytes >= total && c->masks.p ? MS_OK : c->masks.alloc(total);
}

int ms_build_masks(ms_ctx *c, int mode, ms_stream stream)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    if (!c->maps_built) return fail(MS_ERR_STATE, "ms_build_masks: call ms_build_maps first");
    MS_CHECK(mode == 0 || mode == 1, "ms_build_masks: mode must be 0 or 1");
    hipStream_t st = as_stream(stream);
    if (int e = alloc_masks(c)) return e;
    for (int i = 0; i < c->N; ++i) {
        const int aw = c->roi[i].width, ah = c->roi[i].height;
        ms_image mx = view_map_image(c, i, 0), my = view_map_image(c, i, 1);
        k_valid_mask<<<dim3(div_up(aw, 64), div_up(ah, 4)), dim3(64, 4), 0, st>>>(
            (const float *)mx.data, (const float *)my.data, c->map_pitch[i], ah, aw, c->cfg.src_height, c->cfg.src_width,
            (uint8_t *)c->masks.p + c->mask_off[i], aw);
        MS_LAUNCH_CHECK();
    }
    MS_HIP(hipStreamSynchronize(st));
    if (mode == 1) {   // VoronoiSeamFinder on the device (calib.hip): the masks never leave it
        std::vector<uint8_t *> ptr(c->N);
        for (int i = 0; i < c->N; ++i) ptr[i] = (uint8_t *)c->masks.p + c->mask_off[i];
        if (int e = voronoi_seams_device(c->N, c->roi, ptr.data(), st)) return e;
    }
    c->masks_built = true;
    c->blender_ready = false;
    return MS_OK;
}

// Seam-scale calibration, the reference's own pipeline (APP/calibration.cpp:92-135 and 224-237):
//   resize(full, seam_scale) -> warp(image: LINEAR/REFLECT, mask 255: NEAREST/CONSTANT) at seam scale -> download ->
//   GainCompensator::feed -> VoronoiSeamFinder (both on the device here: calib.hip) -> [dilate] -> resize to the compose mask size
//   (INTER_LINEAR) -> bitwise_and with warp(255) at compose scale  => the masks init_gpu receives.
int ms_calibrate_seam(ms_ctx *c, const ms_image *full_imgs, const float *K_seam, const ms_seam_params *prm, double *gains_out, ms_stream stream)
{
    if (!c || !full_imgs || !K_seam || !prm) return fail(MS_ERR_INVALID, "ms_calibrate_seam: null argument");
    if (!c->maps_built) return fail(MS_ERR_STATE, "ms_calibrate_seam: call ms_build_maps first");
    MS_CHECK(prm->seam_scale > 0 && prm->seam_scale <= 1.0 && prm->seam_warp_scale > 0, "ms_calibrate_seam: bad scales");
    hipStream_t st = as_stream(stream);
    const int N = c->N, W = c->cfg.src_width, H = c->cfg.src_height;
    // the seam images are cut from the FULL frames (calibration.cpp:95), which are larger than the context's source size when compose_scale < 1
    const int FW = full_imgs[0].cols, FH = full_imgs[0].rows;
    MS_CHECK(FW >= W && FH >= H, "ms_calibrate_seam: full frames %dx%d smaller than the context's source size %dx%d", FW, FH, W, H);
    const int ws = (int)__builtin_rint(FW * prm->seam_scale), hs = (int)__builtin_rint(FH * prm->seam_scale);   // resize.cpp:74
    MS_CHECK(ws >= 2 && hs >= 2, "ms_calibrate_seam: seam image %dx%d too small", ws, hs);
    if (int e = alloc_masks(c)) return e;
    std::vector<ms_rect> rs(N);
    std::vector<DevBuf> wimg(N), wmask(N);          // the warped seam images and masks of all views stay on the device
    DevBuf seam, mapx, mapy;
    if (int e = seam.alloc((size_t)ws * hs * 3)) return e;
    for (int i = 0; i < N; ++i) {
        MS_CHECK(full_imgs[i].data && full_imgs[i].type == MS_8UC3 && full_imgs[i].rows == FH && full_imgs[i].cols == FW,
                 "ms_calibrate_seam: image %d must be 8UC3 %dx%d like image 0", i, FW, FH);
        ms_image simg{seam.p, (size_t)ws * 3, ws, hs, MS_8UC3};
        if (int e = launch_resize_linear(full_imgs[i], simg, prm->seam_scale, prm->seam_scale, st)) return e;        // calibration.cpp:95
        const float *Ks = K_seam + 9 * i;
        Projector P;
        set_camera_params(P, Ks, c->R[i], nullptr, prm->seam_warp_scale);
        rs[i] = warp_roi(c->cfg.projection, P, ws, hs);
        MS_CHECK(rs[i].width > 0 && rs[i].height > 0 && (size_t)rs[i].width * rs[i].height < (1u << 28), "ms_calibrate_seam: bad seam ROI for view %d", i);
        const size_t npx = (size_t)rs[i].width * rs[i].height;
        if (int e = mapx.alloc(npx * 4)) return e;
        if (int e = mapy.alloc(npx * 4)) return e;
        if (int e = wimg[i].alloc(npx * 3)) return e;
        if (int e = wmask[i].alloc(npx)) return e;
        ms_image mx{mapx.p, (size_t)rs[i].width * 4, rs[i].width, rs[i].height, MS_32FC1}, my{mapy.p, (size_t)rs[i].width * 4, rs[i].width, rs[i].height, MS_32FC1};
        float k_rinv[9];
        k_rinv_gemm(Ks, c->R[i], k_rinv);
        if (int e = launch_build_warp_maps(c->cfg.projection, rs[i].x, rs[i].y, mx, my, k_rinv, nullptr, prm->seam_warp_scale, st)) return e;
        ms_image wi{wimg[i].p, (size_t)rs[i].width * 3, rs[i].width, rs[i].height, MS_8UC3};
        if (int e = launch_remap(simg, mx, my, wi, MS_INTER_LINEAR, MS_BORDER_REFLECT, st)) return e;                   // calibration.cpp:118
        k_valid_mask<<<dim3(div_up(rs[i].width, 64), div_up(rs[i].height, 4)), dim3(64, 4), 0, st>>>(                // calibration.cpp:122
            (const float *)mapx.p, (const float *)mapy.p, rs[i].width, rs[i].height, rs[i].width, hs, ws, (uint8_t *)wmask[i].p, rs[i].width);
        MS_LAUNCH_CHECK();
        MS_HIP(hipStreamSynchronize(st));            // (mapx / mapy / seam are reused by the next view)
    }
    std::vector<const uint8_t *> ip(N), mp(N);
    std::vector<uint8_t *> mpw(N);
    for (int i = 0; i < N; ++i) { ip[i] = (const uint8_t *)wimg[i].p; mp[i] = (const uint8_t *)wmask[i].p; mpw[i] = (uint8_t *)wmask[i].p; }
    if (prm->estimate_gains || gains_out) {                                                                          // calibration.cpp:131
        std::vector<double> g(N, 1.0);
        if (int e = estimate_gains_device(N, rs.data(), ip.data(), mp.data(), g.data(), st)) return e;               // GainCompensator::feed on the device; N doubles come back
        for (int i = 0; i < N; ++i) {
            if (gains_out) gains_out[i] = g[i];
            if (prm->estimate_gains) if (int e = ms_set_gain(c, i, g[i])) return e;
        }
    }
    if (int e = voronoi_seams_device(N, rs.data(), mpw.data(), st)) return e;                                         // calibration.cpp:134-135, on the device
    DevBuf dil, big;
    for (int i = 0; i < N; ++i) {
        const size_t npx = (size_t)rs[i].width * rs[i].height;
        const int aw = c->roi[i].width, ah = c->roi[i].height;
        ms_image sm{wmask[i].p, (size_t)rs[i].width, rs[i].width, rs[i].height, MS_8UC1};
        if (prm->dilate) {                                                                                            // calibration.cpp:231-232
            if (int e = dil.alloc(npx)) return e;
            ms_image dm{dil.p, (size_t)rs[i].width, rs[i].width, rs[i].height, MS_8UC1};
            if (int e = launch_dilate3(sm, dm, st)) return e;
            sm = dm;
        }
        if (int e = big.alloc((size_t)aw * ah)) return e;
        ms_image bm{big.p, (size_t)aw, aw, ah, MS_8UC1};
        if (int e = launch_resize_linear(sm, bm, 0, 0, st)) return e;                                                // calibration.cpp:236
        ms_image mx = view_map_image(c, i, 0), my = view_map_image(c, i, 1);
        uint8_t *dstm = (uint8_t *)c->masks.p + c->mask_off[i];
        k_valid_mask<<<dim3(div_up(aw, 64), div_up(ah, 4)), dim3(64, 4), 0, st>>>(                                   // calibration.cpp:224-227
            (const float *)mx.data, (const float *)my.data, c->map_pitch[i], ah, aw, H, W, dstm, aw);
        MS_LAUNCH_CHECK();
        ms_image vm{dstm, (size_t)aw, aw, ah, MS_8UC1};
        if (int e = launch_and_8u(bm, vm, vm, st)) return e;                                                         // calibration.cpp:237
        MS_HIP(hipStreamSynchronize(st));
    }
    c->masks_built = true;
    c->blender_ready = false;
    return MS_OK;
}

int ms_set_mask(ms_ctx *c, int view, const uint8_t *mask_host, size_t step)
{
    if (int e = ctx_check_view(c, view)) return e;
    if (!c->maps_built) return fail(MS_ERR_STATE, "ms_set_mask: call ms_build_maps first");
    MS_CHECK(mask_host && step >= (size_t)c->roi[view].width, "ms_set_mask: bad mask/step");
    if (int e = alloc_masks(c)) return e;
    c->use_eff[view] = false;
    MS_HIP(hipMemcpy2D((uint8_t *)c->masks.p + c->mask_off[view], c->roi[view].width, mask_host, step,
                       c->roi[view].width, c->roi[view].height, hipMemcpyHostToDevice));
    c->masks_built = true;   // caller is responsible for setting every view
    c->blender_ready = false;
    return MS_OK;
}

// the mask init_gpu sees for a view: the caller's / seam finder's, or its re-warp through the CPW mesh (update_mask)
static const uint8_t *blend_mask_ptr(const ms_ctx *c, int v)
{
    return (c->use_eff[v] ? (const uint8_t *)c->masks_eff.p : (const uint8_t *)c->masks.p) + c->mask_off[v];
}

int ms_init_blender(ms_ctx *c, ms_stream stream)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    if (!c->maps_built || !c->masks_built) return fail(MS_ERR_STATE, "ms_init_blender: maps and masks must be built first");
    hipStream_t st = as_stream(stream);
    std::lock_guard<std::recursive_mutex> tables_lk(c->tables_mu);      // no ms_stitch enqueue while the tables are rebuilt ...
    if (c->stitch_pending) MS_HIP(hipEventSynchronize(c->last_stitch)); // ... and none still reading the old ones on the GPU
    const int nb = c->bg.num_bands, N = c->N, F = c->cfg.max_frames;

    // ---- layout of every per-view level -------------------------------------------------------
    c->h_views.assign(N, ViewDesc{});
    size_t w_total = 0;
    // (64 bytes of lead in front of each frame's view pyramids: fetch_row11 reads the 4 bytes before a row's first pixel unconditionally)
    long long g0_total = 64, gl_total = 64, stage_total = 0;
    c->max_pw = c->max_ph = c->max_aw = c->max_ah = 0;
    for (int v = 0; v < N; ++v) {
        ViewDesc &V = c->h_views[v];
        V.aw = c->roi[v].width; V.ah = c->roi[v].height;
        V.top = c->pad[v].top; V.left = c->pad[v].left;
        V.pw = V.aw + c->pad[v].left + c->pad[v].right;
        V.ph = V.ah + c->pad[v].top + c->pad[v].bottom;
        V.gain = (float)c->gain[v];
        V.xmap = (const float *)c->maps.p + c->map_off[v];
        V.ymap = V.xmap + (size_t)V.ah * c->map_pitch[v];
        V.map_pitch = c->map_pitch[v];
        V.coltab = (const float2 *)c->tabs.p + c->tab_off[v];
        V.rowtab = V.coltab + round_up(V.aw, 4);
        V.wp = c->wparams[v];
        V.proj = c->cfg.projection;
        V.s1_off = stage_total;
        V.s1_pitch = round_up(V.aw * 3, 4);
        stage_total += (long long)round_up(V.s1_pitch * V.ah + 8, 16);
        c->max_pw = std::max(c->max_pw, V.pw); c->max_ph = std::max(c->max_ph, V.ph);
        c->max_aw = std::max(c->max_aw, V.aw); c->max_ah = std::max(c->max_ah, V.ah);
        int w = V.pw, h = V.ph, xt = c->pad[v].x_tl, yt = c->pad[v].y_tl;
        for (int l = 0; l <= nb; ++l) {
            LevelDesc &L = V.lv[l];
            L.w = w; L.h = h; L.x_tl = xt; L.y_tl = yt;
            L.pitch = round_up(w, 16);          // byte planes at every level (the Gaussian levels of 8-bit pixels stay in [0, 255])
            L.wpitch = round_up(w, 4);
            c->w_off[v][l] = w_total;
            w_total += (size_t)h * L.wpitch;
            if (l == 0) { L.off = g0_total; g0_total += 3LL * h * L.pitch; }
            else        { L.off = gl_total; gl_total += 3LL * h * L.pitch; }
            w = (w + 1) / 2; h = (h + 1) / 2; xt /= 2; yt /= 2;
        }
    }
    for (int l = 0; l < nb; ++l) {
        bool ok = true;
        for (int v = 0; v < N; ++v) {
            const LevelDesc &L = c->h_views[v].lv[l];
            ok = ok && (L.w % 8 == 0) && (L.w >= 8) && (L.h >= 3);
        }
        c->down_vec[l] = ok;
    }
    {   // coarse levels that the tile kernel cannot take (width not a multiple of 8) go to one fused launch if a plane chain fits in LDS
        int t0 = 0;
        while (t0 < nb && c->down_vec[t0]) ++t0;
        auto plan_tail = [&](int l0, int sw, int *out_l0, int *out_lds, int *out_strips) {
            *out_l0 = -1; *out_lds = 0; *out_strips = 1;
            if (l0 < 1 || l0 >= nb) return;          // level l0 is int16 (l0 >= 1); the u8 level 0 never starts a tail
            int need = 0, strips = 1;
            for (int v = 0; v < N; ++v) {
                int w[MAX_LEVELS + 1], a[MAX_LEVELS + 1], b[MAX_LEVELS + 1];
                for (int l = l0; l <= nb; ++l) w[l] = c->h_views[v].lv[l].w;
                const int ns = div_up(w[nb], sw);
                strips = std::max(strips, ns);
                for (int sidx = 0; sidx < ns; ++sidx) {
                    tail_range(w, l0, nb, sidx, sw, a, b);
                    int bytes = 0;
                    for (int l = l0; l <= nb; ++l) bytes += ((b[l] - a[l]) * c->h_views[v].lv[l].h + 15) & ~15;
                    need = std::max(need, bytes);
                }
            }
            if (need <= 64 * 1024) { *out_l0 = l0; *out_lds = need; *out_strips = strips; }
        };
        plan_tail(t0, TAIL_STRIP, &c->tail_l0, &c->tail_lds, &c->tail_strips);
        int l0b = -1;
        plan_tail(t0, TAIL_STRIP_BATCH, &l0b, &c->tail_lds_b, &c->tail_strips_b);      // the batch form (wide strips); falls back to the narrow one where a wide strip does not fit LDS
        if (l0b != c->tail_l0) { c->tail_lds_b = c->tail_lds; c->tail_strips_b = c->tail_strips; c->tail_sw_b = TAIL_STRIP; } else c->tail_sw_b = TAIL_STRIP_BATCH;
    }
    for (int l = 0; l < nb; ++l) {
        int qw = c->bg.dst_roi.width >> l, qh = c->bg.dst_roi.height >> l;
        bool ok = (qw % 8 == 0) && (qh % 2 == 0);
        for (int v = 0; v < N; ++v) {
            const LevelDesc &L = c->h_views[v].lv[l];
            ok = ok && (L.w % 8 == 0) && (L.x_tl % 8 == 0) && (L.h % 2 == 0) && (L.y_tl % 2 == 0);
        }
        c->blend_vec[l] = ok;
    }
    c->w_total = w_total;
    if (int e = c->weights.alloc(w_total * sizeof(float))) return e;
    MS_HIP(hipMemsetAsync(c->weights.p, 0, w_total * sizeof(float), st));
    {
        size_t m_total = 0;
        for (int v = 0; v < N; ++v) {
            ViewDesc &V = c->h_views[v];
            V.wm0_pitch = round_up(V.pw, 16);
            c->wm0_off[v] = m_total;
            m_total += (size_t)V.wm0_pitch * V.ph;
        }
        c->wm0_total = m_total + 64;
        if (int e = c->wm0.alloc(m_total + 64)) return e;
        MS_HIP(hipMemsetAsync(c->wm0.p, 0, m_total + 64, st));
        for (int v = 0; v < N; ++v) {
            ViewDesc &V = c->h_views[v];
            V.wm0 = (const uint8_t *)c->wm0.p + c->wm0_off[v];
            MS_HIP(hipMemcpy2DAsync((uint8_t *)c->wm0.p + c->wm0_off[v] + (size_t)V.top * V.wm0_pitch + V.left, V.wm0_pitch,
                                    blend_mask_ptr(c, v), V.aw, V.aw, V.ah, hipMemcpyDeviceToDevice, st));
        }
    }
    for (int v = 0; v < N; ++v)
        for (int l = 0; l <= nb; ++l) c->h_views[v].lv[l].wgt = (const float *)c->weights.p + c->w_off[v][l];

    // ---- pano levels ----------------------------------------------------------------------------
    PanoDesc &P = c->pano;
    P = PanoDesc{};
    P.nb = nb; P.n_views = N;
    size_t den_total = 0;
    long long cl_total = 32, pacc_total = 0;       // (int16 elements: 64 bytes of lead, as for the view pyramids -- up_rows_load)
    {
        int w = c->bg.dst_roi.width, h = c->bg.dst_roi.height;
        for (int l = 0; l <= nb; ++l) {
            P.qw[l] = w; P.qh[l] = h; P.qpitch[l] = round_up(w, 8); P.dpitch[l] = round_up(w, 4);
            c->den_off[l] = den_total; den_total += (size_t)h * P.dpitch[l];
            P.coff[l] = cl_total;
            if (l >= 1) cl_total += 3LL * h * P.qpitch[l];
            P.poff[l] = pacc_total;
            pacc_total += 3LL * h * P.qpitch[l];
            w = (w + 1) / 2; h = (h + 1) / 2;
        }
    }
    P.fw = c->bg.dst_roi_final.width; P.fh = c->bg.dst_roi_final.height;
    P.mask_pitch = round_up(P.fw, 16);      // rows start on 16-byte boundaries: the level-0 band kernel reads 8 mask bytes per lane with one aligned load
    P.alpha = (float)(1. / 255.);
    P.canvas_x = c->canvas_x; P.canvas_y = c->canvas_y; P.out_w = c->cfg.out_width; P.out_h = c->cfg.out_height;
    P.i_y0 = std::max(0, c->canvas_y & ~1); P.i_rows = std::max(0, std::min(c->cfg.out_height & ~1, (c->canvas_y + P.fh + 1) & ~1) - P.i_y0);
    c->den_total = den_total;
    if (int e = c->den.alloc(den_total * sizeof(float))) return e;
    if (int e = c->result_mask.alloc((size_t)P.mask_pitch * P.fh)) return e;
    MS_HIP(hipMemsetAsync(c->den.p, 0, den_total * sizeof(float), st));
    for (int l = 0; l <= nb; ++l) P.den[l] = (const float *)c->den.p + c->den_off[l];
    P.mask = (const uint8_t *)c->result_mask.p;
    {   // fused coarse band chain: from the first band the vectorised kernel cannot take up to the coarsest one
        int t = 0;
        while (t < nb && c->blend_vec[t]) ++t;
        auto plan_btail = [&](int tt, int bw, int *out_t, int *out_lds, int *out_strips) {
            *out_t = -1; *out_lds = 0; *out_strips = 0;
            bool ok = tt >= 1 && tt < nb && c->cfg.debug_simple_kernels == 0;
            for (int l = tt; ok && l < nb; ++l) {          // quads: even band sizes and even-aligned view rects below the coarsest band
                ok = P.qw[l] % 2 == 0 && P.qh[l] % 2 == 0;
                for (int v = 0; ok && v < N; ++v) {
                    const LevelDesc &L = c->h_views[v].lv[l];
                    ok = L.w % 2 == 0 && L.h % 2 == 0 && L.x_tl % 2 == 0 && L.y_tl % 2 == 0;
                }
            }
            if (!ok) return;
            const int strips = div_up(P.qw[tt], bw);
            size_t need = 0;
            for (int sidx = 0; sidx < strips; ++sidx) {
                int a[MAX_LEVELS + 1], b[MAX_LEVELS + 1];
                btail_range(P.qw, tt, nb, sidx, bw, a, b);
                size_t bytes = 0;
                for (int l = tt; l <= nb; ++l) bytes += ((size_t)(b[l] - a[l]) * P.qh[l] + 7) / 8 * 8 * sizeof(int16_t);      // every band of the strip, band tt included (its D term waits in LDS for the collapse)
                need = std::max(need, bytes);
            }
            if (need <= 60 * 1024) { *out_t = tt; *out_lds = (int)need + 64; *out_strips = strips; }
        };
        plan_btail(t, BTAIL_W, &c->btail_t, &c->btail_lds, &c->btail_strips);
        plan_btail(t - 1, BTAIL_W, &c->btail2_t, &c->btail2_lds, &c->btail2_strips);
    }

    // ---- init_gpu per view, in view order: weight = mask/255 -> constant border -> nb x pyrDown,
    //      and the weight sums (the `dst_w += w` of addSrcWeightKernel32F, blenders.cpp:729-746)
    DevBuf wm;   // scratch weight_map of the largest view
    if (int e = wm.alloc((size_t)c->max_aw * c->max_ah * sizeof(float))) return e;
    for (int v = 0; v < N; ++v) {
        ViewDesc &V = c->h_views[v];
        ms_image mask{(void *)blend_mask_ptr(c, v), (size_t)V.aw, V.aw, V.ah, MS_8UC1};
        ms_image wmap{wm.p, (size_t)V.aw * sizeof(float), V.aw, V.ah, MS_32FC1};
        if (c->feather_sharpness >= 0.f) {            // FeatherBlender::feed -> createWeightMap (blenders.cpp:156, 944-951), host side like the reference
            std::vector<uint8_t> hm((size_t)V.aw * V.ah);
            std::vector<float> hw((size_t)V.aw * V.ah);
            MS_HIP(hipMemcpyAsync(hm.data(), mask.data, hm.size(), hipMemcpyDeviceToHost, st));
            MS_HIP(hipStreamSynchronize(st));
            feather_weight_map(hm.data(), V.ah, V.aw, c->feather_sharpness, hw.data());
            MS_HIP(hipMemcpyAsync(wm.p, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice, st));
            MS_HIP(hipStreamSynchronize(st));
        } else if (int e = launch_convert(mask, wmap, 1. / 255., st)) return e;                // blenders.cpp:412
        auto level_img = [&](int l) {
            const LevelDesc &L = V.lv[l];
            return ms_image{(void *)L.wgt, (size_t)L.wpitch * sizeof(float), L.w, L.h, MS_32FC1};
        };
        ms_image l0 = level_img(0);
        if (int e = launch_copy_make_border(wmap, l0, V.top, V.left, MS_BORDER_CONSTANT, st)) return e;   // :420
        for (int l = 0; l < nb; ++l) {
            ms_image a = level_img(l), b = level_img(l + 1);
            if (int e = launch_pyr_down(a, b, st)) return e;                                  // :422-423
        }
        for (int l = 0; l <= nb; ++l) {
            const LevelDesc &L = V.lv[l];
            float *d = (float *)c->den.p + c->den_off[l] + (size_t)L.y_tl * P.dpitch[l] + L.x_tl;
            k_acc_weight<<<dim3(div_up(L.w, 64), div_up(L.h, 4)), dim3(64, 4), 0, st>>>(L.wgt, L.wpitch, L.h, L.w, d, P.dpitch[l]);
            MS_LAUNCH_CHECK();
        }
    }
    for (int l = 0; l <= nb; ++l) {
        k_finish_den<<<dim3(div_up(P.qw[l], 64), div_up(P.qh[l], 4)), dim3(64, 4), 0, st>>>(
            (float *)c->den.p + c->den_off[l], P.dpitch[l], P.qh[l], P.qw[l],
            l == 0 ? (uint8_t *)c->result_mask.p : nullptr, P.mask_pitch, P.fh, P.fw);
        MS_LAUNCH_CHECK();
    }
    MS_HIP(hipStreamSynchronize(st));
    wm.release();

    // ---- per-batch buffers ----------------------------------------------------------------------
    c->g0_stride = (g0_total + 255) / 256 * 256;
    c->gl_stride = (gl_total + 127) / 128 * 128;
    c->cl_stride = (cl_total + 127) / 128 * 128;
    c->pacc_stride = (pacc_total + 127) / 128 * 128;
    c->stage_stride = (stage_total + 255) / 256 * 256;
    if (int e = c->g0.alloc((size_t)c->g0_stride * F + 64)) return e;
    if (int e = c->gl.alloc((size_t)c->gl_stride * F + 64)) return e;
    if (int e = c->cl.alloc((size_t)c->cl_stride * F * sizeof(int16_t) + 64)) return e;
    // invariant the packed band arithmetic relies on: every value in the view pyramids is in [0,255], written or not
    MS_HIP(hipMemsetAsync(c->g0.p, 0, (size_t)c->g0_stride * F + 64, st));
    MS_HIP(hipMemsetAsync(c->gl.p, 0, (size_t)c->gl_stride * F + 64, st));
    MS_HIP(hipMemsetAsync(c->cl.p, 0, (size_t)c->cl_stride * F * sizeof(int16_t) + 64, st));
    if (c->cfg.enable_cpw) {
        if (int e = c->stage.alloc((size_t)c->stage_stride * F)) return e;
        size_t mtotal = 0;
        for (int v = 0; v < N; ++v) { c->mesh_off[v] = mtotal; mtotal += (size_t)2 * c->h_views[v].ah * c->map_pitch[v]; }
        if (c->mesh[0].bytes != mtotal * sizeof(float) || !c->mesh[0].p) {      // (a re-initialisation with the same geometry -- ms_update_mask -- keeps the meshes)
            for (int b = 0; b < 2; ++b) if (int e = c->mesh[b].alloc(mtotal * sizeof(float))) return e;
            for (int v = 0; v < N; ++v) { c->mesh_active[v] = 0; c->mesh_set[v] = false; }
        }
    }
    if (int e = c->view_tab.alloc(sizeof(ViewDesc) * N)) return e;
    MS_HIP(hipMemcpy(c->view_tab.p, c->h_views.data(), sizeof(ViewDesc) * N, hipMemcpyHostToDevice));
    c->col_begin = c->col_end = 0;
    if (c->cfg.col_shards > 1) {      // shard k composites the pano-ROI columns [bound(k), bound(k+1)); boundaries on multiples of 16 columns
        const int S = c->cfg.col_shards, k = c->cfg.col_shard_index, fw = c->pano.fw;
        auto bound = [&](int i) { return i <= 0 ? 0 : (i >= S ? fw : (int)((long long)i * fw / S) / 16 * 16); };
        c->col_begin = bound(k); c->col_end = bound(k + 1);
        if (c->col_end <= c->col_begin) return fail(MS_ERR_INVALID, "ms_init_blender: a %d-column panorama is too narrow for %d column shards", fw, S);
    }
    if (int e = build_plan(c)) return e;
    {   // a (re-)initialisation makes the primary tables current; the alternate copy is set up (not filled) for the enqueue-only ms_update_mask
        std::lock_guard<std::mutex> mk(c->mesh_mu);
        c->tab_active = 0; c->tab_wait = false;
    }
    if (c->cfg.update_mask_margin > 0) {
        ms_ctx::AltTables &A = c->alt;
        if (int e = A.weights.alloc(c->w_total * sizeof(float))) return e;
        if (int e = A.wm0.alloc(c->wm0_total)) return e;
        if (int e = A.den.alloc(c->den_total * sizeof(float))) return e;
        if (int e = A.result_mask.alloc((size_t)c->pano.mask_pitch * c->pano.fh)) return e;
        if (int e = A.pure_maps.alloc(std::max<size_t>(1, c->pure_total))) return e;
        if (int e = A.view_tab.alloc(sizeof(ViewDesc) * N)) return e;
        A.h_views = c->h_views;
        for (int v = 0; v < N; ++v) {
            A.h_views[v].wm0 = (const uint8_t *)A.wm0.p + c->wm0_off[v];
            for (int l = 0; l <= nb; ++l) A.h_views[v].lv[l].wgt = (const float *)A.weights.p + c->w_off[v][l];
        }
        MS_HIP(hipMemcpy(A.view_tab.p, A.h_views.data(), sizeof(ViewDesc) * N, hipMemcpyHostToDevice));
        A.pano = c->pano;
        for (int l = 0; l <= nb; ++l) A.pano.den[l] = (const float *)A.den.p + c->den_off[l];
        A.pano.mask = (const uint8_t *)A.result_mask.p;
        for (int l = 0; l < nb; ++l) A.pano.pure[l] = c->pure_off[l] ? (const uint8_t *)A.pure_maps.p + (c->pure_off[l] - 1) : nullptr;
        if (int e = c->mask_tmp.alloc((size_t)c->max_aw * c->max_ah)) return e;
        if (int e = c->wm_scratch.alloc((size_t)c->max_aw * c->max_ah * sizeof(float))) return e;
        if (c->masks_eff.bytes != c->masks.bytes || !c->masks_eff.p) {
            if (int e = c->masks_eff.alloc(c->masks.bytes)) return e;
            for (int v = 0; v < N; ++v) c->use_eff[v] = false;
        }
        for (int v = 0; v < N; ++v)      // the effective mask of a view that has no re-warped one is the mask itself
            if (!c->use_eff[v])
                MS_HIP(hipMemcpyAsync((uint8_t *)c->masks_eff.p + c->mask_off[v], (const uint8_t *)c->masks.p + c->mask_off[v], (size_t)c->roi[v].width * c->roi[v].height, hipMemcpyDeviceToDevice, st));
        if (!c->tab_ready) MS_HIP(hipEventCreateWithFlags(&c->tab_ready, hipEventDisableTiming));
        if (!c->disp_dev.p) {
            if (int e = c->disp_dev.alloc(2 * MAX_VIEWS * sizeof(unsigned))) return e;
            MS_HIP(hipMemsetAsync(c->disp_dev.p, 0xff, 2 * MAX_VIEWS * sizeof(unsigned), st));      // "unbounded" until measured
        }
    }
    if (c->cfg.self_check != 0) {
        // the band kernels' shared-reciprocal division against the compiler's IEEE a / d, over every distinct denominator these tables hold
        // and all int16 numerators (common.hpp, DivBy): the bit-exactness claim rests on this check, not on an argument about the sequence
        std::vector<float> hd(den_total);
        MS_HIP(hipMemcpy(hd.data(), c->den.p, den_total * sizeof(float), hipMemcpyDeviceToHost));
        std::sort(hd.begin(), hd.end());
        hd.erase(std::unique(hd.begin(), hd.end()), hd.end());
        hd.erase(hd.begin(), std::upper_bound(hd.begin(), hd.end(), 0.f));        // (row padding of the tables: never read)
        for (size_t i = 0; i < hd.size(); i += 65535) {
            const int n = (int)std::min<size_t>(65535, hd.size() - i);
            const int bad = ms_selftest_divide(hd.data() + i, n, stream);
            if (bad < 0) return bad;
            if (bad > 0) return fail(MS_ERR_INVALID, "ms_init_blender: shared-reciprocal division differs from IEEE division for %d (numerator, denominator) pairs", bad);
        }
    }
    // the tables are complete when this returns, whichever stream the next ms_stitch runs on (calibration-time call: a host wait costs nothing here,
    // and a stitch enqueued on another stream right after a rebuild on the recalibration thread must not read half-built tables)
    MS_HIP(hipStreamSynchronize(st));
    c->blender_ready = true;
    return MS_OK;
}

int ms_init_feather(ms_ctx *c, float sharpness, ms_stream stream)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    if (!c->maps_built) return fail(MS_ERR_STATE, "ms_init_feather: maps and masks must be built first");
    MS_CHECK(c->bg.num_bands == 0, "ms_init_feather: create the context with num_bands = 0 (FeatherBlender has a single band)");
    MS_CHECK(sharpness > 0.f, "ms_init_feather: sharpness must be positive");
    c->feather_sharpness = sharpness;
    return ms_init_blender(c, stream);
}

// max over the view of |x_mesh - x|, |y_mesh - y| (NaN = hole of convertMeshesToMap: samples nothing, ignored); non-negative floats
// order like their bit patterns, so one atomicMax on the bits reduces the launch
__global__ void __launch_bounds__(256) k_mesh_disp(const float *__restrict__ mx, const float *__restrict__ my, int pitch, int ah, int aw, unsigned *out)
{
    float d = 0.f;
    for (int y = blockIdx.x; y < ah; y += gridDim.x)            // a block walks rows, grid-stride: few atomics on the one result word
        for (int x = threadIdx.x; x < aw; x += 256) {
            const float a = fabsf(mx[(size_t)y * pitch + x] - (float)x), b = fabsf(my[(size_t)y * pitch + x] - (float)y);
            d = fmaxf(d, fmaxf(a == a ? a : 0.f, b == b ? b : 0.f));
        }
    for (int o = 32; o > 0; o >>= 1) d = fmaxf(d, __shfl_xor(d, o));
    if ((threadIdx.x & 63) == 0 && d > 0.f) atomicMax(out, __float_as_uint(d));
}
static int measure_mesh_disp(ms_ctx *c, int view, int tgt, hipStream_t st)
{
    if (!c->disp_dev.p) {
        if (int e = c->disp_dev.alloc(2 * MAX_VIEWS * sizeof(unsigned))) return e;
        MS_HIP(hipMemsetAsync(c->disp_dev.p, 0xff, 2 * MAX_VIEWS * sizeof(unsigned), st));      // "unbounded" until measured
    }
    const int aw = c->roi[view].width, ah = c->roi[view].height;
    const float *base = (const float *)c->mesh[tgt].p + c->mesh_off[view];
    unsigned *word = (unsigned *)c->disp_dev.p + 2 * view + tgt;
    MS_HIP(hipMemsetAsync(word, 0, sizeof(unsigned), st));
    k_mesh_disp<<<std::min(ah, 128), 256, 0, st>>>(base, base + (size_t)ah * c->map_pitch[view], c->map_pitch[view], ah, aw, word);
    MS_LAUNCH_CHECK();
    return MS_OK;          // (no host synchronisation: the stage-1 kernel reads the word on the device, ms_get_mesh_displacement waits on mesh_ready)
}

// ---- CPW mesh maps ------------------------------------------------------------------------------
static ms_image mesh_image(const ms_ctx *c, int buf, int v, int which)
{
    const int ah = c->roi[v].height, aw = c->roi[v].width;
    float *base = (float *)c->mesh[buf].p + c->mesh_off[v] + (which ? (size_t)ah * c->map_pitch[v] : 0);
    return ms_image{base, (size_t)c->map_pitch[v] * sizeof(float), aw, ah, MS_32FC1};
}

// Callers hold mesh_update_mu (one update at a time).  mesh_mu is taken here only around the bookkeeping ms_stitch shares: ms_stitch holds it from
// picking the active buffers up to recording last_stitch, so `last_stitch` seen here covers every stitch that may still read the inactive buffer.
static hipEvent_t mesh_ready_event(ms_ctx *c, int view) { return c->mesh_ready_via_chain[view] ? c->mesh_chain : c->mesh_ready[view]; }      // (caller holds mesh_mu or is the only updater)
static int mesh_begin_update(ms_ctx *c, int view, int *target, hipStream_t st, bool enqueue_waits = true)
{
    if (!c->blender_ready || !c->cfg.enable_cpw) return fail(MS_ERR_STATE, "mesh update needs enable_cpw and ms_init_blender");
    std::lock_guard<std::mutex> lk(c->mesh_mu);
    *target = c->mesh_set[view] ? 1 - c->mesh_active[view] : c->mesh_active[view];
    if (!c->mesh_ready[view]) MS_HIP(hipEventCreateWithFlags(&c->mesh_ready[view], hipEventDisableTiming));
    if (!c->mesh_chain) MS_HIP(hipEventCreateWithFlags(&c->mesh_chain, hipEventDisableTiming));
    // the inactive buffer may still be read by a stitch enqueued before the previous swap: the update waits for it ON THE GPU
    // (the reference releases its mutex before the async remap finishes, timed.cpp:98-103); no host synchronisation anywhere
    if (enqueue_waits) {      // (ms_set_meshes: once for all views of the call)
        if (c->stitch_pending) MS_HIP(hipStreamWaitEvent(st, c->last_stitch, 0));
        if (c->mesh_chain_set) MS_HIP(hipStreamWaitEvent(st, c->mesh_chain, 0));
    }
    return MS_OK;
}
// one view's update: its own event + the chain, both recorded BEFORE the view is published (ms_set_meshes publishes all its views itself, behind one record of the chain)
static int mesh_end_update(ms_ctx *c, int view, int tgt, hipStream_t st, bool measure = true)
{
    if (measure) if (int e = measure_mesh_disp(c, view, tgt, st)) return e;
    std::lock_guard<std::mutex> lk(c->mesh_mu);
    MS_HIP(hipEventRecord(c->mesh_ready[view], st));
    MS_HIP(hipEventRecord(c->mesh_chain, st));
    c->mesh_ready_via_chain[view] = false;
    c->mesh_chain_set = true;
    c->mesh_wait[view] = true;
    c->mesh_active[view] = tgt;
    c->mesh_set[view] = true;
    return MS_OK;
}

int ms_set_mesh_maps(ms_ctx *c, int view, const ms_image *xm, const ms_image *ym, ms_stream stream)
{
    if (int e = ctx_check_view(c, view)) return e;
    MS_CHECK(xm && ym && xm->data && ym->data, "ms_set_mesh_maps: null image");
    std::lock_guard<std::mutex> lk(c->mesh_update_mu);
    int tgt;
    hipStream_t st = as_stream(stream);
    const int aw = c->roi[view].width, ah = c->roi[view].height;
    MS_CHECK(xm->rows == ah && xm->cols == aw && ym->rows == ah && ym->cols == aw && xm->type == MS_32FC1 && ym->type == MS_32FC1,
             "ms_set_mesh_maps: maps must be 32FC1 %dx%d", aw, ah);
    if (int e = mesh_begin_update(c, view, &tgt, st)) return e;
    ms_image dx = mesh_image(c, tgt, view, 0), dy = mesh_image(c, tgt, view, 1);
    MS_HIP(hipMemcpy2DAsync(dx.data, dx.step, xm->data, xm->step, (size_t)aw * 4, ah, hipMemcpyDeviceToDevice, st));
    MS_HIP(hipMemcpy2DAsync(dy.data, dy.step, ym->data, ym->step, (size_t)aw * 4, ah, hipMemcpyDeviceToDevice, st));
    return mesh_end_update(c, view, tgt, st);
}

int ms_set_mesh(ms_ctx *c, int view, const float *mesh_x, const float *mesh_y, int N, int M, ms_stream stream)
{
    if (int e = ctx_check_view(c, view)) return e;
    MS_CHECK(mesh_x && mesh_y && N >= 2 && M >= 2, "ms_set_mesh: need an N x M (>= 2x2) vertex mesh");
    std::lock_guard<std::mutex> lk(c->mesh_update_mu);       // (ms_stitch never takes this one: the waits and allocations below cannot stall it)
    int tgt;
    hipStream_t st = as_stream(stream);
    const int aw = c->roi[view].width, ah = c->roi[view].height, hw = aw / 2, hh = ah / 2;
    MS_CHECK(hw >= 2 && hh >= 2, "ms_set_mesh: view too small");
    MS_CHECK((long long)aw * ah < (1ll << 24) && aw < 65536 && ah < 65536, "ms_set_mesh: view %dx%d exceeds the scatter accumulators ([count:24 | sum:40])", aw, ah);
    // scratch: vertex mesh x|y, then two half-resolution accumulators used in turn (the launch that fills one clears the other)
    const size_t n_small = (size_t)N * M, n_half = (size_t)hw * hh;
    size_t half_cap = 0;
    for (int v = 0; v < c->N; ++v) half_cap = std::max(half_cap, (size_t)(c->roi[v].width / 2) * (c->roi[v].height / 2));
    if (!c->mesh_tmp.p || c->mesh_small_cap < n_small || c->mesh_half_cap < half_cap || c->mesh_stage_floats < 2 * n_small) {   // first call or a larger mesh: drain earlier updates
        if (c->mesh_chain_set) MS_HIP(hipEventSynchronize(c->mesh_chain));
        const size_t bytes = 2 * n_small * sizeof(float) + 4 * half_cap * sizeof(unsigned long long) + 16;
        if (int e = c->mesh_tmp.alloc(bytes)) return e;
        MS_HIP(hipMemsetAsync(c->mesh_tmp.p, 0, bytes, st));      // on the update's own stream: a plain hipMemset runs on the NULL stream, asynchronously to the host, and non-blocking streams do not wait for it
        c->mesh_small_cap = n_small; c->mesh_half_cap = half_cap; c->mesh_dirty = 0; c->mesh_parity = 0;
        if (c->mesh_stage_floats < 2 * n_small) {
            if (c->mesh_stage) (void)hipHostFree(c->mesh_stage);
            c->mesh_stage = nullptr; c->mesh_stage_floats = 0;
            MS_HIP(hipHostMalloc((void **)&c->mesh_stage, (ms_ctx::MESH_STAGE_GENS + 1) * (2 * n_small * sizeof(float) * MAX_VIEWS), hipHostMallocDefault));      // (the ring + ms_set_mesh's own generation behind it)
            for (bool &b_ : c->mesh_stage_ev_set) b_ = false;
            c->mesh_stage_gen = 0;
            c->mesh_stage_floats = 2 * n_small;
        }
    }
    if (!c->disp_dev.p) {
        if (int e = c->disp_dev.alloc(2 * MAX_VIEWS * sizeof(unsigned))) return e;
        MS_HIP(hipMemsetAsync(c->disp_dev.p, 0xff, 2 * MAX_VIEWS * sizeof(unsigned), st));      // "unbounded" until measured
    }
    if (int e = mesh_begin_update(c, view, &tgt, st)) return e;
    float *sm_x = (float *)c->mesh_tmp.p, *sm_y = sm_x + n_small;
    unsigned long long *acc0 = (unsigned long long *)(((uintptr_t)(sm_x + 2 * c->mesh_small_cap) + 7) & ~(uintptr_t)7);
    unsigned long long *acc = acc0 + (size_t)c->mesh_parity * 2 * c->mesh_half_cap, *other = acc0 + (size_t)(1 - c->mesh_parity) * 2 * c->mesh_half_cap;
    unsigned long long *ax = acc, *ay = acc + n_half;
    // the caller's arrays may be freed right after return: stage them in pinned memory (one slot per view; a slot is reused only by
    // the next update of the same view, whose copy of the previous one finished long before -- checked on its event)
    bool slot_busy;
    hipEvent_t busy_ev = nullptr;
    { std::lock_guard<std::mutex> mk(c->mesh_mu); slot_busy = c->mesh_ready[view] && (c->mesh_wait[view] || c->mesh_set[view]); if (slot_busy) busy_ev = mesh_ready_event(c, view); }
    if (slot_busy && busy_ev) MS_HIP(hipEventSynchronize(busy_ev));
    // (generation MESH_STAGE_GENS, outside the ring ms_set_meshes walks: a batched call can never overwrite a slot whose single-view copy has not run yet -- ADVICE r05)
    float *stg = c->mesh_stage + ((size_t)ms_ctx::MESH_STAGE_GENS * MAX_VIEWS + view) * c->mesh_stage_floats;
    memcpy(stg, mesh_x, n_small * 4);
    memcpy(stg + n_small, mesh_y, n_small * 4);
    MS_HIP(hipMemcpyAsync(sm_x, stg, 2 * n_small * 4, hipMemcpyHostToDevice, st));
    unsigned *word = (unsigned *)c->disp_dev.p + 2 * view + tgt;
    const dim3 grid(div_up(aw, 64), div_up(ah, 4)), blk(64, 4);
    k_mesh_expand_scatter<<<dim3(div_up(aw, 64), div_up(ah, 4 * MESH_SR)), blk, 0, st>>>(sm_x, sm_y, N, M, aw, ah, ax, ay, hw, hh, other, c->mesh_dirty, word);        // meshwarper.cpp:838-869
    MS_LAUNCH_CHECK();
    ms_image dx = mesh_image(c, tgt, view, 0), dy = mesh_image(c, tgt, view, 1);
    const int tiles_x = div_up(aw, 64), n_tiles = tiles_x * div_up(ah, 4 * MESH_TR);
    k_mesh_mean_resize<<<std::min(n_tiles, 1024), blk, 0, st>>>(ax, ay, hw, hh, (float *)dx.data, (float *)dy.data, c->map_pitch[view], aw, ah, tiles_x, n_tiles, word);   // :870-883
    MS_LAUNCH_CHECK();
    c->mesh_dirty = 2 * n_half;
    c->mesh_parity ^= 1;
    return mesh_end_update(c, view, tgt, st, false);
}

// MeshWarper::convertMeshesToMap as the reference calls it -- for ALL views at once (meshwarper.cpp:823-886 loops over the images): the same two kernels
// with blockIdx.z / .y = view, i.e. two launches per recalibration instead of two per view.  The per-view launches are latency-bound (47 us each for a
// 960 x 627 view): six views cost 0.3 ms of GPU time one after the other, about a third of that together.  Same arithmetic, bit-identical maps.
// mesh_x / mesh_y: HOST, num_views meshes of N x M back to back.
int ms_set_meshes(ms_ctx *c, const float *mesh_x, const float *mesh_y, int N, int M, ms_stream stream)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    MS_CHECK(mesh_x && mesh_y && N >= 2 && M >= 2, "ms_set_meshes: need N x M (>= 2x2) vertex meshes");
    if (!c->blender_ready || !c->cfg.enable_cpw) return fail(MS_ERR_STATE, "ms_set_meshes needs enable_cpw and ms_init_blender");
    std::lock_guard<std::mutex> lk(c->mesh_update_mu);
    hipStream_t st = as_stream(stream);
    const int NV = c->N;
    const size_t n_small = (size_t)N * M;
    std::vector<size_t> acc_off(NV + 1, 0);                  // 64-bit words: per view [ax | ay] x two parities
    int max_aw = 0, max_ah = 0, max_tiles = 0;
    for (int v = 0; v < NV; ++v) {
        const int aw = c->roi[v].width, ah = c->roi[v].height, hw = aw / 2, hh = ah / 2;
        MS_CHECK(hw >= 2 && hh >= 2, "ms_set_meshes: view %d too small", v);
        MS_CHECK((long long)aw * ah < (1ll << 24) && aw < 65536 && ah < 65536, "ms_set_meshes: view %dx%d exceeds the scatter accumulators ([count:24 | sum:40])", aw, ah);
        acc_off[v + 1] = acc_off[v] + 4 * (size_t)hw * hh;
        max_aw = std::max(max_aw, aw); max_ah = std::max(max_ah, ah);
        max_tiles = std::max(max_tiles, div_up(aw, 64) * div_up(ah, 4 * MESH_TR));
    }
    const size_t sm_floats = 2 * n_small * NV, sm_bytes = (sm_floats * sizeof(float) + 15) & ~(size_t)15;
    if (!c->mesh_all.p || c->mesh_all_small != n_small || c->mesh_stage_floats < 2 * n_small) {      // first call or another mesh size: drain earlier updates, (re)allocate
        if (c->mesh_chain_set) MS_HIP(hipEventSynchronize(c->mesh_chain));
        const size_t bytes = sm_bytes + acc_off[NV] * sizeof(unsigned long long);
        if (int e = c->mesh_all.alloc(bytes)) return e;
        MS_HIP(hipMemsetAsync(c->mesh_all.p, 0, bytes, st));      // (stream-ordered: see ms_set_mesh)
        c->mesh_all_small = n_small; c->mesh_all_parity = 0; c->mesh_all_dirty = false;
        if (c->mesh_stage_floats < 2 * n_small) {
            if (c->mesh_stage) (void)hipHostFree(c->mesh_stage);
            c->mesh_stage = nullptr; c->mesh_stage_floats = 0;
            MS_HIP(hipHostMalloc((void **)&c->mesh_stage, (ms_ctx::MESH_STAGE_GENS + 1) * (2 * n_small * sizeof(float) * MAX_VIEWS), hipHostMallocDefault));      // (the ring + ms_set_mesh's own generation behind it)
            for (bool &b_ : c->mesh_stage_ev_set) b_ = false;
            c->mesh_stage_gen = 0;
            c->mesh_stage_floats = 2 * n_small;
        }
    }
    if (!c->disp_dev.p) {
        if (int e = c->disp_dev.alloc(2 * MAX_VIEWS * sizeof(unsigned))) return e;
        MS_HIP(hipMemsetAsync(c->disp_dev.p, 0xff, 2 * MAX_VIEWS * sizeof(unsigned), st));
    }
    int tgt[MAX_VIEWS];
    for (int v = 0; v < NV; ++v) if (int e = mesh_begin_update(c, v, &tgt[v], st, v == 0)) return e;
    // stage the caller's arrays in pinned memory (the view's slot is free once its previous update's copy has run: checked on its event), one copy for all views
    float *sm = (float *)c->mesh_all.p;
    unsigned long long *acc0 = (unsigned long long *)((char *)c->mesh_all.p + sm_bytes);
    MeshJobs T{};
    const int p = c->mesh_all_parity;
    // the staging generation this call fills: free once the copy of the call a ring ago has run (its own event, recorded right behind that copy).  Until round 5 the host waited
    // here for the PREVIOUS update, twelve events were recorded and as many waits enqueued per call: a recalibration every 60 frames cost config 3 10.5 % of its frame rate with
    // 4 % of GPU work in it; now 4.5 % (same-box A/B 22.8 k -> 24.3 k frames/s, profiles/r05_experiments.txt)
    const int gen = c->mesh_stage_gen;
    if (c->mesh_stage_ev_set[gen]) MS_HIP(hipEventSynchronize(c->mesh_stage_ev[gen]));      // (blocks only when the caller is a whole ring of updates ahead of the GPU)
    float *stage_gen = c->mesh_stage + (size_t)gen * MAX_VIEWS * c->mesh_stage_floats;
    for (int v = 0; v < NV; ++v) {
        float *stg = stage_gen + (size_t)v * c->mesh_stage_floats;
        memcpy(stg, mesh_x + (size_t)v * n_small, n_small * 4);
        memcpy(stg + n_small, mesh_y + (size_t)v * n_small, n_small * 4);
        const int aw = c->roi[v].width, ah = c->roi[v].height, hw = aw / 2, hh = ah / 2;
        const size_t n_half = (size_t)hw * hh;
        unsigned long long *mine = acc0 + acc_off[v] + (size_t)p * 2 * n_half, *other = acc0 + acc_off[v] + (size_t)(1 - p) * 2 * n_half;
        ms_image dx = mesh_image(c, tgt[v], v, 0), dy = mesh_image(c, tgt[v], v, 1);
        MeshJob &J = T.j[v];
        J.smx = sm + (size_t)v * 2 * n_small; J.smy = J.smx + n_small;
        J.ax = mine; J.ay = mine + n_half;
        J.clear = other; J.n_clear = c->mesh_all_dirty ? 2 * n_half : 0;
        J.disp_word = (unsigned *)c->disp_dev.p + 2 * v + tgt[v];
        J.dx = (float *)dx.data; J.dy = (float *)dy.data;
        J.aw = aw; J.ah = ah; J.hw = hw; J.hh = hh; J.pitch = c->map_pitch[v];
        J.tiles_x = div_up(aw, 64); J.n_tiles = J.tiles_x * div_up(ah, 4 * MESH_TR);
    }
    if (c->mesh_stage_floats == 2 * n_small)
        MS_HIP(hipMemcpyAsync(sm, stage_gen, sm_floats * sizeof(float), hipMemcpyHostToDevice, st));        // the slots are back to back
    else
        for (int v = 0; v < NV; ++v)
            MS_HIP(hipMemcpyAsync(sm + (size_t)v * 2 * n_small, stage_gen + (size_t)v * c->mesh_stage_floats, 2 * n_small * sizeof(float), hipMemcpyHostToDevice, st));
    if (!c->mesh_stage_ev[gen]) MS_HIP(hipEventCreateWithFlags(&c->mesh_stage_ev[gen], hipEventDisableTiming));
    MS_HIP(hipEventRecord(c->mesh_stage_ev[gen], st));
    c->mesh_stage_ev_set[gen] = true;
    c->mesh_stage_gen = (gen + 1) % ms_ctx::MESH_STAGE_GENS;
    const dim3 blk(64, 4);
    k_mesh_expand_scatter_all<<<dim3(div_up(max_aw, 64), div_up(max_ah, 4 * MESH_SR), NV), blk, 0, st>>>(T, N, M);         // meshwarper.cpp:838-869, every view
    MS_LAUNCH_CHECK();
    k_mesh_mean_resize_all<<<dim3(std::min(max_tiles, 512), NV), blk, 0, st>>>(T);                                           // :870-883
    MS_LAUNCH_CHECK();
    c->mesh_all_dirty = true;
    c->mesh_all_parity ^= 1;
    // Publish every view in ONE mesh_mu critical section, the chain's record FIRST: a stitch on another thread that takes mesh_mu between two views of a
    // per-view publication saw mesh_wait[v] set while mesh_chain still held the PREVIOUS update's record (or none), waited for that, cleared the flag and read the
    // buffer k_mesh_mean_resize_all was still writing.  A stitch now sees either none of this call's views or all of them, behind this call's record.
    {
        std::lock_guard<std::mutex> mk(c->mesh_mu);
        MS_HIP(hipEventRecord(c->mesh_chain, st));
        c->mesh_chain_set = true;
        for (int v = 0; v < NV; ++v) {
            c->mesh_ready_via_chain[v] = true;
            c->mesh_wait[v] = true;
            c->mesh_active[v] = tgt[v];
            c->mesh_set[v] = true;
        }
    }
    return MS_OK;
}

// MeshWarper::interpolateMesh (meshwarper.cpp:337-354) + convertMeshesToMap: the RECALIB_INTERP branch of the recalibration thread
// (timed.cpp:449-457) re-expands start + (end - start) * progress every 30 ms
int ms_set_mesh_interp(ms_ctx *c, int view, const float *x0, const float *y0, const float *x1, const float *y1, int N, int M, float progress, ms_stream stream)
{
    MS_CHECK(x0 && y0 && x1 && y1 && N >= 2 && M >= 2, "ms_set_mesh_interp: need two N x M (>= 2x2) vertex meshes");
    std::vector<float> mx((size_t)N * M), my((size_t)N * M);
    for (size_t i = 0; i < mx.size(); ++i) {
        mx[i] = x0[i] + (x1[i] - x0[i]) * progress;        // fp32, in this order (-ffp-contract=off)
        my[i] = y0[i] + (y1[i] - y0[i]) * progress;
    }
    return ms_set_mesh(c, view, mx.data(), my.data(), N, M, stream);
}

// Enqueue-only update_mask (cfg.update_mask_margin > 0; caller holds mesh_update_mu).  Everything that depends on the masks exists twice; the copy
// ms_stitch does not read is rebuilt on `st` -- behind the stitches enqueued so far, which may still read it from before the previous swap -- and becomes
// the active one under mesh_mu; ms_stitch makes its stream wait for `tab_ready`.  No host synchronisation, no allocation, no new work lists: those
// were planned for any mask within the margin (build_plan), and a mesh that displaces further leaves the effective mask unchanged (k_mask_select).
static int update_mask_async(ms_ctx *c, int view, hipStream_t st)
{
    const int N = c->N, nb = c->pano.nb;
    ms_ctx::AltTables &A = c->alt;
    int from, mesh_idx;
    {
        std::lock_guard<std::mutex> mk(c->mesh_mu);
        from = c->tab_active; mesh_idx = c->mesh_active[view];
        if (c->stitch_pending) MS_HIP(hipStreamWaitEvent(st, c->last_stitch, 0));
        if (c->mesh_ready[view] && !c->mesh_ready_via_chain[view]) MS_HIP(hipStreamWaitEvent(st, c->mesh_ready[view], 0));
        if (c->mesh_chain_set) MS_HIP(hipStreamWaitEvent(st, c->mesh_chain, 0));
        if (c->tab_wait) MS_HIP(hipStreamWaitEvent(st, c->tab_ready, 0));
    }
    const bool to_alt = from == 0;
    float *w_to = (float *)(to_alt ? A.weights.p : c->weights.p);
    const float *w_from = (const float *)(to_alt ? c->weights.p : A.weights.p);
    uint8_t *m0_to = (uint8_t *)(to_alt ? A.wm0.p : c->wm0.p);
    const uint8_t *m0_from = (const uint8_t *)(to_alt ? c->wm0.p : A.wm0.p);
    float *den_to = (float *)(to_alt ? A.den.p : c->den.p);
    uint8_t *rm_to = (uint8_t *)(to_alt ? A.result_mask.p : c->result_mask.p);
    uint8_t *pure_to = (uint8_t *)(to_alt ? A.pure_maps.p : c->pure_maps.p);
    const ViewDesc *vt_to = (const ViewDesc *)(to_alt ? A.view_tab.p : c->view_tab.p);
    const ViewDesc &V = c->h_views[view];
    const PanoDesc &P = c->pano;
    const int aw = V.aw, ah = V.ah;
    // 1. the mask init_gpu received, through the view's active mesh (blenders.cpp:299-301); kept only if the mesh stays within the planned margin
    ms_image src{(uint8_t *)c->masks.p + c->mask_off[view], (size_t)aw, aw, ah, MS_8UC1};
    ms_image tmp{c->mask_tmp.p, (size_t)aw, aw, ah, MS_8UC1};
    ms_image mx = mesh_image(c, mesh_idx, view, 0), my = mesh_image(c, mesh_idx, view, 1);
    if (int e = launch_remap(src, mx, my, tmp, MS_INTER_LINEAR, MS_BORDER_CONSTANT, st)) return e;
    const float lim = (float)c->cfg.update_mask_margin;
    unsigned lim_bits;
    memcpy(&lim_bits, &lim, 4);
    uint8_t *eff = (uint8_t *)c->masks_eff.p + c->mask_off[view];
    k_mask_select<<<std::min(1024, div_up(aw * ah, 256)), 256, 0, st>>>(eff, (const uint8_t *)c->mask_tmp.p, (size_t)aw * ah, (const unsigned *)c->disp_dev.p + 2 * view + mesh_idx, lim_bits);
    MS_LAUNCH_CHECK();
    // 2. the other views' weights are what they are in the active copy; this view's are rebuilt (blenders.cpp:303-314 = init_gpu's chain)
    MS_HIP(hipMemcpyAsync(w_to, w_from, c->w_total * sizeof(float), hipMemcpyDeviceToDevice, st));
    MS_HIP(hipMemcpyAsync(m0_to, m0_from, c->wm0_total, hipMemcpyDeviceToDevice, st));
    MS_HIP(hipMemcpy2DAsync(m0_to + c->wm0_off[view] + (size_t)V.top * V.wm0_pitch + V.left, V.wm0_pitch, eff, aw, aw, ah, hipMemcpyDeviceToDevice, st));
    {
        ms_image mask{eff, (size_t)aw, aw, ah, MS_8UC1};
        ms_image wmap{c->wm_scratch.p, (size_t)aw * sizeof(float), aw, ah, MS_32FC1};
        if (int e = launch_convert(mask, wmap, 1. / 255., st)) return e;
        auto level_img = [&](int l) {
            const LevelDesc &L = V.lv[l];
            return ms_image{(void *)(w_to + c->w_off[view][l]), (size_t)L.wpitch * sizeof(float), L.w, L.h, MS_32FC1};
        };
        ms_image l0 = level_img(0);
        if (int e = launch_copy_make_border(wmap, l0, V.top, V.left, MS_BORDER_CONSTANT, st)) return e;
        for (int l = 0; l < nb; ++l) {
            ms_image a = level_img(l), b = level_img(l + 1);
            if (int e = launch_pyr_down(a, b, st)) return e;
        }
    }
    // 3. weight sums, result mask, owner maps from the new set of weights
    for (int l = 0; l <= nb; ++l) {
        k_den_all<<<dim3(div_up(P.qw[l], 64), div_up(P.qh[l], 4)), dim3(64, 4), 0, st>>>(vt_to, N, l, den_to + c->den_off[l], P.dpitch[l], P.qh[l], P.qw[l],
                                                                                         l == 0 ? rm_to : nullptr, P.mask_pitch, P.fh, P.fw);
        MS_LAUNCH_CHECK();
    }
    if (c->pure_total)
        if (int e = launch_owner_maps(c, vt_to, pure_to, st)) return e;
    {
        std::lock_guard<std::mutex> mk(c->mesh_mu);
        MS_HIP(hipEventRecord(c->tab_ready, st));
        c->tab_active = from ^ 1; c->tab_wait = true;
        c->l0_integer_only = false;
        // the re-warp above READS this view's mesh slot and displacement entry: later mesh updates (any stream) are ordered behind it through the
        // chain event they all wait for, so the second ms_set_mesh from now cannot overwrite the slot under the re-warp
        if (c->mesh_chain) { MS_HIP(hipEventRecord(c->mesh_chain, st)); c->mesh_chain_set = true; }
    }
    c->use_eff[view] = true;
    return MS_OK;
}

// MultiBandBlender::update_mask (blenders.cpp:297-315): the view's mask re-warped through its CPW mesh (remap LINEAR, BORDER_CONSTANT 0),
// then weight map, border, pyrDown chain as in init_gpu.  The reference re-accumulates the weight sums every frame, here they are static
// tables: the sums, the result mask and the work lists are rebuilt too (calibration-time cost; synchronises).  Disabled in the reference's
// main loop ("causes black seams", timed.cpp:598-605) but part of the blender's interface.
int ms_update_mask(ms_ctx *c, int view, ms_stream stream)
{
    if (int e = ctx_check_view(c, view)) return e;
    if (!c->blender_ready || !c->cfg.enable_cpw || !c->mesh_set[view]) return fail(MS_ERR_STATE, "ms_update_mask: needs enable_cpw, ms_init_blender and a mesh for view %d", view);
    hipStream_t st = as_stream(stream);
    std::lock_guard<std::mutex> ulk(c->mesh_update_mu);
    if (c->cfg.update_mask_margin > 0) return update_mask_async(c, view, st);
    std::lock_guard<std::recursive_mutex> tables_lk(c->tables_mu);      // synchronous form: safe against a concurrent ms_stitch, which blocks for the rebuild
    if (c->stitch_pending) MS_HIP(hipEventSynchronize(c->last_stitch));
    if (c->mesh_ready[view]) { hipEvent_t rdy_; { std::lock_guard<std::mutex> mk_(c->mesh_mu); rdy_ = mesh_ready_event(c, view); } if (rdy_) MS_HIP(hipEventSynchronize(rdy_)); }
    if (c->masks_eff.bytes != c->masks.bytes || !c->masks_eff.p) {
        if (int e = c->masks_eff.alloc(c->masks.bytes)) return e;
        MS_HIP(hipMemcpyAsync(c->masks_eff.p, c->masks.p, c->masks.bytes, hipMemcpyDeviceToDevice, st));
    }
    const int aw = c->roi[view].width, ah = c->roi[view].height;
    ms_image src{(uint8_t *)c->masks.p + c->mask_off[view], (size_t)aw, aw, ah, MS_8UC1};
    ms_image dst{(uint8_t *)c->masks_eff.p + c->mask_off[view], (size_t)aw, aw, ah, MS_8UC1};
    int active;
    { std::lock_guard<std::mutex> mk(c->mesh_mu); active = c->mesh_active[view]; }
    ms_image mx = mesh_image(c, active, view, 0), my = mesh_image(c, active, view, 1);
    if (int e = launch_remap(src, mx, my, dst, MS_INTER_LINEAR, MS_BORDER_CONSTANT, st)) return e;
    MS_HIP(hipStreamSynchronize(st));
    c->use_eff[view] = true;
    return ms_init_blender(c, stream);
}

int ms_get_mesh_displacement(ms_ctx *c, int view, float *out_px)
{
    if (int e = ctx_check_view(c, view)) return e;
    MS_CHECK(out_px != nullptr, "ms_get_mesh_displacement: null output");
    std::lock_guard<std::mutex> lk(c->mesh_update_mu);       // no update in flight while we look; ms_stitch is not blocked by this lock
    if (!c->blender_ready || !c->cfg.enable_cpw || !c->mesh_set[view] || !c->disp_dev.p) return fail(MS_ERR_STATE, "ms_get_mesh_displacement: no mesh set for view %d", view);
    if (c->mesh_ready[view]) { hipEvent_t rdy_; { std::lock_guard<std::mutex> mk_(c->mesh_mu); rdy_ = mesh_ready_event(c, view); } if (rdy_) MS_HIP(hipEventSynchronize(rdy_)); }
    float *h = (float *)pinned_scratch().get(sizeof(float));          // pinned landing zone: see PinnedScratch (common.hpp)
    if (!h) return fail(MS_ERR_NOMEM, "ms_get_mesh_displacement: no pinned staging memory");
    MS_HIP(hipMemcpy(h, (const unsigned *)c->disp_dev.p + 2 * view + c->mesh_active[view], sizeof(float), hipMemcpyDeviceToHost));
    *out_px = *h;
    return MS_OK;
}

// ---- the per-frame path --------------------------------------------------------------------------
// launch a kernel whose LAST template argument is the projection, with the context's projection.
// TARGS = the leading template arguments in parentheses, with a trailing comma, e.g. (true, false,) or ()
#define MS_UNPAREN(...) __VA_ARGS__
#define MS_PROJ_LAUNCH(K, TARGS, CFG, ...)                                                                                   \
    do {                                                                                                                    \
        if (c->cfg.projection == MS_PROJ_SPHERICAL) K<MS_UNPAREN TARGS MS_PROJ_SPHERICAL><<<MS_UNPAREN CFG>>>(__VA_ARGS__);         \
        else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) K<MS_UNPAREN TARGS MS_PROJ_CYLINDRICAL><<<MS_UNPAREN CFG>>>(__VA_ARGS__); \
        else K<MS_UNPAREN TARGS MS_PROJ_PLANE><<<MS_UNPAREN CFG>>>(__VA_ARGS__);                                                    \
    } while (0)

// k_stage1_t<PROJ, AL>: the projection comes first there
#define MS_PROJ_AL_LAUNCH(K, AL, NF, CFG, ...)                                                                              \
    do {                                                                                                                    \
        if (c->cfg.projection == MS_PROJ_SPHERICAL) K<MS_PROJ_SPHERICAL, AL, NF><<<MS_UNPAREN CFG>>>(__VA_ARGS__);           \
        else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) K<MS_PROJ_CYLINDRICAL, AL, NF><<<MS_UNPAREN CFG>>>(__VA_ARGS__);  \
        else K<MS_PROJ_PLANE, AL, NF><<<MS_UNPAREN CFG>>>(__VA_ARGS__);                                                      \
    } while (0)

// k_warp_s<CPW, PROJ, NF>: NF frames per lane, or 1 for a one-frame call
#define MS_WARP_S_LAUNCH(CPW_, NF_, LDS_, ...)                                                                                                  \
    do {                                                                                                                                        \
        const dim3 b_(WARP_BX, WARP_WY);                                                                                                        \
        if (F == 1) { const dim3 g_(c->n_warp_tiles, WARP_BY / WARP_WY, 1);                                                                     \
            if (c->cfg.projection == MS_PROJ_SPHERICAL) k_warp_s<CPW_, MS_PROJ_SPHERICAL, 1><<<g_, b_, LDS_, st>>>(__VA_ARGS__);                 \
            else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) k_warp_s<CPW_, MS_PROJ_CYLINDRICAL, 1><<<g_, b_, LDS_, st>>>(__VA_ARGS__);        \
            else k_warp_s<CPW_, MS_PROJ_PLANE, 1><<<g_, b_, LDS_, st>>>(__VA_ARGS__);                                                            \
        } else { const dim3 g_(c->n_warp_tiles, WARP_BY / WARP_WY, div_up(F, NF_));                                                             \
            if (c->cfg.projection == MS_PROJ_SPHERICAL) k_warp_s<CPW_, MS_PROJ_SPHERICAL, NF_><<<g_, b_, LDS_, st>>>(__VA_ARGS__);               \
            else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) k_warp_s<CPW_, MS_PROJ_CYLINDRICAL, NF_><<<g_, b_, LDS_, st>>>(__VA_ARGS__);      \
            else k_warp_s<CPW_, MS_PROJ_PLANE, NF_><<<g_, b_, LDS_, st>>>(__VA_ARGS__);                                                          \
        }                                                                                                                                       \
    } while (0)

static int stitch_impl(ms_ctx *c, int n_frames, const ms_image *views, ms_image *out8u, ms_image *out16s, hipStream_t st,
                       int cap, const char **names, float *ms_out, int *n_rec, ShardArgs S = ShardArgs{}, ms_image *out_i420 = nullptr, bool nv12 = false)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    std::lock_guard<std::recursive_mutex> tables_lk(c->tables_mu);      // a synchronous table rebuild on another thread (ms_update_mask without a margin) waits for this enqueue, and vice versa
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_stitch: call ms_build_maps / masks / ms_init_blender first");
    MS_CHECK(n_frames >= 1 && n_frames <= c->cfg.max_frames, "ms_stitch: n_frames %d not in [1,%d]", n_frames, c->cfg.max_frames);
    const int N = c->N, nb = c->pano.nb, F = n_frames;
    const bool sharded = c->own_mask != ((N >= 32) ? 0xffffffffu : ((1u << N) - 1u));
    if (S.mode == 0 && sharded) return fail(MS_ERR_STATE, "ms_stitch: this context owns a view shard; use ms_stitch_partial / ms_stitch_finish");
    S.own_mask = c->own_mask;
    S.pstride = c->pacc_stride;
    MS_CHECK(S.mode == 2 || views != nullptr, "ms_stitch: null views");
    PanoDesc P = c->pano;              // (the pointers of the active copy of the tables are filled in under the lock below)
    // Frames per call: up to MAX_FRAMES (64 since round 6).  The kernels that read the callers' frames take their pointers in a by-value table of MAX_SRC entries (a kernel's
    // arguments are limited to 4 KiB): those launches -- the projection warp, the first CPW remap, the mesh remap -- go out in chunks of f_chunk = MAX_SRC / views frames (32 for six
    // views, 16 for twelve), each chunk with its own table and the per-frame buffers offset by its first frame; they are the frame's large, bandwidth-bound launches and lose
    // nothing by it.  Every other kernel of the call (the reduce chain, the tails, the band chain: latency-bound launches whose fixed cost a longer batch amortises) covers all frames.
    const int F_all = F, f_chunk = std::min(F, std::max(1, MAX_SRC / N));
    SrcAll src{};
    auto src_chunk = [&](int f0, int nf) { SrcTable t{}; for (int i = 0; i < nf * N; ++i) { t.p[i] = src.p[f0 * N + i]; t.step[i] = src.step[f0 * N + i]; } return t; };
    for (int i = 0; i < F * N && S.mode != 2; ++i) {
        if (!((c->own_mask >> (i % N)) & 1u)) continue;       // another shard's view: not read
        if (!((c->needed_mask >> (i % N)) & 1u)) continue;    // column sharding: no pixel of this view reaches the window
        if (nv12)
            MS_CHECK(views[i].data && views[i].type == MS_8UC1 && views[i].rows == c->cfg.src_height * 3 / 2 && views[i].cols == c->cfg.src_width && views[i].step == views[i % N].step,
                     "ms_stitch_nv12: view %d must be the NV12 planes of a %dx%d frame (8UC1, %d rows), every frame of a view with the same step", i, c->cfg.src_width, c->cfg.src_height, c->cfg.src_height * 3 / 2);
        else
        MS_CHECK(views[i].data && views[i].type == MS_8UC3 && views[i].rows == c->cfg.src_height && views[i].cols == c->cfg.src_width,
                 "ms_stitch: view %d must be 8UC3 %dx%d", i, c->cfg.src_width, c->cfg.src_height);
        src.p[i] = (const uint8_t *)views[i].data;
        src.step[i] = (unsigned)views[i].step;
    }
    if (sharded && S.mode == 1)      // the full-grid fallback kernels touch every view slot: give the unowned ones a valid (ignored) source
        for (int f = 0; f < F; ++f) {
            int owned = 0;
            while (!((c->own_mask >> owned) & 1u)) ++owned;
            for (int v = 0; v < N; ++v)
                if (!((c->own_mask >> v) & 1u)) { src.p[f * N + v] = src.p[f * N + owned]; src.step[f * N + v] = src.step[f * N + owned]; }
        }
    OutTable out{};
    for (int f = 0; f < F; ++f) {
        if (out8u && out8u[f].data) {
            MS_CHECK(out8u[f].type == MS_8UC3 && out8u[f].rows == P.out_h && out8u[f].cols == P.out_w,
                     "ms_stitch: out8u[%d] must be 8UC3 %dx%d", f, P.out_w, P.out_h);
            out.p8[f] = (uint8_t *)out8u[f].data; out.step8[f] = (unsigned)out8u[f].step;
        }
        if (out16s && out16s[f].data) {
            MS_CHECK(out16s[f].type == MS_16SC3 && out16s[f].rows == P.fh && out16s[f].cols == P.fw,
                     "ms_stitch: out16s[%d] must be 16SC3 %dx%d", f, P.fw, P.fh);
            out.p16[f] = (int16_t *)out16s[f].data; out.step16[f] = (unsigned)out16s[f].step;
        }
        if (out_i420 && out_i420[f].data) {
            MS_CHECK(out_i420[f].type == MS_8UC1 && out_i420[f].cols == P.out_w && out_i420[f].rows == P.i_rows * 3 / 2 && out_i420[f].step == (size_t)P.out_w,
                     "ms_stitch_i420: out[%d] must be a contiguous 8UC1 image of %d x %d (I420 of canvas rows %d..%d)", f, P.out_w, P.i_rows * 3 / 2, P.i_y0, P.i_y0 + P.i_rows - 1);
            out.pi[f] = (uint8_t *)out_i420[f].data;
        }
    }
    if (out_i420) {
        if (!(P.nb >= 1 && c->blend_vec[0] && c->cfg.debug_simple_kernels == 0 && S.mode == 0 && (P.out_w & 1) == 0 && P.i_rows > 0))
            return fail(MS_ERR_UNSUPPORTED, "ms_stitch_i420: needs the tiled level-0 band kernel (>= 1 band, pano width a multiple of 8, no view sharding) and an even canvas width");
    }
    if (nv12 && !(c->warp_tiled && c->cfg.debug_simple_kernels == 0 && c->cfg.cpu_flavour_remap == 0 && S.mode == 0 &&
                  (c->cfg.src_width & 1) == 0 && (c->cfg.src_height & 1) == 0 && c->cfg.src_width >= 4))
        return fail(MS_ERR_UNSUPPORTED, "ms_stitch_nv12: the NV12-sampling kernels cover the tiled projection warp / CPW stage 1 of even-sized frames, unsharded; convert with ms_nv12_to_bgr_batch otherwise");
    // aligned 8-byte windows where every plane, step and the width are multiples of 4 (cameras' frames in ordinary buffers are); the unaligned 2- / 4-byte reads otherwise
    bool nv_al = nv12 && (c->cfg.src_width & 3) == 0 && c->cfg.src_width >= 8 && dev_knob("MS_NV12_ALIGNED", 1) != 0;
    for (int i = 0; i < F * N && nv_al; ++i) if (src.p[i]) nv_al = (((uintptr_t)src.p[i] | src.step[i]) & 3) == 0;
    MeshTable mesh{};
    const bool cpw = c->cfg.enable_cpw != 0;
    DispTable disp{};
    { const float lim = (float)CPW_DMAX; memcpy(&disp.limit_bits, &lim, 4); }
    // CPW: mesh_mu is held from here to the hipEventRecord(last_stitch) at the end of the enqueue (RAII): a mesh update that starts meanwhile
    // sees this stitch in last_stitch before it may touch the buffer this stitch reads.  Only enqueues happen under the lock.
    std::unique_lock<std::mutex> mesh_lk(c->mesh_mu, std::defer_lock);
    if (cpw) {
        mesh_lk.lock();
        hipEvent_t mesh_waited = nullptr;
        for (int v = 0; v < N; ++v) {
            if (!c->mesh_set[v]) return fail(MS_ERR_STATE, "ms_stitch: enable_cpw is set but view %d has no mesh", v);
            disp.p[v] = (const unsigned *)c->disp_dev.p + 2 * v + c->mesh_active[v];
            if (c->mesh_wait[v]) {
                hipEvent_t e = mesh_ready_event(c, v);
                if (e != mesh_waited) { MS_HIP(hipStreamWaitEvent(st, e, 0)); mesh_waited = e; }      // (the views of one ms_set_meshes call share their event: one wait)
                c->mesh_wait[v] = false;
            }
            ms_image mx = mesh_image(c, c->mesh_active[v], v, 0), my = mesh_image(c, c->mesh_active[v], v, 1);
            mesh.x[v] = (const float *)mx.data; mesh.y[v] = (const float *)my.data; mesh.pitch[v] = c->map_pitch[v];
        }
    }
    // which copy of the mask-dependent tables (enqueue-only ms_update_mask; same lock, same hand-over as the meshes)
    const ViewDesc *vt = (const ViewDesc *)c->view_tab.p;
    if (cpw && c->cfg.update_mask_margin > 0) {
        if (c->tab_wait) MS_HIP(hipStreamWaitEvent(st, c->tab_ready, 0));
        if (c->tab_active == 1) {
            vt = (const ViewDesc *)c->alt.view_tab.p;
            for (int l = 0; l <= nb; ++l) P.den[l] = c->alt.pano.den[l];
            for (int l = 0; l < nb; ++l) P.pure[l] = c->alt.pano.pure[l];
            P.mask = c->alt.pano.mask;
        }
    }

    // Occupancy of the warp kernels, set through dynamic LDS nobody touches (bytes per 128-lane workgroup; 160 KiB per CU).  The aligned-read projection warp needs 85 VGPRs
    // (5 waves per SIMD) since its tap addresses became SGPR base + 32-bit offset, and runs 3.7 % FASTER at 4 waves (20 000 bytes -> 8 workgroups per CU): fewer waves
    // thrash the vector cache less (same-box sweep, us per 16 frames: 5 waves 215, 4 waves 207, 3 waves 214; profiles/r03_resize_ab.txt).  The unaligned variant
    // (config 5, 61 VGPRs) and the mesh remap (98 VGPRs = 4 waves anyway) are best left alone.  MS_WARP_LDS=<bytes> overrides all three (A/B).
    static const int warp_lds_env = dev_knob("MS_WARP_LDS", -1);
    const size_t warp_lds = warp_lds_env >= 0 ? (size_t)warp_lds_env : 0, warp_lds_al = warp_lds_env >= 0 ? (size_t)warp_lds_env : (F <= 2 ? 0 : 20000);      // (one or two frames per call: the chip is not full, every wave that fits helps -- 92.5 -> 91.2 us per frame)
    // k_warp_s / the shared form of k_stage1_t: every frame of a view with the same row step and the same address modulo 4 (then a pixel's aligned tap offset is one
    // 32-bit value for all frames of a lane).  True for any sane caller (frames of one camera in buffers of one shape); checked, not assumed.
    static const bool warp_shared_knob = dev_knob("MS_WARP_SHARED", 1) != 0;
    c->last_warp_kernel = MS_WARP_KERNEL_SIMPLE; c->last_stage1_kernel = MS_WARP_KERNEL_NONE;      // (overwritten below by whichever tile kernel is launched)
    bool src_shared = warp_shared_knob && S.mode != 2;
    for (int i = N; i < F * N && src_shared; ++i)
        if (src.p[i]) src_shared = src.p[i % N] && src.step[i] == src.step[i % N] && (((uintptr_t)src.p[i] ^ (uintptr_t)src.p[i % N]) & 3) == 0;
    bool src_steps_equal = warp_shared_knob && S.mode != 2;
    for (int i = N; i < F * N && src_steps_equal; ++i)
        if (src.p[i]) src_steps_equal = src.p[i % N] && src.step[i] == src.step[i % N];
    const bool int_only = c->l0_integer_only.load() && P.pure[0] != nullptr;      // (an enqueue-only mask update clears it before it publishes its tables under mesh_mu: a stale `true` can only meet the old tables)

    // one context has ONE set of per-batch intermediates: calls on a different stream than the previous one are ordered behind it on the GPU
    // (the reference makes a fresh cuda::Stream per stitch_online call, timed.cpp:64, and relies on the NULL stream for ordering)
    if (c->last_stream_set && c->last_stream != st && c->stitch_pending) MS_HIP(hipStreamWaitEvent(st, c->last_stitch, 0));
    c->last_stream = st; c->last_stream_set = true;

    const uint8_t *g0 = (const uint8_t *)c->g0.p;
    uint8_t *gl = (uint8_t *)c->gl.p;
    int16_t *cl = (int16_t *)c->cl.p;
    const dim3 blk(64, 4);

    std::vector<hipEvent_t> ev;
    int rec = 0;
    auto mark = [&](const char *name) -> int {
        if (!names) return MS_OK;
        hipEvent_t e;
        MS_HIP(hipEventCreate(&e));
        MS_HIP(hipEventRecord(e, st));
        ev.push_back(e);
        if (name && rec < cap) names[rec++] = name;
        return MS_OK;
    };
    if (int e = mark(nullptr)) return e;

    static const char *down_names[MAX_LEVELS] = {"k_down_l0", "k_down_l1", "k_down_l2", "k_down_l3", "k_down_l4", "k_down_l5", "k_down_l6", "k_down_l7"};
    static const char *blend_names[MAX_LEVELS] = {"k_blend_l0", "k_blend_l1", "k_blend_l2", "k_blend_l3", "k_blend_l4", "k_blend_l5", "k_blend_l6", "k_blend_l7"};
    if (S.mode != 2) {       // (finish mode starts from the partial sums: no warp, no pyramids)
    if (cpw) {
        for (int fc0 = 0; fc0 < F_all; fc0 += f_chunk) {      // (source-table chunks: see f_chunk above)
            const int F = std::min(f_chunk, F_all - fc0);
            const SrcTable src = src_chunk(fc0, F);
            uint8_t *const stage_w = (uint8_t *)c->stage.p + (size_t)fc0 * c->stage_stride, *const g0_w = (uint8_t *)c->g0.p + (size_t)fc0 * c->g0_stride;
            (void)stage_w; (void)g0_w;
        if (nv12) {        // CPW stage 1 straight from the cameras' NV12 planes (k_stage1_nv12)
            const dim3 b_(WARP_BX, S1_BY_NV);
#define MS_S1NV_LAUNCH(NF_, ALN_)                                                                                                                                     \
    do {                                                                                                                                                              \
        const dim3 g_(c->n_stage1_tiles, 1, div_up(F, NF_));                                                                                                          \
        if (c->cfg.projection == MS_PROJ_SPHERICAL) k_stage1_nv12<MS_PROJ_SPHERICAL, NF_, ALN_><<<g_, b_, 0, st>>>((const WarpTile *)c->stage1_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride, disp, F); \
        else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) k_stage1_nv12<MS_PROJ_CYLINDRICAL, NF_, ALN_><<<g_, b_, 0, st>>>((const WarpTile *)c->stage1_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride, disp, F); \
        else k_stage1_nv12<MS_PROJ_PLANE, NF_, ALN_><<<g_, b_, 0, st>>>((const WarpTile *)c->stage1_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride, disp, F); \
    } while (0)
            c->last_stage1_kernel = MS_WARP_KERNEL_NV12;
            if (nv_al) { if (F == 1) MS_S1NV_LAUNCH(1, true); else MS_S1NV_LAUNCH(2, true); }
            else { if (F == 1) MS_S1NV_LAUNCH(1, false); else MS_S1NV_LAUNCH(2, false); }
#undef MS_S1NV_LAUNCH
        } else
        if (c->cfg.debug_simple_kernels == 0)
        {
            // frames per lane of the first CPW remap: three where the source is sampled about 1 : 1 (VALU-bound), two otherwise (see k_stage1_t)
            static const int s1_env = dev_knob("MS_S1_NF", 0);
            // (round 5: the shared-offset form takes three wherever it runs -- config 3's first remap 406 -> 387-392 us per 32 frames, 385 -> 364 per 30; the per-frame form keeps the minification rule)
            const int s1_nf = s1_env == 2 || s1_env == 3 ? s1_env : ((c->warp_aligned && src_shared) || (c->warp_minification > 0 && c->warp_minification < 1.5)) ? 3 : 2;
            static const int s1_lds = dev_knob("MS_S1_LDS", 0);      // occupancy A/B knob (dynamic LDS nobody touches)
#define MS_S1_LAUNCH(AL, NF) MS_PROJ_AL_LAUNCH(k_stage1_t, AL, NF, (dim3(c->n_stage1_tiles, 1, div_up(F, NF)), dim3(WARP_BX, S1_BY), s1_lds, st), \
                    (const WarpTile *)c->stage1_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride, disp, F)
#define MS_S1S_LAUNCH(NF)                                                                                                                                     \
    do {                                                                                                                                                      \
        const dim3 g_(c->n_stage1_tiles, 1, div_up(F, NF)), b_(WARP_BX, S1_BY);                                                                               \
        if (c->cfg.projection == MS_PROJ_SPHERICAL) k_stage1_s<MS_PROJ_SPHERICAL, NF><<<g_, b_, s1_lds, st>>>((const WarpTile *)c->stage1_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride, disp, F); \
        else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) k_stage1_s<MS_PROJ_CYLINDRICAL, NF><<<g_, b_, s1_lds, st>>>((const WarpTile *)c->stage1_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride, disp, F); \
        else k_stage1_s<MS_PROJ_PLANE, NF><<<g_, b_, s1_lds, st>>>((const WarpTile *)c->stage1_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride, disp, F); \
    } while (0)
            if (c->warp_aligned && src_shared) { c->last_stage1_kernel = MS_WARP_KERNEL_SHARED_ALIGNED; if (s1_nf == 3) MS_S1S_LAUNCH(3); else MS_S1S_LAUNCH(2); }
            else if (c->warp_aligned) { c->last_stage1_kernel = MS_WARP_KERNEL_PER_FRAME_ALIGNED; if (s1_nf == 3) MS_S1_LAUNCH(true, 3); else MS_S1_LAUNCH(true, 2); }
            else { c->last_stage1_kernel = MS_WARP_KERNEL_PER_FRAME_UNALIGNED; if (s1_nf == 3) MS_S1_LAUNCH(false, 3); else MS_S1_LAUNCH(false, 2); }
#undef MS_S1_LAUNCH
#undef MS_S1S_LAUNCH
        }
        else {
            c->last_stage1_kernel = MS_WARP_KERNEL_SIMPLE;
            k_remap_gain<<<dim3(div_up(c->max_aw, 64), div_up(c->max_ah, 4), F * N), blk, 0, st>>>(
                vt, N, src, c->cfg.src_height, c->cfg.src_width, stage_w, c->stage_stride);
        }
        }
        MS_LAUNCH_CHECK();
        if (int e = mark("k_remap_gain")) return e;
        for (int fc0 = 0; fc0 < F_all; fc0 += f_chunk) {      // (source-table chunks: see f_chunk above)
            const int F = std::min(f_chunk, F_all - fc0);
            const SrcTable src = src_chunk(fc0, F);
            uint8_t *const stage_w = (uint8_t *)c->stage.p + (size_t)fc0 * c->stage_stride, *const g0_w = (uint8_t *)c->g0.p + (size_t)fc0 * c->g0_stride;
            (void)stage_w; (void)g0_w;
        // the stage images of a view have the same pitch and the same address modulo 4 in every frame BY CONSTRUCTION (stage_stride is a multiple of 256): the mesh remap
        // always takes the shared-offset form.  Its unshared twin (k_warp_t<CPW>) was unreachable in the shipped library and untested: dev-knob builds only.
        if (c->warp_tiled && c->cfg.debug_simple_kernels == 0 && warp_shared_knob) {
            MS_CHECK((c->stage_stride & 3) == 0, "internal: stage image stride %lld not a multiple of 4", c->stage_stride);
            c->last_warp_kernel = MS_WARP_KERNEL_SHARED_ALIGNED;
            MS_WARP_S_LAUNCH(true, warp_nf(true), warp_lds, (const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, (const uint8_t *)stage_w, c->stage_stride, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
        }
#ifdef MS_DEV_KNOBS
        else if (c->warp_tiled && c->cfg.debug_simple_kernels == 0) {
            c->last_warp_kernel = MS_WARP_KERNEL_PER_FRAME_ALIGNED;
            MS_PROJ_LAUNCH(k_warp_t, (true, true,), (dim3(c->n_warp_tiles, WARP_BY / WARP_WY, div_up(F, warp_nf(true))), dim3(WARP_BX, WARP_WY), warp_lds, st), (const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, (const uint8_t *)stage_w, c->stage_stride, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
        }
#endif
        else
            k_warp<true><<<dim3(div_up(c->max_pw, 64), div_up(c->max_ph, 4), F * N), blk, 0, st>>>(
                vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, (const uint8_t *)stage_w, c->stage_stride, g0_w, c->g0_stride);
        }
    } else {
        for (int fc0 = 0; fc0 < F_all; fc0 += f_chunk) {      // (source-table chunks: see f_chunk above)
            const int F = std::min(f_chunk, F_all - fc0);
            const SrcTable src = src_chunk(fc0, F);
            uint8_t *const stage_w = (uint8_t *)c->stage.p + (size_t)fc0 * c->stage_stride, *const g0_w = (uint8_t *)c->g0.p + (size_t)fc0 * c->g0_stride;
            (void)stage_w; (void)g0_w;
        if (c->warp_tiled && c->cfg.debug_simple_kernels == 0) {
        // opt-in (ms_config.warp_lds_stage = 1; MS_WARP_ASYNC=1 in a dev-knob build): source tiles staged in LDS by asynchronous LDS-DMA, persistent waves (k_warp_a) -- bit-identical,
        // measured slower than the direct gathers on config 2 (353 vs 242 us per 16 frames: at its 1.6-2.1 x minification only 54 % of the tiles' source
        // boxes fit a staging buffer and 10 waves per CU cannot hide what 20 do; profiles/r02_warp_probes.txt)
        static const bool env_on = dev_knob("MS_WARP_ASYNC", 0) != 0;
        bool staged = c->warp_lds_tiles > 0 && (c->cfg.warp_lds_stage == 1 || (env_on && c->cfg.warp_lds_stage != 2));
        for (int i = 0; i < F * N; ++i) staged = staged && ((uintptr_t)src.p[i] & 15) == 0;   // chunk copies start on 16-byte lines of the buffer
        if (nv12) {        // the cameras' NV12 planes sampled directly (k_warp_nv12): no BGR image in between
            const dim3 b_(WARP_BX, WARP_WY);
#define MS_NV12_LAUNCH(NF_, ALN_)                                                                                                                                     \
    do {                                                                                                                                                              \
        const dim3 g_(c->n_warp_tiles, WARP_BY / WARP_WY, div_up(F, NF_));                                                                                            \
        if (c->cfg.projection == MS_PROJ_SPHERICAL) k_warp_nv12<MS_PROJ_SPHERICAL, NF_, ALN_><<<g_, b_, 0, st>>>((const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F); \
        else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) k_warp_nv12<MS_PROJ_CYLINDRICAL, NF_, ALN_><<<g_, b_, 0, st>>>((const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F); \
        else k_warp_nv12<MS_PROJ_PLANE, NF_, ALN_><<<g_, b_, 0, st>>>((const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F); \
    } while (0)
            c->last_warp_kernel = MS_WARP_KERNEL_NV12;
            if (nv_al) { if (F == 1) MS_NV12_LAUNCH(1, true); else MS_NV12_LAUNCH(2, true); }
            else { if (F == 1) MS_NV12_LAUNCH(1, false); else MS_NV12_LAUNCH(2, false); }
#undef MS_NV12_LAUNCH
        } else if (staged) {
            c->last_warp_kernel = MS_WARP_KERNEL_LDS_STAGED;
            const long long items = (long long)c->n_warp_tiles * F;
            const int grid = (int)std::min<long long>(items, (long long)c->n_cus * (160 * 1024 / (2 * WA_BUF_BYTES)));
            MS_PROJ_LAUNCH(k_warp_a, (), (dim3(grid), dim3(64), 2 * WA_BUF_BYTES, st), (const WarpTile *)c->warp_tiles.p, c->n_warp_tiles, vt, N, src, c->cfg.src_height, c->cfg.src_width, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
        } else
        {
            // (with shared offsets the aligned form wins at every minification measured: config 5 -- 2.7 x, k_warp_t's unaligned territory -- 757 -> 730 / 738 -> 699 us per 16 frames;
            //  the minification rule of build_plan still picks between k_warp_t's two forms when the frames do not share step and alignment)
            static const int force_al = dev_knob("MS_WARP_ALIGNED", -1);
            c->last_warp_kernel = ((c->warp_aligned || force_al != 0) && src_shared) ? MS_WARP_KERNEL_SHARED_ALIGNED : c->warp_aligned ? MS_WARP_KERNEL_PER_FRAME_ALIGNED
                                  : (src_steps_equal && F > 1 && dev_knob("MS_WARP_SHARED_U", 1)) ? MS_WARP_KERNEL_SHARED_UNALIGNED : MS_WARP_KERNEL_PER_FRAME_UNALIGNED;
            if ((c->warp_aligned || force_al != 0) && src_shared)
                MS_WARP_S_LAUNCH(false, WARP_NF_S, warp_lds_al, (const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
            else if (c->warp_aligned)
                MS_PROJ_LAUNCH(k_warp_t, (false, true,), (dim3(c->n_warp_tiles, WARP_BY / WARP_WY, div_up(F, WARP_NF)), dim3(WARP_BX, WARP_WY), warp_lds_al, st), (const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
            else if (src_steps_equal && F > 1 && dev_knob("MS_WARP_SHARED_U", 1)) {      // the unaligned-read form with shared offsets (config 5): only the row step has to agree
                const dim3 g_(c->n_warp_tiles, WARP_BY / WARP_WY, div_up(F, WARP_NF)), b_(WARP_BX, WARP_WY);
                if (c->cfg.projection == MS_PROJ_SPHERICAL) k_warp_s<false, MS_PROJ_SPHERICAL, WARP_NF, false><<<g_, b_, warp_lds, st>>>((const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
                else if (c->cfg.projection == MS_PROJ_CYLINDRICAL) k_warp_s<false, MS_PROJ_CYLINDRICAL, WARP_NF, false><<<g_, b_, warp_lds, st>>>((const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
                else k_warp_s<false, MS_PROJ_PLANE, WARP_NF, false><<<g_, b_, warp_lds, st>>>((const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
            } else
                MS_PROJ_LAUNCH(k_warp_t, (false, false,), (dim3(c->n_warp_tiles, WARP_BY / WARP_WY, div_up(F, WARP_NF)), dim3(WARP_BX, WARP_WY), warp_lds, st), (const WarpTile *)c->warp_tiles.p, vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride, (const float2 *)c->tabs.p, F);
        }
    } else if (c->cfg.cpu_flavour_remap != 0) {
        k_warp<false, true><<<dim3(div_up(c->max_pw, 64), div_up(c->max_ph, 4), F * N), blk, 0, st>>>(
            vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride);
    } else {
        k_warp<false><<<dim3(div_up(c->max_pw, 64), div_up(c->max_ph, 4), F * N), blk, 0, st>>>(
            vt, N, src, c->cfg.src_height, c->cfg.src_width, mesh, nullptr, 0, g0_w, c->g0_stride);
    }
        }
    }
    MS_LAUNCH_CHECK();
    if (int e = mark("k_warp")) return e;

    static const int tail_sw_knob = dev_knob("MS_TAIL_SW", 0);      // A/B: force the strip width of the reduce tail (16 = the round-4 form)
    const bool dt_wide = F > 2 && tail_sw_knob != TAIL_STRIP;
    const int dt_l0 = c->tail_l0, dt_lds = dt_wide ? c->tail_lds_b : c->tail_lds, dt_strips = dt_wide ? c->tail_strips_b : c->tail_strips, dt_sw = dt_wide ? c->tail_sw_b : TAIL_STRIP;     // (starting the reduce tail a level finer was measured slower even for one frame)
    for (int l = 0; l < nb; ++l) {
        if (l == dt_l0 && c->cfg.debug_simple_kernels == 0) {
            k_down_tail<<<dim3(F * N * 3 * dt_strips), (F <= 2 || dt_sw > TAIL_STRIP) ? dim3(64, 16) : blk, dt_lds, st>>>(vt, N, l, nb, dt_strips, dt_sw, gl, c->gl_stride, c->own_mask & c->needed_mask);
            MS_LAUNCH_CHECK();
            if (int e = mark("k_down_tail")) return e;
            break;
        }
        const int ow = (std::max(c->max_pw >> l, 1) + 1) / 2, oh = (std::max(c->max_ph >> l, 1) + 1) / 2;
        if (c->down_vec[l] && c->cfg.debug_simple_kernels == 0) {   // level-l widths are multiples of 8: tile list, DOWN_ROWS (4) rows x 4 cols per lane
            const dim3 g(c->n_down_tiles[l], 3, F), b(32, 8);
            static const int down_lds = dev_knob("MS_DOWN_LDS", 0);      // occupancy A/B knob (dynamic LDS nobody touches)
            if (l == 0) k_down_t<true><<<g, b, down_lds, st>>>((const DownTile *)c->down_tiles[l].p, vt, l, g0, c->g0_stride, gl, c->gl_stride);
            else        k_down_t<false><<<g, b, 0, st>>>((const DownTile *)c->down_tiles[l].p, vt, l, gl, c->gl_stride, gl, c->gl_stride);
        } else {
            const dim3 g(div_up(ow, 64), div_up(oh, 4), F * N * 3);
            if (l == 0) k_down<<<g, blk, 0, st>>>(vt, N, l, g0, c->g0_stride, gl, c->gl_stride);
            else        k_down<<<g, blk, 0, st>>>(vt, N, l, gl, c->gl_stride, gl, c->gl_stride);
        }
        MS_LAUNCH_CHECK();
        if (int e = mark(down_names[l])) return e;
    }
    }   // S.mode != 2
    if (nb == 0) {       // single band (FeatherBlender weights or plain mask weights): no pyramid, one pass over the pano
        if (S.mode != 0) return fail(MS_ERR_UNSUPPORTED, "view sharding needs num_bands >= 1");
        k_single_band<<<dim3(div_up(P.fw, 64), div_up(P.fh, 4), F), blk, 0, st>>>(vt, P, g0, c->g0_stride, out);
        MS_LAUNCH_CHECK();
        if (int e = mark("k_single_band")) return e;
    } else {
#define MS_MODE_LAUNCH(K, G, B, ...)                                     \
    do {                                                                \
        if (S.mode == 0) K<0><<<G, B, 0, st>>>(__VA_ARGS__);            \
        else if (S.mode == 1) K<1><<<G, B, 0, st>>>(__VA_ARGS__);       \
        else K<2><<<G, B, 0, st>>>(__VA_ARGS__);                        \
    } while (0)
#define MS_MODE_LAUNCH2(K, L0, G, B, ...)                               \
    do {                                                                \
        if (S.mode == 0) K<L0, 0><<<G, B, 0, st>>>(__VA_ARGS__);        \
        else if (S.mode == 1) K<L0, 1><<<G, B, 0, st>>>(__VA_ARGS__);   \
        else K<L0, 2><<<G, B, 0, st>>>(__VA_ARGS__);                    \
    } while (0)
    int l_first = nb - 1;
    static const bool bt2_off = dev_knob("MS_LIVE_TAIL", 1) == 0;
    const bool bt2 = F <= 2 && c->btail2_t >= 0 && !bt2_off;      // live mode (1-2 frames per call): launches dominate, the band tail starts one band finer
    const int bt_t = bt2 ? c->btail2_t : c->btail_t, bt_lds = bt2 ? c->btail2_lds : c->btail_lds, bt_strips = bt2 ? c->btail2_strips : c->btail_strips, bt_w = BTAIL_W;
    if (S.mode == 0 && bt_t >= 0) {      // bands nb .. bt_t in one launch
        int s0 = 0, s1 = bt_strips;
        if (c->col_end > c->col_begin) {      // column sharding: only the strips of band bt_t that hold a column the window depends on
            int ra = c->col_begin, rb = c->col_end;
            for (int l = 0; l < bt_t; ++l) { ra = std::max(ra / 2 - 1, 0); rb = std::min((rb + 1) / 2 + 1, P.qw[l + 1]); }
            s0 = std::min(ra / bt_w, bt_strips - 1); s1 = std::max(std::min(div_up(rb, bt_w), bt_strips), s0 + 1);
        }
        static const int bt_rows = dev_knob("MS_BTAIL_ROWS", 0);      // (A/B: lane rows per workgroup)
        const dim3 bt_blk(64, bt_rows > 0 ? bt_rows : (F <= 2 ? 16 : 4));      // live mode: a strip's few hundred quads per band on 1024 lanes instead of 256 -- fewer serial rounds
        if (F <= 2) k_blend_tail<true><<<dim3(s1 - s0, 3, F), bt_blk, bt_lds, st>>>(vt, P, bt_t, gl, c->gl_stride, cl, c->cl_stride, s0, bt_w);
        else        k_blend_tail<false><<<dim3(s1 - s0, 3, F), bt_blk, bt_lds, st>>>(vt, P, bt_t, gl, c->gl_stride, cl, c->cl_stride, s0, bt_w);
        MS_LAUNCH_CHECK();
        if (int e = mark("k_blend_tail")) return e;
        l_first = bt_t - 1;
    } else {
        MS_MODE_LAUNCH(k_blend_top, dim3(div_up(P.qw[nb], 64), div_up(P.qh[nb], 4), F), blk, vt, P, g0, c->g0_stride, gl, c->gl_stride, cl, c->cl_stride, S);
        MS_LAUNCH_CHECK();
        if (int e = mark(blend_names[nb])) return e;
    }
    for (int l = l_first; l >= 0; --l) {
        if (c->blend_vec[l] && c->cfg.debug_simple_kernels == 0) {
            const dim3 g(c->n_blend_tiles[l], 4, F), b(32, 2);
            static const bool no_cls = dev_knob("MS_BLEND_CLS", 1) == 0;      // MS_BLEND_CLS=0: always the all-classes build (A/B)
            // level 0 without general cells (the usual case: binary Voronoi seam masks): the build that has no general path -- 88 instead of 120 VGPRs, 5 waves per SIMD
            static const int blend_lds = dev_knob("MS_BLEND_LDS", 0);      // occupancy A/B knob
            if (l == 0 && S.mode == 0 && int_only && !no_cls) k_blend8<true, 0, 1><<<g, b, blend_lds, st>>>((const BlendTile *)c->blend_tiles[l].p, vt, P, l, g0, c->g0_stride, gl, c->gl_stride, cl, c->cl_stride, out, S);
            else if (l == 0) MS_MODE_LAUNCH2(k_blend8, true, g, b, (const BlendTile *)c->blend_tiles[l].p, vt, P, l, g0, c->g0_stride, gl, c->gl_stride, cl, c->cl_stride, out, S);
            else        MS_MODE_LAUNCH2(k_blend8, false, g, b, (const BlendTile *)c->blend_tiles[l].p, vt, P, l, g0, c->g0_stride, gl, c->gl_stride, cl, c->cl_stride, out, S);
        } else {
            const dim3 g(div_up(P.qw[l] / 2, 64), div_up(P.qh[l] / 2, 4), F);
            if (l == 0) MS_MODE_LAUNCH2(k_blend, true, g, blk, vt, P, l, g0, c->g0_stride, gl, c->gl_stride, cl, c->cl_stride, out, S);
            else        MS_MODE_LAUNCH2(k_blend, false, g, blk, vt, P, l, g0, c->g0_stride, gl, c->gl_stride, cl, c->cl_stride, out, S);
        }
        MS_LAUNCH_CHECK();
        if (int e = mark(blend_names[l])) return e;
    }
    }   // nb > 0
    MS_HIP(hipEventRecord(c->last_stitch, st));
    c->stitch_pending = true;

    if (names) {
        MS_HIP(hipEventSynchronize(ev.back()));
        for (size_t i = 1; i < ev.size() && (int)i - 1 < cap; ++i) {
            float t = 0.f;
            MS_HIP(hipEventElapsedTime(&t, ev[i - 1], ev[i]));
            ms_out[i - 1] = t;
        }
        for (auto e : ev) (void)hipEventDestroy(e);
        if (n_rec) *n_rec = rec;
    }
    return MS_OK;
}

int ms_stitch(ms_ctx *c, int n_frames, const ms_image *views, ms_image *out8u, ms_image *out16s, ms_stream stream)
{
    return stitch_impl(c, n_frames, views, out8u, out16s, as_stream(stream), 0, nullptr, nullptr, nullptr);
}

int ms_stitch_nv12(ms_ctx *c, int n_frames, const ms_image *views_nv12, ms_image *out8u, ms_image *out16s, ms_stream stream)
{
    return stitch_impl(c, n_frames, views_nv12, out8u, out16s, as_stream(stream), 0, nullptr, nullptr, nullptr, ShardArgs{}, nullptr, true);
}

int ms_stitch_i420(ms_ctx *c, int n_frames, const ms_image *views, ms_image *out_i420, ms_stream stream)
{
    if (!out_i420) return fail(MS_ERR_INVALID, "ms_stitch_i420: null output");
    return stitch_impl(c, n_frames, views, nullptr, nullptr, as_stream(stream), 0, nullptr, nullptr, nullptr, ShardArgs{}, out_i420);
}

int ms_get_col_window(const ms_ctx *c, int *begin, int *end)
{
    if (!c || !begin || !end) return fail(MS_ERR_INVALID, "ms_get_col_window: null argument");
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_get_col_window: call ms_init_blender first");
    const bool windowed = c->col_end > c->col_begin;
    *begin = windowed ? c->col_begin : 0;
    *end = windowed ? c->col_end : c->pano.fw;
    return MS_OK;
}

int ms_get_needed_views(const ms_ctx *c, unsigned *mask)
{
    if (!c || !mask) return fail(MS_ERR_INVALID, "ms_get_needed_views: null argument");
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_get_needed_views: call ms_init_blender first");
    const unsigned all = (c->N >= 32) ? 0xffffffffu : ((1u << c->N) - 1u);
    *mask = c->own_mask & c->needed_mask & all;
    return MS_OK;
}

int ms_get_i420_rows(const ms_ctx *c, int *first_canvas_row, int *rows)
{
    if (!c || !first_canvas_row || !rows) return fail(MS_ERR_INVALID, "ms_get_i420_rows: null argument");
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_get_i420_rows: call ms_init_blender first");
    *first_canvas_row = c->pano.i_y0; *rows = c->pano.i_rows;
    return MS_OK;
}

int ms_stitch_timed(ms_ctx *c, int n_frames, const ms_image *views, ms_image *out8u, ms_image *out16s, ms_stream stream,
                    int cap, const char **names, float *ms_out)
{
    if (!names || !ms_out || cap <= 0) return fail(MS_ERR_INVALID, "ms_stitch_timed: need output arrays");
    int n = 0;
    const int e = stitch_impl(c, n_frames, views, out8u, out16s, as_stream(stream), cap, names, ms_out, &n);
    return e ? e : n;
}

// ---- the reference's call shape: stitch_online(view) x N, then blend (timed.cpp:56-152) ------------------------------------
// feed_online's per-view work (remap, gain, pyramids, accumulate) is not run per view here: the compositor batches all views of a
// frame into the same launches, so ms_feed only records the view and ms_blend runs the frame.  The images must stay valid (and
// unmodified) until ms_blend returns -- they do in the reference's stitch_one, whose full_imgs outlive the blend call.
int ms_feed(ms_ctx *c, int view, const ms_image *img, ms_stream)
{
    if (int e = ctx_check_view(c, view)) return e;
    MS_CHECK(img && img->data && img->type == MS_8UC3 && img->rows == c->cfg.src_height && img->cols == c->cfg.src_width,
             "ms_feed: view %d must be 8UC3 %dx%d", view, c->cfg.src_width, c->cfg.src_height);
    c->fed[view] = *img;
    c->fed_mask |= 1u << view;
    return MS_OK;
}

int ms_blend(ms_ctx *c, ms_image *out8u, ms_image *out16s, ms_stream stream)
{
    if (!c) return fail(MS_ERR_INVALID, "null context");
    const unsigned all = (c->N >= 32) ? 0xffffffffu : ((1u << c->N) - 1u);
    if ((c->fed_mask & all) != all) return fail(MS_ERR_STATE, "ms_blend: only views 0x%x of 0x%x were fed since the last blend", c->fed_mask, all);
    c->fed_mask = 0;
    return stitch_impl(c, 1, c->fed, out8u, out16s, as_stream(stream), 0, nullptr, nullptr, nullptr);
}

// ---- view sharding: partial sums on every rank, finish on the sink (BASELINE configs[4]) -------------
size_t ms_partial_bytes(const ms_ctx *c)
{
    return (c && c->blender_ready) ? (size_t)c->pacc_stride * sizeof(int16_t) : 0;
}

int ms_stitch_partial(ms_ctx *c, int n_frames, const ms_image *views, void *partial_out, ms_stream stream)
{
    MS_CHECK(c && partial_out && ((uintptr_t)partial_out & 15) == 0, "ms_stitch_partial: 16-byte aligned output buffer required");
    ShardArgs S{};
    S.mode = 1;
    S.pout = (int16_t *)partial_out;
    return stitch_impl(c, n_frames, views, nullptr, nullptr, as_stream(stream), 0, nullptr, nullptr, nullptr, S);
}

int ms_stitch_finish(ms_ctx *c, int n_frames, const void *const *partials, int n_partials, ms_image *out8u, ms_image *out16s, ms_stream stream)
{
    MS_CHECK(c && partials && n_partials >= 1 && n_partials <= 4, "ms_stitch_finish: 1..4 partial buffers required");
    ShardArgs S{};
    S.mode = 2;
    S.n_parts = n_partials;
    for (int i = 0; i < n_partials; ++i) {
        MS_CHECK(partials[i] && ((uintptr_t)partials[i] & 15) == 0, "ms_stitch_finish: partial %d must be a 16-byte aligned device buffer", i);
        S.part[i] = (const int16_t *)partials[i];
    }
    return stitch_impl(c, n_frames, nullptr, out8u, out16s, as_stream(stream), 0, nullptr, nullptr, nullptr, S);
}

// ---- read-back -----------------------------------------------------------------------------------
int ms_get_view_geom(const ms_ctx *c, int view, ms_view_geom *g)
{
    if (int e = ctx_check_view(c, view)) return e;
    if (!c->maps_built) return fail(MS_ERR_STATE, "ms_get_view_geom: call ms_build_maps first");
    MS_CHECK(g, "null output");
    g->roi = c->roi[view];
    const ViewPad &p = c->pad[view];
    g->top = p.top; g->left = p.left; g->bottom = p.bottom; g->right = p.right;
    g->x_tl = p.x_tl; g->y_tl = p.y_tl; g->x_br = p.x_br; g->y_br = p.y_br;
    return MS_OK;
}

int ms_get_pano_geom(const ms_ctx *c, ms_pano_geom *g)
{
    if (!c || !g) return fail(MS_ERR_INVALID, "null argument");
    if (!c->maps_built) return fail(MS_ERR_STATE, "ms_get_pano_geom: call ms_build_maps first");
    g->num_bands = c->bg.num_bands; g->dst_roi_final = c->bg.dst_roi_final; g->dst_roi = c->bg.dst_roi;
    g->canvas_x = c->canvas_x; g->canvas_y = c->canvas_y;
    return MS_OK;
}

int ms_get_maps(const ms_ctx *c, int view, ms_image *xm, ms_image *ym)
{
    if (int e = ctx_check_view(c, view)) return e;
    if (!c->maps_built) return fail(MS_ERR_STATE, "ms_get_maps: call ms_build_maps first");
    if (xm) *xm = view_map_image(c, view, 0);
    if (ym) *ym = view_map_image(c, view, 1);
    return MS_OK;
}

int ms_get_mask(const ms_ctx *c, int view, ms_image *m)
{
    if (int e = ctx_check_view(c, view)) return e;
    if (!c->masks_built) return fail(MS_ERR_STATE, "ms_get_mask: masks not built");
    MS_CHECK(m, "null output");
    *m = ms_image{(uint8_t *)c->masks.p + c->mask_off[view], (size_t)c->roi[view].width, c->roi[view].width, c->roi[view].height, MS_8UC1};
    return MS_OK;
}

int ms_get_weight_level(const ms_ctx *c, int view, int level, ms_image *w)
{
    if (int e = ctx_check_view(c, view)) return e;
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_get_weight_level: call ms_init_blender first");
    MS_CHECK(w && level >= 0 && level <= c->pano.nb, "ms_get_weight_level: bad level %d", level);
    int active; bool wait;                                            // the pair is swapped by the recalibration thread under mesh_mu
    { std::lock_guard<std::mutex> mk(const_cast<ms_ctx *>(c)->mesh_mu); active = c->tab_active; wait = c->tab_wait; }
    if (wait) MS_HIP(hipEventSynchronize(c->tab_ready));              // an enqueue-only ms_update_mask may still be filling the active copy
    const LevelDesc &L = (active == 1 ? c->alt.h_views : c->h_views)[view].lv[level];
    *w = ms_image{(void *)L.wgt, (size_t)L.wpitch * sizeof(float), L.w, L.h, MS_32FC1};
    return MS_OK;
}

int ms_get_mesh_maps(const ms_ctx *c, int view, ms_image *xm, ms_image *ym)
{
    if (int e = ctx_check_view(c, view)) return e;
    if (!c->blender_ready || !c->cfg.enable_cpw || !c->mesh_set[view]) return fail(MS_ERR_STATE, "ms_get_mesh_maps: no mesh set for view %d", view);
    if (c->mesh_ready[view]) { ms_ctx *mc_ = const_cast<ms_ctx *>(c); hipEvent_t rdy_; { std::lock_guard<std::mutex> mk_(mc_->mesh_mu); rdy_ = mesh_ready_event(mc_, view); } if (rdy_) MS_HIP(hipEventSynchronize(rdy_)); }       // updates are asynchronous: the maps are final after this
    if (xm) *xm = mesh_image(c, c->mesh_active[view], view, 0);
    if (ym) *ym = mesh_image(c, c->mesh_active[view], view, 1);
    return MS_OK;
}

int ms_get_result_mask(ms_ctx *c, ms_image *m)
{
    if (!c || !m) return fail(MS_ERR_INVALID, "null argument");
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_get_result_mask: call ms_init_blender first");
    int active; bool wait;
    { std::lock_guard<std::mutex> mk(c->mesh_mu); active = c->tab_active; wait = c->tab_wait; }
    if (wait) MS_HIP(hipEventSynchronize(c->tab_ready));
    *m = ms_image{active == 1 ? c->alt.result_mask.p : c->result_mask.p, (size_t)c->pano.mask_pitch, c->pano.fw, c->pano.fh, MS_8UC1};
    return MS_OK;
}

// ---- calibration tables as a blob (the reference recomputes its calibration at every start, timed.cpp:553) ---------------------
// What a context's static tables are derived FROM: the configuration, per view the camera (K, R), the exposure gain and the blend mask, and the blender kind.  Every
// table the per-frame kernels read (projection tables, weight pyramids, weight sums, owner maps, work lists) is a deterministic function of these on one build of
// the library, so the blob stores the inputs and ms_load_tables replays ms_create / ms_set_camera / ms_set_gain / ms_build_maps / ms_set_mask / ms_init_blender
// (or ms_init_feather).  A start-up then skips the seam-scale calibration (stitch_calib), and every rank of a multi-GPU run builds identical tables from one
// blob.  CPW meshes are run-time state (ms_set_meshes) and are not part of it.
namespace {
struct TablesHeader {
    char magic[8];            // "MSTBL02" (round 5: the checksum covers the header too; blobs of sharded contexts are refused)
    unsigned header_bytes, config_bytes;
    int n_views, blender_kind; // 0 = multiband / plain masks (ms_init_blender), 1 = FeatherBlender weights (ms_init_feather)
    float feather_sharpness;
    unsigned reserved;
    unsigned long long total_bytes, checksum;      // FNV-1a over the WHOLE blob with this field zero: a flipped bit in the embedded ms_config (projection, bands, out size ..)
                                                   // would otherwise pass ms_create's range checks and silently build other tables
    ms_config cfg;
};
struct TablesView { float K[9], R[9]; double gain; int aw, ah, roi_x, roi_y; };
unsigned long long fnv1a(const unsigned char *p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
}  // namespace

int ms_save_tables(ms_ctx *c, void *buf, size_t cap, size_t *bytes_out)
{
    if (!c || !bytes_out) return fail(MS_ERR_INVALID, "ms_save_tables: null argument");
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_save_tables: call ms_init_blender / ms_init_feather first");
    // a blob replays its context's configuration verbatim: one of a column / view shard would give EVERY loader that shard's window.  The unsharded
    // context is what "one blob for every rank" means; a rank that needs a shard creates it from the same K / R / gains / masks itself.
    if (c->cfg.col_shards > 1 || c->cfg.view_shards > 1) return fail(MS_ERR_UNSUPPORTED, "ms_save_tables: this context is a column / view shard; save the tables of an unsharded context");
    size_t need = sizeof(TablesHeader);
    for (int v = 0; v < c->N; ++v) need += sizeof(TablesView) + (size_t)c->roi[v].width * c->roi[v].height;
    *bytes_out = need;
    if (!buf) return MS_OK;                          // size query
    MS_CHECK(cap >= need, "ms_save_tables: buffer of %zu bytes, %zu needed", cap, need);
    std::lock_guard<std::recursive_mutex> tables_lk(c->tables_mu);
    unsigned char *o = static_cast<unsigned char *>(buf);
    TablesHeader h{};
    memcpy(h.magic, "MSTBL02", 8);
    h.header_bytes = (unsigned)sizeof(TablesHeader); h.config_bytes = (unsigned)sizeof(ms_config);
    h.n_views = c->N; h.blender_kind = c->feather_sharpness >= 0.f ? 1 : 0; h.feather_sharpness = c->feather_sharpness;
    h.total_bytes = need; h.cfg = c->cfg;
    size_t at = sizeof(TablesHeader);
    for (int v = 0; v < c->N; ++v) {
        TablesView tv{};
        memcpy(tv.K, c->K[v], sizeof(tv.K)); memcpy(tv.R, c->R[v], sizeof(tv.R));
        tv.gain = c->gain[v]; tv.aw = c->roi[v].width; tv.ah = c->roi[v].height; tv.roi_x = c->roi[v].x; tv.roi_y = c->roi[v].y;
        memcpy(o + at, &tv, sizeof(tv)); at += sizeof(tv);
        // the mask ms_init_blender was given (the caller's / the seam finder's; a re-warp through a CPW mesh is run-time state like the mesh itself)
        MS_HIP(hipMemcpy(o + at, (const uint8_t *)c->masks.p + c->mask_off[v], (size_t)tv.aw * tv.ah, hipMemcpyDeviceToHost));
        at += (size_t)tv.aw * tv.ah;
    }
    h.checksum = 0;
    memcpy(o, &h, sizeof(h));
    h.checksum = fnv1a(o, need);
    memcpy(o, &h, sizeof(h));
    return MS_OK;
}

int ms_load_tables(const void *buf, size_t bytes, ms_ctx **out, ms_stream stream)
{
    if (!buf || !out) return fail(MS_ERR_INVALID, "ms_load_tables: null argument");
    *out = nullptr;
    const unsigned char *in = static_cast<const unsigned char *>(buf);
    TablesHeader h;
    MS_CHECK(bytes >= sizeof(h), "ms_load_tables: %zu bytes is not a table blob", bytes);
    memcpy(&h, in, sizeof(h));
    MS_CHECK(memcmp(h.magic, "MSTBL02", 8) == 0, "ms_load_tables: not a table blob of this library (bad magic / older format)");
    MS_CHECK(h.header_bytes == sizeof(TablesHeader) && h.config_bytes == sizeof(ms_config) && h.cfg.struct_size == sizeof(ms_config),
             "ms_load_tables: the blob was written by a library with another ms_config / header layout");
    MS_CHECK(h.total_bytes == bytes && h.n_views >= 1 && h.n_views <= MAX_VIEWS && h.n_views == h.cfg.num_views, "ms_load_tables: truncated or inconsistent blob");
    {      // over the whole blob with the checksum field zero (header and embedded configuration included)
        std::vector<unsigned char> tmp(in, in + bytes);
        TablesHeader z = h;
        z.checksum = 0;
        memcpy(tmp.data(), &z, sizeof(z));
        MS_CHECK(fnv1a(tmp.data(), bytes) == h.checksum, "ms_load_tables: checksum mismatch (corrupt blob)");
    }
    MS_CHECK(h.cfg.col_shards <= 1 && h.cfg.view_shards <= 1, "ms_load_tables: the blob describes a column / view shard");
    size_t at = sizeof(h);
    for (int v = 0; v < h.n_views; ++v) {            // structure check before anything touches the device
        TablesView tv;
        MS_CHECK(at + sizeof(tv) <= bytes, "ms_load_tables: truncated blob");
        memcpy(&tv, in + at, sizeof(tv));
        MS_CHECK(tv.aw > 0 && tv.ah > 0 && at + sizeof(tv) + (size_t)tv.aw * tv.ah <= bytes, "ms_load_tables: truncated blob");
        at += sizeof(tv) + (size_t)tv.aw * tv.ah;
    }
    MS_CHECK(at == bytes, "ms_load_tables: trailing bytes");
    ms_ctx *c = nullptr;
    if (int e = ms_create(&h.cfg, &c)) return e;
    int err = MS_OK;
    at = sizeof(h);
    std::vector<size_t> mask_at((size_t)h.n_views);
    std::vector<TablesView> tvs((size_t)h.n_views);
    for (int v = 0; v < h.n_views && !err; ++v) {
        memcpy(&tvs[v], in + at, sizeof(TablesView));
        mask_at[v] = at + sizeof(TablesView);
        at = mask_at[v] + (size_t)tvs[v].aw * tvs[v].ah;
        err = ms_set_camera(c, v, tvs[v].K, tvs[v].R);
        if (!err) err = ms_set_gain(c, v, tvs[v].gain);
    }
    if (!err) err = ms_build_maps(c, stream);
    for (int v = 0; v < h.n_views && !err; ++v) {
        // the warped-view rectangles are recomputed from K, R: a blob whose masks do not fit them comes from a library whose geometry differs
        if (c->roi[v].width != tvs[v].aw || c->roi[v].height != tvs[v].ah || c->roi[v].x != tvs[v].roi_x || c->roi[v].y != tvs[v].roi_y)
            err = fail(MS_ERR_INVALID, "ms_load_tables: view %d warps to %dx%d at (%d,%d) here, the blob holds %dx%d at (%d,%d)", v, c->roi[v].width, c->roi[v].height,
                       c->roi[v].x, c->roi[v].y, tvs[v].aw, tvs[v].ah, tvs[v].roi_x, tvs[v].roi_y);
        else err = ms_set_mask(c, v, in + mask_at[v], (size_t)tvs[v].aw);
    }
    if (!err) err = h.blender_kind == 1 ? ms_init_feather(c, h.feather_sharpness, stream) : ms_init_blender(c, stream);
    if (err) { ms_destroy(c); return err; }
    *out = c;
    return MS_OK;
}

int ms_get_band_cells(ms_ctx *c, int level, unsigned *owned, unsigned *exclusive, unsigned *general)
{
    if (!c || !owned || !exclusive || !general) return fail(MS_ERR_INVALID, "null argument");
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_get_band_cells: call ms_init_blender first");
    MS_CHECK(level >= 0 && level < c->pano.nb, "ms_get_band_cells: band %d not in [0, %d)", level, c->pano.nb);
    int active; bool wait;
    { std::lock_guard<std::mutex> mk(c->mesh_mu); active = c->tab_active; wait = c->tab_wait; }
    if (wait) MS_HIP(hipEventSynchronize(c->tab_ready));
    const int pw_ = div_up(c->pano.qw[level], 64), ph_ = div_up(c->pano.qh[level], 16);
    *owned = *exclusive = 0; *general = (unsigned)(pw_ * ph_);
    if (!c->pure_off[level]) return MS_OK;
    std::vector<uint8_t> h((size_t)pw_ * ph_);
    const uint8_t *src = (const uint8_t *)(active == 1 ? c->alt.pure_maps.p : c->pure_maps.p) + (c->pure_off[level] - 1);
    MS_HIP(hipMemcpy(h.data(), src, h.size(), hipMemcpyDeviceToHost));
    *general = 0;
    for (uint8_t b : h) { if (b == 255) ++*general; else if (b == 254) ++*exclusive; else ++*owned; }
    return MS_OK;
}

int ms_get_stitch_kernels(ms_ctx *c, int *warp_kernel, int *stage1_kernel)
{
    if (!c || !warp_kernel || !stage1_kernel) return fail(MS_ERR_INVALID, "null argument");
    *warp_kernel = c->last_warp_kernel; *stage1_kernel = c->last_stage1_kernel;
    return MS_OK;
}

int ms_get_plan_stats(ms_ctx *c, ms_plan_stats *out)
{
    if (!c || !out) return fail(MS_ERR_INVALID, "null argument");
    if (!c->blender_ready) return fail(MS_ERR_STATE, "ms_get_plan_stats: call ms_init_blender first");
    MS_CHECK(out->struct_size == sizeof(ms_plan_stats), "ms_get_plan_stats: struct_size %u, this library's ms_plan_stats has %zu bytes", out->struct_size, sizeof(ms_plan_stats));
    ms_plan_stats s{};
    s.struct_size = (unsigned)sizeof(s);
    s.warp_tile_w = WARP_TW; s.warp_tile_h = WARP_TH; s.n_warp_tiles = c->n_warp_tiles;
    s.n_stage1_tiles = c->n_stage1_tiles; s.n_stage1_reachable = c->n_stage1_reachable;
    s.down_tile_w = DOWN_TW; s.down_tile_h = DOWN_TH; s.blend_tile_w = BLEND_TW; s.blend_tile_h = BLEND_TH;
    for (int l = 0; l < 8 && l < MAX_LEVELS; ++l) { s.n_down_tiles[l] = c->n_down_tiles[l]; s.n_blend_tiles[l] = c->n_blend_tiles[l]; }
    *out = s;
    return MS_OK;
}

static int calib_grid()
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return 16 * cus;
}

int ms_calib_copy(const void *src, void *dst, size_t bytes, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(src && dst && bytes >= 16 && bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "ms_calib_copy: 16-byte aligned buffers and size required");
    k_calib_copy<<<calib_grid(), 256, 0, as_stream(stream)>>>((const u32x4_t *)src, (u32x4_t *)dst, bytes / 16);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

int ms_calib_read(const void *src, size_t bytes, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(src && bytes >= 16 && bytes % 16 == 0 && ((uintptr_t)src & 15) == 0, "ms_calib_read: 16-byte aligned buffer and size required");
    unsigned *sink = (unsigned *)device_scratch().get(16);
    if (!sink) return fail(MS_ERR_NOMEM, "ms_calib_read: no device scratch");
    k_calib_read<<<calib_grid(), 256, 0, as_stream(stream)>>>((const u32x4_t *)src, sink, bytes / 16);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

int ms_calib_shape(void *buf, size_t bytes, int shape, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(buf && bytes >= 4096 && bytes % 128 == 0 && ((uintptr_t)buf & 127) == 0 && shape >= 0 && shape <= 2, "ms_calib_shape: 128-byte aligned buffer and size, shape 0..2");
    unsigned *sink = (unsigned *)device_scratch().get(16);
    if (!sink) return fail(MS_ERR_NOMEM, "ms_calib_shape: no device scratch");
    if (shape == 0) k_calib_shape<0><<<dim3(calib_grid()), 256, 0, as_stream(stream)>>>((const uint8_t *)buf, (uint8_t *)buf, sink, bytes);
    else if (shape == 1) k_calib_shape<1><<<dim3(calib_grid()), 256, 0, as_stream(stream)>>>((const uint8_t *)buf, (uint8_t *)buf, sink, bytes);
    else k_calib_shape<2><<<dim3(calib_grid(), 4), 256, 0, as_stream(stream)>>>((const uint8_t *)buf, (uint8_t *)buf, sink, bytes);
    MS_LAUNCH_CHECK();
    return MS_OK;
}

int ms_selftest_cvt_u8(unsigned long long *mismatches_out, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(mismatches_out != nullptr, "ms_selftest_cvt_u8: null output");
    hipStream_t st = as_stream(stream);
    DevBuf cnt;
    if (int e = cnt.alloc(sizeof(unsigned long long))) return e;
    MS_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), st));
    k_selftest_cvt_u8<<<65536, 256, 0, st>>>((unsigned long long *)cnt.p);
    MS_LAUNCH_CHECK();
    MS_HIP(hipMemcpyAsync(mismatches_out, cnt.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    MS_HIP(hipStreamSynchronize(st));
    cnt.release();
    return MS_OK;
}

int ms_selftest_divide(const float *dens_host, int n, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(dens_host && n > 0 && n <= 65535, "ms_selftest_divide: bad arguments");
    hipStream_t st = as_stream(stream);
    DevBuf d, cnt;
    if (int e = d.alloc((size_t)n * sizeof(float))) return e;
    if (int e = cnt.alloc(sizeof(unsigned))) return e;
    MS_HIP(hipMemcpyAsync(d.p, dens_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    MS_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned), st));
    k_selftest_divide<<<dim3(256, n), 256, 0, st>>>((const float *)d.p, n, (unsigned *)cnt.p);
    MS_LAUNCH_CHECK();
    unsigned h = 0;
    MS_HIP(hipMemcpyAsync(&h, cnt.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    MS_HIP(hipStreamSynchronize(st));
    d.release(); cnt.release();
    return (int)h;
}

int ms_selftest_divide_range(float d_lo, float d_hi, unsigned long long *mismatches_out, unsigned long long *checked_out, ms_stream stream)
{
    if (int e = require_device()) return e;
    MS_CHECK(mismatches_out && d_lo > 0.f && d_hi >= d_lo && d_hi < 3.0e38f, "ms_selftest_divide_range: needs 0 < d_lo <= d_hi (finite) and an output");
    hipStream_t st = as_stream(stream);
    DevBuf cnt;
    if (int e = cnt.alloc(sizeof(unsigned long long))) return e;
    MS_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), st));
    unsigned b0, b1;
    memcpy(&b0, &d_lo, 4); memcpy(&b1, &d_hi, 4);
    unsigned long long total = 0;
    for (unsigned long long b = b0; b <= b1; b += (1u << 22)) {          // launches of <= 4 M denominators (a fraction of a second each)
        const unsigned n = (unsigned)std::min<unsigned long long>(1u << 22, (unsigned long long)b1 - b + 1);
        k_selftest_divide_range<<<(n + 255) / 256, 256, 0, st>>>((unsigned)b, n, (unsigned long long *)cnt.p);
        MS_LAUNCH_CHECK();
        total += n;
    }
    unsigned long long h = 0;
    MS_HIP(hipMemcpyAsync(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, st));
    MS_HIP(hipStreamSynchronize(st));
    cnt.release();
    *mismatches_out = h;
    if (checked_out) *checked_out = total * 65536ull;
    return MS_OK;
}

}  // extern "C"
