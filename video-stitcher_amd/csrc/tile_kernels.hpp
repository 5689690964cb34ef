// tile_kernels.hpp -- the work-list driven per-frame kernels (included by compositor.hip).
//
// Same arithmetic as the simple kernels in compositor.hip (k_warp / k_down / k_blend), restructured for
// CDNA4: one workgroup per NEEDED tile (no empty waves, zero-weight regions never touched), 4-8 pixels
// per lane so that every global access is a dword or wider, and all loads of a lane issued before the
// first use so a wave pays one memory round trip instead of one per tap/row.
#pragma once
#include "descs.hpp"

namespace ms {

// BORDER_REFLECT without the modulo when -len <= i < 2*len (always true for the blender's borders)
__device__ __forceinline__ int reflect_fast(int i, int len)
{
    const int last = len - 1;
    if (i >= -len && i < 2 * len) return i < 0 ? -i - 1 : (i > last ? 2 * last - i + 1 : i);
    return reflect_idx(i, len);
}

struct Taps {            // one bilinear sample: clamped tap origin, weights, fast-path flag
    int x1, y1;
    float w11, w12, w21, w22;
    bool fast;
};
__device__ __forceinline__ Taps make_taps(float xc, float yc, int srows, int scols)
{
    Taps t;
    t.x1 = f2i_rd(xc); t.y1 = f2i_rd(yc);
    const int x2 = (int)((unsigned)t.x1 + 1u), y2 = (int)((unsigned)t.y1 + 1u);
    const float wx2 = (float)x2 - xc, wx1 = xc - (float)t.x1;
    const float wy2 = (float)y2 - yc, wy1 = yc - (float)t.y1;
    t.w11 = wx2 * wy2; t.w12 = wx1 * wy2; t.w21 = wx2 * wy1; t.w22 = wx1 * wy1;
    t.fast = t.x1 >= 0 && t.x1 < scols - 1 && t.y1 >= 0 && t.y1 < srows - 1;
    return t;
}
// 6 source bytes (two BGR pixels) of one row as a dword + a halfword, from a possibly unaligned address
struct Px2 { unsigned lo; unsigned hi; };
__device__ __forceinline__ Px2 load_px2(const uint8_t *p)
{
    Px2 r;
    unsigned a; uint16_t b;
    __builtin_memcpy(&a, p, 4);
    __builtin_memcpy(&b, p + 4, 2);
    r.lo = a; r.hi = b;
    return r;
}
__device__ __forceinline__ void blend_taps(const Taps &t, const Px2 &r1, const Px2 &r2, float out[3])
{
    // bytes: lo = B1 G1 R1 B2, hi = G2 R2
    const float s11[3] = {(float)(r1.lo & 0xff), (float)((r1.lo >> 8) & 0xff), (float)((r1.lo >> 16) & 0xff)};
    const float s12[3] = {(float)(r1.lo >> 24), (float)(r1.hi & 0xff), (float)((r1.hi >> 8) & 0xff)};
    const float s21[3] = {(float)(r2.lo & 0xff), (float)((r2.lo >> 8) & 0xff), (float)((r2.lo >> 16) & 0xff)};
    const float s22[3] = {(float)(r2.lo >> 24), (float)(r2.hi & 0xff), (float)((r2.hi >> 8) & 0xff)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o = __builtin_fmaf(s11[c], t.w11, 0.f);
        o = __builtin_fmaf(s12[c], t.w12, o);
        o = __builtin_fmaf(s21[c], t.w21, o);
        o = __builtin_fmaf(s22[c], t.w22, o);
        out[c] = o;
    }
}

// ---- Gaussian level 0 (remap + gain [or CPW stage 2] + reflect pad), 4 px per lane ---------------------
template <bool CPW>
__global__ void __launch_bounds__(256) k_warp_t(const WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int n_views,
                                                SrcTable src, int src_rows, int src_cols, MeshTable mesh,
                                                const uint8_t *__restrict__ stage, long long stage_stride,
                                                uint8_t *__restrict__ g0, long long g0_stride)
{
    const WarpTile T = tiles[blockIdx.x];
    const int f = blockIdx.z, v = T.view;
    const ViewDesc &V = views[v];
    const int x = T.x0 + 4 * (int)threadIdx.x, y = T.y0 + (int)threadIdx.y;
    if (x >= V.pw || y >= V.ph) return;
    const int ay = reflect_fast(y - V.top, V.ah);
    const int i0 = x - V.left;
    const float *mxp, *myp;
    int mpitch;
    const uint8_t *sp;
    unsigned sstep;
    int srows, scols;
    if (CPW) {
        mxp = mesh.x[v]; myp = mesh.y[v]; mpitch = mesh.pitch[v];
        sp = stage + (size_t)f * stage_stride + V.s1_off; sstep = (unsigned)V.s1_pitch; srows = V.ah; scols = V.aw;
    } else {
        mxp = V.xmap; myp = V.ymap; mpitch = V.map_pitch;
        sp = src.p[f * n_views + v]; sstep = src.step[f * n_views + v]; srows = src_rows; scols = src_cols;
    }
    float xc[4], yc[4];
    if (i0 >= 0 && i0 + 3 < V.aw) {                 // interior: 4 consecutive map entries (dword-aligned 16-byte loads)
        float4 a, b;
        __builtin_memcpy(&a, mxp + (size_t)ay * mpitch + i0, 16);
        __builtin_memcpy(&b, myp + (size_t)ay * mpitch + i0, 16);
        xc[0] = a.x; xc[1] = a.y; xc[2] = a.z; xc[3] = a.w;
        yc[0] = b.x; yc[1] = b.y; yc[2] = b.z; yc[3] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ax = reflect_fast(i0 + k, V.aw);
            xc[k] = mxp[(size_t)ay * mpitch + ax];
            yc[k] = myp[(size_t)ay * mpitch + ax];
        }
    }
    Taps t[4];
    Px2 r1[4], r2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        t[k] = make_taps(xc[k], yc[k], srows, scols);
        const int xs = min(max(t[k].x1, 0), scols - 2), ys = min(max(t[k].y1, 0), srows - 2);
        const uint8_t *p = sp + (size_t)ys * sstep + (size_t)xs * 3;
        r1[k] = load_px2(p);
        r2[k] = load_px2(p + sstep);
    }
    unsigned packed[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o[3];
        if (t[k].fast) blend_taps(t[k], r1[k], r2[k], o);
        else sample3(sp, sstep, srows, scols, xc[k], yc[k], o);      // image-edge / invalid coordinates: per-tap bounds
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned val = CPW ? (unsigned)sat_u8(o[c]) : (unsigned)sat_u8(__builtin_fmaf(V.gain, (float)sat_u8(o[c]), 0.f));
            packed[c] |= val << (8 * k);
        }
    }
    const LevelDesc &L = V.lv[0];
    uint8_t *d = g0 + (size_t)f * g0_stride + L.off + (size_t)y * L.pitch + x;
    const size_t plane = (size_t)L.h * L.pitch;
    *reinterpret_cast<unsigned *>(d) = packed[0];
    *reinterpret_cast<unsigned *>(d + plane) = packed[1];
    *reinterpret_cast<unsigned *>(d + 2 * plane) = packed[2];
}

// ---- pyrDown, tile list, 2 rows x 4 cols per lane (block 32 x 8) -------------------------------------
template <typename TIN>
__global__ void __launch_bounds__(256) k_down_t(const DownTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int l,
                                                const TIN *__restrict__ gin, long long in_stride,
                                                int16_t *__restrict__ gout, long long out_stride)
{
    const DownTile T = tiles[blockIdx.x];
    const int c = blockIdx.y, f = blockIdx.z, v = T.view;
    const LevelDesc &Li = views[v].lv[l], &Lo = views[v].lv[l + 1];
    const int t = (T.x0 >> 2) + (int)threadIdx.x;
    const int y = T.y0 + 2 * (int)threadIdx.y;
    if (4 * t >= Lo.w || y >= Lo.h) return;
    const TIN *in = gin + (size_t)f * in_stride + Li.off + (size_t)c * Li.h * Li.pitch;
    const bool two = (y + 1) < Lo.h;
    const int sy = 2 * y, last = Li.h - 1;
    int ridx[7];
    ridx[0] = abs(sy - 2); ridx[1] = abs(sy - 1); ridx[2] = sy;
#pragma unroll
    for (int j = 3; j < 7; ++j) { const int r = sy + j - 2; ridx[j] = r > last ? 2 * last - r : r; }
    if (!two) { ridx[5] = ridx[4]; ridx[6] = ridx[4]; }
    typename Row11<TIN>::type raw[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) raw[j] = fetch_row11(in + (size_t)ridx[j] * Li.pitch, t, Li.w);
    int V0[11], V1[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) V0[k] = V1[k] = 0;
    const int w0[7] = {1, 4, 6, 4, 1, 0, 0}, w1[7] = {0, 0, 1, 4, 6, 4, 1};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int r[11];
        unpack_row11(raw[j], t, Li.w, r);
#pragma unroll
        for (int k = 0; k < 11; ++k) { V0[k] += w0[j] * r[k]; V1[k] += w1[j] * r[k]; }
    }
    int16_t *out = gout + (size_t)f * out_stride + Lo.off + (size_t)c * Lo.h * Lo.pitch + (size_t)y * Lo.pitch + 4 * t;
    auto emit = [&](const int *V, int16_t *dst) {
        unsigned o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = (unsigned)(uint16_t)sat_s16(rne_shift(V[2 * i] + 4 * V[2 * i + 1] + 6 * V[2 * i + 2] + 4 * V[2 * i + 3] + V[2 * i + 4], 8));
        *reinterpret_cast<uint2 *>(dst) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
    };
    emit(V0, out);
    if (two) emit(V1, out + Lo.pitch);
}

}  // namespace ms
