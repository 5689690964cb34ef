// tile_kernels.hpp -- the work-list driven per-frame kernels (included by compositor.hip).
//
// Same arithmetic as the simple kernels in compositor.hip (k_warp / k_down / k_blend), restructured for
// CDNA4: one workgroup per NEEDED tile (no empty waves, zero-weight regions never touched), 4-8 pixels
// per lane so that every global access is a dword or wider, and all loads of a lane issued before the
// first use so a wave pays one memory round trip instead of one per tap/row.
#pragma once
#include "descs.hpp"
// Wave priority around the two phases of a gather kernel (s_setprio; session 2 of round 5, profiles/r05_experiments.txt).  Mode 1: a wave that is building addresses and issuing its
// reads runs at priority 3 and drops to 0 for the arithmetic on what came back, so that a new wave gets its reads out past the older waves' long blends instead of taking
// turns with them (the SIMD's arbiter is oldest-first among equal priorities); 3: raised only until the first reads are out; 2: the opposite of 1; 0: no instruction.
// Measured per kernel, same box, alternating: the projection warp 389 -> 376 us per 32 frames with 1 (378 with 3, 389 with 2) -- adopted; the CPW mesh remap +3.4 % and the
// first CPW remap +0 % (config 3) / +3.8 % (shipped rig) SLOWER with 1 -- not bound by misses in flight, the raised waves only delay the stores of the older ones; the level-0
// band kernel -0.8 %, the level-0 reduce +4 % (the scheduling barrier the switch needs splits its load clause), the NV12-sampling warp +2 %.  So: MS_PRIO_WARP applies to the
// projection warp in its aligned shared-offset form alone (a second box: 362-368 -> 356-357 us; config 5's stronger minification: neutral).
#ifndef MS_PRIO_WARP
#define MS_PRIO_WARP 1
#endif
#ifndef MS_PRIO_CPW
#define MS_PRIO_CPW 0
#endif
#ifndef MS_PRIO_S1
#define MS_PRIO_S1 0
#endif
#ifndef MS_PRIO_NV12
#define MS_PRIO_NV12 0
#endif
#ifndef MS_PRIO_BLEND
#define MS_PRIO_BLEND 0
#endif
#define MS_PRIO_LOADS(K) do { if ((K) == 1 || (K) == 3) __builtin_amdgcn_s_setprio(3); else if ((K) == 2) __builtin_amdgcn_s_setprio(0); } while (0)
#define MS_PRIO_RELOADS(K) do { if ((K) == 1) __builtin_amdgcn_s_setprio(3); else if ((K) == 2) __builtin_amdgcn_s_setprio(0); } while (0)
#define MS_PRIO_MATH(K)  do { if ((K) == 1 || (K) == 3) __builtin_amdgcn_s_setprio(0); else if ((K) == 2) __builtin_amdgcn_s_setprio(3); } while (0)

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
    const float fx1 = __builtin_floorf(xc), fy1 = __builtin_floorf(yc);
    t.x1 = (int)fx1; t.y1 = (int)fy1;
    // all 4 taps inside and the 8-byte row reads inside (one unsigned compare per axis)
    t.fast = (unsigned)t.x1 < (unsigned)(scols - 2) && (unsigned)t.y1 < (unsigned)(srows - 1);
    // (float)(x1 + 1) == floor(xc) + 1 and (float)x1 == floor(xc) exactly while |xc| < 2^24; beyond that every tap is outside
    // the image, all four samples are 0 and any finite weight gives 0 (an infinite / NaN one gives NaN -> saturate_cast -> 0)
    const float wx2 = (fx1 + 1.f) - xc, wx1 = xc - fx1;
    const float wy2 = (fy1 + 1.f) - yc, wy1 = yc - fy1;
    t.w11 = wx2 * wy2; t.w12 = wx1 * wy2; t.w21 = wx2 * wy1; t.w22 = wx1 * wy1;
    return t;
}
// The two BGR pixels of a tap row (6 bytes) fetched as ONE unaligned 8-byte load: the TA/L1 cost of a wave-wide
// gather is per instruction, not per byte.  The 2 extra bytes stay inside the image row because the caller only
// uses this for taps with x1 <= cols-3 (the last column pair goes through the bounds-checked path).
struct Px2 { unsigned lo; unsigned hi; };
// (Both tap reads name the GLOBAL address space: the source pointers come out of a by-value table, or through an integer mask, so the compiler cannot tell -- and
// FLAT loads cost full 64-bit VGPR addresses, count on lgkmcnt too, and can only be waited for all together: see assume_global in common.hpp.)
typedef unsigned ms_u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned ms_u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
#define MS_GLOBAL_AS __attribute__((address_space(1)))
typedef float ms_f32x2_g __attribute__((ext_vector_type(2)));
typedef float ms_f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ float2 gload_f2(const float2 *p) { const ms_f32x2_g v = *(const MS_GLOBAL_AS ms_f32x2_g *)(uintptr_t)p; return make_float2(v.x, v.y); }
__device__ __forceinline__ float4 gload_f4(const float2 *p) { const ms_f32x4_a8 v = *(const MS_GLOBAL_AS ms_f32x4_a8 *)(uintptr_t)p; return make_float4(v.x, v.y, v.z, v.w); }   // two table entries, 8-byte aligned
// The UNALIGNED 8-byte form stays a flat load: measured, config 5 (every tile takes this form): global_load_dwordx2 433-437 us per 8 frames, flat_load_dwordx2 392-395
// (the commit before, whose scheduler order made unit 0 wait for all 16 reads: 402-408).  The aligned 12-byte form below gains from being global: config 2 218.9 -> 211.7 us per
// 16 frames, the shipped configuration's stage 1 351 -> 324 (counted waits: unit 0 is blended while unit 1's reads are still in flight).
#ifndef MS_PX2_FLAT
#define MS_PX2_FLAT 1
#endif
__device__ __forceinline__ Px2 load_px2(const uint8_t *p)
{
#if MS_PX2_FLAT
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return Px2{v.x, v.y};
#else
    const ms_u32x2_a1 v = *(const MS_GLOBAL_AS ms_u32x2_a1 *)(uintptr_t)p;
    return Px2{v.x, v.y};
#endif
}
// The same 8 bytes out of ONE dword-aligned 12-byte read (global_load_dwordx3) and two v_alignbyte_b32 by the address's low two bits: an unaligned
// 8-byte gather costs the texture-address path 13 % more than an aligned one (profiles/r02_warp_probes.txt: 242 vs 210 us), the aligned 12-byte form
// with the two extra VALU operations per tap row 223 us.  It reads up to 3 bytes before and 4 bytes after the 8-byte window: fine inside an image
// (the neighbouring pixels / the next row), NOT behind the last image row of a caller's buffer -- tiles that sample it keep the unaligned read.
struct Px3 { unsigned d0, d1, d2; };
__device__ __forceinline__ Px3 load_px3(const uint8_t *p)
{
    const ms_u32x3_a4 v = *(const MS_GLOBAL_AS ms_u32x3_a4 *)((uintptr_t)p & ~(uintptr_t)3);
    return Px3{v.x, v.y, v.z};
}
// the same read as "4-aligned uniform base + 32-bit lane offset": the address stays one VGPR (global_load_dwordx3 v, v_off, s[base:base+1]) instead of a 64-bit
// add and mask per read.  base_al = the image pointer rounded down to 4 bytes, a = (pointer & 3) + byte offset of the tap; a & 3 is the byte shift for px3_to_px2.
typedef const MS_GLOBAL_AS uint8_t *ms_gptr_u8;
__device__ __forceinline__ Px3 load_px3_at(ms_gptr_u8 base_al, unsigned a)
{
    const ms_u32x3_a4 v = *(const MS_GLOBAL_AS ms_u32x3_a4 *)(base_al + (a & ~3u));
    return Px3{v.x, v.y, v.z};
}
__device__ __forceinline__ Px2 px3_to_px2(const Px3 &q, unsigned addr_lo)      // v_alignbyte_b32 uses the low two bits of its shift operand
{
    return Px2{__builtin_amdgcn_alignbyte(q.d1, q.d0, addr_lo), __builtin_amdgcn_alignbyte(q.d2, q.d1, addr_lo)};
}
// uniform base + 32-bit lane offset: the address arithmetic stays in one VGPR (global_load ... v_off, s[base])
__device__ __forceinline__ Px2 load_px2(const uint8_t *base, unsigned off) { return load_px2(base + off); }
// Byte offset of the two tap rows of a sample, clamped so that both 8-byte reads stay inside the image: rows
// (ya, ya+1) with ya = clamp(y1, 0, rows-2), bytes [bl, bl+8) with bl = clamp(3*x1, 0, 3*cols-8).  For an interior
// sample (Taps::fast) this is exactly the tap address.
__device__ __forceinline__ unsigned tap_offset(int x1, int y1, int srows, int scols, unsigned sstep)
{
    const int ya = min(max(y1, 0), srows - 2);
    const int bl = min(3 * min(max(x1, 0), scols), 3 * scols - 8);
    return (unsigned)ya * sstep + (unsigned)bl;
}
// Samples with a tap outside the image (BORDER_CONSTANT 0, border_interpolate.hpp:698-717) reuse the clamped reads: the wanted
// 6 bytes are the loaded 8 shifted by the clamp distance, with zeros shifted in for the columns outside, and rows outside
// are zeroed / swapped.  Pure ALU, run under a wave-level branch only where a wave touches the image border.
__device__ __forceinline__ void fix_border_taps(Px2 &r1, Px2 &r2, int x1, int y1, int srows, int scols)
{
    const int x1c = min(max(x1, -3), scols + 2);                 // anything further out is all zeros anyway
    const int b = 3 * x1c, bl = min(max(b, 0), 3 * scols - 8), sh = b - bl;
    unsigned long long v1 = ((unsigned long long)r1.hi << 32) | r1.lo, v2 = ((unsigned long long)r2.hi << 32) | r2.lo;
    if (sh > 0) { const int n = 8 * min(sh, 7); v1 = sh >= 8 ? 0ull : v1 >> n; v2 = sh >= 8 ? 0ull : v2 >> n; }
    else if (sh < 0) { const int n = 8 * min(-sh, 7); v1 = sh <= -8 ? 0ull : v1 << n; v2 = sh <= -8 ? 0ull : v2 << n; }
    // loaded rows are (ya, ya+1), ya = clamp(y1, 0, rows-2)
    if (y1 == -1) { v2 = v1; v1 = 0ull; }
    else if (y1 == srows - 1) { v1 = v2; v2 = 0ull; }
    else if (y1 < -1 || y1 >= srows) { v1 = 0ull; v2 = 0ull; }
    r1.lo = (unsigned)v1; r1.hi = (unsigned)(v1 >> 32);
    r2.lo = (unsigned)v2; r2.hi = (unsigned)(v2 >> 32);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// bytes of a tap row: lo = B1 G1 R1 B2, hi = G2 R2 x x
__device__ __forceinline__ float px_ch(const Px2 &r, int tap, int c)
{
    const int b = 3 * tap + c;
    const unsigned w = b < 4 ? r.lo : r.hi;
    return (float)((w >> (8 * (b & 3))) & 0xffu);          // v_cvt_f32_ubyteN
}
__device__ __forceinline__ void blend_taps(const Taps &t, const Px2 &r1, const Px2 &r2, float out[3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o = __builtin_fmaf(px_ch(r1, 0, c), t.w11, 0.f);
        o = __builtin_fmaf(px_ch(r1, 1, c), t.w12, o);
        o = __builtin_fmaf(px_ch(r2, 0, c), t.w21, o);
        o = __builtin_fmaf(px_ch(r2, 1, c), t.w22, o);
        out[c] = o;
    }
}
// MS_PK_F32 = 0: the scalar forms, each fma kept a single v_fma_f32 (inline asm: plain -O3 would SLP-pack adjacent fmas back into v_pk_fma_f32)
#ifndef MS_PK_F32
#define MS_PK_F32 0      // measured (same box, 4 pairs of runs): k_warp_t 221.6 us packed, 219.7 us scalar per 16 frames -- v_pk_fma_f32 buys nothing per flop on this part
#endif
__device__ __forceinline__ float fma_single(float a, float b, float c)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void blend_taps_single(const Taps &t, const Px2 &r1, const Px2 &r2, float out[3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o = fma_single(px_ch(r1, 0, c), t.w11, 0.f);
        o = fma_single(px_ch(r1, 1, c), t.w12, o);
        o = fma_single(px_ch(r2, 0, c), t.w21, o);
        o = fma_single(px_ch(r2, 1, c), t.w22, o);
        out[c] = o;
    }
}
// gain * x for two values at once: fma(gain, x, 0) per half, exactly the scalar __builtin_fmaf(gain, x, 0.f)
__device__ __forceinline__ f32x2 gain_pair(float gain, float a, float b)
{
#if MS_PK_F32
    f32x2 g, x, z = {0.f, 0.f};
    g.x = gain; g.y = gain; x.x = a; x.y = b;
    return __builtin_elementwise_fma(g, x, z);
#else
    f32x2 r;
    r.x = fma_single(gain, a, 0.f); r.y = fma_single(gain, b, 0.f);
    return r;
#endif
}
// two samples at once: the same four fmas per channel, issued as v_pk_fma_f32 (two fp32 lanes per instruction)
__device__ __forceinline__ void blend_taps2(const Taps ta, const Taps tb, const Px2 r1a, const Px2 r2a, const Px2 r1b, const Px2 r2b,
                                            float oa[3], float ob[3])
{
#if !MS_PK_F32
    blend_taps_single(ta, r1a, r2a, oa);
    blend_taps_single(tb, r1b, r2b, ob);
    return;
#endif
    f32x2 w11, w12, w21, w22;
    w11.x = ta.w11; w11.y = tb.w11; w12.x = ta.w12; w12.y = tb.w12;
    w21.x = ta.w21; w21.y = tb.w21; w22.x = ta.w22; w22.y = tb.w22;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x2 s11, s12, s21, s22, o = {0.f, 0.f};
        s11.x = px_ch(r1a, 0, c); s11.y = px_ch(r1b, 0, c); s12.x = px_ch(r1a, 1, c); s12.y = px_ch(r1b, 1, c);
        s21.x = px_ch(r2a, 0, c); s21.y = px_ch(r2b, 0, c); s22.x = px_ch(r2a, 1, c); s22.y = px_ch(r2b, 1, c);
        o = __builtin_elementwise_fma(s11, w11, o);
        o = __builtin_elementwise_fma(s12, w12, o);
        o = __builtin_elementwise_fma(s21, w21, o);
        o = __builtin_elementwise_fma(s22, w22, o);
        oa[c] = o.x; ob[c] = o.y;
    }
}

// ---- Gaussian level 0 (remap + gain [or CPW stage 2] + reflect pad), 4 px per lane ---------------------
// Source coordinates of the 4 pixels of a lane.  Non-CPW: rebuilt from the 1-D tables (same fp32 ops as the dense
// x_map/y_map, which are therefore never read per frame); CPW stage 2: read from the dense mesh maps.
template <bool CPW, int PROJ = -1>
__device__ __forceinline__ void warp_coords4(const ViewDesc &V, const MeshTable &mesh, int v, int x, int y, float xc[4], float yc[4])
{
    const int ay = reflect_fast(y - V.top, V.ah);
    const int i0 = x - V.left;
    const bool interior = i0 >= 0 && i0 + 3 < V.aw;
    if (!CPW) {
        const float2 rt = gload_f2(V.rowtab + ay);
        float2 ct[4];
        if (interior) {
            float4 a, b;
            a = gload_f4(V.coltab + i0);
            b = gload_f4(V.coltab + i0 + 2);
            ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) ct[k] = gload_f2(V.coltab + reflect_fast(i0 + k, V.aw));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) warp_combine(PROJ < 0 ? V.proj : PROJ, ct[k], rt, V.wp, xc[k], yc[k]);
    } else {
        const float *mxp = mesh.x[v], *myp = mesh.y[v];
        const int mpitch = mesh.pitch[v];
        if (interior) {
            float4 a, b;
            __builtin_memcpy(&a, __builtin_assume_aligned(mxp + (size_t)ay * mpitch + i0, 4), 16);
            __builtin_memcpy(&b, __builtin_assume_aligned(myp + (size_t)ay * mpitch + i0, 4), 16);
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
    }
}

// Column terms of the 4 pixels of a lane (they do not depend on the row: loaded once per tile, not once per row group)
__device__ __forceinline__ void warp_coltab4(const ViewDesc &V, int x, float2 ct[4])
{
    const int i0 = x - V.left;
    if (i0 >= 0 && i0 + 3 < V.aw) {
        float4 a, b;
        a = gload_f4(V.coltab + i0);
        b = gload_f4(V.coltab + i0 + 2);
        ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) ct[k] = gload_f2(V.coltab + reflect_fast(i0 + k, V.aw));
    }
}

// ---- source tile staged in LDS (k_warp_a) ------------------------------------------------------------------------------
// The source bounding box of a tile is copied row by row, PACKED (3 B/px), in 16-byte chunks that are 16-byte aligned in global
// memory, by LDS-DMA (global_load_lds_dwordx4: lane i of a wave-instruction lands at base + 16 i, so the LDS image is chunk-linear);
// a row takes `np` chunk slots, np odd (rows then start on different LDS banks).
__host__ __device__ __forceinline__ int warp_lds_ncopy(int sw) { return (3 * sw + 15 + 15) / 16; }   // worst-case leading misalignment of 15 bytes
__host__ __device__ __forceinline__ int warp_lds_np(int sw) { return (warp_lds_ncopy(sw) + 1) | 1; }  // + one chunk of slack for the 12-byte reads
#ifndef MS_WA_BUF
#define MS_WA_BUF 8192
#endif
constexpr int WA_BUF_BYTES = MS_WA_BUF;                    // one staged tile; a wave owns two (double buffer): 10 waves per CU

// Bounding box (in source pixels) of every in-image bilinear tap of a tile: run once when the tables are built.
// flags bit 0 = the box fits a staging buffer and does not touch the last image row (whose 16-byte chunks could run past the buffer).
// stage1 != 0: the tiles are CPW stage-1 tiles (origin in warped-view pixels, no reflect pad): only flags bit 3 is of interest there.
__global__ void __launch_bounds__(256) k_tile_bbox(WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int src_rows, int src_cols, int stage1)
{
    __shared__ int s_box[5];
    WarpTile T = tiles[blockIdx.x];
    const ViewDesc &V = views[T.view];
    if (threadIdx.x == 0 && threadIdx.y == 0) { s_box[0] = s_box[2] = 0x7fffffff; s_box[1] = s_box[3] = -1; s_box[4] = -1; }
    __syncthreads();
    const int ox = stage1 ? V.left : 0, oy = stage1 ? V.top : 0;      // stage-1 tiles: the same coordinates through the padded position (identity reflect inside the view)
    const int x = T.x0 + 4 * (int)threadIdx.x + ox;
    for (int y = T.y0 + (int)threadIdx.y + oy; y < T.y0 + oy + WARP_TH; y += (int)blockDim.y)
    if (stage1 ? (x - ox < V.aw && y - oy < V.ah) : (x < V.pw && y < V.ph)) {
        float xc[4], yc[4];
        MeshTable none{};
        warp_coords4<false>(V, none, T.view, min(x, V.pw - 4), y, xc, yc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Taps t = make_taps(xc[k], yc[k], src_rows, src_cols);
            atomicMax(&s_box[4], min(max(f2i_rd(yc[k]), 0), src_rows - 2) + 1);       // lower tap row of the (clamped) read of ANY sample, see tap_offset
            if (t.fast) {
                atomicMin(&s_box[0], t.x1); atomicMax(&s_box[1], t.x1 + 1);
                atomicMin(&s_box[2], t.y1); atomicMax(&s_box[3], t.y1 + 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        T.flags &= ~(1 | 8);
        if (s_box[4] < src_rows - 1) T.flags |= 8;          // no sample of this tile reads the last image row: the aligned 12-byte tap reads stay inside the image
        if (s_box[1] < 0) { T.sx0 = T.sy0 = T.sw = T.sh = 0; }
        else {
            const int sx0 = s_box[0] & ~3;
            const int sw = ((s_box[1] - sx0 + 1) + 3) & ~3, sh = s_box[3] - s_box[2] + 1;
            T.sx0 = (short)sx0; T.sy0 = (short)s_box[2]; T.sw = (short)sw; T.sh = (short)sh;
            if (16 * warp_lds_np(sw) * sh <= WA_BUF_BYTES && s_box[2] + sh < src_rows) T.flags |= 1;
        }
        tiles[blockIdx.x] = T;
    }
}

// One lane owns WARP_NG row groups of 4 consecutive pixels (rows y, y + WARP_TH / WARP_NG, ..) of WARP_NF frames.  Round 3 default: ONE row group, TWO frames
// (same 16 tap-row reads in flight per lane as two row groups of one frame, but the coordinates and weights are built once: see warp_tile_direct).
#ifndef MS_WARP_NG
#define MS_WARP_NG 1
#endif
constexpr int WARP_NG = MS_WARP_NG;
constexpr int WARP_BY = WARP_TH / WARP_NG;    // block = WARP_BX x WARP_BY lanes

// One tile of Gaussian level 0 with the taps gathered straight from global memory (unaligned 8-byte reads): lane (tx, ty) of a
// WARP_BX x WARP_BY arrangement.  Software pipeline over the lane's units of work -- a unit = one row group of one frame -- with the tap
// reads of unit u+1 in flight while unit u is blended.  PROJ = the context's projection (compile-time: no per-pixel branches in the coordinate code).
// NF = frames per lane (1 or 2): the source coordinates of a pixel do not depend on the frame (the projection tables, and with CPW the mesh
// maps, are per context), so with NF = 2 a wave warps the same tile of frames f0 and f0 + 1 and builds each row group's coordinates ONCE --
// a fifth of the kernel's VALU work, and with CPW half of its mesh-map reads -- while the reads in flight per lane stay what they were.
#ifndef MS_WARP_NF_OPAQUE
#define MS_WARP_NF_OPAQUE 0
#endif
template <bool CPW, int PROJ, bool AL = false, int NF = 1, int NG = WARP_NG>      // NG: row groups per lane (block = WARP_BX x WARP_TH / NG lanes); AL: aligned 12-byte tap reads (see Px3); chosen per tile by the caller, never per lane
__device__ __forceinline__ void warp_tile_direct(const WarpTile &T, int f0, int nf, int tx, int ty, const ViewDesc *__restrict__ views, int n_views,
                                                 const SrcTable &src, int src_rows, int src_cols, const MeshTable &mesh,
                                                 const uint8_t *__restrict__ stage, long long stage_stride,
                                                 uint8_t *__restrict__ g0, long long g0_stride, const float2 *__restrict__ tabs)
{
    const int v = T.view;
    const ViewDesc &V = views[v];
    const int x = T.x0 + 4 * tx;
    int ys[NG];
    bool active[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { ys[g] = T.y0 + ty + g * (WARP_TH / NG); active[g] = x < V.pw && ys[g] < V.ph; }
    const uint8_t *sp[NF];
    unsigned sstep[NF];
    int srows, scols;
#pragma unroll
    for (int fi = 0; fi < NF; ++fi) {
        const int f = f0 + (fi < nf ? fi : 0);      // (a missing second frame of an odd batch is never sampled: its units are skipped below)
        if (CPW) { sp[fi] = stage + (size_t)f * stage_stride + V.s1_off; sstep[fi] = (unsigned)V.s1_pitch; }
        else { sp[fi] = src.p[f * n_views + v]; sstep[fi] = src.step[f * n_views + v]; }
    }
    if (CPW) { srows = V.ah; scols = V.aw; } else { srows = src_rows; scols = src_cols; }
    const LevelDesc &L = V.lv[0];
    const size_t plane = (size_t)L.h * L.pitch;
    float xc[2][4], yc[2][4];           // coordinates of a row group (NG <= 2 with NF = 2: one buffer per group)
    Px2 r1[2][4], r2[2][4];             // tap rows of a unit: double-buffered by unit
    // AL: the raw tap reads in flight are 12 aligned bytes per tap row + the 2-bit byte shifts of the unit's 4 pixels
    Px3 q1[AL ? 2 : 1][AL ? 4 : 1], q2[AL ? 2 : 1][AL ? 4 : 1];
    unsigned sh1[2] = {0u, 0u}, sh2[2] = {0u, 0u};
    // the 1-D tables of the projection are read ONCE, up front (column terms of the lane's 4 pixels, row term of each row group):
    // building the coordinates of a group is then pure arithmetic, with no load between it and the tap reads
    float2 ct[4], rt[NG];
    if (!CPW) {
        if (T.flags & 4) {            // interior tile: table addresses come from the tile entry alone, so these loads do not wait for
                                      // the view descriptor (one round trip less on the wave's critical path)
            float4 a, b;
            const float2 *cp = tabs + T.ctab + 4 * tx;
            __builtin_memcpy(&a, __builtin_assume_aligned(cp, 8), 16);
            __builtin_memcpy(&b, __builtin_assume_aligned(cp + 2, 8), 16);
            ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
#pragma unroll
            for (int g = 0; g < NG; ++g) rt[g] = tabs[T.rtab + ty + g * (WARP_TH / NG)];
        } else {
            warp_coltab4(V, min(x, V.pw - 4), ct);
#pragma unroll
            for (int g = 0; g < NG; ++g) rt[g] = gload_f2(V.rowtab + reflect_fast(min(ys[g], V.ph - 1) - V.top, V.ah));
        }
    }
    if (CPW && NG <= 2) {        // the dense mesh maps of both row groups are read up front too (one round trip, not one per group)
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if (active[g]) warp_coords4<CPW, PROJ>(V, mesh, v, x, ys[g], xc[g & 1], yc[g & 1]);
    }
    // units of this lane: (frame fi, group g), u = fi * NG + g -- all groups of the first frame, then all groups of the second: the coordinates of every
    // group are live from its first unit on (16 registers for two groups, as before), the column / row tables die after the first frame's units, and at most
    // two units' tap reads are in flight: the register peak is the one-frame kernel's (the order g-major instead costs 40 VGPRs = a wave per SIMD)
    constexpr int NU = NG * NF;
    auto issue = [&](int u) {
        const int fi = u / NG, g = u % NG, cb = g & 1, lb = u & 1;
        if (fi == 0) {                    // the group's coordinates: built for its first frame, reused by the second
#if defined(MS_PROBE) && MS_PROBE == 3       // roofline probe: affine coordinates instead of the projection (same gather density)
            if (active[g]) { for (int k = 0; k < 4; ++k) { xc[cb][k] = 1.55f * (float)(x + k - V.left) + 20.3f; yc[cb][k] = 1.6f * (float)(ys[g] - V.top) + 10.7f; } }
#else
            if (active[g]) {
                if (CPW) {
                    if (NG > 2) warp_coords4<CPW, PROJ>(V, mesh, v, x, ys[g], xc[cb], yc[cb]);      // (<= 2 groups: already read up front)
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) warp_combine(PROJ, ct[k], rt[g], V.wp, xc[cb][k], yc[cb][k]);
                }
            }
#endif
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) xc[cb][k] = yc[cb][k] = -1.f;
            }
        }
        // (no early exit for the missing second frame of an odd batch: its reads go to frame f0 again and are dropped.  A branch around them would make the
        //  wait-count pass assume either path at the join -- vmcnt(7) instead of vmcnt(15) before unit 0's first use, i.e. a wait for ALL of unit 1's reads)
        const uint8_t *spf = sp[fi];
        const unsigned stf = sstep[fi];
        const unsigned sp_lo = (unsigned)(uintptr_t)spf;
        const ms_gptr_u8 sp_al = (ms_gptr_u8)((uintptr_t)spf & ~(uintptr_t)3);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float xq = xc[cb][k], yq = yc[cb][k];
#if MS_WARP_NF_OPAQUE                        // the second frame re-derives offsets and weights from the coordinates instead of keeping the first frame's alive
            if (fi > 0) asm volatile("" : "+v"(xq), "+v"(yq));      // (common-subexpression elimination across the two frames costs 48 VGPRs = a wave per SIMD)
#endif
            unsigned off = tap_offset(f2i_rd(xq), f2i_rd(yq), srows, scols, stf);
#if defined(MS_PROBE) && MS_PROBE == 1       // roofline probe: same instructions, every tap read from one 4 KiB window (cache resident)
            off &= 0xfffu;
#endif
#if defined(MS_PROBE) && MS_PROBE == 10      // no-gather probe (WRONG pixels): no tap reads at all
            r1[lb][k] = Px2{off, off * 3u}; r2[lb][k] = Px2{off ^ 0x55u, off + 7u};
#else
            if (AL) {
                const unsigned a = (sp_lo & 3u) + off;
                q1[AL ? lb : 0][AL ? k : 0] = load_px3_at(sp_al, a);
                q2[AL ? lb : 0][AL ? k : 0] = load_px3_at(sp_al, a + stf);
                if (k == 0) { sh1[lb] = a & 3u; sh2[lb] = (a + stf) & 3u; }
                else { sh1[lb] |= (a & 3u) << (2 * k); sh2[lb] |= ((a + stf) & 3u) << (2 * k); }
            } else {
                r1[lb][k] = load_px2(spf, off);
                r2[lb][k] = load_px2(spf + stf, off);
            }
#endif
        }
    };
    issue(0);
#ifndef MS_NO_ORDER_BARRIER
    __builtin_amdgcn_sched_barrier(0);      // unit 0's reads go out FIRST: without this the scheduler hoists unit 1's above them, and unit 0's blend then waits for all 16
#endif
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int fi = u / NG, g = u % NG, cb = g & 1, lb = u & 1;
        if (u + 1 < NU) issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (active[g] && (NF == 1 || fi < nf)) {
            unsigned packed[3] = {0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                float o[2][3];
                Taps t[2];
#if !(defined(MS_PROBE) && MS_PROBE == 10)
                if (AL) {       // the pair's 8-byte windows out of the aligned reads, right before use (keeps the raw reads, not both forms, live)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        r1[lb][k + j] = px3_to_px2(q1[AL ? lb : 0][AL ? k + j : 0], sh1[lb] >> (2 * (k + j)));
                        r2[lb][k + j] = px3_to_px2(q2[AL ? lb : 0][AL ? k + j : 0], sh2[lb] >> (2 * (k + j)));
                    }
                }
#endif
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float xq = xc[cb][k + j], yq = yc[cb][k + j];
#if MS_WARP_NF_OPAQUE
                    if (fi > 0) asm volatile("" : "+v"(xq), "+v"(yq));
#endif
                    t[j] = make_taps(xq, yq, srows, scols);
                    if (!t[j].fast) fix_border_taps(r1[lb][k + j], r2[lb][k + j], t[j].x1, t[j].y1, srows, scols);
                }
                blend_taps2(t[0], t[1], r1[lb][k], r2[lb][k], r1[lb][k + 1], r2[lb][k + 1], o[0], o[1]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (CPW) {
                        packed[c] = sat_u8_into(o[0][c], k, packed[c]);
                        packed[c] = sat_u8_into(o[1][c], k + 1, packed[c]);
                    } else {      // convertTo(gain) of the rounded remap result (timed.cpp:94), both pixels of the pair in one v_pk_fma_f32 (fma(gain, x, 0) per half)
                        const f32x2 r = gain_pair(V.gain, (float)sat_u8(o[0][c]), (float)sat_u8(o[1][c]));
                        packed[c] = sat_u8_into(r.x, k, packed[c]);
                        packed[c] = sat_u8_into(r.y, k + 1, packed[c]);
                    }
                }
            }
            uint8_t *d = g0 + (size_t)(f0 + fi) * g0_stride + L.off + (size_t)ys[g] * L.pitch + x;
#if defined(MS_PROBE) && MS_PROBE == 9       // no-store probe: the stores (almost) never execute
            if (packed[0] == 0x12345678u && packed[1] == 0x9abcdef0u && packed[2] == 0x0fedcba9u)
#endif
            {
            *reinterpret_cast<unsigned *>(d) = packed[0];
            *reinterpret_cast<unsigned *>(d + plane) = packed[1];
            *reinterpret_cast<unsigned *>(d + 2 * plane) = packed[2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// AL: aligned 12-byte tap reads (Px3) for the tiles that allow them -- 28 more VGPRs (4 instead of 6 waves per SIMD), so the context picks the kernel:
// aligned where neighbouring samples share dwords (moderate minification: config 2, -6 %; the CPW mesh remap, -10 %), unaligned where every tap
// read is isolated and occupancy matters more (config 5's 2.7x minification: +17 % with AL).  blockIdx.z = a PAIR of frames (WARP_NF = 2): see NF above.
#ifndef MS_WARP_NF
#define MS_WARP_NF 2
#endif
constexpr int WARP_NF = MS_WARP_NF;
// the mesh remap (CPW) also reads its two dense coordinate maps once per lane and frame GROUP: three frames per lane there (same box, 24 frames per launch:
// 214.9 us with two, 204.8 with three, 205.6 with four); the projection warp is best at two (307 / 332 / 345 us for two / three / four)
#ifndef MS_WARP_NF_CPW
#define MS_WARP_NF_CPW 3
#endif
constexpr int warp_nf(bool cpw) { return cpw ? MS_WARP_NF_CPW : WARP_NF; }
// the aligned shared-offset projection warp (k_warp_s<false, ., ., true>: config 2): THREE frames per lane since round 5 -- with the per-frame work down to its arithmetic core
// (round 4) a third frame amortises the per-pixel part once more: 362 -> 333 us per 30 frames on the same box; the per-frame kernels (k_warp_t) and the unaligned shared form keep WARP_NF
#ifndef MS_WARP_NF_S
#define MS_WARP_NF_S 3
#endif
constexpr int WARP_NF_S = MS_WARP_NF_S;
// lane rows per WORKGROUP: a tile's WARP_BY lane rows are split over WARP_BY / WARP_WY workgroups (blockIdx.y).  MS_WARP_ONE_WAVE: 64-lane workgroups
// (the waves of a tile share nothing but the tile record), as k_blend8 got in round 2.
#ifndef MS_WARP_ONE_WAVE
#define MS_WARP_ONE_WAVE 0
#endif
constexpr int WARP_WY = (MS_WARP_ONE_WAVE && WARP_BY * WARP_BX > 64) ? 64 / WARP_BX : WARP_BY;
template <bool CPW, bool AL, int PROJ>
__global__ void __launch_bounds__(WARP_BX * WARP_WY) k_warp_t(const WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int n_views,
                                                         SrcTable src, int src_rows, int src_cols, MeshTable mesh,
                                                         const uint8_t *__restrict__ stage, long long stage_stride,
                                                         uint8_t *__restrict__ g0, long long g0_stride, const float2 *__restrict__ tabs, int n_frames)
{
    const WarpTile T = tiles[blockIdx.x];
    constexpr int NF = warp_nf(CPW);
    const int f0 = (int)blockIdx.z * NF, nf = min(NF, n_frames - f0);
    // aligned tap reads unless a sample of the tile reads the last row of a caller's image (flags bit 3, k_tile_bbox); the CPW stage buffer is ours and padded
    if (AL && (CPW || (T.flags & 8)))
        warp_tile_direct<CPW, PROJ, true, NF>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                                   g0, g0_stride, tabs);
    else
        warp_tile_direct<CPW, PROJ, false, NF>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                                    g0, g0_stride, tabs);
}

// ---- round 4: the frames of a lane share EVERYTHING but the source base -------------------------------------------------------------------
// warp_tile_direct shares the source coordinates between the frames of a lane; the ISA showed that each frame still rebuilt the tap offsets
// (clamps, row * step, the 2-bit byte shifts), the floors, the "all taps inside" tests with their exec-mask branches, and the four bilinear
// weights of every pixel -- about a third of the kernel's VALU cycles (profiles/r04_valu_probe.txt: clamps, conversions, v_mul_lo_u32 and
// the three-operand forms issue at half rate on gfx950).  When the NF source images of a view have the same row step and the same address
// modulo 4 -- the caller's frames of one camera practically always do; the host checks (stitch_impl) and launches k_warp_t otherwise -- the
// aligned tap address of a pixel is ONE 32-bit offset for all frames, added to a per-frame SGPR base by the load itself
// (global_load_dwordx3 v, v_off, s[base]).  So per pixel, once: floor, offset, shifts, weights, border flag; per pixel and frame: two reads,
// four v_alignbyte, 12 conversions, 12 + 3 fmas, 6 packs.  104 VGPRs: the kernel is held at 4 waves per SIMD anyway (see stitch_impl).
// The border test is one wave-level branch per frame (ballot over the lanes' four pixels) instead of one exec-masked region per pixel.
// AL = false: the unaligned 8-byte tap reads of warp_tile_direct (strong minification, config 5: every tap read is isolated and the aligned form's extra registers cost
// more than its dwords save) with the same sharing -- the offset needs only the frames' common row step, whatever their alignment.
template <bool CPW, int PROJ, int NF, bool AL = true>
__device__ __forceinline__ void warp_tile_shared(const WarpTile &T, int f0, int nf, int tx, int ty, const ViewDesc *__restrict__ views, int n_views,
                                                 const SrcTable &src, int src_rows, int src_cols, const MeshTable &mesh,
                                                 const uint8_t *__restrict__ stage, long long stage_stride,
                                                 uint8_t *__restrict__ g0, long long g0_stride, const float2 *__restrict__ tabs)
{
    constexpr int PRIO = CPW ? MS_PRIO_CPW : (AL ? MS_PRIO_WARP : 0);      // (config 5 -- 2.7 x minification -- runs this aligned form too: 705-710 us per 16 frames without the switch, 709-714 with: neutral there;
                                                                           //  the unaligned-read form, taken only when the frames of a view differ in alignment, is left alone)
    MS_PRIO_LOADS(PRIO);
    const int v = T.view;
    const ViewDesc &V = views[v];
    const int x = T.x0 + 4 * tx, y = T.y0 + ty;
    const bool active = x < V.pw && y < V.ph;
    ms_gptr_u8 base[NF];
    const uint8_t *ubase[NF];
    unsigned lo2 = 0u;
#pragma unroll
    for (int fi = 0; fi < NF; ++fi) {
        const int f = f0 + (fi < nf ? fi : 0);      // (a missing frame of a short group is read from frame f0 again and dropped: no branch around the reads, see warp_tile_direct)
        const uint8_t *p = CPW ? stage + (size_t)f * stage_stride + V.s1_off : src.p[f * n_views + v];
        base[fi] = (ms_gptr_u8)((uintptr_t)p & ~(uintptr_t)3);
        ubase[fi] = p;
        if (fi == 0 && AL) lo2 = (unsigned)(uintptr_t)p & 3u;
    }
    const unsigned st = CPW ? (unsigned)V.s1_pitch : src.step[f0 * n_views + v];
    const int srows = CPW ? V.ah : src_rows, scols = CPW ? V.aw : src_cols;
    const LevelDesc &L = V.lv[0];
    const size_t plane = (size_t)L.h * L.pitch;
    float xc[4], yc[4];
    if (CPW) {
        if (active) warp_coords4<true, PROJ>(V, mesh, v, x, y, xc, yc);
    } else {
        float2 ct[4], rt;
        if (T.flags & 4) {            // interior tile: table addresses from the tile entry alone
            float4 a, b;
            const float2 *cp = tabs + T.ctab + 4 * tx;
            __builtin_memcpy(&a, __builtin_assume_aligned(cp, 8), 16);
            __builtin_memcpy(&b, __builtin_assume_aligned(cp + 2, 8), 16);
            ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
            rt = tabs[T.rtab + ty];
        } else {
            warp_coltab4(V, min(x, V.pw - 4), ct);
            rt = gload_f2(V.rowtab + reflect_fast(min(y, V.ph - 1) - V.top, V.ah));
        }
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) warp_combine(PROJ, ct[k], rt, V.wp, xc[k], yc[k]);
        }
    }
    if (!active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xc[k] = yc[k] = -1.f;
    }
    // once per pixel: the aligned offsets of its two tap rows, the byte shifts, the border flag
    unsigned va[4], vb[4], sh1 = 0u, sh2 = 0u;
    bool slow = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x1 = f2i_rd(xc[k]), y1 = f2i_rd(yc[k]);
        slow = slow || !((unsigned)x1 < (unsigned)(scols - 2) && (unsigned)y1 < (unsigned)(srows - 1));
        const unsigned a = lo2 + tap_offset(x1, y1, srows, scols, st), b = a + st;
        va[k] = AL ? (a & ~3u) : a; vb[k] = AL ? (b & ~3u) : b;
        sh1 |= (a & 3u) << (2 * k); sh2 |= (b & 3u) << (2 * k);
    }
    Px3 q1[AL ? 2 : 1][AL ? 4 : 1], q2[AL ? 2 : 1][AL ? 4 : 1];
    Px2 u1[AL ? 1 : 2][AL ? 1 : 4], u2[AL ? 1 : 2][AL ? 1 : 4];
    auto issue = [&](int fi) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (AL) {
#ifdef MS_PROBE_TAPS      // request-path probe (WRONG pixels): only every MS_PROBE_TAPS-th pixel of a lane reads its taps, the others blend a copy -- same VALU, same stores, 1/n of the tap reads
                if (CPW && (k % MS_PROBE_TAPS) != 0) { q1[AL ? fi & 1 : 0][AL ? k : 0] = q1[AL ? fi & 1 : 0][AL ? k - k % MS_PROBE_TAPS : 0]; q2[AL ? fi & 1 : 0][AL ? k : 0] = q2[AL ? fi & 1 : 0][AL ? k - k % MS_PROBE_TAPS : 0]; continue; }
#endif
                const ms_u32x3_a4 r1 = *(const MS_GLOBAL_AS ms_u32x3_a4 *)(base[fi] + va[k]);
                const ms_u32x3_a4 r2 = *(const MS_GLOBAL_AS ms_u32x3_a4 *)(base[fi] + vb[k]);
                q1[AL ? fi & 1 : 0][AL ? k : 0] = Px3{r1.x, r1.y, r1.z};
                q2[AL ? fi & 1 : 0][AL ? k : 0] = Px3{r2.x, r2.y, r2.z};
            } else {
                u1[AL ? 0 : fi & 1][AL ? 0 : k] = load_px2(ubase[fi], va[k]);
                u2[AL ? 0 : fi & 1][AL ? 0 : k] = load_px2(ubase[fi], vb[k]);
            }
        }
    };
    issue(0);
    __builtin_amdgcn_sched_barrier(0);      // frame 0's reads go out first
    if (NF > 1) issue(1);
    __builtin_amdgcn_sched_barrier(0);
    MS_PRIO_MATH(PRIO);
    // ... and while they are in flight: the weights (frame-invariant)
    Taps t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = make_taps(xc[k], yc[k], srows, scols);
    const bool any_slow = __builtin_amdgcn_ballot_w64(active && slow) != 0ull;
    const float gain = V.gain;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int fi = 0; fi < NF; ++fi) {
        const int b = fi & 1;
        if (active && fi < nf) {
            unsigned packed[3] = {0, 0, 0};
            Px2 r1[4], r2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (AL) {
                    r1[k] = px3_to_px2(q1[AL ? b : 0][AL ? k : 0], sh1 >> (2 * k));
                    r2[k] = px3_to_px2(q2[AL ? b : 0][AL ? k : 0], sh2 >> (2 * k));
                } else { r1[k] = u1[AL ? 0 : b][AL ? 0 : k]; r2[k] = u2[AL ? 0 : b][AL ? 0 : k]; }
            }
            if (any_slow) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (!t[k].fast) fix_border_taps(r1[k], r2[k], t[k].x1, t[k].y1, srows, scols);
            }
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                float o[2][3];
                blend_taps2(t[k], t[k + 1], r1[k], r2[k], r1[k + 1], r2[k + 1], o[0], o[1]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (CPW) {
                        packed[c] = sat_u8_into(o[0][c], k, packed[c]);
                        packed[c] = sat_u8_into(o[1][c], k + 1, packed[c]);
                    } else {      // convertTo(gain) of the rounded remap result (timed.cpp:94)
                        const f32x2 r = gain_pair(gain, (float)sat_u8(o[0][c]), (float)sat_u8(o[1][c]));
                        packed[c] = sat_u8_into(r.x, k, packed[c]);
                        packed[c] = sat_u8_into(r.y, k + 1, packed[c]);
                    }
                }
            }
#ifdef MS_PROBE_EXTRA_VALU      // headroom probe (WRONG pixels by a hair: the extra work is folded into the result so that it cannot be dropped): N more half-rate VALU per frame
            { unsigned e = packed[0];
#pragma unroll
              for (int i_ = 0; i_ < MS_PROBE_EXTRA_VALU; ++i_) e = __builtin_amdgcn_perm(e, packed[1] + i_, 0x07020500u);
              if (e == 0x12345678u) packed[2] ^= 1u; }
#endif
            uint8_t *d = g0 + (size_t)(f0 + fi) * g0_stride + L.off + (size_t)y * L.pitch + x;
            *reinterpret_cast<unsigned *>(d) = packed[0];
            *reinterpret_cast<unsigned *>(d + plane) = packed[1];
            *reinterpret_cast<unsigned *>(d + 2 * plane) = packed[2];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (fi + 2 < NF) { MS_PRIO_RELOADS(PRIO); issue(fi + 2); __builtin_amdgcn_sched_barrier(0); MS_PRIO_MATH(PRIO); }      // (three frames per lane: the third frame's reads reuse frame 0's registers)
    }
}

template <bool CPW, int PROJ, int NF, bool AL = true>      // NF = warp_nf(CPW) frames per lane; 1 for one-frame calls (the reference's call shape: no reads for a frame that is not there)
__global__ void __launch_bounds__(WARP_BX * WARP_WY) k_warp_s(const WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int n_views,
                                                         SrcTable src, int src_rows, int src_cols, MeshTable mesh,
                                                         const uint8_t *__restrict__ stage, long long stage_stride,
                                                         uint8_t *__restrict__ g0, long long g0_stride, const float2 *__restrict__ tabs, int n_frames)
{
    const WarpTile T = tiles[blockIdx.x];
    const int f0 = (int)blockIdx.z * NF, nf = min(NF, n_frames - f0);
    if (!AL) {
        if (NF == 3 && nf == 2)
            warp_tile_shared<CPW, PROJ, 2, false>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                                  g0, g0_stride, tabs);
        else if (NF >= 2 && nf == 1)
            warp_tile_shared<CPW, PROJ, 1, false>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                                  g0, g0_stride, tabs);
        else
            warp_tile_shared<CPW, PROJ, NF, false>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                                   g0, g0_stride, tabs);
    } else if (CPW || (T.flags & 8)) {     // (a tile that samples the last row of a caller's image keeps the unaligned reads: see Px3)
        // The last frame group of a call may be short (32 frames = 10 groups of three + one of two).  warp_tile_shared issues the reads of all NF frames whatever nf is (no branch
        // around reads: the waits stay counted), so a short group used to cost a full one -- 11 groups x 33 us instead of 10.67 (round 5: profiles/r05_experiments.txt).  The
        // group index is uniform over the workgroup: a short group takes the instantiation for its own frame count.
        if (NF == 3 && nf == 2)
            warp_tile_shared<CPW, PROJ, 2>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                           g0, g0_stride, tabs);
        else if (NF >= 2 && nf == 1)
            warp_tile_shared<CPW, PROJ, 1>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                           g0, g0_stride, tabs);
        else
            warp_tile_shared<CPW, PROJ, NF>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                            g0, g0_stride, tabs);
    } else
        warp_tile_direct<CPW, PROJ, false, NF>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, src_rows, src_cols, mesh, stage, stage_stride,
                                               g0, g0_stride, tabs);
}

// the 1-D projection tables of a lane for tile T: column terms of its 4 pixels, row term of its row (see warp_tile_direct)
__device__ __forceinline__ void warp_tabs_load(const WarpTile &T, const ViewDesc *__restrict__ views, const float2 *__restrict__ tabs, int tx, int ty, float2 ct[4], float2 &rt)
{
    if (T.flags & 4) {            // interior tile: table addresses from the tile entry alone
        float4 a, b;
        const float2 *cp = tabs + T.ctab + 4 * tx;
        __builtin_memcpy(&a, __builtin_assume_aligned(cp, 8), 16);
        __builtin_memcpy(&b, __builtin_assume_aligned(cp + 2, 8), 16);
        ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
        rt = tabs[T.rtab + ty];
    } else {
        const ViewDesc &V = views[T.view];
        warp_coltab4(V, min(T.x0 + 4 * tx, V.pw - 4), ct);
        rt = gload_f2(V.rowtab + reflect_fast(min(T.y0 + ty, V.ph - 1) - V.top, V.ah));
    }
}

// ---- round 4: the projection warp sampling the cameras' NV12 frames directly --------------------------------------------
// The capture threads of the reference convert every camera frame with cvtColor(COLOR_YUV2BGR_NV12) before the stitcher sees it (APP/networking.cpp:45-47,
// defs.h:10-17); ms_nv12_to_bgr[_batch] does that on the device as a pass of its own (18.7 MB read + 37.3 MB written per 6 x 1080p frame, which k_warp_t then reads
// again).  Here the warp reads the Y plane and the interleaved UV plane itself: every bilinear tap is converted with the SAME integer formula (nv12_bgr below =
// nv12_to_bgr_cell of prims.hip = YUV420sp2RGB888Invoker of imgproc/src/color.cpp) and the four converted taps go through the same fp32 bilinear in the same order,
// so the result is bit-identical to ms_nv12_to_bgr_batch + ms_stitch (tests/test_compositor_gpu.py::test_nv12_direct_*).  Per pixel and frame: two 2-byte reads
// (Y of the two tap rows), two 4-byte reads (the UV pairs under them) -- 12 bytes instead of 24, no BGR image at all.  Offsets, weights and flags are per pixel,
// shared by the frames of a lane (the frames of a view must share their row step: checked by the host).
constexpr int S1_BY_NV = WARP_TH;          // lane rows of a stage-1 workgroup (= S1_BY, declared further down)
typedef unsigned short ms_u16_a1 __attribute__((aligned(1)));
typedef unsigned ms_u32_a2 __attribute__((aligned(2)));
struct NvRGB { float b, g, r; };
__device__ __forceinline__ NvRGB nv12_bgr(unsigned Y, unsigned uvp)      // uvp: U in bits 0..7, V in bits 8..15
{
    constexpr int SH = 20, CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527;
    const int u = (int)(uvp & 0xffu) - 128, v = (int)((uvp >> 8) & 0xffu) - 128;
    const int ruv = (1 << (SH - 1)) + CVR * v, guv = (1 << (SH - 1)) + CVG * v + CUG * u, buv = (1 << (SH - 1)) + CUB * u;
    const int yy = max(0, (int)Y - 16) * CY;
    NvRGB o;
    o.b = (float)min(max((yy + buv) >> SH, 0), 255);
    o.g = (float)min(max((yy + guv) >> SH, 0), 255);
    o.r = (float)min(max((yy + ruv) >> SH, 0), 255);
    return o;
}
// one tap with an explicit bounds test (border samples only): BORDER_CONSTANT 0 in BGR space, as the remap of the converted image would see it
__device__ __forceinline__ NvRGB nv12_tap_checked(ms_gptr_u8 base, unsigned st, int rows, int cols, int xx, int yy)
{
    NvRGB z{0.f, 0.f, 0.f};
    if ((unsigned)xx >= (unsigned)cols || (unsigned)yy >= (unsigned)rows) return z;
    const unsigned Y = base[(unsigned)yy * st + (unsigned)xx];
    const unsigned o = (unsigned)(rows + (yy >> 1)) * st + (unsigned)(xx & ~1);
    return nv12_bgr(Y, (unsigned)base[o] | ((unsigned)base[o + 1] << 8));
}
// ALN: every read an ALIGNED 8-byte window (global_load_dwordx2 at a multiple of 4) and one 64-bit shift -- an unaligned gather costs the texture-address path more than an aligned one of
// twice the size (config 2, BGR: unaligned 8-byte tap reads 456 us, aligned 12-byte ones 375).  Needs 4-byte aligned planes, steps and widths (the host checks): then both tap rows
// and both UV rows of a pixel share their shifts, and clamping the UV window to [row end - 8, row end) keeps the last row's reads inside the caller's buffer.
// S1: the tile is a CPW stage-1 tile (k_stage1_t's geometry: view pixels, no reflect pad) and the result goes to the interleaved 8UC3 stage image the mesh remap samples --
// images[i] = gain(remap(cvtColor(nv12_i), x_map, y_map)) (networking.cpp:45-47 + timed.cpp:90-94) without the BGR frame.
template <int PROJ, int NF, bool ALN, bool S1>
__device__ __forceinline__ void nv12_tile(const WarpTile &T, int f0, int nf, int tx, int ty, const ViewDesc *__restrict__ views, int n_views,
                                          const SrcTable &src, int rows, int cols, uint8_t *__restrict__ g0, long long g0_stride, const float2 *__restrict__ tabs)
{
    constexpr int PRIO = S1 ? 0 : MS_PRIO_NV12;
    MS_PRIO_LOADS(PRIO);
    const int v = T.view;
    const ViewDesc &V = views[v];
    const int x = T.x0 + 4 * tx, y = T.y0 + ty;
    const bool active = S1 ? (x < V.aw && y < V.ah) : (x < V.pw && y < V.ph);
    if (S1 && !active) return;
    ms_gptr_u8 base[NF];
#pragma unroll
    for (int fi = 0; fi < NF; ++fi) base[fi] = (ms_gptr_u8)(uintptr_t)src.p[(f0 + (fi < nf ? fi : 0)) * n_views + v];
    const unsigned st = src.step[f0 * n_views + v];
    const LevelDesc &L = V.lv[0];
    const size_t plane = (size_t)L.h * L.pitch;
    float xc[4], yc[4];
    {
        float2 ct[4], rt;
        if (S1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ct[k] = gload_f2(V.coltab + min(x + k, V.aw - 1));
            rt = gload_f2(V.rowtab + y);
        } else warp_tabs_load(T, views, tabs, tx, ty, ct, rt);
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) warp_combine(PROJ, ct[k], rt, V.wp, xc[k], yc[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) xc[k] = yc[k] = -1.f;
        }
    }
    // once per pixel: offsets of the two Y reads and the two UV reads, which UV pair each tap column takes, the border flag
    unsigned oy[4], ou1[4], ou2[4], sel = 0u, shf = 0u;
    bool slow = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x1 = f2i_rd(xc[k]), y1 = f2i_rd(yc[k]);
        slow = slow || !((unsigned)x1 < (unsigned)(cols - 1) && (unsigned)y1 < (unsigned)(rows - 1));
        const int x1c = min(max(x1, 0), cols - 2), y1c = min(max(y1, 0), rows - 2);
        const int p0 = min(x1c & ~1, cols - 4);                  // the 4-byte UV window [p0, p0 + 4) stays inside the row
        oy[k] = (unsigned)y1c * st + (unsigned)x1c;
        ou1[k] = (unsigned)(rows + (y1c >> 1)) * st + (unsigned)p0;
        ou2[k] = (unsigned)(rows + ((y1c + 1) >> 1)) * st + (unsigned)p0;
        sel |= ((unsigned)(((x1c & ~1) - p0) >> 1) | ((unsigned)((((x1c + 1) & ~1) - p0) >> 1) << 1)) << (2 * k);      // bit 0: pair of tap column x1, bit 1: of x1 + 1
        if (ALN) {      // aligned windows: Y at oy & ~3 (both rows: the step is a multiple of 4), UV at min(ou & ~3, row end - 8); the byte shifts (<= 3 and <= 4) once per pixel
            const unsigned ca = min((unsigned)p0 & ~3u, (unsigned)cols - 8u), su = (unsigned)p0 - ca;
            shf |= ((oy[k] & 3u) | (su << 2)) << (5 * k);
            oy[k] &= ~3u;
            ou1[k] = ou1[k] - (unsigned)p0 + ca;
            ou2[k] = ou2[k] - (unsigned)p0 + ca;
        }
    }
    typedef unsigned ms_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
    unsigned qy1[2][4], qy2[2][4], qu1[2][4], qu2[2][4];
    unsigned hy1[ALN ? 2 : 1][ALN ? 4 : 1], hy2[ALN ? 2 : 1][ALN ? 4 : 1], hu1[ALN ? 2 : 1][ALN ? 4 : 1], hu2[ALN ? 2 : 1][ALN ? 4 : 1];      // ALN: the windows' upper dwords
    auto issue = [&](int fi) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ALN) {
                const ms_u32x2_a4 a = *(const MS_GLOBAL_AS ms_u32x2_a4 *)(base[fi] + oy[k]), b = *(const MS_GLOBAL_AS ms_u32x2_a4 *)(base[fi] + oy[k] + st);
                const ms_u32x2_a4 c = *(const MS_GLOBAL_AS ms_u32x2_a4 *)(base[fi] + ou1[k]), d = *(const MS_GLOBAL_AS ms_u32x2_a4 *)(base[fi] + ou2[k]);
                qy1[fi & 1][k] = a.x; hy1[ALN ? fi & 1 : 0][ALN ? k : 0] = a.y; qy2[fi & 1][k] = b.x; hy2[ALN ? fi & 1 : 0][ALN ? k : 0] = b.y;
                qu1[fi & 1][k] = c.x; hu1[ALN ? fi & 1 : 0][ALN ? k : 0] = c.y; qu2[fi & 1][k] = d.x; hu2[ALN ? fi & 1 : 0][ALN ? k : 0] = d.y;
            } else {
                qy1[fi & 1][k] = *(const MS_GLOBAL_AS ms_u16_a1 *)(base[fi] + oy[k]);
                qy2[fi & 1][k] = *(const MS_GLOBAL_AS ms_u16_a1 *)(base[fi] + oy[k] + st);
                qu1[fi & 1][k] = *(const MS_GLOBAL_AS ms_u32_a2 *)(base[fi] + ou1[k]);
                qu2[fi & 1][k] = *(const MS_GLOBAL_AS ms_u32_a2 *)(base[fi] + ou2[k]);
            }
        }
    };
    issue(0);
    __builtin_amdgcn_sched_barrier(0);
    if (NF > 1) issue(1);
    __builtin_amdgcn_sched_barrier(0);
    MS_PRIO_MATH(PRIO);
    Taps t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = make_taps(xc[k], yc[k], rows + 1, cols + 1);      // (weights only; `fast` of Taps is not used here: see `slow`)
    const bool any_slow = __builtin_amdgcn_ballot_w64(active && slow) != 0ull;
    const float gain = V.gain;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int fi = 0; fi < NF; ++fi) {
        const int b = fi & 1;
        if (active && fi < nf) {
            unsigned packed[3] = {0, 0, 0};      // S1: the 12 interleaved output bytes; else one dword per colour plane
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned s0 = (sel >> (2 * k)) & 1u, s1 = (sel >> (2 * k + 1)) & 1u;
                unsigned u1 = qu1[b][k], u2 = qu2[b][k], y1p = qy1[b][k], y2p = qy2[b][k];
                if (ALN) {      // the wanted bytes out of the aligned windows: one 64-bit shift each
                    const unsigned sy = 8u * ((shf >> (5 * k)) & 3u), su = 8u * ((shf >> (5 * k + 2)) & 7u);
                    y1p = (unsigned)((((unsigned long long)hy1[ALN ? b : 0][ALN ? k : 0] << 32) | y1p) >> sy) & 0xffffu;
                    y2p = (unsigned)((((unsigned long long)hy2[ALN ? b : 0][ALN ? k : 0] << 32) | y2p) >> sy) & 0xffffu;
                    u1 = (unsigned)((((unsigned long long)hu1[ALN ? b : 0][ALN ? k : 0] << 32) | u1) >> su);
                    u2 = (unsigned)((((unsigned long long)hu2[ALN ? b : 0][ALN ? k : 0] << 32) | u2) >> su);
                }
                NvRGB a11 = nv12_bgr(y1p & 0xffu, s0 ? (u1 >> 16) : u1), a12 = nv12_bgr(y1p >> 8, s1 ? (u1 >> 16) : u1);
                NvRGB a21 = nv12_bgr(y2p & 0xffu, s0 ? (u2 >> 16) : u2), a22 = nv12_bgr(y2p >> 8, s1 ? (u2 >> 16) : u2);
                if (any_slow) {
                    const int x1 = f2i_rd(xc[k]), y1 = f2i_rd(yc[k]);
                    if (!((unsigned)x1 < (unsigned)(cols - 1) && (unsigned)y1 < (unsigned)(rows - 1))) {      // a tap outside the image: every tap again, with its bounds test
                        a11 = nv12_tap_checked(base[fi], st, rows, cols, x1, y1);     a12 = nv12_tap_checked(base[fi], st, rows, cols, x1 + 1, y1);
                        a21 = nv12_tap_checked(base[fi], st, rows, cols, x1, y1 + 1); a22 = nv12_tap_checked(base[fi], st, rows, cols, x1 + 1, y1 + 1);
                    }
                }
                // the fp32 bilinear of remap.cu / filters.hpp in the reference's tap order (blend_taps), per channel
                float o[3];
                o[0] = fma_single(a22.b, t[k].w22, fma_single(a21.b, t[k].w21, fma_single(a12.b, t[k].w12, fma_single(a11.b, t[k].w11, 0.f))));
                o[1] = fma_single(a22.g, t[k].w22, fma_single(a21.g, t[k].w21, fma_single(a12.g, t[k].w12, fma_single(a11.g, t[k].w11, 0.f))));
                o[2] = fma_single(a22.r, t[k].w22, fma_single(a21.r, t[k].w21, fma_single(a12.r, t[k].w12, fma_single(a11.r, t[k].w11, 0.f))));
#pragma unroll
                for (int c = 0; c < 3; ++c) {      // convertTo(gain) of the rounded remap result (timed.cpp:94)
                    const int i = S1 ? 3 * k + c : 0;
                    if (S1) packed[i >> 2] = sat_u8_into(fma_single(gain, (float)sat_u8(o[c]), 0.f), i & 3, packed[i >> 2]);
                    else packed[c] = sat_u8_into(fma_single(gain, (float)sat_u8(o[c]), 0.f), k, packed[c]);
                }
            }
            if (S1) {      // (g0 / g0_stride: the stage buffer and its per-frame stride)
                uint8_t *d = g0 + (size_t)(f0 + fi) * g0_stride + V.s1_off + (size_t)y * V.s1_pitch + (size_t)x * 3;   // 12 B per lane, dword aligned
                if (x + 3 < V.aw) __builtin_memcpy(__builtin_assume_aligned(d, 4), packed, 12);
                else {
                    for (int k = 0; k < 4 && x + k < V.aw; ++k)
                        for (int c = 0; c < 3; ++c) { const int i = 3 * k + c; d[i] = (uint8_t)(packed[i >> 2] >> (8 * (i & 3))); }
                }
            } else {
                uint8_t *d = g0 + (size_t)(f0 + fi) * g0_stride + L.off + (size_t)y * L.pitch + x;
                *reinterpret_cast<unsigned *>(d) = packed[0];
                *reinterpret_cast<unsigned *>(d + plane) = packed[1];
                *reinterpret_cast<unsigned *>(d + 2 * plane) = packed[2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (fi + 2 < NF) { MS_PRIO_RELOADS(PRIO); issue(fi + 2); __builtin_amdgcn_sched_barrier(0); MS_PRIO_MATH(PRIO); }
    }
}
template <int PROJ, int NF, bool ALN>
__global__ void __launch_bounds__(WARP_BX * WARP_WY) k_warp_nv12(const WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int n_views,
                                                            SrcTable src, int rows, int cols, uint8_t *__restrict__ g0, long long g0_stride,
                                                            const float2 *__restrict__ tabs, int n_frames)
{
    const WarpTile T = tiles[blockIdx.x];
    const int f0 = (int)blockIdx.z * NF, nf = min(NF, n_frames - f0);
    nv12_tile<PROJ, NF, ALN, false>(T, f0, nf, (int)threadIdx.x, (int)(threadIdx.y + blockIdx.y * WARP_WY), views, n_views, src, rows, cols, g0, g0_stride, tabs);
}
template <int PROJ, int NF, bool ALN>
__global__ void __launch_bounds__(WARP_BX * S1_BY_NV) k_stage1_nv12(const WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int n_views,
                                                               SrcTable src, int rows, int cols, uint8_t *__restrict__ stage, long long stage_stride, DispTable disp, int n_frames)
{
    const WarpTile T = tiles[blockIdx.x];
    if (!(T.flags & 2) && *disp.p[T.view] <= disp.limit_bits) return;      // the mesh of this view moves no sample this far: stage 2 never reads this tile (k_stage1_t)
    const int f0 = (int)blockIdx.z * NF, nf = min(NF, n_frames - f0);
    nv12_tile<PROJ, NF, ALN, true>(T, f0, nf, (int)threadIdx.x, (int)threadIdx.y, views, n_views, src, rows, cols, stage, stage_stride, nullptr);
}

// ---- the same tiles with the source staged in LDS by asynchronous LDS-DMA: persistent, self-pipelined waves -------------------
// Measured on the direct kernel (profiles/r02_warp_probes.txt): the gathers cost per lane-dword the texture-address path handles
// (an unaligned 8-byte tap read touches 3 dwords, 48 per lane and tile), and they do not overlap the kernel's other half, its VALU
// work, at 5 waves per SIMD.  Here a wave stages the bounding box of its NEXT tile (about 4.6 KiB for a 32 x 16 tile at the 1.6 x 1.7
// minification of config 2 = 18 coalesced dwords per lane) with global_load_lds_dwordx4 -- no VGPRs, no ds_write, completion counted
// on vmcnt -- while it samples the CURRENT tile out of LDS (ds_read2_b32 + ds_read_b32 per tap row, v_alignbyte to the pixel).
// The buffers are private to the wave: no barrier anywhere, only the wave's own s_waitcnt.  Tiles whose box does not fit (strong
// minification) or touches the last image row, and samples with a tap outside the image, take the direct path.
// Same fp32 operations on the same bytes as warp_tile_direct: bit-identical results.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)      // lds_dst: wave-uniform LDS byte address; lane i lands at + 16 i
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wa_stage(const WarpTile &T, const uint8_t *sp, unsigned sstep, unsigned lds_buf, int lane)
{
    const int np = warp_lds_np(T.sw), ncopy = warp_lds_ncopy(T.sw), total = np * T.sh;
    const float rnp = 1.0f / (float)np;
    const uint8_t *tile0 = sp + (size_t)T.sy0 * sstep + 3 * (int)T.sx0;
    for (int q0 = 0; q0 < total; q0 += 64) {
        const int q = q0 + lane;
        const int r = (int)(((float)q + 0.5f) * rnp), c = q - r * np;       // (q + 0.5) / np is >= 0.5 / np away from an integer: exact for q < 2^16
        if (q < total && c < ncopy) {
            const uintptr_t row = (uintptr_t)(tile0 + (size_t)r * sstep);
            glds16((const void *)((row & ~(uintptr_t)15) + 16u * (unsigned)c), lds_buf + 16u * (unsigned)q0);
        }
    }
}
// the three dwords around the 6 tap bytes at LDS byte offset `off`, as the (lo, hi) pair an unaligned 8-byte global read returns
__device__ __forceinline__ Px2 lds_px2(const uint8_t *lds, unsigned off)
{
    const unsigned *w = reinterpret_cast<const unsigned *>(lds + (off & ~3u));
    const unsigned d0 = w[0], d1 = w[1], d2 = w[2];
    const unsigned sh = off & 3u;
    Px2 px;
    px.lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    px.hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return px;
}

constexpr int WA_NG = 2, WA_BY = WARP_TH / WA_NG;      // k_warp_a: one 64-lane wave per tile = WARP_BX x WA_BY lanes, two row groups per lane
// column terms of the lane's 4 pixels and row terms of its row groups, for tile T (loads only: consumed one iteration later)
__device__ __forceinline__ void wa_tables(const WarpTile &T, const ViewDesc *__restrict__ views, const float2 *__restrict__ tabs, int tx, int ty,
                                          float2 ct[4], float2 rt[WA_NG])
{
    if (T.flags & 4) {
        float4 a, b;
        const float2 *cp = tabs + T.ctab + 4 * tx;
        __builtin_memcpy(&a, __builtin_assume_aligned(cp, 8), 16);
        __builtin_memcpy(&b, __builtin_assume_aligned(cp + 2, 8), 16);
        ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
#pragma unroll
        for (int g = 0; g < WA_NG; ++g) rt[g] = tabs[T.rtab + ty + g * WA_BY];
    } else {
        const ViewDesc &V = views[T.view];
        warp_coltab4(V, min(T.x0 + 4 * tx, V.pw - 4), ct);
#pragma unroll
        for (int g = 0; g < WA_NG; ++g) rt[g] = gload_f2(V.rowtab + reflect_fast(min(T.y0 + ty + g * WA_BY, V.ph - 1) - V.top, V.ah));
    }
}

template <int PROJ>
__global__ void __launch_bounds__(64) k_warp_a(const WarpTile *__restrict__ tiles, int n_tiles, const ViewDesc *__restrict__ views, int n_views,
                                               SrcTable src, int srows, int scols, uint8_t *__restrict__ g0, long long g0_stride,
                                               const float2 *__restrict__ tabs, int n_frames)
{
    extern __shared__ uint4 s_wa[];                           // 2 x WA_BUF_BYTES
    const uint8_t *lds = reinterpret_cast<const uint8_t *>(s_wa);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)s_wa);
    const int lane = (int)threadIdx.x, tx = lane & (WARP_BX - 1), ty = lane / WARP_BX;
    const int stride = (int)gridDim.x;
    auto advance = [&](int &t_, int &f_) { t_ += stride; while (t_ >= n_tiles) { t_ -= n_tiles; ++f_; } };
    int t = (int)blockIdx.x - stride, f = 0;
    advance(t, f);
    if (f >= n_frames) return;
    int tn = t, fn = f;
    advance(tn, fn);
    WarpTile T = tiles[t], Tn = tiles[fn < n_frames ? tn : t];
    float2 ct[4], rt[WA_NG];
    wa_tables(T, views, tabs, tx, ty, ct, rt);
    int buf = 0;
    if (T.flags & 1) wa_stage(T, src.p[f * n_views + T.view], src.step[f * n_views + T.view], lds0, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): the first tile has landed
    MeshTable none{};
    for (;;) {
        // the NEXT tile of this wave: its tables (ordinary loads, consumed one iteration later) and then its staging copy into the other
        // buffer, in flight while this tile is sampled.  Nothing issued after the copy is waited for before the s_waitcnt below.
        const bool more = fn < n_frames;
        float2 ctn[4], rtn[WA_NG];
        if (more) {
            wa_tables(Tn, views, tabs, tx, ty, ctn, rtn);
            if (Tn.flags & 1) wa_stage(Tn, src.p[fn * n_views + Tn.view], src.step[fn * n_views + Tn.view], lds0 + (unsigned)((buf ^ 1) * WA_BUF_BYTES), lane);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) ctn[k] = ct[k];
#pragma unroll
            for (int g = 0; g < WA_NG; ++g) rtn[g] = rt[g];
        }
        int tnn = tn, fnn = fn;                                // descriptor two tiles ahead (scalar load: its latency hides behind this tile)
        advance(tnn, fnn);
        const WarpTile Tnn = tiles[fnn < n_frames ? tnn : t];
        if (!(T.flags & 1)) {
            warp_tile_direct<false, PROJ, false, 1, WA_NG>(T, f, 1, tx, ty, views, n_views, src, srows, scols, none, nullptr, 0, g0, g0_stride, tabs);
            __builtin_amdgcn_s_waitcnt(0x0F70);               // the staging copy of the next tile has landed
        } else {
            const int v = T.view;
            const ViewDesc &V = views[v];
            const int x = T.x0 + 4 * tx;
            const uint8_t *sp = src.p[f * n_views + v];
            const unsigned sstep = src.step[f * n_views + v];
            const uint8_t *lb = lds + buf * WA_BUF_BYTES;
            const unsigned pitch = 16u * (unsigned)warp_lds_np(T.sw);
            const unsigned a0 = (unsigned)(((uintptr_t)sp + (size_t)T.sy0 * sstep + 3u * (unsigned)T.sx0) & 15u), astep = sstep & 15u;
            int ys[WA_NG];
            bool active[WA_NG];
#pragma unroll
            for (int g = 0; g < WA_NG; ++g) { ys[g] = T.y0 + ty + g * WA_BY; active[g] = x < V.pw && ys[g] < V.ph; }
            float xc[WA_NG][4], yc[WA_NG][4];
            Px2 r1[WA_NG][4], r2[WA_NG][4];
            bool slow = false;
#pragma unroll
            for (int g = 0; g < WA_NG; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (active[g]) warp_combine(PROJ, ct[k], rt[g], V.wp, xc[g][k], yc[g][k]);
                    else xc[g][k] = yc[g][k] = -1.f;
                    const int x1 = f2i_rd(xc[g][k]), y1 = f2i_rd(yc[g][k]);
                    const bool fast = (unsigned)x1 < (unsigned)(scols - 2) && (unsigned)y1 < (unsigned)(srows - 1);
                    slow = slow || (active[g] && !fast);
                    // inside the staged box for every sample the box was built from (active and fast); clamped so that the others read valid LDS
                    const int lx = min(max(x1 - (int)T.sx0, 0), (int)T.sw - 2), ly = min(max(y1 - (int)T.sy0, 0), (int)T.sh - 2);
                    const unsigned o1 = (unsigned)ly * pitch + ((a0 + (unsigned)ly * astep) & 15u) + 3u * (unsigned)lx;
                    const unsigned o2 = (unsigned)(ly + 1) * pitch + ((a0 + (unsigned)(ly + 1) * astep) & 15u) + 3u * (unsigned)lx;
                    r1[g][k] = lds_px2(lb, o1);
                    r2[g][k] = lds_px2(lb, o2);
                }
            if (__builtin_amdgcn_ballot_w64(slow)) {            // a tap outside the image somewhere in the wave: those samples re-read from global memory
#pragma unroll
                for (int g = 0; g < WA_NG; ++g)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const Taps tt = make_taps(xc[g][k], yc[g][k], srows, scols);
                        if (active[g] && !tt.fast) {
                            const unsigned off = tap_offset(tt.x1, tt.y1, srows, scols, sstep);
                            r1[g][k] = load_px2(sp, off);
                            r2[g][k] = load_px2(sp + sstep, off);
                            fix_border_taps(r1[g][k], r2[g][k], tt.x1, tt.y1, srows, scols);
                        }
                    }
            }
            const LevelDesc &L = V.lv[0];
            const size_t plane = (size_t)L.h * L.pitch;
            unsigned packed[WA_NG][3];
#pragma unroll
            for (int g = 0; g < WA_NG; ++g) {
                packed[g][0] = packed[g][1] = packed[g][2] = 0u;
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    float o[2][3];
                    const Taps t0 = make_taps(xc[g][k], yc[g][k], srows, scols), t1 = make_taps(xc[g][k + 1], yc[g][k + 1], srows, scols);
                    blend_taps2(t0, t1, r1[g][k], r2[g][k], r1[g][k + 1], r2[g][k + 1], o[0], o[1]);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            packed[g][c] = sat_u8_into(__builtin_fmaf(V.gain, (float)sat_u8(o[j][c]), 0.f), k + j, packed[g][c]);
                }
            }
            // everything this wave has asked of the vector memory path has landed: the next tile's tables and staging copy (they had this
            // tile's sampling to arrive) and, long ago, the previous tile's stores.  Only then are this tile's stores issued: they stay
            // in flight across the next iteration.
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int g = 0; g < WA_NG; ++g)
                if (active[g]) {
                    uint8_t *d = g0 + (size_t)f * g0_stride + L.off + (size_t)ys[g] * L.pitch + x;
                    *reinterpret_cast<unsigned *>(d) = packed[g][0];
                    *reinterpret_cast<unsigned *>(d + plane) = packed[g][1];
                    *reinterpret_cast<unsigned *>(d + 2 * plane) = packed[g][2];
                }
        }
        if (!more) break;
        T = Tn; t = tn; f = fn; Tn = Tnn; tn = tnn; fn = fnn; buf ^= 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) ct[k] = ctn[k];
#pragma unroll
        for (int g = 0; g < WA_NG; ++g) rt[g] = rtn[g];
    }
}

// ---- CPW stage 1: images[i] = gain(remap(full_img, x_map, y_map)) (timed.cpp:90-94), 4 px per lane --------------
// Same sampling code as k_warp_t without the reflect pad; interleaved 8UC3 output (the stage-2 remap samples it).  One row of 4 pixels per lane,
// S1_NF frames per lane: the projection coordinates and bilinear weights are built once and used for both frames (as in warp_tile_direct), the tap
// reads of frame fi + 1 are in flight while frame fi is blended.
constexpr int S1_BY = WARP_TH;          // lane rows of the block (the launch uses the same): one tile row per lane
template <int PROJ, bool AL, int S1_NF>
__device__ __forceinline__ void stage1_tile(const WarpTile &T, int f0, int nf, const ViewDesc *__restrict__ views, int n_views,
                                            const SrcTable &src, int srows, int scols, uint8_t *__restrict__ stage, long long stage_stride)
{
    const int v = T.view;
    const ViewDesc &V = views[v];
    const int x = T.x0 + 4 * (int)threadIdx.x, y = T.y0 + (int)threadIdx.y;
    if (x >= V.aw || y >= V.ah) return;
    const uint8_t *sp[S1_NF];
    unsigned sstep[S1_NF];
#pragma unroll
    for (int fi = 0; fi < S1_NF; ++fi) {
        const int f = f0 + (fi < nf ? fi : 0);
        sp[fi] = src.p[f * n_views + v]; sstep[fi] = src.step[f * n_views + v];
    }
    float2 ct[4];
    if (x + 3 < V.aw) {
        float4 a, b;
        a = gload_f4(V.coltab + x);
        b = gload_f4(V.coltab + x + 2);
        ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) ct[k] = gload_f2(V.coltab + min(x + k, V.aw - 1));
    }
    const float2 rt = gload_f2(V.rowtab + y);
    float xc[4], yc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) warp_combine(PROJ, ct[k], rt, V.wp, xc[k], yc[k]);
    Px2 r1[2][4], r2[2][4];
    Px3 q1[AL ? 2 : 1][AL ? 4 : 1], q2[AL ? 2 : 1][AL ? 4 : 1];      // AL: aligned 12-byte tap reads, as in warp_tile_direct
    unsigned sh1[2] = {0u, 0u}, sh2[2] = {0u, 0u};
    auto issue = [&](int fi) {      // (unconditional, also for the missing second frame of an odd batch: see warp_tile_direct)
        const int b = fi & 1;
        const uint8_t *spf = sp[fi];
        const unsigned stf = sstep[fi], sp_lo = (unsigned)(uintptr_t)spf;
        const ms_gptr_u8 sp_al = (ms_gptr_u8)((uintptr_t)spf & ~(uintptr_t)3);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned off = tap_offset(f2i_rd(xc[k]), f2i_rd(yc[k]), srows, scols, stf);
            if (AL) {
                const unsigned a = (sp_lo & 3u) + off;
                q1[AL ? b : 0][AL ? k : 0] = load_px3_at(sp_al, a);
                q2[AL ? b : 0][AL ? k : 0] = load_px3_at(sp_al, a + stf);
                if (k == 0) { sh1[b] = a & 3u; sh2[b] = (a + stf) & 3u; }
                else { sh1[b] |= (a & 3u) << (2 * k); sh2[b] |= ((a + stf) & 3u) << (2 * k); }
            } else {
                r1[b][k] = load_px2(spf, off);
                r2[b][k] = load_px2(spf + stf, off);
            }
        }
    };
    issue(0);
    __builtin_amdgcn_sched_barrier(0);      // (frame 0's reads first: see warp_tile_direct)
#pragma unroll
    for (int fi = 0; fi < S1_NF; ++fi) {
        const int b = fi & 1;
        if (fi + 1 < S1_NF) issue(fi + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (fi >= nf) continue;
        unsigned w[3] = {0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            float o[2][3];
            Taps t[2];
            if (AL) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    r1[b][k + j] = px3_to_px2(q1[AL ? b : 0][AL ? k + j : 0], sh1[b] >> (2 * (k + j)));
                    r2[b][k + j] = px3_to_px2(q2[AL ? b : 0][AL ? k + j : 0], sh2[b] >> (2 * (k + j)));
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                t[j] = make_taps(xc[k + j], yc[k + j], srows, scols);
                if (!t[j].fast) fix_border_taps(r1[b][k + j], r2[b][k + j], t[j].x1, t[j].y1, srows, scols);
            }
            blend_taps2(t[0], t[1], r1[b][k], r2[b][k], r1[b][k + 1], r2[b][k + 1], o[0], o[1]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x2 r = gain_pair(V.gain, (float)sat_u8(o[0][c]), (float)sat_u8(o[1][c]));
                const int i0 = 3 * k + c, i1 = 3 * (k + 1) + c;          // bytes of the 12 interleaved output bytes
                w[i0 >> 2] = sat_u8_into(r.x, i0 & 3, w[i0 >> 2]);
                w[i1 >> 2] = sat_u8_into(r.y, i1 & 3, w[i1 >> 2]);
            }
        }
        uint8_t *d = stage + (size_t)(f0 + fi) * stage_stride + V.s1_off + (size_t)y * V.s1_pitch + (size_t)x * 3;   // 12 B per lane, dword aligned
        if (x + 3 < V.aw) __builtin_memcpy(__builtin_assume_aligned(d, 4), w, 12);
        else {
            for (int k = 0; k < 4 && x + k < V.aw; ++k)
                for (int c = 0; c < 3; ++c) { const int i = 3 * k + c; d[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3))); }
        }
    }
}

// S1_NF frames per lane (coordinates, offsets and weights built once per lane): 2 where the remap is gather-bound (config 3: 338 / 344 / 358 us per 24 frames with 2 / 3 / 4),
// 3 where the source is sampled about 1 : 1 and the kernel is VALU-bound (the shipped cylindrical rig: 494 / 438 / 430): ms_ctx::stage1_nf picks by the tiles' minification
template <int PROJ, bool AL, int S1_NF>
__global__ void __launch_bounds__(WARP_BX * S1_BY) k_stage1_t(const WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int n_views,
                                                             SrcTable src, int srows, int scols, uint8_t *__restrict__ stage, long long stage_stride, DispTable disp, int n_frames)
{
    const WarpTile T = tiles[blockIdx.x];
    // the mesh of this view moves no sample further than the bound the plan assumed: stage 2 never reads this tile
    if (!(T.flags & 2) && *disp.p[T.view] <= disp.limit_bits) return;
    const int f0 = (int)blockIdx.z * S1_NF, nf = min(S1_NF, n_frames - f0);
    if (AL && (T.flags & 8)) stage1_tile<PROJ, true, S1_NF>(T, f0, nf, views, n_views, src, srows, scols, stage, stage_stride);
    else stage1_tile<PROJ, false, S1_NF>(T, f0, nf, views, n_views, src, srows, scols, stage, stage_stride);
}

// the shared-offset form of stage1_tile (see warp_tile_shared): aligned reads, offsets / shifts / weights / border flag once per pixel, per frame only the SGPR base
template <int PROJ, int S1_NF>
__device__ __forceinline__ void stage1_tile_shared(const WarpTile &T, int f0, int nf, const ViewDesc *__restrict__ views, int n_views,
                                                   const SrcTable &src, int srows, int scols, uint8_t *__restrict__ stage, long long stage_stride)
{
    MS_PRIO_LOADS(MS_PRIO_S1);
    const int v = T.view;
    const ViewDesc &V = views[v];
    const int x = T.x0 + 4 * (int)threadIdx.x, y = T.y0 + (int)threadIdx.y;
    if (x >= V.aw || y >= V.ah) return;
    ms_gptr_u8 base[S1_NF];
#pragma unroll
    for (int fi = 0; fi < S1_NF; ++fi) base[fi] = (ms_gptr_u8)((uintptr_t)src.p[(f0 + (fi < nf ? fi : 0)) * n_views + v] & ~(uintptr_t)3);
    const unsigned lo2 = (unsigned)(uintptr_t)src.p[f0 * n_views + v] & 3u, st = src.step[f0 * n_views + v];
    float2 ct[4];
    if (x + 3 < V.aw) {
        float4 a, b;
        a = gload_f4(V.coltab + x);
        b = gload_f4(V.coltab + x + 2);
        ct[0] = make_float2(a.x, a.y); ct[1] = make_float2(a.z, a.w); ct[2] = make_float2(b.x, b.y); ct[3] = make_float2(b.z, b.w);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) ct[k] = gload_f2(V.coltab + min(x + k, V.aw - 1));
    }
    const float2 rt = gload_f2(V.rowtab + y);
    float xc[4], yc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) warp_combine(PROJ, ct[k], rt, V.wp, xc[k], yc[k]);
    unsigned va[4], vb[4], sh1 = 0u, sh2 = 0u;
    bool slow = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x1 = f2i_rd(xc[k]), y1 = f2i_rd(yc[k]);
        slow = slow || !((unsigned)x1 < (unsigned)(scols - 2) && (unsigned)y1 < (unsigned)(srows - 1));
        const unsigned a = lo2 + tap_offset(x1, y1, srows, scols, st), b = a + st;
        va[k] = a & ~3u; vb[k] = b & ~3u;
        sh1 |= (a & 3u) << (2 * k); sh2 |= (b & 3u) << (2 * k);
    }
    Px3 q1[2][4], q2[2][4];
    auto issue = [&](int fi) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#ifdef MS_PROBE_TAPS
            if ((k % MS_PROBE_TAPS) != 0) { q1[fi & 1][k] = q1[fi & 1][k - k % MS_PROBE_TAPS]; q2[fi & 1][k] = q2[fi & 1][k - k % MS_PROBE_TAPS]; continue; }
#endif
            const ms_u32x3_a4 r1 = *(const MS_GLOBAL_AS ms_u32x3_a4 *)(base[fi] + va[k]);
            const ms_u32x3_a4 r2 = *(const MS_GLOBAL_AS ms_u32x3_a4 *)(base[fi] + vb[k]);
            q1[fi & 1][k] = Px3{r1.x, r1.y, r1.z};
            q2[fi & 1][k] = Px3{r2.x, r2.y, r2.z};
        }
    };
    issue(0);
    __builtin_amdgcn_sched_barrier(0);
    if (S1_NF > 1) issue(1);
    __builtin_amdgcn_sched_barrier(0);
    MS_PRIO_MATH(MS_PRIO_S1);
    Taps t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = make_taps(xc[k], yc[k], srows, scols);
    const bool any_slow = __builtin_amdgcn_ballot_w64(slow) != 0ull;
    const float gain = V.gain;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int fi = 0; fi < S1_NF; ++fi) {
        const int b = fi & 1;
        if (fi < nf) {
            unsigned w[3] = {0u, 0u, 0u};
            Px2 r1[4], r2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r1[k] = px3_to_px2(q1[b][k], sh1 >> (2 * k));
                r2[k] = px3_to_px2(q2[b][k], sh2 >> (2 * k));
            }
            if (any_slow) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (!t[k].fast) fix_border_taps(r1[k], r2[k], t[k].x1, t[k].y1, srows, scols);
            }
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                float o[2][3];
                blend_taps2(t[k], t[k + 1], r1[k], r2[k], r1[k + 1], r2[k + 1], o[0], o[1]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x2 r = gain_pair(gain, (float)sat_u8(o[0][c]), (float)sat_u8(o[1][c]));
                    const int i0 = 3 * k + c, i1 = 3 * (k + 1) + c;          // bytes of the 12 interleaved output bytes
                    w[i0 >> 2] = sat_u8_into(r.x, i0 & 3, w[i0 >> 2]);
                    w[i1 >> 2] = sat_u8_into(r.y, i1 & 3, w[i1 >> 2]);
                }
            }
            uint8_t *d = stage + (size_t)(f0 + fi) * stage_stride + V.s1_off + (size_t)y * V.s1_pitch + (size_t)x * 3;   // 12 B per lane, dword aligned
            if (x + 3 < V.aw) __builtin_memcpy(__builtin_assume_aligned(d, 4), w, 12);
            else {
                for (int k = 0; k < 4 && x + k < V.aw; ++k)
                    for (int c = 0; c < 3; ++c) { const int i = 3 * k + c; d[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3))); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (fi + 2 < S1_NF) { MS_PRIO_RELOADS(MS_PRIO_S1); issue(fi + 2); __builtin_amdgcn_sched_barrier(0); MS_PRIO_MATH(MS_PRIO_S1); }
    }
}

template <int PROJ, int S1_NF>
__global__ void __launch_bounds__(WARP_BX * S1_BY) k_stage1_s(const WarpTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int n_views,
                                                             SrcTable src, int srows, int scols, uint8_t *__restrict__ stage, long long stage_stride, DispTable disp, int n_frames)
{
    const WarpTile T = tiles[blockIdx.x];
    if (!(T.flags & 2) && *disp.p[T.view] <= disp.limit_bits) return;
    const int f0 = (int)blockIdx.z * S1_NF, nf = min(S1_NF, n_frames - f0);
    if (T.flags & 8) {
        if (S1_NF == 3 && nf == 2) stage1_tile_shared<PROJ, 2>(T, f0, nf, views, n_views, src, srows, scols, stage, stage_stride);      // (a short last group: see k_warp_s)
        else if (S1_NF >= 2 && nf == 1) stage1_tile_shared<PROJ, 1>(T, f0, nf, views, n_views, src, srows, scols, stage, stage_stride);
        else stage1_tile_shared<PROJ, S1_NF>(T, f0, nf, views, n_views, src, srows, scols, stage, stage_stride);
    } else stage1_tile<PROJ, false, S1_NF>(T, f0, nf, views, n_views, src, srows, scols, stage, stage_stride);
}

// ---- pyrDown, tile list, DOWN_ROWS (4) rows x 4 cols per lane (block 32 x 8): 11 input rows for 4 output rows (2 rows per lane: 7 for 2, 16 % slower) ----
// Packed 16-bit arithmetic: every input value is in [0,255] (8-bit pixels, or a Gaussian level of them -- see the invariant at
// up_2x8_pk), so the vertical sums are <= 16*255 and the full 5x5 sums <= 256*255 = 65280 < 2^16: two columns share one
// register and plain 32-bit adds never carry between the halves.  Same result as rne_shift(sum, 8) per pixel
// (pyr_down.cu:55-174: the fp32 sums are exact, saturate_cast<short> rounds half-even).
// A lane needs input columns 8t-2 .. 8t+8 = taps v0..v10; "even" taps E_i = v_{2i} (i = 0..5), "odd" taps O_i = v_{2i+1} (i = 0..4);
//   out_i = E_i + 6 E_{i+1} + E_{i+2} + 4 (O_i + O_{i+1}),  i = 0..3.
struct Down7 { unsigned a[7]; };     // one input row (or a vertical sum of rows): see the two loaders for the layouts
// u8 row, window bytes 8t-4 .. 8t+11 (d0..d3): a0 = (x,E0) a1 = (E1,E2) a2 = (E3,E4) a3 = (E5,x) a4 = (x,O0) a5 = (O1,O2) a6 = (O3,O4)
__device__ __forceinline__ Down7 down_row(const Row11u8 &o, int t, int w)
{
    unsigned d0 = o.b.x;
    const unsigned d1 = o.b.y, d2 = o.b.z, d3 = o.b.w;
    if (t == 0) d0 = __builtin_amdgcn_perm(0u, d1, 0x01020c0cu);     // columns -2, -1 (taps v0, v1) mirror to columns 2, 1 (BORDER_REFLECT_101): byte2 <- col 2, byte3 <- col 1
    Down7 r;
    r.a[0] = d0 & 0x00ff00ffu; r.a[1] = d1 & 0x00ff00ffu; r.a[2] = d2 & 0x00ff00ffu; r.a[3] = d3 & 0x00ff00ffu;
    r.a[4] = (d0 >> 8) & 0x00ff00ffu; r.a[5] = (d1 >> 8) & 0x00ff00ffu; r.a[6] = (d2 >> 8) & 0x00ff00ffu;
    if (8 * t + 8 >= w) r.a[3] = r.a[2] >> 16;              // column w mirrors to w-2: E5 = E4
    return r;
}
__device__ __forceinline__ unsigned rne8_pk(unsigned s)
{
    const unsigned t = pk_lshr16(s, 8) & 0x00010001u;
    return pk_lshr16(s + t + 0x007f007fu, 8);          // (sums <= 65280 + 128: the halves stay below 2^16)
}
// horizontal pass on a vertical sum; returns the four outputs as bytes of one dword (each is <= 255)
__device__ __forceinline__ unsigned down_hpass(const Down7 &v)
{
    const unsigned e01 = __builtin_amdgcn_alignbyte(v.a[1], v.a[0], 2), e12 = v.a[1], e23 = __builtin_amdgcn_alignbyte(v.a[2], v.a[1], 2);
    const unsigned e34 = v.a[2], e45 = __builtin_amdgcn_alignbyte(v.a[3], v.a[2], 2);
    const unsigned o01 = __builtin_amdgcn_alignbyte(v.a[5], v.a[4], 2), o12 = v.a[5], o23 = __builtin_amdgcn_alignbyte(v.a[6], v.a[5], 2), o34 = v.a[6];
    const unsigned s01 = mad6(e12, e01 + e23 + 4u * (o01 + o12));
    const unsigned s23 = mad6(e34, e23 + e45 + 4u * (o23 + o34));
    return __builtin_amdgcn_perm(rne8_pk(s23), rne8_pk(s01), 0x06040200u);
}

// Input and output are both byte planes (level 0 = the gained warp output, levels >= 1 = the view's Gaussian levels: all in [0, 255]).
// L0 only names the launch (level 0 vs the coarser levels) so that kernel traces tell them apart; the code is the same.
template <bool L0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) k_down_t(const DownTile *__restrict__ tiles, const ViewDesc *__restrict__ views, int l,
                                                const uint8_t *__restrict__ gin, long long in_stride,
                                                uint8_t *__restrict__ gout, long long out_stride)
{
    const DownTile T = tiles[blockIdx.x];
    const int c = blockIdx.y, f = blockIdx.z, v = T.view;
    const int ty = (int)threadIdx.y;              // (one-wave workgroups, as in k_blend8, change nothing here: 68.6 vs 68.4 us)
    const LevelDesc &Li = views[v].lv[l], &Lo = views[v].lv[l + 1];
    constexpr int RO = DOWN_ROWS, RI = 2 * RO + 3;          // output rows per lane, input rows they need
    const int t = (T.x0 >> 2) + (int)threadIdx.x;
    const int y = T.y0 + RO * ty;
    if (4 * t >= Lo.w || y >= Lo.h) return;
    const uint8_t *in = gin + (size_t)f * in_stride + Li.off + (size_t)c * Li.h * Li.pitch;
    const int nrow = min(RO, Lo.h - y);                     // valid output rows of this lane
    const int sy = 2 * y, last = Li.h - 1;
    int ridx[RI];
    ridx[0] = abs(sy - 2); ridx[1] = abs(sy - 1); ridx[2] = sy;
#pragma unroll
    for (int j = 3; j < RI; ++j) { const int r = sy + j - 2; ridx[j] = r > last ? 2 * last - r : r; }
#pragma unroll
    for (int j = 5; j < RI; ++j) if (j > 2 * nrow + 2) ridx[j] = ridx[4];      // rows only the missing outputs would read: any valid row
    Row11u8 raw[RI];
#pragma unroll
    for (int j = 0; j < RI; ++j) raw[j] = fetch_row11(in + (size_t)ridx[j] * Li.pitch, t, Li.w);
    Down7 r[RI];
#pragma unroll
    for (int j = 0; j < RI; ++j) r[j] = down_row(raw[j], t, Li.w);
    uint8_t *out = gout + (size_t)f * out_stride + Lo.off + (size_t)c * Lo.h * Lo.pitch + (size_t)y * Lo.pitch + 4 * t;
#pragma unroll
    for (int o = 0; o < RO; ++o) {                          // vertical 1 4 6 4 1 of input rows 2o .. 2o+4
        Down7 vv;
#pragma unroll
        for (int k = 0; k < 7; ++k) vv.a[k] = mad6(r[2 * o + 2].a[k], (r[2 * o].a[k] + r[2 * o + 4].a[k]) + 4u * (r[2 * o + 1].a[k] + r[2 * o + 3].a[k]));
        if (o < nrow) *reinterpret_cast<unsigned *>(out + (size_t)o * Lo.pitch) = down_hpass(vv);
    }
}

}  // namespace ms
