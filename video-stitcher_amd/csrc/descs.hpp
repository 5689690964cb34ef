// descs.hpp -- device-visible descriptor tables of the compositor (static after calibration).
#pragma once
#include "common.hpp"

namespace ms {

constexpr int MAX_LEVELS = 8;     // num_bands <= 7
constexpr int MAX_VIEWS = 16;
constexpr int MAX_SRC = 192;      // frames * views of ONE by-value source table (a launch of the kernels that read the callers' frames: stitch_impl sends them out in chunks)
constexpr int MAX_FRAMES = 64;    // frames per ms_stitch call

struct LevelDesc {
    int w, h, pitch;              // level size; pitch in elements
    int x_tl, y_tl;               // position inside the padded pano at this level
    long long off;                // element offset of plane 0 inside the per-frame pyramid buffer
    const float *wgt;             // weight pyramid level (static)
    int wpitch;
};
struct ViewDesc {
    int aw, ah;                   // warped view size (warpRoi)
    int top, left;                // reflect border
    int pw, ph;                   // padded size
    float gain;
    const float *xmap, *ymap;     // projection maps (static), pitch in elements
    int map_pitch;
    const float2 *coltab, *rowtab; // 1-D terms of the backward map (per ROI column / row); the fused path rebuilds
    WarpParams wp;                 // the coordinates from these (bit-identical to xmap/ymap) instead of reading 8 B/px
    int proj;
    const uint8_t *wm0;           // padded 8-bit mask = level-0 weight before the 1/255 scale (static)
    int wm0_pitch;
    long long s1_off;             // byte offset of the CPW stage-1 image inside the per-frame stage buffer
    int s1_pitch;                 // its row pitch in bytes (multiple of 4)
    LevelDesc lv[MAX_LEVELS];
};
struct PanoDesc {
    int nb, n_views;
    int qw[MAX_LEVELS], qh[MAX_LEVELS], qpitch[MAX_LEVELS];
    long long coff[MAX_LEVELS];   // element offset of collapsed level l (l >= 1) in the per-frame buffer
    long long poff[MAX_LEVELS];   // element offset of level l (l >= 0) in a per-frame PARTIAL accumulator buffer (view sharding)
    const float *den[MAX_LEVELS]; // sum_v w_v + 1e-5f (static)
    int dpitch[MAX_LEVELS];
    float alpha;                  // (float)(1./255.): level-0 weight = fmaf(alpha, mask, 0)
    const uint8_t *mask;          // gpu_dst_mask_ over dst_roi_final
    int mask_pitch;
    int fw, fh;                   // dst_roi_final size
    int canvas_x, canvas_y, out_w, out_h;
    int i_y0, i_rows;             // even-aligned canvas rows covering the pano ROI: what the I420 output holds
    // per level: one byte per 64 x 16 cell of the pano = the view that OWNS it (exactly one view has non-zero weights there, all of them
    // exactly 1.0f, hence w_sum + 1e-5 == 1.00001f), or 255.  nullptr where the level has no such map.
    const uint8_t *pure[MAX_LEVELS];
    int ppitch[MAX_LEVELS];
};
// View sharding (BASELINE configs[4]): the weighted accumulation over views is a sum of int16 terms, so a
// rank that owns a subset of the views writes its partial sums (mode 1) and the sink adds the partials of all ranks
// (wrap-around int16, order independent => bit-identical to the single-GPU result), then normalises and collapses (mode 2).
struct ShardArgs {
    int mode;                     // 0 = whole frame on this GPU, 1 = write partial sums of the owned views, 2 = finish from partials
    unsigned own_mask;            // views this rank accumulates (modes 0/1)
    int n_parts;                  // mode 2: number of partial buffers
    const int16_t *part[4];
    int16_t *pout;                // mode 1: where the partial sums go
    long long pstride;            // elements per frame in a partial buffer
};
struct SrcTable { const uint8_t *p[MAX_SRC]; unsigned step[MAX_SRC]; };
struct SrcAll { const uint8_t *p[MAX_FRAMES * MAX_VIEWS]; unsigned step[MAX_FRAMES * MAX_VIEWS]; };      // host side: every frame of a call
struct MeshTable { const float *x[MAX_VIEWS]; const float *y[MAX_VIEWS]; int pitch[MAX_VIEWS]; };
// max |mesh map - identity| of the active mesh of each view, as float bits in device memory (written by ms_set_mesh), and the bound
// under which CPW stage 1 may skip the tiles stage 2 cannot reach (WarpTile::flags bit 1 = reachable within that bound)
struct DispTable { const unsigned *p[MAX_VIEWS]; unsigned limit_bits; };
struct OutTable { uint8_t *p8[MAX_FRAMES]; unsigned step8[MAX_FRAMES]; int16_t *p16[MAX_FRAMES]; unsigned step16[MAX_FRAMES];
                  uint8_t *pi[MAX_FRAMES]; };       // pi: planar I420 of the canvas rows [i_y0, i_y0 + i_rows) (ms_stitch_i420)


// ---- work lists (built once in ms_init_blender, plan.cpp-style host code in compositor.hip) ----------
// Every per-frame kernel is driven by a list of tiles that are actually needed: tiles of a view whose
// weights are zero at every band (most of the +-pi-straddling view, the seam-cut outer parts of the others)
// are never produced or read.  Skipping them is exact: a zero weight contributes trunc(L * 0) == 0.
// out tile origin + source bbox (LDS-staged variant) + where the tile's projection tables start.  flags: bit 0 = bbox fits the LDS budget,
// bit 1 = (CPW stage 1) reachable tile, bit 2 = every pixel of the tile lies inside the warped view in x and y (no reflect padding),
// in which case ctab / rtab index the column / row terms of the tile's first column / row in the context's table buffer
struct WarpTile { short view, flags; short x0, y0; short sx0, sy0, sw, sh; int ctab, rtab; };
struct DownTile { short view, pad; short x0, y0; };                              // output (level l+1) tile origin
struct BlendTile { short x0, y0; unsigned view_mask; };                          // pano tile origin + contributing views

#ifndef MS_WARP_TW
#define MS_WARP_TW 32
#define MS_WARP_TH 16
#endif
constexpr int WARP_TW = MS_WARP_TW, WARP_TH = MS_WARP_TH;      // k_warp_t tile (4 px per lane)
constexpr int WARP_BX = WARP_TW / 4;            // lanes across a tile row
#ifndef MS_DOWN_ROWS
#define MS_DOWN_ROWS 4
#endif
constexpr int DOWN_ROWS = MS_DOWN_ROWS;                      // output rows per lane of k_down_t (2 r + 3 input rows are read for r output rows)
constexpr int DOWN_TW = 128, DOWN_TH = 8 * DOWN_ROWS;       // k_down_t output tile (4 x DOWN_ROWS px per thread, 32 x 8 threads)
constexpr int BLEND_TW = 256, BLEND_TH = 16;   // k_blend8_t tile (8 x 2 px per thread, 32 x 8 threads)

}  // namespace ms
