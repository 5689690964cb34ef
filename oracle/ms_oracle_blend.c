/*
 * ms_oracle_blend.c -- CPU ORACLE (test infrastructure only; see ms_oracle.h header).
 * Calibration-time helpers (distance transform, Voronoi seams, CPW mesh -> backward map) and the
 * fork's GPU MultiBandBlender {prepare, init_gpu, feed_online, blend(gpuOut)} composed from the
 * per-kernel restatements in ms_oracle_prims.c, in the reference's call order.
 */
#include "ms_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * cv::distanceTransform(src, dst, DIST_L1, 3) -> distanceTransform_3x3 with mask (1, 2)
 * OCV/imgproc/src/distransform.cpp:46-48 (DIST_SHIFT 16, INIT_DIST0), :70-137 (two passes)
 */
void orc_distance_transform_l1(const uint8_t *src, size_t sstep, int rows, int cols, float *dst, size_t dstep)
{
    const int DIST_SHIFT = 16;
    const int INIT_DIST0 = (INT_MAX >> 2);
    const int HV = 1 << DIST_SHIFT, DG = 2 << DIST_SHIFT;
    const float scale = 1.f / (1 << DIST_SHIFT);
    const int step = cols + 2;
    int *temp = (int *)malloc(sizeof(int) * (size_t)step * (size_t)(rows + 2));
    for (int j = 0; j < step; ++j) { temp[j] = INIT_DIST0; temp[(size_t)(rows + 1) * step + j] = INIT_DIST0; }
    for (int i = 0; i < rows; ++i) {
        const uint8_t *s = src + (size_t)i * sstep;
        int *tmp = temp + (size_t)(i + 1) * step + 1;
        tmp[-1] = tmp[cols] = INIT_DIST0;
        for (int j = 0; j < cols; ++j) {
            if (!s[j]) tmp[j] = 0;
            else {
                int t0 = tmp[j - step - 1] + DG;
                int t = tmp[j - step] + HV; if (t0 > t) t0 = t;
                t = tmp[j - step + 1] + DG; if (t0 > t) t0 = t;
                t = tmp[j - 1] + HV; if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
        }
    }
    for (int i = rows - 1; i >= 0; --i) {
        float *d = (float *)((char *)dst + (size_t)i * dstep);
        int *tmp = temp + (size_t)(i + 1) * step + 1;
        for (int j = cols - 1; j >= 0; --j) {
            int t0 = tmp[j];
            if (t0 > HV) {
                int t = tmp[j + step + 1] + DG; if (t0 > t) t0 = t;
                t = tmp[j + step] + HV; if (t0 > t) t0 = t;
                t = tmp[j + step - 1] + DG; if (t0 > t) t0 = t;
                t = tmp[j + 1] + HV; if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
            d[j] = (float)(t0 * scale);
        }
    }
    free(temp);
}

/* overlapRoi  OCV/stitching/src/util.cpp:100-112 */
static int overlap_roi(int x1, int y1, int x2, int y2, int w1, int h1, int w2, int h2, orc_rect *roi)
{
    int x_tl = x1 > x2 ? x1 : x2, y_tl = y1 > y2 ? y1 : y2;
    int x_br = (x1 + w1 < x2 + w2) ? x1 + w1 : x2 + w2;
    int y_br = (y1 + h1 < y2 + h2) ? y1 + h1 : y2 + h2;
    if (x_tl < x_br && y_tl < y_br) { roi->x = x_tl; roi->y = y_tl; roi->width = x_br - x_tl; roi->height = y_br - y_tl; return 1; }
    return 0;
}

/* VoronoiSeamFinder::findInPair  OCV/stitching/src/seam_finders.cpp:111-160 */
static void voronoi_pair(uint8_t *mask1, int w1, int h1, int tl1x, int tl1y,
                         uint8_t *mask2, int w2, int h2, int tl2x, int tl2y, orc_rect roi)
{
    const int gap = 10;
    const int R = roi.height + 2 * gap, C = roi.width + 2 * gap;
    uint8_t *sub1 = (uint8_t *)malloc((size_t)R * C), *sub2 = (uint8_t *)malloc((size_t)R * C);
    uint8_t *z1 = (uint8_t *)malloc((size_t)R * C), *z2 = (uint8_t *)malloc((size_t)R * C);
    float *d1 = (float *)malloc(sizeof(float) * (size_t)R * C), *d2 = (float *)malloc(sizeof(float) * (size_t)R * C);
    for (int y = -gap; y < roi.height + gap; ++y)
        for (int x = -gap; x < roi.width + gap; ++x) {
            int y1 = roi.y - tl1y + y, x1 = roi.x - tl1x + x;
            sub1[(size_t)(y + gap) * C + x + gap] =
                (y1 >= 0 && x1 >= 0 && y1 < h1 && x1 < w1) ? mask1[(size_t)y1 * w1 + x1] : 0;
            int y2 = roi.y - tl2y + y, x2 = roi.x - tl2x + x;
            sub2[(size_t)(y + gap) * C + x + gap] =
                (y2 >= 0 && x2 >= 0 && y2 < h2 && x2 < w2) ? mask2[(size_t)y2 * w2 + x2] : 0;
        }
    /* collision = (s1 != 0) & (s2 != 0); unique = sub with collision zeroed;
     * distanceTransform(unique == 0): zero pixels of the *input* are where unique != 0 */
    for (size_t i = 0; i < (size_t)R * C; ++i) {
        const int col = (sub1[i] != 0) && (sub2[i] != 0);
        const uint8_t u1 = col ? 0 : sub1[i], u2 = col ? 0 : sub2[i];
        z1[i] = (u1 == 0) ? 255 : 0;
        z2[i] = (u2 == 0) ? 255 : 0;
    }
    orc_distance_transform_l1(z1, (size_t)C, R, C, d1, sizeof(float) * (size_t)C);
    orc_distance_transform_l1(z2, (size_t)C, R, C, d2, sizeof(float) * (size_t)C);
    for (int y = 0; y < roi.height; ++y)
        for (int x = 0; x < roi.width; ++x) {
            const size_t i = (size_t)(y + gap) * C + x + gap;
            if (d1[i] < d2[i]) mask2[(size_t)(roi.y - tl2y + y) * w2 + (roi.x - tl2x + x)] = 0;
            else mask1[(size_t)(roi.y - tl1y + y) * w1 + (roi.x - tl1x + x)] = 0;
        }
    free(sub1); free(sub2); free(z1); free(z2); free(d1); free(d2);
}

/* PairwiseSeamFinder::run  seam_finders.cpp:71-83 */
void orc_voronoi_seams(int n, const int *cx, const int *cy, const int *w, const int *h, uint8_t **masks)
{
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            orc_rect roi;
            if (overlap_roi(cx[i], cy[i], cx[j], cy[j], w[i], h[i], w[j], h[j], &roi))
                voronoi_pair(masks[i], w[i], h[i], cx[i], cy[i], masks[j], w[j], h[j], cx[j], cy[j], roi);
        }
}

/* MeshWarper::convertMeshesToMap (one view)  APP/meshwarper.cpp:823-886 */
void orc_convert_mesh_to_map(const float *mesh_x, const float *mesh_y, int N, int M,
                             int width, int height, float *map_x, float *map_y)
{
    const int scale = 2;
    const int hw = width / scale, hh = height / scale;
    float *big_x = (float *)malloc(sizeof(float) * (size_t)width * height);
    float *big_y = (float *)malloc(sizeof(float) * (size_t)width * height);
    orc_custom_resize_32f(mesh_x, sizeof(float) * (size_t)M, N, M, big_x, sizeof(float) * (size_t)width, height, width);
    orc_custom_resize_32f(mesh_y, sizeof(float) * (size_t)M, N, M, big_y, sizeof(float) * (size_t)width, height, width);
    float *sum_x = (float *)calloc((size_t)hw * hh, sizeof(float));
    float *sum_y = (float *)calloc((size_t)hw * hh, sizeof(float));
    float *cnt = (float *)calloc((size_t)hw * hh, sizeof(float));
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const float fx = big_x[(size_t)y * width + x], fy = big_y[(size_t)y * width + x];
            /* (int)float / scale: truncation then integer division (meshwarper.cpp:861-862).
             * Non-finite / out-of-int-range values fall outside the bounds check. */
            if (!(fx > -2147483648.f && fx < 2147483648.f && fy > -2147483648.f && fy < 2147483648.f)) continue;
            const int x_ = (int)fx / scale, y_ = (int)fy / scale;
            if (x_ >= 0 && y_ >= 0 && x_ < hw && y_ < hh) {
                sum_x[(size_t)y_ * hw + x_] += (float)x;
                sum_y[(size_t)y_ * hw + x_] += (float)y;
                cnt[(size_t)y_ * hw + x_] += 1.f;
            }
        }
    for (size_t i = 0; i < (size_t)hw * hh; ++i) { /* 0/0 -> NaN holes, as the reference */
        sum_x[i] = sum_x[i] / cnt[i];
        sum_y[i] = sum_y[i] / cnt[i];
    }
    orc_custom_resize_32f(sum_x, sizeof(float) * (size_t)hw, hh, hw, map_x, sizeof(float) * (size_t)width, height, width);
    orc_custom_resize_32f(sum_y, sizeof(float) * (size_t)hw, hh, hw, map_y, sizeof(float) * (size_t)width, height, width);
    free(big_x); free(big_y); free(sum_x); free(sum_y); free(cnt);
}

/* ---------------------------------------------------------------------------------------------
 * MultiBandBlender, GPU branch of the fork (can_use_gpu_ == true, weight_type_ == CV_32F)
 */
typedef struct { void *data; size_t step; int rows, cols; } img_t;

struct orc_blender {
    int n;
    orc_blend_geom g;
    int *cx, *cy, *w, *h;
    orc_view_geom *vg;
    int n_init;
    img_t *dst_lap;    /* [nb+1] 16SC3  gpu_dst_pyr_laplace_  */
    img_t *dst_w;      /* [nb+1] 32FC1  gpu_dst_band_weights_ */
    img_t **wpyr;      /* [n][nb+1] 32FC1 gpu_weight_pyr_gauss_vec_ */
    img_t **spyr;      /* [n][nb+1] 16SC3 gpu_src_pyr_laplace_vec   */
    int flavour;       /* 0: GPU branch (CUDA arithmetic); 1: CPU branch (cv::pyrDown / pyrUp rounding, weight pyramid per feed) */
    img_t *wmap;       /* [n] flavour 1: mask / 255 as feed receives it (the CPU feed rebuilds the weight pyramid on every call) */
};

static void pd16(const orc_blender *b, const img_t *s, img_t *d)
{
    if (b->flavour) orc_cv_pyr_down_16s((const int16_t *)s->data, s->step, s->rows, s->cols, 3, (int16_t *)d->data, d->step);
    else orc_pyr_down_16s((const int16_t *)s->data, s->step, s->rows, s->cols, 3, (int16_t *)d->data, d->step);
}
static void pu16(const orc_blender *b, const img_t *s, img_t *d)
{
    if (b->flavour) orc_cv_pyr_up_16s((const int16_t *)s->data, s->step, s->rows, s->cols, 3, (int16_t *)d->data, d->step);
    else orc_pyr_up_16s((const int16_t *)s->data, s->step, s->rows, s->cols, 3, (int16_t *)d->data, d->step);
}
static void pd32(const orc_blender *b, const img_t *s, img_t *d)
{
    if (b->flavour) orc_cv_pyr_down_32f((const float *)s->data, s->step, s->rows, s->cols, (float *)d->data, d->step);
    else orc_pyr_down_32f((const float *)s->data, s->step, s->rows, s->cols, (float *)d->data, d->step);
}
void orc_blender_set_flavour(orc_blender *b, int flavour) { b->flavour = flavour; }

static img_t img_alloc(int rows, int cols, int elem)
{
    img_t m;
    m.rows = rows; m.cols = cols; m.step = (size_t)cols * elem;
    m.data = calloc((size_t)rows * cols, (size_t)elem);
    return m;
}

orc_blender *orc_blender_create(int n, int num_bands, const int *cx, const int *cy, const int *w, const int *h)
{
    orc_blender *b = (orc_blender *)calloc(1, sizeof(*b));
    b->n = n;
    b->cx = (int *)malloc(sizeof(int) * n); b->cy = (int *)malloc(sizeof(int) * n);
    b->w = (int *)malloc(sizeof(int) * n); b->h = (int *)malloc(sizeof(int) * n);
    memcpy(b->cx, cx, sizeof(int) * n); memcpy(b->cy, cy, sizeof(int) * n);
    memcpy(b->w, w, sizeof(int) * n); memcpy(b->h, h, sizeof(int) * n);
    /* Blender::prepare(corners, sizes) -> prepare(resultRoi(...))  blenders.cpp:82-85 */
    orc_blender_prepare(orc_result_roi(n, cx, cy, w, h), num_bands, &b->g);
    const int nb = b->g.num_bands;
    b->vg = (orc_view_geom *)calloc(n, sizeof(orc_view_geom));
    /* blenders.cpp:257-273: dst pyramids, level i = ((rows+1)/2, (cols+1)/2) of level i-1, zeroed */
    b->dst_lap = (img_t *)calloc(nb + 1, sizeof(img_t));
    b->dst_w = (img_t *)calloc(nb + 1, sizeof(img_t));
    int r = b->g.dst_roi.height, c = b->g.dst_roi.width;
    for (int i = 0; i <= nb; ++i) {
        b->dst_lap[i] = img_alloc(r, c, 6);
        b->dst_w[i] = img_alloc(r, c, 4);
        r = (r + 1) / 2; c = (c + 1) / 2;
    }
    b->wpyr = (img_t **)calloc(n, sizeof(img_t *));
    b->spyr = (img_t **)calloc(n, sizeof(img_t *));
    b->wmap = (img_t *)calloc(n, sizeof(img_t));
    for (int v = 0; v < n; ++v) {
        b->wpyr[v] = (img_t *)calloc(nb + 1, sizeof(img_t));
        b->spyr[v] = (img_t *)calloc(nb + 1, sizeof(img_t));
    }
    return b;
}

void orc_blender_destroy(orc_blender *b)
{
    if (!b) return;
    const int nb = b->g.num_bands;
    for (int i = 0; i <= nb; ++i) { free(b->dst_lap[i].data); free(b->dst_w[i].data); }
    for (int v = 0; v < b->n; ++v) {
        for (int i = 0; i <= nb; ++i) { free(b->wpyr[v][i].data); free(b->spyr[v][i].data); }
        free(b->wpyr[v]); free(b->spyr[v]); free(b->wmap[v].data);
    }
    free(b->wmap);
    free(b->wpyr); free(b->spyr); free(b->dst_lap); free(b->dst_w);
    free(b->cx); free(b->cy); free(b->w); free(b->h); free(b->vg);
    free(b);
}

void orc_blender_get_geom(const orc_blender *b, orc_blend_geom *g) { *g = b->g; }
void orc_blender_get_view_geom(const orc_blender *b, int view, orc_view_geom *vg) { *vg = b->vg[view]; }

/* MultiBandBlender::init_gpu  blenders.cpp:344-461 */
void orc_blender_init_view(orc_blender *b, int v, const uint8_t *mask, size_t mstep)
{
    const int nb = b->g.num_bands;
    orc_blender_view_geom(&b->g, b->cx[v], b->cy[v], b->w[v], b->h[v], &b->vg[v]);
    const orc_view_geom *vg = &b->vg[v];
    /* mask.convertTo(weight_map, CV_32F, 1./255.)  :412 */
    img_t wm = img_alloc(b->h[v], b->w[v], 4);
    orc_convert_8u_32f_scale(mask, mstep, (float *)wm.data, wm.step, wm.rows, wm.cols, 1. / 255.);
    /* copyMakeBorder(..., BORDER_CONSTANT) :420 ; nb x pyrDown :422-423 */
    const int pr = b->h[v] + vg->top + vg->bottom, pc = b->w[v] + vg->left + vg->right;
    for (int i = 0; i <= nb; ++i) { free(b->wpyr[v][i].data); free(b->spyr[v][i].data); }
    b->wpyr[v][0] = img_alloc(pr, pc, 4);
    orc_copy_make_border_const_32f((const float *)wm.data, wm.step, wm.rows, wm.cols,
                                   (float *)b->wpyr[v][0].data, b->wpyr[v][0].step,
                                   vg->top, vg->bottom, vg->left, vg->right);
    free(b->wmap[v].data);
    b->wmap[v] = wm;
    b->spyr[v][0] = img_alloc(pr, pc, 6);
    for (int i = 0; i < nb; ++i) {
        const img_t *s = &b->wpyr[v][i];
        b->wpyr[v][i + 1] = img_alloc((s->rows + 1) / 2, (s->cols + 1) / 2, 4);
        pd32(b, s, &b->wpyr[v][i + 1]);
        b->spyr[v][i + 1] = img_alloc((s->rows + 1) / 2, (s->cols + 1) / 2, 6);
    }
}

/* the pyramid part shared by feed_online (8U image) and the CPU feed (16S image): Gaussian chain, Laplacian in place, weighted accumulate */
static void feed_from_level0(orc_blender *b, int v)
{
    const int nb = b->g.num_bands;
    const orc_view_geom *vg = &b->vg[v];
    img_t *sp = b->spyr[v];
    for (int i = 0; i < nb; ++i) pd16(b, &sp[i], &sp[i + 1]);
    for (int i = 0; i < nb; ++i) {
        img_t up = img_alloc(sp[i + 1].rows * 2, sp[i + 1].cols * 2, 6);
        pu16(b, &sp[i + 1], &up);
        orc_sub_16s((const int16_t *)sp[i].data, sp[i].step, (const int16_t *)up.data, up.step,
                    (int16_t *)sp[i].data, sp[i].step, sp[i].rows, sp[i].cols * 3);
        free(up.data);
    }
    int y_tl = vg->y_tl, y_br = vg->y_br, x_tl = vg->x_tl, x_br = vg->x_br;
    for (int i = 0; i <= nb; ++i) {
        img_t *dl = &b->dst_lap[i], *dw = &b->dst_w[i];
        orc_add_src_weight_32f((const int16_t *)sp[i].data, sp[i].step,
                               (const float *)b->wpyr[v][i].data, b->wpyr[v][i].step,
                               (int16_t *)((char *)dl->data + (size_t)y_tl * dl->step) + 3 * x_tl, dl->step,
                               (float *)((char *)dw->data + (size_t)y_tl * dw->step) + x_tl, dw->step,
                               y_br - y_tl, x_br - x_tl);
        x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
    }
}

/* MultiBandBlender::feed, CPU branch (blenders.cpp:585-696) for a 16SC3 image: copyMakeBorder(REFLECT) :587, createLaplacePyr :595
 * (16S branch :997-1008), the weight map's Gaussian pyramid REBUILT on every call :603-624, accumulate :633-690 */
void orc_blender_feed_cpu(orc_blender *b, int v, const int16_t *img, size_t step)
{
    const int nb = b->g.num_bands;
    const orc_view_geom *vg = &b->vg[v];
    img_t *sp = b->spyr[v];
    for (int y = 0; y < sp[0].rows; ++y) {          /* BORDER_REFLECT on 16SC3: same index map as the 8U version */
        int sy = y - vg->top;
        sy = sy < 0 ? -sy - 1 : (sy >= b->h[v] ? 2 * b->h[v] - sy - 1 : sy);
        const int16_t *s = (const int16_t *)((const char *)img + (size_t)sy * step);
        int16_t *d = (int16_t *)((char *)sp[0].data + (size_t)y * sp[0].step);
        for (int x = 0; x < sp[0].cols; ++x) {
            int sx = x - vg->left;
            sx = sx < 0 ? -sx - 1 : (sx >= b->w[v] ? 2 * b->w[v] - sx - 1 : sx);
            d[3 * x] = s[3 * sx]; d[3 * x + 1] = s[3 * sx + 1]; d[3 * x + 2] = s[3 * sx + 2];
        }
    }
    orc_copy_make_border_const_32f((const float *)b->wmap[v].data, b->wmap[v].step, b->wmap[v].rows, b->wmap[v].cols,
                                   (float *)b->wpyr[v][0].data, b->wpyr[v][0].step, vg->top, vg->bottom, vg->left, vg->right);
    for (int i = 0; i < nb; ++i) pd32(b, &b->wpyr[v][i], &b->wpyr[v][i + 1]);
    feed_from_level0(b, v);
}

/* MultiBandBlender::feed_online  blenders.cpp:700-749 */
void orc_blender_feed(orc_blender *b, int v, const uint8_t *img, size_t step)
{
    const int nb = b->g.num_bands;
    const orc_view_geom *vg = &b->vg[v];
    img_t *sp = b->spyr[v];
    /* copyMakeBorder BORDER_REFLECT :711 */
    img_t bord = img_alloc(sp[0].rows, sp[0].cols, 3);
    orc_copy_make_border_reflect(img, step, b->h[v], b->w[v], 3, (uint8_t *)bord.data, bord.step,
                                 vg->top, vg->bottom, vg->left, vg->right);
    /* convertTo CV_16S :713 */
    orc_convert_8u_16s((const uint8_t *)bord.data, bord.step, (int16_t *)sp[0].data, sp[0].step, sp[0].rows, sp[0].cols * 3);
    free(bord.data);
    /* pyrDown chain :714-715, pyrUp + subtract in place :716-720, weighted accumulate :722-746 */
    (void)nb; (void)vg;
    feed_from_level0(b, v);
}

/* MultiBandBlender::blend(dst, dst_mask, gpuOut, true)  blenders.cpp:758-832 */
void orc_blender_blend(orc_blender *b, int16_t *out, size_t ostep, uint8_t *out_mask, size_t mstep)
{
    const int nb = b->g.num_bands;
    const int fw = b->g.dst_roi_final.width, fh = b->g.dst_roi_final.height;
    /* normalise every level over its whole extent :767-783 */
    for (int i = 0; i <= nb; ++i)
        orc_normalize_32f((const float *)b->dst_w[i].data, b->dst_w[i].step,
                          (int16_t *)b->dst_lap[i].data, b->dst_lap[i].step, b->dst_w[i].rows, b->dst_w[i].cols);
    /* collapse :786-790 */
    for (int i = nb; i > 0; --i) {
        img_t *s = &b->dst_lap[i], *d = &b->dst_lap[i - 1];
        img_t up = img_alloc(s->rows * 2, s->cols * 2, 6);
        pu16(b, s, &up);
        orc_add_16s((const int16_t *)up.data, up.step, (const int16_t *)d->data, d->step,
                    (int16_t *)d->data, d->step, d->rows, d->cols * 3);
        free(up.data);
    }
    /* masks :803,808 ; setTo :810 ; copy out :811 */
    uint8_t *dmask = (uint8_t *)malloc((size_t)fw * fh), *inv = (uint8_t *)malloc((size_t)fw * fh);
    orc_compare_gt_32f((const float *)b->dst_w[0].data, b->dst_w[0].step, 1e-5f, dmask, (size_t)fw, fh, fw);
    orc_compare_eq_8u(dmask, (size_t)fw, 0, inv, (size_t)fw, fh, fw);
    orc_set_zero_masked_16sc3((int16_t *)b->dst_lap[0].data, b->dst_lap[0].step, inv, (size_t)fw, fh, fw);
    for (int y = 0; y < fh; ++y) {
        memcpy((char *)out + (size_t)y * ostep, (char *)b->dst_lap[0].data + (size_t)y * b->dst_lap[0].step, (size_t)fw * 6);
        if (out_mask) memcpy(out_mask + (size_t)y * mstep, dmask + (size_t)y * fw, (size_t)fw);
    }
    free(dmask); free(inv);
    /* clear accumulators :827-831 */
    for (int i = 0; i <= nb; ++i) {
        memset(b->dst_lap[i].data, 0, b->dst_lap[i].step * (size_t)b->dst_lap[i].rows);
        memset(b->dst_w[i].data, 0, b->dst_w[i].step * (size_t)b->dst_w[i].rows);
    }
}

const float *orc_blender_weight_level(const orc_blender *b, int view, int level, int *rows, int *cols, size_t *step)
{
    const img_t *m = &b->wpyr[view][level];
    *rows = m->rows; *cols = m->cols; *step = m->step;
    return (const float *)m->data;
}

const int16_t *orc_blender_src_level(const orc_blender *b, int view, int level, int *rows, int *cols, size_t *step)
{
    const img_t *m = &b->spyr[view][level];
    *rows = m->rows; *cols = m->cols; *step = m->step;
    return (const int16_t *)m->data;
}

/* The reference's CPU per-view stage (the surveyor's harness, SURVEY App. D: remap + gain + convert 79 ms, prepare + feed 95 ms per frame):
 * cv::remap(INTER_LINEAR, BORDER_CONSTANT) in its fixed-point CPU arithmetic -> convertTo(same type, gain) -> convertTo(CV_16S) -> feed */
void orc_stitch_online_cpu(orc_blender *b, int v, const uint8_t *src, size_t sstep, int srows, int scols,
                           const float *xmap, const float *ymap, double gain)
{
    const int w = b->w[v], h = b->h[v];
    const size_t step3 = (size_t)w * 3, stepf = sizeof(float) * (size_t)w;
    uint8_t *img = (uint8_t *)malloc(step3 * (size_t)h);
    orc_cv_remap_linear_8u(src, sstep, srows, scols, 3, xmap, stepf, ymap, stepf, img, step3, h, w);
    orc_convert_scale_8u(img, step3, img, step3, h, w * 3, gain);
    int16_t *img16 = (int16_t *)malloc(step3 * 2 * (size_t)h);
    orc_convert_8u_16s(img, step3, img16, step3 * 2, h, w * 3);
    orc_blender_feed_cpu(b, v, img16, step3 * 2);
    free(img); free(img16);
}

/* stitch_online  APP/timed.cpp:56-121 (compose_scale == 1 branch :90, gain :94, CPW :96-104, feed :116) */
void orc_stitch_online(orc_blender *b, int v, const uint8_t *src, size_t sstep, int srows, int scols,
                       const float *xmap, const float *ymap, double gain,
                       const float *xmesh, const float *ymesh, uint8_t *warped_out)
{
    const int w = b->w[v], h = b->h[v];
    const size_t step3 = (size_t)w * 3, stepf = sizeof(float) * (size_t)w;
    uint8_t *img = (uint8_t *)malloc(step3 * h);
    orc_remap_linear_8uc3(src, sstep, srows, scols, xmap, stepf, ymap, stepf, img, step3, h, w);
    orc_convert_scale_8u(img, step3, img, step3, h, w * 3, gain);
    if (xmesh && ymesh) {
        uint8_t *warped = (uint8_t *)malloc(step3 * h);
        orc_remap_linear_8uc3(img, step3, h, w, xmesh, stepf, ymesh, stepf, warped, step3, h, w);
        free(img);
        img = warped;
    }
    if (warped_out) memcpy(warped_out, img, step3 * h);
    orc_blender_feed(b, v, img, step3);
    free(img);
}

/* cv::solve(A, b, x, DECOMP_LU) for CV_64F: closed forms for n <= 3 (OCV/core/src/lapack.cpp:1107-1237), otherwise
 * hal::LU64f = LUImpl (OCV/core/src/matrix_decomp.cpp:52-112) with partial pivoting, in the reference's operation order. */
static int solve_lu64(double *A, double *b, int n)
{
    if (n == 1) { if (A[0] == 0.) return 0; b[0] = b[0] / A[0]; return 1; }
    if (n == 2) {
        double d = A[0] * A[3] - A[1] * A[2];
        if (d == 0.) return 0;
        d = 1. / d;
        const double t = (b[0] * A[3] - b[1] * A[1]) * d;
        b[1] = (b[1] * A[0] - b[0] * A[2]) * d;
        b[0] = t;
        return 1;
    }
    if (n == 3) {
#define S(i, j) A[(i) * 3 + (j)]
        double d = S(0, 0) * (S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) - S(0, 1) * (S(1, 0) * S(2, 2) - S(1, 2) * S(2, 0)) +
                   S(0, 2) * (S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0));
        if (d == 0.) return 0;
        d = 1. / d;
        double t[3];
        t[0] = ((S(1, 1) * S(2, 2) - S(1, 2) * S(2, 1)) * b[0] + (S(0, 2) * S(2, 1) - S(0, 1) * S(2, 2)) * b[1] + (S(0, 1) * S(1, 2) - S(0, 2) * S(1, 1)) * b[2]) * d;
        t[1] = ((S(1, 2) * S(2, 0) - S(1, 0) * S(2, 2)) * b[0] + (S(0, 0) * S(2, 2) - S(0, 2) * S(2, 0)) * b[1] + (S(0, 2) * S(1, 0) - S(0, 0) * S(1, 2)) * b[2]) * d;
        t[2] = ((S(1, 0) * S(2, 1) - S(1, 1) * S(2, 0)) * b[0] + (S(0, 1) * S(2, 0) - S(0, 0) * S(2, 1)) * b[1] + (S(0, 0) * S(1, 1) - S(0, 1) * S(1, 0)) * b[2]) * d;
#undef S
        b[0] = t[0]; b[1] = t[1]; b[2] = t[2];
        return 1;
    }
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < n; ++i) {
        int k = i;
        for (int j = i + 1; j < n; ++j) if (fabs(A[j * n + i]) > fabs(A[k * n + i])) k = j;
        if (fabs(A[k * n + i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < n; ++j) { double t = A[i * n + j]; A[i * n + j] = A[k * n + j]; A[k * n + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        const double d = -1 / A[i * n + i];
        for (int j = i + 1; j < n; ++j) {
            const double alpha = A[j * n + i] * d;
            for (int q = i + 1; q < n; ++q) A[j * n + q] += alpha * A[i * n + q];
            b[j] += alpha * b[i];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double sv = b[i];
        for (int q = i + 1; q < n; ++q) sv -= A[i * n + q] * b[q];
        b[i] = sv / A[i * n + i];
    }
    return 1;
}

/* GainCompensator::feed  OCV/stitching/src/exposure_compensate.cpp:71-145.  images: 8UC3 contiguous (h x w x 3),
 * masks: 8UC1 contiguous; overlap pixels are those where both masks equal 255. */
int orc_gain_compensator(int n, const int *cx, const int *cy, const int *w, const int *h,
                         const uint8_t *const *images, const uint8_t *const *masks, double *gains)
{
    int *N = (int *)calloc((size_t)n * n, sizeof(int));
    double *I = (double *)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            orc_rect roi;
            if (!overlap_roi(cx[i], cy[i], cx[j], cy[j], w[i], h[i], w[j], h[j], &roi)) continue;
            int cnt = 0;
            double s1 = 0, s2 = 0;
            for (int y = 0; y < roi.height; ++y)
                for (int x = 0; x < roi.width; ++x) {
                    const size_t p1 = (size_t)(roi.y - cy[i] + y) * w[i] + (roi.x - cx[i] + x);
                    const size_t p2 = (size_t)(roi.y - cy[j] + y) * w[j] + (roi.x - cx[j] + x);
                    if (masks[i][p1] == 255 && masks[j][p2] == 255) {
                        ++cnt;
                        const uint8_t *a = images[i] + 3 * p1, *b = images[j] + 3 * p2;
                        s1 += sqrt((double)(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]));
                        s2 += sqrt((double)(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]));
                    }
                }
            N[i * n + j] = N[j * n + i] = cnt > 1 ? cnt : 1;
            I[i * n + j] = s1 / N[i * n + j];
            I[j * n + i] = s2 / N[i * n + j];
        }
    const double alpha = 0.01, beta = 100;
    double *A = (double *)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n; ++i) gains[i] = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            gains[i] += beta * N[i * n + j];
            A[i * n + i] += beta * N[i * n + j];
            if (j == i) continue;
            A[i * n + i] += 2 * alpha * I[i * n + j] * I[i * n + j] * N[i * n + j];
            A[i * n + j] -= 2 * alpha * I[i * n + j] * I[j * n + i] * N[i * n + j];
        }
    const int ok = solve_lu64(A, gains, n);
    free(N); free(I); free(A);
    return ok;
}


/* ====================================================================================================================
 * FeatherBlender -- the CPU blender of BASELINE configs[0].  TEST INFRASTRUCTURE ONLY.
 * Follows OCV stitching/src/blenders.cpp:
 *   createWeightMap            :944-951   distanceTransform(mask, DIST_L1, 3) * sharpness, threshold(.., 1, THRESH_TRUNC)
 *   Blender::prepare(Rect)     :89-95     dst 16SC3 zeros, dst_mask 8U zeros over dst_roi (= resultRoi(corners, sizes), :82-86)
 *   FeatherBlender::prepare    :139-144   dst_weight_map 32F zeros
 *   FeatherBlender::feed       :147-178   dst += (short)(src * w) per channel, dst_w += w
 *   FeatherBlender::blend      :181-186   normalizeUsingWeightMap (:899-912), mask = dst_w > WEIGHT_EPS (1e-5f), Blender::blend:
 *   Blender::blend             :125-135   dst.setTo(0, dst_mask == 0)
 * multiply(32F, scalar) works in float (the scalar is converted to the array depth), so weight = min(d * sharpness, 1) in fp32.
 * static_cast<short>(float) truncates toward zero (values here are far inside the short range).
 * ==================================================================================================================== */
void orc_feather_weight_map(const uint8_t *mask, size_t mstep, int rows, int cols, float sharpness, float *w, size_t wstep)
{
    orc_distance_transform_l1(mask, mstep, rows, cols, w, wstep);
    for (int y = 0; y < rows; ++y) {
        float *r = (float *)((char *)w + (size_t)y * wstep);
        for (int x = 0; x < cols; ++x) {
            const float t = r[x] * sharpness;
            r[x] = t > 1.f ? 1.f : t;                      /* THRESH_TRUNC */
        }
    }
}

/* dst / dst_w cover dst_roi (rows x cols = roi_h x roi_w); img is 16SC3 (the caller's convertTo(CV_16S) of the warped 8U view) */
void orc_feather_feed(const int16_t *img, size_t istep, const float *w, size_t wstep, int rows, int cols, int dx, int dy,
                      int16_t *dst, size_t dstep, float *dst_w, size_t dwstep)
{
    for (int y = 0; y < rows; ++y) {
        const int16_t *s = (const int16_t *)((const char *)img + (size_t)y * istep);
        const float *wr = (const float *)((const char *)w + (size_t)y * wstep);
        int16_t *d = (int16_t *)((char *)dst + (size_t)(dy + y) * dstep);
        float *dw = (float *)((char *)dst_w + (size_t)(dy + y) * dwstep);
        for (int x = 0; x < cols; ++x) {
            for (int c = 0; c < 3; ++c)
                d[3 * (dx + x) + c] = (int16_t)(d[3 * (dx + x) + c] + (int16_t)((float)s[3 * x + c] * wr[x]));
            dw[dx + x] += wr[x];
        }
    }
}

void orc_feather_blend(int16_t *dst, size_t dstep, const float *dst_w, size_t dwstep, int rows, int cols, uint8_t *mask, size_t mstep)
{
    for (int y = 0; y < rows; ++y) {
        int16_t *d = (int16_t *)((char *)dst + (size_t)y * dstep);
        const float *dw = (const float *)((const char *)dst_w + (size_t)y * dwstep);
        uint8_t *m = mask + (size_t)y * mstep;
        for (int x = 0; x < cols; ++x) {
            for (int c = 0; c < 3; ++c) d[3 * x + c] = (int16_t)((float)d[3 * x + c] / (dw[x] + 1e-5f));
            m[x] = dw[x] > 1e-5f ? 255 : 0;
            if (!m[x]) d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = 0;
        }
    }
}
