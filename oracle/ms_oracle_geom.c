/*
 * ms_oracle_geom.c -- CPU ORACLE (test infrastructure only; see ms_oracle.h header).
 * Warper geometry (projector parameters, forward/backward maps, result ROI) and the
 * MultiBandBlender padding geometry, restated from the reference's host code.
 * PINNED: exact integer equality with SURVEY.md Appendix C known answers
 * (tests/golden/geometry_kats.json, tests/test_oracle_geometry.py).
 */
#include "ms_oracle.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <string.h>

#define ORC_PI 3.1415926535897932384626433832795 /* CV_PI  OCV/core/include/opencv2/core/cvdef.h */

/* cv::Mat::inv() 3x3 CV_32F closed form  OCV/core/src/lapack.cpp:749-751,980-1007 */
static int inv3_f32(const float *m, float *out)
{
#define M(i, j) m[(i) * 3 + (j)]
    double d = M(0, 0) * ((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) -
               M(0, 1) * ((double)M(1, 0) * M(2, 2) - (double)M(1, 2) * M(2, 0)) +
               M(0, 2) * ((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0));
    if (d == 0.) { memset(out, 0, 9 * sizeof(float)); return 0; }
    d = 1. / d;
    double t[9];
    t[0] = (((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) * d);
    t[1] = (((double)M(0, 2) * M(2, 1) - (double)M(0, 1) * M(2, 2)) * d);
    t[2] = (((double)M(0, 1) * M(1, 2) - (double)M(0, 2) * M(1, 1)) * d);
    t[3] = (((double)M(1, 2) * M(2, 0) - (double)M(1, 0) * M(2, 2)) * d);
    t[4] = (((double)M(0, 0) * M(2, 2) - (double)M(0, 2) * M(2, 0)) * d);
    t[5] = (((double)M(0, 2) * M(1, 0) - (double)M(0, 0) * M(1, 2)) * d);
    t[6] = (((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0)) * d);
    t[7] = (((double)M(0, 1) * M(2, 0) - (double)M(0, 0) * M(2, 1)) * d);
    t[8] = (((double)M(0, 0) * M(1, 1) - (double)M(0, 1) * M(1, 0)) * d);
    for (int i = 0; i < 9; ++i) out[i] = (float)t[i];
#undef M
    return 1;
}

/* cv::gemm small-matrix fast path for 3x3 CV_32F, flags == 0  OCV/core/src/matmul.cpp:979-991:
 * float t = a0*b0 + a1*b1 + a2*b2 (fp32, left to right), d = (float)(t*alpha(double 1) + 0*beta) */
static void mul3_f32(const float *a, const float *b, float *d)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float t = a[i * 3 + 0] * b[0 * 3 + j];
            t = t + a[i * 3 + 1] * b[1 * 3 + j];
            t = t + a[i * 3 + 2] * b[2 * 3 + j];
            d[i * 3 + j] = (float)((double)t * 1.0 + 0.0);
        }
}

/* ProjectorBase::setCameraParams  OCV/stitching/src/warpers.cpp:49-79 */
void orc_set_camera_params(orc_projector *p, const float *K, const float *R, const float *T, float scale)
{
    float kinv[9];
    memcpy(p->k, K, sizeof(p->k));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) p->rinv[i * 3 + j] = R[j * 3 + i];      /* Rinv = R.t() */
    inv3_f32(K, kinv);
    mul3_f32(R, kinv, p->r_kinv);                                          /* R * K.inv()  */
    mul3_f32(K, p->rinv, p->k_rinv);                                       /* K * Rinv     */
    if (T) memcpy(p->t, T, sizeof(p->t)); else p->t[0] = p->t[1] = p->t[2] = 0.f;
    p->scale = scale;
}

/* warpers_cuda.cpp:108,136,164: Mat K_Rinv = K * R.t() -- a GEMM with the transpose flag,
 * which takes cv::gemm's generic path (GEMMSingleMul<float,double>: double accumulator). */
void orc_k_rinv_gpu(const float *K, const float *R, float *k_rinv)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)K[i * 3 + k] * (double)R[j * 3 + k];
            k_rinv[i * 3 + j] = (float)s;
        }
}

/* {Plane,Spherical,Cylindrical}Projector::mapForward
 * OCV/stitching/include/opencv2/stitching/detail/warpers_inl.hpp:213-226,244-254,278-287 */
void orc_map_forward(int proj, const orc_projector *p, float x, float y, float *u, float *v)
{
    const float *r = p->r_kinv;
    float x_ = r[0] * x + r[1] * y + r[2];
    float y_ = r[3] * x + r[4] * y + r[5];
    float z_ = r[6] * x + r[7] * y + r[8];
    if (proj == ORC_PROJ_PLANE) {
        x_ = p->t[0] + x_ / z_ * (1 - p->t[2]);
        y_ = p->t[1] + y_ / z_ * (1 - p->t[2]);
        *u = p->scale * x_;
        *v = p->scale * y_;
    } else if (proj == ORC_PROJ_SPHERICAL) {
        *u = p->scale * atan2f(x_, z_);
        float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        *v = p->scale * ((float)ORC_PI - acosf(w == w ? w : 0));
    } else {
        *u = p->scale * atan2f(x_, z_);
        *v = p->scale * y_ / sqrtf(x_ * x_ + z_ * z_);
    }
}

/* mapBackward (CPU flavour)  warpers_inl.hpp:229-241,257-275,290-307 */
void orc_map_backward(int proj, const orc_projector *p, float u, float v, float *x, float *y)
{
    const float *k = p->k_rinv;
    float x_, y_, z_, z;
    if (proj == ORC_PROJ_PLANE) {
        u = u / p->scale - p->t[0];
        v = v / p->scale - p->t[1];
        *x = k[0] * u + k[1] * v + k[2] * (1 - p->t[2]);
        *y = k[3] * u + k[4] * v + k[5] * (1 - p->t[2]);
        z = k[6] * u + k[7] * v + k[8] * (1 - p->t[2]);
        *x /= z; *y /= z;
        return;
    }
    u /= p->scale;
    v /= p->scale;
    if (proj == ORC_PROJ_SPHERICAL) {
        float sinv = sinf((float)ORC_PI - v);
        x_ = sinv * sinf(u);
        y_ = cosf((float)ORC_PI - v);
        z_ = sinv * cosf(u);
    } else {
        x_ = sinf(u);
        y_ = v;
        z_ = cosf(u);
    }
    *x = k[0] * x_ + k[1] * y_ + k[2] * z_;
    *y = k[3] * x_ + k[4] * y_ + k[5] * z_;
    z = k[6] * x_ + k[7] * y_ + k[8] * z_;
    if (z > 0) { *x /= z; *y /= z; }
    else *x = *y = -1;
}

typedef struct { float tl_u, tl_v, br_u, br_v; } fbox;
static inline void fbox_add(fbox *b, float u, float v)
{
    /* (std::min)(a, b) = (b < a) ? b : a  -- keeps a when b is NaN */
    b->tl_u = (u < b->tl_u) ? u : b->tl_u; b->tl_v = (v < b->tl_v) ? v : b->tl_v;
    b->br_u = (b->br_u < u) ? u : b->br_u; b->br_v = (b->br_v < v) ? v : b->br_v;
}

/* RotationWarperBase::detectResultRoi (all pixels)  warpers_inl.hpp:150-173 -- plane warper */
static void roi_all(int proj, const orc_projector *p, int w, int h, fbox *b)
{
    float u, v;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            orc_map_forward(proj, p, (float)x, (float)y, &u, &v);
            fbox_add(b, u, v);
        }
}

/* RotationWarperBase::detectResultRoiByBorder  warpers_inl.hpp:176-210 */
static void roi_border(int proj, const orc_projector *p, int w, int h, fbox *b)
{
    float u, v;
    for (float x = 0; x < w; ++x) {
        orc_map_forward(proj, p, x, 0, &u, &v); fbox_add(b, u, v);
        orc_map_forward(proj, p, x, (float)(h - 1), &u, &v); fbox_add(b, u, v);
    }
    for (int y = 0; y < h; ++y) {
        orc_map_forward(proj, p, 0, (float)y, &u, &v); fbox_add(b, u, v);
        orc_map_forward(proj, p, (float)(w - 1), (float)y, &u, &v); fbox_add(b, u, v);
    }
}

void orc_detect_result_roi(int proj, const orc_projector *p, int src_w, int src_h,
                           int *tl_x, int *tl_y, int *br_x, int *br_y)
{
    fbox b = {FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (proj == ORC_PROJ_PLANE) {
        roi_all(proj, p, src_w, src_h, &b);
    } else if (proj == ORC_PROJ_CYLINDRICAL) {
        /* CylindricalWarper::detectResultRoi -> ByBorder  detail/warpers.hpp:287-290 */
        roi_border(proj, p, src_w, src_h, &b);
    } else {
        /* SphericalWarper::detectResultRoi  OCV/stitching/src/warpers.cpp:277-318:
         * border walk -> (int) truncation -> back to float -> pole fix-ups -> (int) truncation */
        roi_border(proj, p, src_w, src_h, &b);
        float tl_uf = (float)(int)b.tl_u, tl_vf = (float)(int)b.tl_v;
        float br_uf = (float)(int)b.br_u, br_vf = (float)(int)b.br_v;
        float x = p->rinv[1], y = p->rinv[4], z = p->rinv[7];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                const float pv = (float)(ORC_PI * p->scale);
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, pv);
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, pv);
            }
        }
        x = p->rinv[1]; y = -p->rinv[4]; z = p->rinv[7];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, 0.f);
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, 0.f);
            }
        }
        b.tl_u = tl_uf; b.tl_v = tl_vf; b.br_u = br_uf; b.br_v = br_vf;
    }
    *tl_x = (int)b.tl_u; *tl_y = (int)b.tl_v;
    *br_x = (int)b.br_u; *br_y = (int)b.br_v;
}

/* RotationWarperBase::buildMaps (CPU)  warpers_inl.hpp:65-90 */
void orc_build_maps_cpu(int proj, const orc_projector *p, int tl_x, int tl_y, int rows, int cols,
                        float *mapx, size_t mxstep, float *mapy, size_t mystep)
{
    for (int v = 0; v < rows; ++v) {
        float *mx = (float *)((char *)mapx + (size_t)v * mxstep);
        float *my = (float *)((char *)mapy + (size_t)v * mystep);
        for (int u = 0; u < cols; ++u)
            orc_map_backward(proj, p, (float)(tl_x + u), (float)(tl_y + v), &mx[u], &my[u]);
    }
}

/* detail::resultRoi(corners, sizes)  OCV/stitching/src/util.cpp:125-138 */
orc_rect orc_result_roi(int n, const int *cx, const int *cy, const int *w, const int *h)
{
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {
        if (cx[i] < tlx) tlx = cx[i];
        if (cy[i] < tly) tly = cy[i];
        if (cx[i] + w[i] > brx) brx = cx[i] + w[i];
        if (cy[i] + h[i] > bry) bry = cy[i] + h[i];
    }
    orc_rect r = {tlx, tly, brx - tlx, bry - tly};
    return r;
}

/* MultiBandBlender::prepare(Rect)  OCV/stitching/src/blenders.cpp:237-252 */
void orc_blender_prepare(orc_rect dst_roi, int actual_num_bands, orc_blend_geom *g)
{
    g->dst_roi_final = dst_roi;
    double max_len = (double)(dst_roi.width > dst_roi.height ? dst_roi.width : dst_roi.height);
    int lim = (int)ceil(log(max_len) / log(2.0));
    g->num_bands = actual_num_bands < lim ? actual_num_bands : lim;
    const int m = 1 << g->num_bands;
    dst_roi.width += (m - dst_roi.width % m) % m;
    dst_roi.height += (m - dst_roi.height % m) % m;
    g->dst_roi = dst_roi;
}

/* MultiBandBlender::init_gpu geometry prologue  blenders.cpp:353-387, 425-428 */
void orc_blender_view_geom(const orc_blend_geom *g, int tl_x, int tl_y, int mask_cols, int mask_rows,
                           orc_view_geom *vg)
{
    const int nb = g->num_bands;
    const orc_rect d = g->dst_roi;
    const int d_brx = d.x + d.width, d_bry = d.y + d.height;
    const int gap = 3 * (1 << nb);
    int tlnx = d.x > tl_x - gap ? d.x : tl_x - gap;
    int tlny = d.y > tl_y - gap ? d.y : tl_y - gap;
    int brnx = d_brx < tl_x + mask_cols + gap ? d_brx : tl_x + mask_cols + gap;
    int brny = d_bry < tl_y + mask_rows + gap ? d_bry : tl_y + mask_rows + gap;
    tlnx = d.x + (((tlnx - d.x) >> nb) << nb);
    tlny = d.y + (((tlny - d.y) >> nb) << nb);
    int width = brnx - tlnx, height = brny - tlny;
    const int m = 1 << nb;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    brnx = tlnx + width;
    brny = tlny + height;
    int dy = brny - d_bry > 0 ? brny - d_bry : 0;
    int dx = brnx - d_brx > 0 ? brnx - d_brx : 0;
    tlnx -= dx; brnx -= dx;
    tlny -= dy; brny -= dy;
    vg->top = tl_y - tlny;
    vg->left = tl_x - tlnx;
    vg->bottom = brny - tl_y - mask_rows;
    vg->right = brnx - tl_x - mask_cols;
    vg->y_tl = tlny - d.y; vg->y_br = brny - d.y;
    vg->x_tl = tlnx - d.x; vg->x_br = brnx - d.x;
}
