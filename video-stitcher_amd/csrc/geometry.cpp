// geometry.cpp -- host-side geometry of the compositor (the reference computes all of this on the
// host as well): projector parameters, result-ROI detection, pano ROI, blender padding, Voronoi seams.
// Reference: OCV/stitching/src/warpers.cpp:49-79,277-318; detail/warpers_inl.hpp:136-307;
// src/util.cpp:100-138; src/blenders.cpp:237-252,353-387,425-428; src/seam_finders.cpp:71-160;
// OCV/imgproc/src/distransform.cpp:70-137.  fp32 operation order follows the reference so that the
// (int) truncations land on the same integers (SURVEY App. C known answers).
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <vector>
#include "launchers.hpp"

namespace ms {

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;   // CV_PI

using M3 = float[9];

// cv::Mat::inv() 3x3 fp32: closed form evaluated in double (lapack.cpp:749-751,980-1007)
void inv3(const float *m, float *o)
{
    auto M = [&](int i, int j) { return m[i * 3 + j]; };
    double det = M(0, 0) * ((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) -
                 M(0, 1) * ((double)M(1, 0) * M(2, 2) - (double)M(1, 2) * M(2, 0)) +
                 M(0, 2) * ((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0));
    if (det == 0.) { std::fill(o, o + 9, 0.f); return; }
    const double d = 1. / det;
    o[0] = (float)(((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) * d);
    o[1] = (float)(((double)M(0, 2) * M(2, 1) - (double)M(0, 1) * M(2, 2)) * d);
    o[2] = (float)(((double)M(0, 1) * M(1, 2) - (double)M(0, 2) * M(1, 1)) * d);
    o[3] = (float)(((double)M(1, 2) * M(2, 0) - (double)M(1, 0) * M(2, 2)) * d);
    o[4] = (float)(((double)M(0, 0) * M(2, 2) - (double)M(0, 2) * M(2, 0)) * d);
    o[5] = (float)(((double)M(0, 2) * M(1, 0) - (double)M(0, 0) * M(1, 2)) * d);
    o[6] = (float)(((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0)) * d);
    o[7] = (float)(((double)M(0, 1) * M(2, 0) - (double)M(0, 0) * M(2, 1)) * d);
    o[8] = (float)(((double)M(0, 0) * M(1, 1) - (double)M(0, 1) * M(1, 0)) * d);
}

// cv::gemm 3x3 fp32 fast path, flags == 0 (matmul.cpp:979-991): fp32 products summed left to right
void mul3(const float *a, const float *b, float *d)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float t = a[i * 3] * b[j];
            t = t + a[i * 3 + 1] * b[3 + j];
            t = t + a[i * 3 + 2] * b[6 + j];
            d[i * 3 + j] = t;
        }
}

struct Box {
    float tl_u = FLT_MAX, tl_v = FLT_MAX, br_u = -FLT_MAX, br_v = -FLT_MAX;
    void add(float u, float v)
    {
        tl_u = std::min(tl_u, u); tl_v = std::min(tl_v, v);
        br_u = std::max(br_u, u); br_v = std::max(br_v, v);
    }
};

// Source pixel -> panorama coordinates: the viewing ray of the pixel (R K^-1 applied to (x, y, 1)) projected onto the plane / cylinder / sphere.
// The formulas are the projectors' of detail/warpers_inl.hpp:244-307; what is pinned here is the fp32 evaluation order, because the ROI corners are
// integer truncations of these values and must land on the reference's (SURVEY App. C known answers, tests/test_geometry_kats.py).
struct Ray { float x, y, z; };
inline Ray pixel_ray(const Projector &p, float px, float py)
{
    const float *m = p.r_kinv;
    return Ray{m[0] * px + m[1] * py + m[2], m[3] * px + m[4] * py + m[5], m[6] * px + m[7] * py + m[8]};
}
void map_forward(int proj, const Projector &p, float px, float py, float &u, float &v)
{
    const Ray d = pixel_ray(p, px, py);
    if (proj == MS_PROJ_PLANE) {            // perspective division onto the plane z = 1 - t_z, shifted by the translation
        const float depth = 1 - p.t[2];
        u = p.scale * (p.t[0] + d.x / d.z * depth);
        v = p.scale * (p.t[1] + d.y / d.z * depth);
        return;
    }
    u = p.scale * atan2f(d.x, d.z);          // longitude: shared by the cylinder and the sphere
    if (proj == MS_PROJ_SPHERICAL) {
        const float cos_polar = d.y / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
        v = p.scale * (static_cast<float>(kPi) - acosf(std::isnan(cos_polar) ? 0.f : cos_polar));      // a zero ray (0 / 0) counts as the equator
    } else {
        v = p.scale * d.y / sqrtf(d.x * d.x + d.z * d.z);
    }
}

void walk_border(int proj, const Projector &p, int w, int h, Box &b)
{
    float u, v;
    for (float x = 0; x < w; ++x) {
        map_forward(proj, p, x, 0, u, v); b.add(u, v);
        map_forward(proj, p, x, static_cast<float>(h - 1), u, v); b.add(u, v);
    }
    for (int y = 0; y < h; ++y) {
        map_forward(proj, p, 0, static_cast<float>(y), u, v); b.add(u, v);
        map_forward(proj, p, static_cast<float>(w - 1), static_cast<float>(y), u, v); b.add(u, v);
    }
}

}  // namespace

void set_camera_params(Projector &p, const float *K, const float *R, const float *T, float scale)
{
    std::copy(K, K + 9, p.k);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) p.rinv[i * 3 + j] = R[j * 3 + i];
    float kinv[9];
    inv3(K, kinv);
    mul3(R, kinv, p.r_kinv);
    mul3(K, p.rinv, p.k_rinv);
    for (int i = 0; i < 3; ++i) p.t[i] = T ? T[i] : 0.f;
    p.scale = scale;
}

// warpers_cuda.cpp:108-109: Mat K_Rinv = K * R.t(); Mat R_Kinv = R * K.inv();
// the transposed product goes through cv::gemm's generic path (double accumulator)
void k_rinv_gemm(const float *K, const float *R, float *out)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (double)K[i * 3 + k] * (double)R[j * 3 + k];
            out[i * 3 + j] = (float)s;
        }
}
void r_kinv_gemm(const float *K, const float *R, float *out)
{
    float kinv[9];
    inv3(K, kinv);
    mul3(R, kinv, out);
}

ms_rect warp_roi(int proj, const Projector &p, int src_w, int src_h)
{
    Box b;
    if (proj == MS_PROJ_PLANE) {             // RotationWarperBase::detectResultRoi: every pixel
        float u, v;
        for (int y = 0; y < src_h; ++y)
            for (int x = 0; x < src_w; ++x) { map_forward(proj, p, (float)x, (float)y, u, v); b.add(u, v); }
    } else {
        walk_border(proj, p, src_w, src_h, b);
    }
    if (proj == MS_PROJ_SPHERICAL) {         // pole fix-up of SphericalWarper::detectResultRoi
        float tl_uf = (float)(int)b.tl_u, tl_vf = (float)(int)b.tl_v;
        float br_uf = (float)(int)b.br_u, br_vf = (float)(int)b.br_v;
        for (int pole = 0; pole < 2; ++pole) {
            const float x = p.rinv[1], y = pole ? -p.rinv[4] : p.rinv[4], z = p.rinv[7];
            if (!(y > 0.f)) continue;
            const float x_ = (p.k[0] * x + p.k[1] * y) / z + p.k[2];
            const float y_ = p.k[4] * y / z + p.k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                const float pv = pole ? 0.f : static_cast<float>(kPi * p.scale);
                tl_uf = std::min(tl_uf, 0.f); tl_vf = std::min(tl_vf, pv);
                br_uf = std::max(br_uf, 0.f); br_vf = std::max(br_vf, pv);
            }
        }
        b.tl_u = tl_uf; b.tl_v = tl_vf; b.br_u = br_uf; b.br_v = br_vf;
    }
    const int tlx = (int)b.tl_u, tly = (int)b.tl_v, brx = (int)b.br_u, bry = (int)b.br_v;
    return ms_rect{tlx, tly, brx - tlx + 1, bry - tly + 1};   // warpRoi: Rect(tl, br + 1)
}

ms_rect result_roi(int n, const ms_rect *r)
{
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {
        tlx = std::min(tlx, r[i].x); tly = std::min(tly, r[i].y);
        brx = std::max(brx, r[i].x + r[i].width); bry = std::max(bry, r[i].y + r[i].height);
    }
    return ms_rect{tlx, tly, brx - tlx, bry - tly};
}

BlendGeom blender_prepare(ms_rect roi, int actual_num_bands)
{
    BlendGeom g;
    g.dst_roi_final = roi;
    const double max_len = (double)std::max(roi.width, roi.height);
    g.num_bands = std::min(actual_num_bands, (int)std::ceil(std::log(max_len) / std::log(2.0)));
    const int m = 1 << g.num_bands;
    roi.width += (m - roi.width % m) % m;
    roi.height += (m - roi.height % m) % m;
    g.dst_roi = roi;
    return g;
}

ViewPad blender_view_pad(const BlendGeom &g, int tl_x, int tl_y, int mc, int mr)
{
    const int nb = g.num_bands, m = 1 << nb, gap = 3 * m;
    const ms_rect &d = g.dst_roi;
    const int dbx = d.x + d.width, dby = d.y + d.height;
    int ax = std::max(d.x, tl_x - gap), ay = std::max(d.y, tl_y - gap);
    int bx = std::min(dbx, tl_x + mc + gap), by = std::min(dby, tl_y + mr + gap);
    ax = d.x + (((ax - d.x) >> nb) << nb);
    ay = d.y + (((ay - d.y) >> nb) << nb);
    int w = bx - ax, h = by - ay;
    w += (m - w % m) % m;
    h += (m - h % m) % m;
    bx = ax + w; by = ay + h;
    const int dy = std::max(by - dby, 0), dx = std::max(bx - dbx, 0);
    ax -= dx; bx -= dx; ay -= dy; by -= dy;
    ViewPad v;
    v.top = tl_y - ay; v.left = tl_x - ax;
    v.bottom = by - tl_y - mr; v.right = bx - tl_x - mc;
    v.x_tl = ax - d.x; v.y_tl = ay - d.y; v.x_br = bx - d.x; v.y_br = by - d.y;
    return v;
}

// ---- L1 distance transform (FeatherBlender's createWeightMap; the Voronoi seams use the device version in calib.hip) ------
namespace {

// cv::distanceTransform(DIST_L1, 3): two-pass chamfer, fixed point 16.16 (distransform.cpp:70-137)
void dist_l1(const std::vector<uint8_t> &src, int rows, int cols, std::vector<float> &dst)
{
    constexpr int INIT = INT_MAX >> 2, HV = 1 << 16, DG = 2 << 16;
    const int step = cols + 2;
    std::vector<int> temp((size_t)step * (rows + 2), INIT);
    for (int i = 0; i < rows; ++i) {
        int *t = temp.data() + (size_t)(i + 1) * step + 1;
        const uint8_t *s = src.data() + (size_t)i * cols;
        for (int j = 0; j < cols; ++j) {
            if (!s[j]) { t[j] = 0; continue; }
            int a = t[j - step - 1] + DG;
            a = std::min(a, t[j - step] + HV);
            a = std::min(a, t[j - step + 1] + DG);
            a = std::min(a, t[j - 1] + HV);
            t[j] = a;
        }
    }
    dst.resize((size_t)rows * cols);
    for (int i = rows - 1; i >= 0; --i) {
        int *t = temp.data() + (size_t)(i + 1) * step + 1;
        for (int j = cols - 1; j >= 0; --j) {
            int a = t[j];
            if (a > HV) {
                a = std::min(a, t[j + step + 1] + DG);
                a = std::min(a, t[j + step] + HV);
                a = std::min(a, t[j + step - 1] + DG);
                a = std::min(a, t[j + 1] + HV);
                t[j] = a;
            }
            dst[(size_t)i * cols + j] = (float)(a * (1.f / (1 << 16)));
        }
    }
}

}  // namespace

// createWeightMap (blenders.cpp:944-951): distanceTransform(mask, DIST_L1, 3) * sharpness, threshold(.., 1, THRESH_TRUNC); fp32
void feather_weight_map(const uint8_t *mask, int rows, int cols, float sharpness, float *w)
{
    std::vector<uint8_t> m(mask, mask + (size_t)rows * cols);
    std::vector<float> d;
    dist_l1(m, rows, cols, d);
    for (size_t i = 0; i < d.size(); ++i) {
        const float t = d[i] * sharpness;
        w[i] = t > 1.f ? 1.f : t;
    }
}

// (VoronoiSeamFinder and GainCompensator::feed run on the device: calib.hip)

// calibrateCameras + scale bookkeeping (APP/calibration.cpp:28-68, 101-116, 147-181, 269-288), in the reference's types
int calibrate_cameras(const ms_rig_params &q, ms_rig &r)
{
    memset(&r, 0, sizeof(r));
    const double area = (double)q.src_width * (double)q.src_height;
    r.work_scale = q.work_megapix < 0 ? 1.0 : std::min(1.0, std::sqrt(q.work_megapix * 1e6 / area));       // :269-277
    r.seam_scale = std::min(1.0, std::sqrt(q.seam_megapix * 1e6 / area));                                  // :280
    r.seam_work_aspect = r.seam_scale / r.work_scale;                                                       // :281
    r.compose_scale = q.compose_megapix > 0 ? std::min(1.0, std::sqrt(q.compose_megapix * 1e6 / area)) : 1.0;   // :147-150 (compose_scale starts at 1)
    r.compose_work_aspect = r.compose_scale / r.work_scale;                                                 // :153
    const double PI = 3.14159265358979323846;
    const double fov = q.hfov_deg * PI / 180.0, focal_tmp = 1.0 / std::tan(fov * 0.5);                      // :31-32
    const double ppx = ((double)q.src_width * r.work_scale) / 2.0, ppy = ((double)q.src_height * r.work_scale) / 2.0, focal = focal_tmp * ppx;   // :57-66
    r.warped_image_scale = (float)focal;                                                                    // :288
    r.seam_warp_scale = (float)((double)r.warped_image_scale * r.seam_work_aspect);                         // :101 (float * double -> double -> float)
    r.compose_warp_scale = r.warped_image_scale * (float)r.compose_work_aspect;                             // :156
    r.resize_input = std::fabs(r.compose_scale - 1) > 1e-1;                                                 // :161
    r.compose_width = r.resize_input ? (int)std::nearbyint((double)q.src_width * r.compose_scale) : q.src_width;     // cvRound :163-164
    r.compose_height = r.resize_input ? (int)std::nearbyint((double)q.src_height * r.compose_scale) : q.src_height;
    const float swa = (float)r.seam_work_aspect;
    for (int i = 0; i < q.num_views; ++i) {
        const float rot = (float)(2.0 * PI * (double)(float)i / (double)q.num_views);                       // :35
        const float c = (float)std::cos((double)rot), s = (float)std::sin((double)rot);
        const float R[9] = {c, 0.f, s, 0.f, 1.f, 0.f, -s, 0.f, c};                                          // Rz * Ry * Rx, Rx = Rz = I (exact)
        memcpy(r.R[i], R, sizeof(R));
        // CameraParams::K(): (focal, 0, ppx; 0, focal * aspect, ppy; 0, 0, 1) in double, then convertTo(CV_32F)
        const float Kw[9] = {(float)focal, 0.f, (float)ppx, 0.f, (float)focal, (float)ppy, 0.f, 0.f, 1.f};
        memcpy(r.K_seam[i], Kw, sizeof(Kw));
        r.K_seam[i][0] *= swa; r.K_seam[i][2] *= swa; r.K_seam[i][4] *= swa; r.K_seam[i][5] *= swa;         // :112-116
        const double fc = focal * r.compose_work_aspect, px = ppx * r.compose_work_aspect, py = ppy * r.compose_work_aspect;   // :171-173
        const float Kc[9] = {(float)fc, 0.f, (float)px, 0.f, (float)fc, (float)py, 0.f, 0.f, 1.f};
        memcpy(r.K_compose[i], Kc, sizeof(Kc));
    }
    return MS_OK;
}

void num_bands_rule(int pano_w, int pano_h, float blend_strength, float *blend_width, int *num_bands)
{
    const float bw = std::sqrt((float)((long long)pano_w * pano_h)) * blend_strength / 100.f;               // :183
    *blend_width = bw;
    *num_bands = bw < 1.f ? 0 : (int)(std::ceil(std::log((double)bw) / std::log(2.)) - 1.);                 // :185-193
}

}  // namespace ms
