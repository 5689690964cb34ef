// ms_shim.hpp -- thin C++ shim over the C-ABI (include/ms_stitch.h) that re-exposes the reference's call
// surface: cv::cuda::-style free functions on any GpuMat-like type and a MultiBandBlender-style compositor
// object, so that 360_stitcher/timed.cpp / calibration.cpp style callers change an include and a namespace.
//
// "GpuMat-like" = anything with  data, step, rows, cols, type()  (cv::cuda::GpuMat qualifies unchanged:
// sources/modules/core/include/opencv2/core/cuda.hpp:283-303).  Header-only, no OpenCV dependency.
// Errors become exceptions here (the reference throws cv::Exception), never across the C boundary.
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ms_stitch.h"

namespace msshim {

struct Error : std::runtime_error { int code; Error(int c, const char *m) : std::runtime_error(m), code(c) {} };
inline void check(int rc) { if (rc < 0) throw Error(rc, ms_last_error()); }

template <class Mat> inline ms_image wrap(const Mat &m)
{
    return ms_image{(void *)m.data, (size_t)m.step, m.cols, m.rows, m.type()};
}

// ---- cv::cuda:: free functions (dst must be pre-created, as GpuMat::create would) ------------------------
namespace cuda {
template <class Mat> void remap(const Mat &src, Mat &dst, const Mat &xmap, const Mat &ymap, int interpolation, int borderMode = MS_BORDER_CONSTANT, ms_stream s = nullptr)
{ ms_image a = wrap(src), x = wrap(xmap), y = wrap(ymap), d = wrap(dst); check(ms_remap(&a, &x, &y, &d, interpolation, borderMode, s)); }
template <class Mat> void resize(const Mat &src, Mat &dst, double fx, double fy, ms_stream s = nullptr)
{ ms_image a = wrap(src), d = wrap(dst); check(ms_resize_linear(&a, &d, fx, fy, s)); }
// every view of a frame in one launch (the loop of cuda::resize calls of stitch_online, timed.cpp:75-85)
template <class Mat> void resize(const std::vector<Mat> &src, std::vector<Mat> &dst, double fx, double fy, ms_stream s = nullptr)
{
    std::vector<ms_image> a(src.size()), d(dst.size());
    for (size_t i = 0; i < src.size(); ++i) { a[i] = wrap(src[i]); d[i] = wrap(dst[i]); }
    check(ms_resize_linear_batch(a.data(), d.data(), (int)src.size(), fx, fy, s));
}
template <class Mat> void copyMakeBorder(const Mat &src, Mat &dst, int top, int bottom, int left, int right, int borderType, ms_stream s = nullptr)
{ ms_image a = wrap(src), d = wrap(dst); check(ms_copy_make_border(&a, &d, top, bottom, left, right, borderType, s)); }
template <class Mat> void pyrDown(const Mat &src, Mat &dst, ms_stream s = nullptr)
{ ms_image a = wrap(src), d = wrap(dst); check(ms_pyr_down(&a, &d, s)); }
template <class Mat> void pyrUp(const Mat &src, Mat &dst, ms_stream s = nullptr)
{ ms_image a = wrap(src), d = wrap(dst); check(ms_pyr_up(&a, &d, s)); }
template <class Mat> void subtract(const Mat &a, const Mat &b, Mat &dst, ms_stream s = nullptr)
{ ms_image x = wrap(a), y = wrap(b), d = wrap(dst); check(ms_subtract_16s(&x, &y, &d, s)); }
template <class Mat> void add(const Mat &a, const Mat &b, Mat &dst, ms_stream s = nullptr)
{ ms_image x = wrap(a), y = wrap(b), d = wrap(dst); check(ms_add_16s(&x, &y, &d, s)); }
template <class Mat> void convertTo(const Mat &src, Mat &dst, double alpha = 1.0, ms_stream s = nullptr)
{
    ms_image a = wrap(src), d = wrap(dst);
    check(a.type == d.type ? ms_convert_scale_8u(&a, &d, alpha, s) : ms_convert(&a, &d, alpha, s));
}
}  // namespace cuda

// device::blend launchers (blenders.cpp:48-61)
template <class Mat> void addSrcWeightGpu32F(const Mat &src, const Mat &w, Mat &dst, Mat &dst_w, int rc_w, int rc_h, ms_stream s = nullptr)
{ ms_image a = wrap(src), b = wrap(w), d = wrap(dst), e = wrap(dst_w); check(ms_add_src_weight_32f(&a, &b, &d, &e, rc_w, rc_h, s)); }
template <class Mat> void normalizeUsingWeightMapGpu32F(const Mat &w, Mat &src, int width, int height, ms_stream s = nullptr)
{ ms_image a = wrap(w), d = wrap(src); check(ms_normalize_using_weight_32f(&a, &d, width, height, s)); }
// consume()'s resize + black bars + cvtColor(BGR2YUV_I420) (timed.cpp:251-316) in one pass; i420: contiguous 8UC1 (out_h * 3 / 2) x out_w.  Returns image_height.
template <class Mat> int consume_i420(const Mat &pano8u, Mat &i420, int out_w, int out_h, bool keep_aspect_ratio = true, ms_stream s = nullptr)
{
    ms_image a = wrap(pano8u), b = wrap(i420);
    int ih = 0;
    check(ms_consume_i420(&a, &b, out_w, out_h, keep_aspect_ratio ? 1 : 0, &ih, s));
    return ih;
}

// custom_resize (APP/calibration.h:15)
template <class Mat> void custom_resize(const Mat &in, Mat &out, ms_stream s = nullptr)
{ ms_image a = wrap(in), d = wrap(out); check(ms_custom_resize_32f(&a, &d, s)); }

// ---- the compositor: stitch_calib tables + stitch_one ----------------------------------------------------------
class Compositor {
public:
    // num_bands as MultiBandBlender(try_gpu, num_bands); projection MS_PROJ_CYLINDRICAL is what calibration.cpp:100 ships
    Compositor(int num_views, int src_w, int src_h, int projection, float warp_scale, int num_bands, bool enable_local,
               int out_w, int out_h, int frames_in_flight = 1, int update_mask_margin = 0)
    {
        ms_config c{};
        c.update_mask_margin = update_mask_margin;      // > 0: update_mask() only enqueues (timed.cpp:598-605 could call it after every mesh swap)
        c.struct_size = (unsigned)sizeof(ms_config);
        c.num_views = num_views; c.src_width = src_w; c.src_height = src_h; c.projection = projection; c.warp_scale = warp_scale;
        c.num_bands = num_bands; c.enable_cpw = enable_local; c.out_width = out_w; c.out_height = out_h; c.max_frames = frames_in_flight;
        check(ms_create(&c, &ctx_));
        n_ = num_views;
    }
    ~Compositor() { ms_destroy(ctx_); }
    Compositor(const Compositor &) = delete;
    Compositor &operator=(const Compositor &) = delete;

    // calibrateCameras + warpImages (calibration.cpp:28-248): K, R are the 3x3 CV_32F matrices of cameras[i]
    void setCamera(int i, const float *K, const float *R) { check(ms_set_camera(ctx_, i, K, R)); }
    void setGain(int i, double g) { check(ms_set_gain(ctx_, i, g)); }                 // gc->gains()[i]
    void buildMaps(ms_stream s = nullptr) { check(ms_build_maps(ctx_, s)); }          // gpu_warper->buildMaps + blender->prepare
    void buildMasks(bool voronoi_seams = true, ms_stream s = nullptr) { check(ms_build_masks(ctx_, voronoi_seams ? 1 : 0, s)); }
    void setMask(int i, const uint8_t *host_mask, size_t step) { check(ms_set_mask(ctx_, i, host_mask, step)); }
    void init_gpu(ms_stream s = nullptr) { check(ms_init_blender(ctx_, s)); }         // mb->init_gpu for every view
    void update_mask(int img_num, ms_stream s = nullptr) { check(ms_update_mask(ctx_, img_num, s)); }   // mb->update_mask(idx, x_mesh, y_mesh): uses the view's active mesh
    // MeshWarper::convertMeshesToMap (meshwarper.cpp:823): callable from the recalibration thread
    void convertMeshToMap(int i, const float *mesh_x, const float *mesh_y, int N, int M, ms_stream s = nullptr)
    { check(ms_set_mesh(ctx_, i, mesh_x, mesh_y, N, M, s)); }
    // ... and for all views at once, as MeshWarper::convertMeshesToMap itself loops over the images: num_views meshes of N x M back to back, two launches in all
    void convertMeshesToMap(const float *mesh_x, const float *mesh_y, int N, int M, ms_stream s = nullptr) { check(ms_set_meshes(ctx_, mesh_x, mesh_y, N, M, s)); }

    // stitch_one (timed.cpp:123-152): full_imgs = the NUM_IMAGES uploaded frames; out = caller-owned ring slot
    template <class Mat> void stitch_one(const std::vector<Mat> &full_imgs, Mat *out8u, Mat *out16s, ms_stream s = nullptr)
    {
        std::vector<ms_image> v;
        for (const Mat &m : full_imgs) v.push_back(wrap(m));
        ms_image o8{}, o16{};
        if (out8u) o8 = wrap(*out8u);
        if (out16s) o16 = wrap(*out16s);
        check(ms_stitch(ctx_, (int)(v.size() / n_), v.data(), out8u ? &o8 : nullptr, out16s ? &o16 : nullptr, s));
    }
    // stitch_one on the cameras' NV12 frames (defs.h:10-17; the capture threads' cvtColor(YUV2BGR_NV12), networking.cpp:45-47, folded into the warp): nv12_imgs[i] = 8UC1 (rows * 3 / 2) x cols
    template <class Mat> void stitch_one_nv12(const std::vector<Mat> &nv12_imgs, Mat *out8u, Mat *out16s, ms_stream s = nullptr)
    {
        std::vector<ms_image> v;
        for (const Mat &m : nv12_imgs) v.push_back(wrap(m));
        ms_image o8{}, o16{};
        if (out8u) o8 = wrap(*out8u);
        if (out16s) o16 = wrap(*out16s);
        check(ms_stitch_nv12(ctx_, (int)(v.size() / n_), v.data(), out8u ? &o8 : nullptr, out16s ? &o16 : nullptr, s));
    }
    // stitch_one + consume()'s cvtColor(BGR2YUV_I420) (timed.cpp:308-316) in one: i420[f] = contiguous 8UC1 (rows * 3 / 2) x out_w, rows from i420Rows()
    template <class Mat> void stitch_one_i420(const std::vector<Mat> &full_imgs, std::vector<Mat> &i420, ms_stream s = nullptr)
    {
        std::vector<ms_image> v, o;
        for (const Mat &m : full_imgs) v.push_back(wrap(m));
        for (const Mat &m : i420) o.push_back(wrap(m));
        check(ms_stitch_i420(ctx_, (int)(v.size() / n_), v.data(), o.data(), s));
    }
    void i420Rows(int &first_canvas_row, int &rows) const { check(ms_get_i420_rows(ctx_, &first_canvas_row, &rows)); }
    // MultiBandBlender::feed_online(img, idx, stream) / blend(dst, dst_mask, gpuOut, true) call shape (timed.cpp:110,137)
    template <class Mat> void feed_online(const Mat &img, int idx, ms_stream s = nullptr) { ms_image v = wrap(img); check(ms_feed(ctx_, idx, &v, s)); }
    template <class Mat> void blend(Mat *out8u, Mat *out16s, ms_stream s = nullptr)
    {
        ms_image o8{}, o16{};
        if (out8u) o8 = wrap(*out8u);
        if (out16s) o16 = wrap(*out16s);
        check(ms_blend(ctx_, out8u ? &o8 : nullptr, out16s ? &o16 : nullptr, s));
    }
    ms_pano_geom panoGeom() const { ms_pano_geom g; check(ms_get_pano_geom(ctx_, &g)); return g; }
    ms_view_geom viewGeom(int i) const { ms_view_geom g; check(ms_get_view_geom(ctx_, i, &g)); return g; }
    ms_ctx *raw() { return ctx_; }
    // warpImages' seam-scale half (calibration.cpp:92-135, 224-237): resize, seam warps, gains, Voronoi seams, dilate, resize up, AND
    template <class Mat> void calibrateSeam(const std::vector<Mat> &full_imgs, const float *K_seam, const ms_seam_params &prm, double *gains_out = nullptr, ms_stream s = nullptr)
    {
        std::vector<ms_image> v;
        for (const Mat &m : full_imgs) v.push_back(wrap(m));
        check(ms_calibrate_seam(ctx_, v.data(), K_seam, &prm, gains_out, s));
    }

private:
    ms_ctx *ctx_ = nullptr;
    int n_ = 0;
};


// ---- calibrateCameras / stitch_calib (360_stitcher/calibration.cpp:28-68, 252-311) ---------------------------------------------------
// The rig model and every scale of the calibration, in product code: cameras[i].{focal, ppx, ppy, R} at compose and seam scale plus
// work / seam / compose scales and warper scales (ms_calibrate_cameras restates calibration.cpp:28-68, 101-116, 147-181, 269-288).
inline ms_rig calibrateCameras(int num_views, int full_w, int full_h, double hfov_deg = 90.0,
                               double work_megapix = 0.6, double seam_megapix = 0.01, double compose_megapix = 1.4)      // defs.h:51-53
{
    ms_rig_params q{num_views, full_w, full_h, hfov_deg, work_megapix, seam_megapix, compose_megapix};
    ms_rig rig;
    check(ms_calibrate_cameras(&q, &rig));
    return rig;
}

struct Calibration {
    ms_rig rig;
    int num_bands = 0;             // mb->setNumBands(...) of calibration.cpp:193
    float blend_width = 0.f;
    ms_rect pano_roi{};            // resultRoi(corners, sizes)
    std::vector<double> gains;     // GainCompensator::gains()
};

// stitch_calib (calibration.cpp:252-311) up to the blender being ready for stitch_one: rig + scales (calibrateCameras), compose-scale ROIs and the
// num_bands rule (:163-194), warp maps + blender->prepare (:196-221), the seam-scale pipeline with exposure gains (:92-135, 224-237),
// init_gpu for every view (:240).  full_imgs: the first frame of every camera, device 8UC3 at full size.  With compose_scale more than 10 %
// from 1 the returned compositor expects frames already resized to rig.compose_width x compose_height (msshim::cuda::resize, timed.cpp:75-85).
// projection: the app ships MS_PROJ_CYLINDRICAL (calibration.cpp:100,156); out_w / out_h: canvas for the 8U output, 0 = none.
// update_mask_margin > 0 (CPW only): update_mask() on the returned compositor only enqueues, so the recalibration thread may call it after every
// mesh swap while frames flow; with 0 it is the synchronous rebuild, which makes a concurrent stitch_one wait (never race) for its ~50 ms.
template <class Mat>
std::unique_ptr<Compositor> stitch_calib(const std::vector<Mat> &full_imgs, int projection, bool enable_local, Calibration &cal,
                                         double hfov_deg = 90.0, double work_megapix = 0.6, double seam_megapix = 0.01, double compose_megapix = 1.4,
                                         float blend_strength = 5.f, int out_w = 0, int out_h = 0, int frames_in_flight = 1,
                                         int num_bands_override = -1, ms_stream s = nullptr, int update_mask_margin = 0)
{
    const int n = (int)full_imgs.size();
    if (n < 1) throw Error(MS_ERR_INVALID, "stitch_calib: no images");
    cal.rig = calibrateCameras(n, full_imgs[0].cols, full_imgs[0].rows, hfov_deg, work_megapix, seam_megapix, compose_megapix);
    const ms_rig &rig = cal.rig;
    std::vector<ms_rect> rois(n);
    for (int i = 0; i < n; ++i)            // warper->warpRoi(sz, K, R) at compose scale :163-181
        check(ms_warp_roi(projection, rig.K_compose[i], rig.R[i], rig.compose_warp_scale, rig.compose_width, rig.compose_height, &rois[i]));
    check(ms_result_roi(n, rois.data(), &cal.pano_roi));
    check(ms_num_bands_rule(cal.pano_roi.width, cal.pano_roi.height, blend_strength, &cal.blend_width, &cal.num_bands));       // :183-194
    if (num_bands_override >= 0) cal.num_bands = num_bands_override;
    if (out_w < 0 || out_h < 0) {          // fit the 8U canvas to the panorama ROI (the compositor places pano x = 0 at canvas column out_w / 2)
        const ms_rect &r = cal.pano_roi;
        out_w = (2 * std::max(std::abs(r.x), std::abs(r.x + r.width)) + 1) & ~1;
        out_h = projection == MS_PROJ_SPHERICAL ? ((r.y + r.height + 1) & ~1) : ((2 * std::max(std::abs(r.y), std::abs(r.y + r.height)) + 1) & ~1);
    }
    std::unique_ptr<Compositor> comp(new Compositor(n, rig.compose_width, rig.compose_height, projection, rig.compose_warp_scale, cal.num_bands,
                                                    enable_local, out_w, out_h, frames_in_flight, enable_local ? update_mask_margin : 0));
    for (int i = 0; i < n; ++i) comp->setCamera(i, rig.K_compose[i], rig.R[i]);
    comp->buildMaps(s);                                                                                          // :196, :221
    ms_seam_params sp{rig.seam_scale, rig.seam_warp_scale, enable_local ? 1 : 0, 1};
    cal.gains.assign(n, 1.0);
    comp->calibrateSeam(full_imgs, &rig.K_seam[0][0], sp, cal.gains.data(), s);                                  // :92-135, :224-237
    comp->init_gpu(s);                                                                                           // :240
    return comp;
}

// ---- matchFeatures' descriptor matching (featurefinder.cpp:50-66): dm->knnMatch(f1.descriptors, f2.descriptors, matches, 2) on the device,
// then the 0.7 ratio test.  query / train: GpuMat-like 8UC1 descriptor matrices (upload of ImageFeatures::descriptors).
// Calls emit(queryIdx, trainIdx, distance) for every kept pair, in query order -- push_back(DMatch(...)) in the caller.
template <class Mat, class Emit> void knnRatioMatches(const Mat &query, const Mat &train, Emit emit, ms_stream s = nullptr)
{
    ms_image q = wrap(query), t = wrap(train);
    std::vector<int> idx((size_t)2 * q.rows), dist((size_t)2 * q.rows);
    check(ms_knn_match_hamming2(&q, &t, idx.data(), dist.data(), s));
    for (int i = 0; i < q.rows; ++i)
        if (idx[2 * i + 1] >= 0 && (float)dist[2 * i] < 0.7 * (float)dist[2 * i + 1]) emit(i, idx[2 * i], (float)dist[2 * i]);
}

// ---- featurefinder (360_stitcher/featurefinder.cpp, featurefinder.h:7-13) over the device front-end ----------------------------------------
// Mat = a GpuMat-like device image type with create(rows, cols, type) (cv::cuda::GpuMat qualifies).  ImageFeatures / MatchesInfo have the fields
// of cv::detail::ImageFeatures / MatchesInfo that the application reads; descriptors stay on the device (the reference downloads them and matches
// on the CPU; here matching runs on the device too).
namespace featurefinder {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave; };
struct Size { int width, height; };
struct DMatch { int queryIdx, trainIdx; float distance; };
template <class Mat> struct ImageFeatures { int img_idx = 0; Size img_size{0, 0}; std::vector<KeyPoint> keypoints; Mat descriptors; };
struct MatchesInfo {
    int src_img_idx = -1, dst_img_idx = -1;
    std::vector<DMatch> matches;
    std::vector<unsigned char> inliers_mask;
    int num_inliers = 0;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // empty Mat in the reference when no model was found: have_H = false
    bool have_H = false;
    double confidence = 0;
};

// createMesh's feature masks (meshwarper.cpp:82-115): overlap bands of 400 px on both sides (the hard-coded split of view 3 when the rig has 6 views), minus black pixels
template <class Mat> void featureMasks(const std::vector<Mat> &images, std::vector<Mat> &masks, int overlap = 400, ms_stream s = nullptr)
{
    masks.resize(images.size());
    for (size_t idx = 0; idx < images.size(); ++idx) {
        const int cols = images[idx].cols, rows = images[idx].rows;
        masks[idx].create(rows, cols, MS_8UC1);
        int ax0 = 0, bx0 = cols - overlap;
        if (idx == 3 && images.size() == 6) {        // "Opencv splits the third video (idx == 3) in the middle"
            const float split_l = images[0].cols * 0.25f * 2.0f, split_r = -images[0].cols * 0.25f * 2.0f;
            ax0 = (int)split_l - overlap; bx0 = cols + (int)split_r;
        }
        ms_image im = wrap(images[idx]), mk = wrap(masks[idx]);
        check(ms_feature_mask(&im, ax0, overlap, bx0, overlap, &mk, s));
    }
}

// findFeatures (featurefinder.cpp:13-46) with work_scale < 0 (what createMesh passes): BGR2GRAY, cuda::ORB::create(2500, 1.2f, 8)->detectAndCompute
template <class Mat> void findFeatures(const std::vector<Mat> &images, const std::vector<Mat> &masks, std::vector<ImageFeatures<Mat>> &features, ms_stream s = nullptr,
                                       int nfeatures = 2500, float scale_factor = 1.2f, int nlevels = 8)
{
    features.resize(images.size());
    ms_orb_params prm;
    check(ms_orb_default_params(&prm));
    prm.nfeatures = nfeatures; prm.scale_factor = scale_factor; prm.nlevels = nlevels;
    Mat gray;
    std::vector<float> kp((size_t)6 * nfeatures);
    for (size_t i = 0; i < images.size(); ++i) {
        gray.create(images[i].rows, images[i].cols, MS_8UC1);
        ms_image im = wrap(images[i]), g = wrap(gray);
        check(ms_bgr_to_gray(&im, &g, s));                                             // cuda::cvtColor(gpu_img, gpu_img, CV_BGR2GRAY)
        features[i].descriptors.create(nfeatures, 32, MS_8UC1);
        ms_image d = wrap(features[i].descriptors), mk{};
        const bool have_mask = i < masks.size() && masks[i].data;
        if (have_mask) mk = wrap(masks[i]);
        int n = 0;
        check(ms_orb_detect_and_compute(&g, have_mask ? &mk : nullptr, &prm, kp.data(), nfeatures, &d, &n, s));
        features[i].img_idx = (int)i;
        features[i].img_size = Size{images[i].cols, images[i].rows};
        features[i].keypoints.resize(n);
        for (int k = 0; k < n; ++k) {
            const float *r = &kp[6 * (size_t)k];
            features[i].keypoints[k] = KeyPoint{{r[0], r[1]}, r[5], r[3], r[2], (int)r[4]};
        }
    }
}

// the body shared by matchFeatures (featurefinder.cpp:48-108) and matchFeaturesTemporal (:110-170): knnMatch(k = 2) + 0.7 ratio, centred points,
// findHomography(RANSAC), inlier count; confidence ends as 1 in both
template <class Mat> void matchPair(const ImageFeatures<Mat> &f1, const ImageFeatures<Mat> &f2, MatchesInfo &pm, ms_stream s = nullptr)
{
    pm.matches.clear(); pm.inliers_mask.clear(); pm.num_inliers = 0; pm.have_H = false;
    const int n1 = (int)f1.keypoints.size(), n2 = (int)f2.keypoints.size();
    if (n1 > 0 && n2 > 0) {
        ms_image q = wrap(f1.descriptors), t = wrap(f2.descriptors);
        q.rows = n1; t.rows = n2;
        std::vector<int> idx((size_t)2 * n1), dist((size_t)2 * n1);
        check(ms_knn_match_hamming2(&q, &t, idx.data(), dist.data(), s));
        for (int j = 0; j < n1; ++j)
            if (idx[2 * j + 1] >= 0 && (float)dist[2 * j] < 0.7 * (float)dist[2 * j + 1]) pm.matches.push_back(DMatch{j, idx[2 * j], (float)dist[2 * j]});
    }
    if (!pm.matches.empty()) {
        std::vector<float> src(2 * pm.matches.size()), dst(2 * pm.matches.size());
        for (size_t j = 0; j < pm.matches.size(); ++j) {
            const DMatch &m = pm.matches[j];
            src[2 * j] = f1.keypoints[m.queryIdx].pt.x - f1.img_size.width * 0.5f; src[2 * j + 1] = f1.keypoints[m.queryIdx].pt.y - f1.img_size.height * 0.5f;
            dst[2 * j] = f2.keypoints[m.trainIdx].pt.x - f2.img_size.width * 0.5f; dst[2 * j + 1] = f2.keypoints[m.trainIdx].pt.y - f2.img_size.height * 0.5f;
        }
        pm.inliers_mask.assign(pm.matches.size(), 0);
        const int rc = ms_find_homography_ransac(src.data(), dst.data(), (int)pm.matches.size(), 0, 0, 0, pm.H, pm.inliers_mask.data(), &pm.num_inliers, s);
        check(rc);
        pm.have_H = rc == 0;
    }
    pm.confidence = 1;
}
template <class Mat> void matchFeatures(const std::vector<ImageFeatures<Mat>> &features, std::vector<MatchesInfo> &pairwise_matches, ms_stream s = nullptr)
{
    const int n = (int)features.size();
    for (size_t i = 0; i < pairwise_matches.size(); ++i) {
        const int idx1 = (int)i, idx2 = i == 0 ? n - 1 : (int)i - 1;
        pairwise_matches[i].src_img_idx = idx1; pairwise_matches[i].dst_img_idx = idx2;
        matchPair(features[idx1], features[idx2], pairwise_matches[i], s);
    }
}
template <class Mat> void matchFeaturesTemporal(const std::vector<ImageFeatures<Mat>> &features, const std::vector<ImageFeatures<Mat>> &prev_features,
                                                std::vector<MatchesInfo> &pairwise_matches, ms_stream s = nullptr)
{
    for (size_t i = 0; i < pairwise_matches.size(); ++i) {
        pairwise_matches[i].src_img_idx = pairwise_matches[i].dst_img_idx = (int)i;
        matchPair(features[i], prev_features[i], pairwise_matches[i], s);
    }
}
}  // namespace featurefinder

// ---- MeshWarper: createMesh's host logic around ms_create_mesh (360_stitcher/meshwarper.cpp:158-335) --------------------------
// Works on anything shaped like cv::detail::ImageFeatures (.img_size, .keypoints[i].pt) and cv::detail::MatchesInfo (.src_img_idx,
// .dst_img_idx, .matches[i].queryIdx/.trainIdx, .inliers_mask, .num_inliers): msshim::featurefinder's types above, or the caller's own.
// Differences from the reference, all stated: NUM_IMAGES is the run-time view count (the literal 5 of filterMatches is num_images - 1);
// prev_avg starts at 0 (an uninitialised member in the reference); old matches are kept as resolved positions instead of indices into a
// copy of the old ImageFeatures.
class MeshWarper {
public:
    static const int RECALIB_THRESH = 15;            // defs.h:49
    static const int MAX_FEATURES_PER_IMAGE = 100;   // defs.h:60

    MeshWarper(int num_images, int mesh_width, int mesh_height, float focal_length, double compose_scale, double work_scale)
        : n_(num_images), old_(num_images), prev_(num_images), prev_avg_(2 * num_images, 0.f), have_old_(num_images, 0)
    {
        check(ms_mesh_default_params(&prm_));
        prm_.mesh_cols = mesh_width; prm_.mesh_rows = mesh_height; prm_.focal_length = focal_length;
        prm_.compose_scale = compose_scale; prm_.work_scale = work_scale;
    }
    ms_mesh_params &params() { return prm_; }

    // theta of a camera pair in camera steps, meshwarper.cpp:617-629 / :914-925 (before the * 2 pi / 6, which filterMatches never applies);
    // params().theta_rule = 1 (evenly spaced ring): the wrapped index difference, in radians
    float theta(int src, int dst) const
    {
        if (prm_.theta_rule == 1) {
            int d = dst - src;
            if (d > n_ / 2.0) d -= n_;
            if (d < -n_ / 2.0) d += n_;
            return (float)(d * 2 * 3.1415926535897932384626 / n_);
        }
        float t = (float)(dst - src);
        if (src == 0 && dst == n_ - 1 && prm_.wrap_around) t = -1;
        if (src == 3) t = 4.25f;
        if (src == 4) t = -0.25f;
        return t;
    }

    // filterMatches, meshwarper.cpp:888-946: inliers of the kept camera pairs whose offset is plausible for the rig
    template <class Features, class Matches>
    void filterMatches(const std::vector<Matches> &pairwise_matches, const std::vector<Features> &features, std::vector<std::vector<ms_mesh_match>> &filt) const
    {
        filt.assign(n_, {});
        for (const Matches &pw : pairwise_matches) {
            const int src = pw.src_img_idx, dst = pw.dst_img_idx;
            if (!pw.matches.size() || !pw.num_inliers) continue;
            if (dst != n_ - 1 || (src != 0 && dst == n_ - 1))
                if (src - dst - 1 != 0) continue;
            for (size_t i = 0; i < pw.inliers_mask.size(); ++i) {
                if (!pw.inliers_mask[i]) continue;
                const auto &p1 = features[src].keypoints[pw.matches[i].queryIdx].pt;
                const auto &p2 = features[dst].keypoints[pw.matches[i].trainIdx].pt;
                const float scale = (float)(prm_.compose_scale / prm_.work_scale);
                const float max_x_dist = theta(src, dst) * prm_.focal_length * scale;
                if (std::fabs(p1.y - p2.y) > 40) continue;
                if (std::fabs(max_x_dist - (p1.x - p2.x)) > 300) continue;
                filt[src].push_back(ms_mesh_match{p1.x, p1.y, p2.x, p2.y, dst});
            }
        }
    }

    // the match lists createMesh hands to calcLocalTerm / calcGlobalTerm (meshwarper.cpp:158-292), and the state update (:313-334)
    template <class Features, class Matches>
    std::vector<std::vector<ms_mesh_match>> select(const std::vector<Features> &features, const std::vector<Matches> &pairwise_matches)
    {
        std::vector<std::vector<ms_mesh_match>> all, sel(n_), use(n_);
        filterMatches(pairwise_matches, features, all);
        for (int v = 0; v < n_; ++v)
            sel[v].assign(all[v].begin(), all[v].begin() + std::min<size_t>(MAX_FEATURES_PER_IMAGE, all[v].size()));
        // average x of the matched points per image half: [2 idx] = as source (left partner), [2 idx + 1] = as destination
        std::vector<float> sum(2 * n_, 0.f), cnt(2 * n_, 0.f), avg(2 * n_, 0.f);
        for (int v = 0; v < n_; ++v)
            for (const ms_mesh_match &m : sel[v]) { sum[v * 2] += m.x1; sum[m.dst * 2 + 1] += m.x2; cnt[v * 2]++; cnt[m.dst * 2 + 1]++; }
        for (int k = 0; k < 2 * n_; ++k) if (cnt[k] != 0) avg[k] = sum[k] / cnt[k];
        std::vector<char> use_old(n_, 0);
        for (int v = 0; v < n_; ++v) {
            const int v2 = v == 0 ? n_ - 1 : v - 1;
            const float d = std::fabs(avg[v * 2] - avg[v2 * 2 + 1]), dp = std::fabs(prev_avg_[v * 2] - prev_avg_[v2 * 2 + 1]);
            const bool found = avg[v * 2] != 0 && avg[v2 * 2 + 1] != 0, found_prev = prev_avg_[v * 2] != 0 && prev_avg_[v2 * 2 + 1] != 0;
            use_old[v] = std::fabs(d - dp) < RECALIB_THRESH || (!found && found_prev);
            use[v] = use_old[v] ? old_[v] : sel[v];
        }
        for (int v = 0; v < n_; ++v) {
            prev_[v] = sel[v];
            const int v2 = v == 0 ? n_ - 1 : v - 1;
            if (use_old[v] && !prev_[v].empty() && have_old_[v]) continue;
            old_[v] = sel[v];   have_old_[v] = 1;  prev_avg_[v * 2] = avg[v * 2];   prev_avg_[v * 2 + 1] = avg[v * 2 + 1];
            old_[v2] = sel[v2]; have_old_[v2] = 1; prev_avg_[v2 * 2] = avg[v2 * 2]; prev_avg_[v2 * 2 + 1] = avg[v2 * 2 + 1];
        }
        return use;
    }

    // createMesh from the matching step on: warped = images[idx] (the remapped frames, device 8UC3); meshes come back as n x N x M host floats
    template <class Mat, class Features, class Matches>
    ms_mesh_info createMesh(const std::vector<Mat> &warped, const std::vector<Features> &features, const std::vector<Matches> &pairwise_matches,
                            std::vector<float> &mesh_x, std::vector<float> &mesh_y, ms_stream s = nullptr)
    {
        const std::vector<std::vector<ms_mesh_match>> use = select(features, pairwise_matches);
        std::vector<ms_image> views;
        std::vector<ms_mesh_match> flat;
        std::vector<int> count;
        for (const Mat &m : warped) views.push_back(wrap(m));
        for (const auto &l : use) { flat.insert(flat.end(), l.begin(), l.end()); count.push_back((int)l.size()); }
        mesh_x.assign((size_t)n_ * prm_.mesh_rows * prm_.mesh_cols, 0.f);
        mesh_y.assign(mesh_x.size(), 0.f);
        ms_mesh_info info{};
        check(ms_create_mesh(n_, views.data(), flat.data(), count.data(), nullptr, nullptr, &prm_, mesh_x.data(), mesh_y.data(), &info, s));
        return info;
    }

    // calibrateMeshWarp (meshwarper.cpp:356-376): createMesh + convertMeshesToMap into the compositor
    template <class Mat, class Features, class Matches>
    ms_mesh_info calibrateMeshWarp(Compositor &comp, const std::vector<Mat> &warped, const std::vector<Features> &features,
                                   const std::vector<Matches> &pairwise_matches, ms_stream s = nullptr)
    {
        std::vector<float> mx, my;
        const ms_mesh_info info = createMesh(warped, features, pairwise_matches, mx, my, s);
        const size_t per = (size_t)prm_.mesh_rows * prm_.mesh_cols;
        for (int v = 0; v < n_; ++v) comp.convertMeshToMap(v, mx.data() + v * per, my.data() + v * per, prm_.mesh_rows, prm_.mesh_cols, s);
        return info;
    }

private:
    int n_;
    ms_mesh_params prm_{};
    std::vector<std::vector<ms_mesh_match>> old_, prev_;    // old_matches / prev_matches, resolved to positions
    std::vector<float> prev_avg_;
    std::vector<char> have_old_;                             // !old_features.at(idx).empty()
};

}  // namespace msshim
