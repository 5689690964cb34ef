// ms_shim.hpp -- thin C++ shim over the C-ABI (include/ms_stitch.h) that re-exposes the reference's call
// surface: cv::cuda::-style free functions on any GpuMat-like type and a MultiBandBlender-style compositor
// object, so that 360_stitcher/timed.cpp / calibration.cpp style callers change an include and a namespace.
//
// "GpuMat-like" = anything with  data, step, rows, cols, type()  (cv::cuda::GpuMat qualifies unchanged:
// sources/modules/core/include/opencv2/core/cuda.hpp:283-303).  Header-only, no OpenCV dependency.
// Errors become exceptions here (the reference throws cv::Exception), never across the C boundary.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ms_stitch.h"

namespace msshim {

struct Error : std::runtime_error { int code; Error(int c, const char *m) : std::runtime_error(m), code(c) {} };
inline void check(int rc) { if (rc < 0) throw Error(rc, ms_last_error()); }

template <class Mat> inline ms_image wrap(const Mat &m)
{
    return ms_image{(void *)m.data, (size_t)m.step, m.rows, m.cols, m.type()};
}

// ---- cv::cuda:: free functions (dst must be pre-created, as GpuMat::create would) ------------------------
namespace cuda {
template <class Mat> void remap(const Mat &src, Mat &dst, const Mat &xmap, const Mat &ymap, int interpolation, int borderMode = MS_BORDER_CONSTANT, ms_stream s = nullptr)
{ ms_image a = wrap(src), x = wrap(xmap), y = wrap(ymap), d = wrap(dst); check(ms_remap(&a, &x, &y, &d, interpolation, borderMode, s)); }
template <class Mat> void resize(const Mat &src, Mat &dst, double fx, double fy, ms_stream s = nullptr)
{ ms_image a = wrap(src), d = wrap(dst); check(ms_resize_linear(&a, &d, fx, fy, s)); }
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
// custom_resize (APP/calibration.h:15)
template <class Mat> void custom_resize(const Mat &in, Mat &out, ms_stream s = nullptr)
{ ms_image a = wrap(in), d = wrap(out); check(ms_custom_resize_32f(&a, &d, s)); }

// ---- the compositor: stitch_calib tables + stitch_one ----------------------------------------------------------
class Compositor {
public:
    // num_bands as MultiBandBlender(try_gpu, num_bands); projection MS_PROJ_CYLINDRICAL is what calibration.cpp:100 ships
    Compositor(int num_views, int src_w, int src_h, int projection, float warp_scale, int num_bands, bool enable_local,
               int out_w, int out_h, int frames_in_flight = 1)
    {
        ms_config c{};
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

private:
    ms_ctx *ctx_ = nullptr;
    int n_ = 0;
};

}  // namespace msshim
