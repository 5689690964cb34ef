// api.cpp -- extern "C" wrappers of the image-op entry points (include/ms_stitch.h sections 1-2):
// argument validation (the reference's CV_Assert / CV_Error become status codes), then the launcher.
#include <cstdarg>
#include "launchers.hpp"

namespace ms {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

PinnedScratch &pinned_scratch() { static thread_local PinnedScratch s; return s; }
DeviceScratch &device_scratch() { static thread_local DeviceScratch s; return s; }

int require_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MS_ERR_NO_DEVICE, "no HIP device visible: libmsstitch has no CPU fallback");
    }
    return MS_OK;
}

static int check_img(const ms_image *m, const char *what)
{
    if (!m || !m->data) return fail(MS_ERR_INVALID, "%s: null image", what);
    if (m->rows <= 0 || m->cols <= 0) return fail(MS_ERR_INVALID, "%s: empty image %dx%d", what, m->cols, m->rows);
    static const int esz[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    const size_t row = (size_t)m->cols * (((m->type >> 3) & 7) + 1) * esz[m->type & 7];
    if (m->step < row) return fail(MS_ERR_INVALID, "%s: step %zu smaller than a row (%zu bytes)", what, m->step, row);
    return MS_OK;
}
static int same_size(const ms_image *a, const ms_image *b, const char *what)
{
    if (a->rows != b->rows || a->cols != b->cols) return fail(MS_ERR_INVALID, "%s: size mismatch %dx%d vs %dx%d", what, a->cols, a->rows, b->cols, b->rows);
    return MS_OK;
}

}  // namespace ms

using namespace ms;

#define PRE(fn)  if (int e_ = require_device()) return e_;
#define IMG(m, fn) if (int e_ = check_img(m, fn)) return e_;
#define SAME(a, b, fn) if (int e_ = same_size(a, b, fn)) return e_;

extern "C" {

const char *ms_last_error(void) { return g_err; }
const char *ms_version(void) { return "msstitch 0.1 (gfx950)"; }
int ms_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int ms_remap(const ms_image *src, const ms_image *xmap, const ms_image *ymap, ms_image *dst, int interp, int border_type, ms_stream s)
{
    PRE() IMG(src, "ms_remap src") IMG(xmap, "ms_remap xmap") IMG(ymap, "ms_remap ymap") IMG(dst, "ms_remap dst")
    MS_CHECK(xmap->type == MS_32FC1 && ymap->type == MS_32FC1, "ms_remap: maps must be 32FC1");   // remap.cpp:81
    SAME(xmap, ymap, "ms_remap maps") SAME(xmap, dst, "ms_remap dst")
    MS_CHECK(dst->type == src->type, "ms_remap: dst type must equal src type");
    MS_CHECK(src->data != dst->data, "ms_remap: in-place remap is not supported (the reference allocates a new dst)");
    return launch_remap(*src, *xmap, *ymap, *dst, interp, border_type, as_stream(s));
}

int ms_resize_linear(const ms_image *src, ms_image *dst, double fx, double fy, ms_stream s)
{
    PRE() IMG(src, "ms_resize_linear src") IMG(dst, "ms_resize_linear dst")
    MS_CHECK(dst->type == src->type, "ms_resize_linear: type mismatch");
    MS_CHECK((fx > 0 && fy > 0) || (fx == 0 && fy == 0), "ms_resize_linear: fx, fy must both be > 0 or both be 0");   // resize.cpp:70
    if (fx > 0) {
        const int w = (int)__builtin_rint(src->cols * fx), h = (int)__builtin_rint(src->rows * fy);   // saturate_cast<int>(double) = cvRound
        MS_CHECK(dst->cols == w && dst->rows == h, "ms_resize_linear: dst must be %dx%d for fx=%g fy=%g (resize.cpp:74)", w, h, fx, fy);
    }
    return launch_resize_linear(*src, *dst, fx, fy, as_stream(s));
}

int ms_resize_linear_batch(const ms_image *src, ms_image *dst, int n, double fx, double fy, ms_stream s)
{
    PRE()
    MS_CHECK(src && dst && n >= 1, "ms_resize_linear_batch: null argument / empty batch");
    MS_CHECK((fx > 0 && fy > 0) || (fx == 0 && fy == 0), "ms_resize_linear_batch: fx, fy must both be > 0 or both be 0");
    for (int i = 0; i < n; ++i) {
        MS_CHECK(src[i].data && dst[i].data && src[i].type == MS_8UC3 && dst[i].type == MS_8UC3, "ms_resize_linear_batch: image %d must be a device 8UC3 image", i);
        MS_CHECK(src[i].rows == src[0].rows && src[i].cols == src[0].cols && src[i].step == src[0].step && dst[i].rows == dst[0].rows && dst[i].cols == dst[0].cols &&
                 dst[i].step == dst[0].step, "ms_resize_linear_batch: all images of a batch must share one geometry (image %d differs)", i);
    }
    MS_CHECK(dst[0].rows != src[0].rows || dst[0].cols != src[0].cols, "ms_resize_linear_batch: equal sizes (cuda::resize copies: use the images as they are)");
    if (fx > 0) {
        const int w = (int)__builtin_rint(src[0].cols * fx), h = (int)__builtin_rint(src[0].rows * fy);
        MS_CHECK(dst[0].cols == w && dst[0].rows == h, "ms_resize_linear_batch: dst must be %dx%d for fx=%g fy=%g (resize.cpp:74)", w, h, fx, fy);
    }
    return launch_resize_linear_batch(src, dst, n, fx, fy, as_stream(s));
}

int ms_convert_scale_8u(const ms_image *src, ms_image *dst, double alpha, ms_stream s)
{
    PRE() IMG(src, "ms_convert_scale_8u src") IMG(dst, "ms_convert_scale_8u dst") SAME(src, dst, "ms_convert_scale_8u")
    MS_CHECK((src->type & 7) == 0 && dst->type == src->type, "ms_convert_scale_8u: 8U images of equal type required");
    return launch_convert_scale_8u(*src, *dst, alpha, as_stream(s));
}

int ms_convert(const ms_image *src, ms_image *dst, double alpha, ms_stream s)
{
    PRE() IMG(src, "ms_convert src") IMG(dst, "ms_convert dst") SAME(src, dst, "ms_convert")
    MS_CHECK((src->type >> 3) == (dst->type >> 3), "ms_convert: channel count mismatch");
    return launch_convert(*src, *dst, alpha, as_stream(s));
}

int ms_copy_make_border(const ms_image *src, ms_image *dst, int top, int bottom, int left, int right, int border, ms_stream s)
{
    PRE() IMG(src, "ms_copy_make_border src") IMG(dst, "ms_copy_make_border dst")
    MS_CHECK(top >= 0 && bottom >= 0 && left >= 0 && right >= 0, "ms_copy_make_border: negative border");
    MS_CHECK(dst->rows == src->rows + top + bottom && dst->cols == src->cols + left + right && dst->type == src->type,
             "ms_copy_make_border: dst must be %dx%d of the source type", src->cols + left + right, src->rows + top + bottom);
    return launch_copy_make_border(*src, *dst, top, left, border, as_stream(s));
}

int ms_pyr_down(const ms_image *src, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_pyr_down src") IMG(dst, "ms_pyr_down dst")
    MS_CHECK(dst->rows == (src->rows + 1) / 2 && dst->cols == (src->cols + 1) / 2 && dst->type == src->type,
             "ms_pyr_down: dst must be %dx%d (pyramids.cpp:88)", (src->cols + 1) / 2, (src->rows + 1) / 2);
    return launch_pyr_down(*src, *dst, as_stream(s));
}

int ms_pyr_up(const ms_image *src, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_pyr_up src") IMG(dst, "ms_pyr_up dst")
    MS_CHECK(dst->rows == src->rows * 2 && dst->cols == src->cols * 2 && dst->type == src->type,
             "ms_pyr_up: dst must be %dx%d (pyramids.cpp:128)", src->cols * 2, src->rows * 2);
    return launch_pyr_up(*src, *dst, as_stream(s));
}

int ms_subtract_16s(const ms_image *a, const ms_image *b, ms_image *dst, ms_stream s)
{
    PRE() IMG(a, "ms_subtract_16s a") IMG(b, "ms_subtract_16s b") IMG(dst, "ms_subtract_16s dst") SAME(a, b, "ms_subtract_16s") SAME(a, dst, "ms_subtract_16s")
    MS_CHECK((a->type & 7) == 3 && a->type == b->type && a->type == dst->type, "ms_subtract_16s: 16S images of equal type required");
    return launch_sub_16s(*a, *b, *dst, as_stream(s));
}

int ms_add_16s(const ms_image *a, const ms_image *b, ms_image *dst, ms_stream s)
{
    PRE() IMG(a, "ms_add_16s a") IMG(b, "ms_add_16s b") IMG(dst, "ms_add_16s dst") SAME(a, b, "ms_add_16s") SAME(a, dst, "ms_add_16s")
    MS_CHECK((a->type & 7) == 3 && a->type == b->type && a->type == dst->type, "ms_add_16s: 16S images of equal type required");
    return launch_add_16s(*a, *b, *dst, as_stream(s));
}

int ms_add_src_weight_32f(const ms_image *src, const ms_image *w, ms_image *dst, ms_image *dstw, int rcw, int rch, ms_stream s)
{
    PRE() IMG(src, "ms_add_src_weight_32f src") IMG(w, "ms_add_src_weight_32f weight") IMG(dst, "ms_add_src_weight_32f dst") IMG(dstw, "ms_add_src_weight_32f dst_weight")
    MS_CHECK(src->type == MS_16SC3 && dst->type == MS_16SC3 && w->type == MS_32FC1 && dstw->type == MS_32FC1, "ms_add_src_weight_32f: 16SC3 / 32FC1 required");
    MS_CHECK(rcw > 0 && rch > 0 && rcw <= src->cols && rch <= src->rows && rcw <= w->cols && rch <= w->rows &&
             rcw <= dst->cols && rch <= dst->rows && rcw <= dstw->cols && rch <= dstw->rows, "ms_add_src_weight_32f: rect exceeds an operand");
    return launch_add_src_weight(*src, *w, *dst, *dstw, rcw, rch, as_stream(s));
}

int ms_normalize_using_weight_32f(const ms_image *w, ms_image *src, int width, int height, ms_stream s)
{
    PRE() IMG(w, "ms_normalize_using_weight_32f weight") IMG(src, "ms_normalize_using_weight_32f src")
    MS_CHECK(src->type == MS_16SC3 && w->type == MS_32FC1, "ms_normalize_using_weight_32f: 16SC3 / 32FC1 required");
    MS_CHECK(width > 0 && height > 0 && width <= src->cols && height <= src->rows && width <= w->cols && height <= w->rows, "ms_normalize_using_weight_32f: extent exceeds an operand");
    return launch_normalize(*w, *src, width, height, as_stream(s));
}

int ms_add_src_weight_16s(const ms_image *src, const ms_image *w, ms_image *dst, ms_image *dstw, int rcw, int rch, ms_stream s)
{
    PRE() IMG(src, "ms_add_src_weight_16s src") IMG(w, "ms_add_src_weight_16s weight") IMG(dst, "ms_add_src_weight_16s dst") IMG(dstw, "ms_add_src_weight_16s dst_weight")
    MS_CHECK(src->type == MS_16SC3 && dst->type == MS_16SC3 && w->type == MS_16SC1 && dstw->type == MS_16SC1, "ms_add_src_weight_16s: 16SC3 / 16SC1 required");
    MS_CHECK(rcw > 0 && rch > 0 && rcw <= src->cols && rch <= src->rows && rcw <= w->cols && rch <= w->rows &&
             rcw <= dst->cols && rch <= dst->rows && rcw <= dstw->cols && rch <= dstw->rows, "ms_add_src_weight_16s: rect exceeds an operand");
    return launch_add_src_weight_16s(*src, *w, *dst, *dstw, rcw, rch, as_stream(s));
}

int ms_normalize_using_weight_16s(const ms_image *w, ms_image *src, int width, int height, ms_stream s)
{
    PRE() IMG(w, "ms_normalize_using_weight_16s weight") IMG(src, "ms_normalize_using_weight_16s src")
    MS_CHECK(src->type == MS_16SC3 && w->type == MS_16SC1, "ms_normalize_using_weight_16s: 16SC3 / 16SC1 required");
    MS_CHECK(width > 0 && height > 0 && width <= src->cols && height <= src->rows && width <= w->cols && height <= w->rows, "ms_normalize_using_weight_16s: extent exceeds an operand");
    return launch_normalize_16s(*w, *src, width, height, as_stream(s));
}

int ms_compare_gt_32f(const ms_image *src, float thr, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_compare_gt_32f src") IMG(dst, "ms_compare_gt_32f dst") SAME(src, dst, "ms_compare_gt_32f")
    MS_CHECK(src->type == MS_32FC1 && dst->type == MS_8UC1, "ms_compare_gt_32f: 32FC1 -> 8UC1");
    return launch_compare_gt_32f(*src, thr, *dst, as_stream(s));
}

int ms_compare_eq_8u(const ms_image *src, int val, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_compare_eq_8u src") IMG(dst, "ms_compare_eq_8u dst") SAME(src, dst, "ms_compare_eq_8u")
    MS_CHECK(src->type == MS_8UC1 && dst->type == MS_8UC1, "ms_compare_eq_8u: 8UC1 -> 8UC1");
    return launch_compare_eq_8u(*src, val, *dst, as_stream(s));
}

int ms_set_zero_masked_16sc3(ms_image *img, const ms_image *mask, ms_stream s)
{
    PRE() IMG(img, "ms_set_zero_masked_16sc3 img") IMG(mask, "ms_set_zero_masked_16sc3 mask") SAME(img, mask, "ms_set_zero_masked_16sc3")
    MS_CHECK(img->type == MS_16SC3 && mask->type == MS_8UC1, "ms_set_zero_masked_16sc3: 16SC3 image, 8UC1 mask");
    return launch_zero_masked(*img, *mask, as_stream(s));
}

int ms_bitwise_and_8u(const ms_image *a, const ms_image *b, ms_image *dst, ms_stream s)
{
    PRE() IMG(a, "ms_bitwise_and_8u a") IMG(b, "ms_bitwise_and_8u b") IMG(dst, "ms_bitwise_and_8u dst") SAME(a, b, "ms_bitwise_and_8u") SAME(a, dst, "ms_bitwise_and_8u")
    MS_CHECK(a->type == MS_8UC1 && b->type == MS_8UC1 && dst->type == MS_8UC1, "ms_bitwise_and_8u: 8UC1 required");
    return launch_and_8u(*a, *b, *dst, as_stream(s));
}

int ms_dilate3x3_8u(const ms_image *src, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_dilate3x3_8u src") IMG(dst, "ms_dilate3x3_8u dst") SAME(src, dst, "ms_dilate3x3_8u")
    MS_CHECK(src->type == MS_8UC1 && dst->type == MS_8UC1 && src->data != dst->data, "ms_dilate3x3_8u: distinct 8UC1 images required");
    return launch_dilate3(*src, *dst, as_stream(s));
}

int ms_build_warp_maps(int projection, int tl_u, int tl_v, ms_image *mx, ms_image *my, const float *k_rinv, const float *r_kinv,
                       const float *t, float scale, ms_stream s)
{
    (void)r_kinv;   // uploaded by the reference but unused by its kernels (build_warp_maps.cu:61)
    PRE() IMG(mx, "ms_build_warp_maps map_x") IMG(my, "ms_build_warp_maps map_y") SAME(mx, my, "ms_build_warp_maps")
    MS_CHECK(mx->type == MS_32FC1 && my->type == MS_32FC1 && k_rinv, "ms_build_warp_maps: 32FC1 maps and k_rinv required");
    return launch_build_warp_maps(projection, tl_u, tl_v, *mx, *my, k_rinv, t, scale, as_stream(s));
}

int ms_nv12_to_bgr(const ms_image *src, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_nv12_to_bgr src") IMG(dst, "ms_nv12_to_bgr dst")
    MS_CHECK(src->type == MS_8UC1 && dst->type == MS_8UC3, "ms_nv12_to_bgr: 8UC1 planes -> 8UC3");
    MS_CHECK(dst->cols % 2 == 0 && dst->rows % 2 == 0 && src->cols == dst->cols && src->rows == dst->rows * 3 / 2,
             "ms_nv12_to_bgr: src must be (rows*3/2) x cols of an even-sized dst (color.cpp: CV_Assert)");
    return launch_nv12_to_bgr(*src, *dst, as_stream(s));
}

int ms_nv12_to_bgr_batch(const ms_image *src, ms_image *dst, int n, ms_stream s)
{
    PRE()
    MS_CHECK(src && dst && n >= 1, "ms_nv12_to_bgr_batch: null argument / empty batch");
    for (int i = 0; i < n; ++i) {
        MS_CHECK(src[i].data && dst[i].data && src[i].type == MS_8UC1 && dst[i].type == MS_8UC3, "ms_nv12_to_bgr_batch: image %d: 8UC1 planes -> 8UC3", i);
        MS_CHECK(dst[i].cols % 2 == 0 && dst[i].rows % 2 == 0 && src[i].cols == dst[i].cols && src[i].rows == dst[i].rows * 3 / 2,
                 "ms_nv12_to_bgr_batch: src %d must be (rows*3/2) x cols of an even-sized dst", i);
        MS_CHECK(src[i].rows == src[0].rows && src[i].cols == src[0].cols && src[i].step == src[0].step && dst[i].step == dst[0].step,
                 "ms_nv12_to_bgr_batch: all images of a batch must share one geometry (image %d differs)", i);
    }
    return launch_nv12_to_bgr_batch(src, dst, n, as_stream(s));
}

int ms_bgr_to_i420(const ms_image *src, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_bgr_to_i420 src") IMG(dst, "ms_bgr_to_i420 dst")
    MS_CHECK(src->type == MS_8UC3 && dst->type == MS_8UC1, "ms_bgr_to_i420: 8UC3 -> 8UC1 planes");
    MS_CHECK(src->cols % 2 == 0 && src->rows % 2 == 0, "ms_bgr_to_i420: width and height must be even (color.cpp: CV_Assert)");
    MS_CHECK(dst->cols == src->cols && dst->rows == src->rows * 3 / 2 && dst->step == (size_t)dst->cols,
             "ms_bgr_to_i420: dst must be a contiguous 8UC1 image of %d x %d", src->cols, src->rows * 3 / 2);
    return launch_bgr_to_i420(*src, *dst, as_stream(s));
}

int ms_consume_i420(const ms_image *pano8u, ms_image *dst, int out_width, int out_height, int keep_aspect_ratio, int *image_height, ms_stream s)
{
    PRE() IMG(pano8u, "ms_consume_i420 src") IMG(dst, "ms_consume_i420 dst")
    MS_CHECK(pano8u->type == MS_8UC3 && dst->type == MS_8UC1, "ms_consume_i420: 8UC3 -> 8UC1 planes");
    MS_CHECK(out_width >= 2 && out_height >= 2 && out_width % 2 == 0 && out_height % 2 == 0, "ms_consume_i420: output size must be even");
    MS_CHECK(dst->cols == out_width && dst->rows == out_height * 3 / 2 && dst->step == (size_t)dst->cols,
             "ms_consume_i420: dst must be a contiguous 8UC1 image of %d x %d", out_width, out_height * 3 / 2);
    int ih = out_height;
    if (keep_aspect_ratio) {      // timed.cpp:256-266: width is the restricting dimension, the height follows the aspect ratio (rounded), capped
        ih = (int)((double)out_width / (double)pano8u->cols * pano8u->rows + 0.5);
        if (ih > out_height) ih = out_height;
    }
    MS_CHECK(ih >= 1, "ms_consume_i420: the panorama is too flat for a %d-column output", out_width);
    if (image_height) *image_height = ih;
    return launch_consume_i420(*pano8u, *dst, out_width, out_height, ih, out_height / 2 - ih / 2, as_stream(s));      // timed.cpp:287
}

int ms_bgr_to_gray(const ms_image *src, ms_image *dst, ms_stream s)
{
    PRE() IMG(src, "ms_bgr_to_gray src") IMG(dst, "ms_bgr_to_gray dst")
    MS_CHECK(src->type == MS_8UC3 && dst->type == MS_8UC1, "ms_bgr_to_gray: 8UC3 -> 8UC1");
    SAME(src, dst, "ms_bgr_to_gray")
    return launch_bgr_to_gray(*src, *dst, as_stream(s));
}

int ms_bgr_to_i420_batch(const ms_image *src, ms_image *dst, int n, ms_stream s)
{
    PRE()
    MS_CHECK(src && dst && n >= 0, "ms_bgr_to_i420_batch: null argument");
    for (int i = 0; i < n; ++i) {
        MS_CHECK(src[i].data && dst[i].data && src[i].type == MS_8UC3 && dst[i].type == MS_8UC1, "ms_bgr_to_i420_batch: frame %d: 8UC3 -> 8UC1 planes", i);
        MS_CHECK(src[i].cols == src[0].cols && src[i].rows == src[0].rows && src[i].step == src[0].step, "ms_bgr_to_i420_batch: frame %d differs in geometry from frame 0", i);
        MS_CHECK(src[i].cols % 2 == 0 && src[i].rows % 2 == 0, "ms_bgr_to_i420_batch: width and height must be even (color.cpp: CV_Assert)");
        MS_CHECK(dst[i].cols == src[i].cols && dst[i].rows == src[i].rows * 3 / 2 && dst[i].step == (size_t)dst[i].cols,
                 "ms_bgr_to_i420_batch: dst %d must be a contiguous 8UC1 image of %d x %d", i, src[i].cols, src[i].rows * 3 / 2);
    }
    if (n == 0) return MS_OK;
    return launch_bgr_to_i420_batch(src, dst, n, as_stream(s));
}

int ms_custom_resize_32f(const ms_image *in, ms_image *out, ms_stream s)
{
    PRE() IMG(in, "ms_custom_resize_32f in") IMG(out, "ms_custom_resize_32f out")
    MS_CHECK(in->type == MS_32FC1 && out->type == MS_32FC1, "ms_custom_resize_32f: 32FC1 required");
    return launch_custom_resize(*in, *out, as_stream(s));
}

int ms_warp_roi(int projection, const float *K, const float *R, float scale, int src_w, int src_h, ms_rect *roi)
{
    MS_CHECK(K && R && roi && src_w > 0 && src_h > 0 && projection >= MS_PROJ_PLANE && projection <= MS_PROJ_SPHERICAL, "ms_warp_roi: bad argument");
    Projector p;
    set_camera_params(p, K, R, nullptr, scale);
    *roi = warp_roi(projection, p, src_w, src_h);
    return MS_OK;
}

int ms_result_roi(int n, const ms_rect *rois, ms_rect *roi)
{
    MS_CHECK(n > 0 && rois && roi, "ms_result_roi: bad argument");
    *roi = result_roi(n, rois);
    return MS_OK;
}

int ms_calibrate_cameras(const ms_rig_params *q, ms_rig *rig)
{
    MS_CHECK(q && rig, "ms_calibrate_cameras: null argument");
    MS_CHECK(q->num_views >= 1 && q->num_views <= MS_MAX_VIEWS && q->src_width > 1 && q->src_height > 1, "ms_calibrate_cameras: bad rig (%d views, %dx%d)", q->num_views, q->src_width, q->src_height);
    MS_CHECK(q->hfov_deg > 0 && q->hfov_deg < 180 && q->seam_megapix > 0 && q->work_megapix != 0, "ms_calibrate_cameras: bad field of view / megapixel budgets");
    return calibrate_cameras(*q, *rig);
}

int ms_num_bands_rule(int pano_width, int pano_height, float blend_strength, float *blend_width, int *num_bands)
{
    MS_CHECK(pano_width > 0 && pano_height > 0 && blend_width && num_bands, "ms_num_bands_rule: bad argument");
    num_bands_rule(pano_width, pano_height, blend_strength, blend_width, num_bands);
    return MS_OK;
}

}  // extern "C"
