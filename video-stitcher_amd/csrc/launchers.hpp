// launchers.hpp -- internal host-side launchers shared by the C-ABI wrappers (api.cpp) and the
// compositor (compositor.hip).  The analogue of the reference's host<->.cu seam
// (device::imgproc::*_gpu / device::blend::* free functions taking PtrStepSz by value).
#pragma once
#include "common.hpp"

namespace ms {

int launch_remap(const ms_image &src, const ms_image &xm, const ms_image &ym, ms_image &dst, int interp, int border, hipStream_t st);
int launch_resize_linear(const ms_image &src, ms_image &dst, double fx, double fy, hipStream_t st);
int launch_resize_linear_batch(const ms_image *src, ms_image *dst, int n, double fx, double fy, hipStream_t st);
int launch_convert_scale_8u(const ms_image &src, ms_image &dst, double alpha, hipStream_t st);
int launch_convert(const ms_image &src, ms_image &dst, double alpha, hipStream_t st);
int launch_sub_16s(const ms_image &a, const ms_image &b, ms_image &dst, hipStream_t st);
int launch_add_16s(const ms_image &a, const ms_image &b, ms_image &dst, hipStream_t st);
int launch_and_8u(const ms_image &a, const ms_image &b, ms_image &dst, hipStream_t st);
int launch_compare_gt_32f(const ms_image &src, float thr, ms_image &dst, hipStream_t st);
int launch_compare_eq_8u(const ms_image &src, int val, ms_image &dst, hipStream_t st);
int launch_copy_make_border(const ms_image &src, ms_image &dst, int top, int left, int border_type, hipStream_t st);
int launch_pyr_down(const ms_image &src, ms_image &dst, hipStream_t st);
int launch_pyr_up(const ms_image &src, ms_image &dst, hipStream_t st);
int launch_add_src_weight(const ms_image &src, const ms_image &w, ms_image &dst, ms_image &dstw, int rcw, int rch, hipStream_t st);
int launch_add_src_weight_16s(const ms_image &src, const ms_image &w, ms_image &dst, ms_image &dstw, int rcw, int rch, hipStream_t st);
int launch_normalize_16s(const ms_image &w, ms_image &src, int width, int height, hipStream_t st);
int launch_normalize(const ms_image &w, ms_image &src, int width, int height, hipStream_t st);
int launch_zero_masked(ms_image &img, const ms_image &mask, hipStream_t st);
int launch_dilate3(const ms_image &src, ms_image &dst, hipStream_t st);
int launch_build_warp_maps(int proj, int tl_u, int tl_v, ms_image &mx, ms_image &my, const float *k_rinv, const float *t, float scale, hipStream_t st);
int launch_nv12_to_bgr(const ms_image &src, ms_image &dst, hipStream_t st);
int launch_nv12_to_bgr_batch(const ms_image *src, ms_image *dst, int n, hipStream_t st);
int launch_bgr_to_i420(const ms_image &src, ms_image &dst, hipStream_t st);
int launch_consume_i420(const ms_image &src, ms_image &dst, int out_w, int out_h, int ih, int y_off, hipStream_t st);
int launch_bgr_to_gray(const ms_image &src, ms_image &dst, hipStream_t st);
int launch_bgr_to_i420_batch(const ms_image *src, ms_image *dst, int n, hipStream_t st);
int launch_custom_resize(const ms_image &in, ms_image &out, hipStream_t st);

// host geometry (geometry.cpp)
struct Projector { float k[9], rinv[9], r_kinv[9], k_rinv[9], t[3], scale; };
void set_camera_params(Projector &p, const float *K, const float *R, const float *T, float scale);
void k_rinv_gemm(const float *K, const float *R, float *k_rinv);   // warpers_cuda.cpp: K * R.t()
void r_kinv_gemm(const float *K, const float *R, float *r_kinv);
ms_rect warp_roi(int proj, const Projector &p, int src_w, int src_h);
int calibrate_cameras(const ms_rig_params &q, ms_rig &r);
void num_bands_rule(int pano_w, int pano_h, float blend_strength, float *blend_width, int *num_bands);
ms_rect result_roi(int n, const ms_rect *rois);

struct BlendGeom { int num_bands; ms_rect dst_roi_final, dst_roi; };
struct ViewPad { int top, left, bottom, right, x_tl, y_tl, x_br, y_br; };
BlendGeom blender_prepare(ms_rect dst_roi, int actual_num_bands);
ViewPad blender_view_pad(const BlendGeom &g, int tl_x, int tl_y, int mask_cols, int mask_rows);
// VoronoiSeamFinder over host masks (contiguous, h x w each), in place
// VoronoiSeamFinder over device masks, in place
int voronoi_seams_device(int n, const ms_rect *rois, uint8_t *const *masks_dev, hipStream_t st);                 // calib.hip
int estimate_gains_device(int n, const ms_rect *rois, const uint8_t *const *images_dev, const uint8_t *const *masks_dev, double *gains_host, hipStream_t st);
void feather_weight_map(const uint8_t *mask, int rows, int cols, float sharpness, float *w);

}  // namespace ms
