/*
 * ms_stitch.h -- C-ABI of libmsstitch.so: the MI355X-native (gfx950, hand-written HIP) per-frame
 * 360-degree stitching compositor.  Drop-in boundary for the hot path of ultravideo/video-stitcher
 * (360_stitcher/timed.cpp stitch_online/stitch_one -> OpenCV-3.4-fork MultiBandBlender
 * feed_online/blend and the cv::cuda:: image ops underneath).
 *
 * The reference has no FFI layer; its seam is (1) the host<->.cu launcher boundary -- free functions
 * taking POD PtrStepSz<T> {data, step, cols, rows} + cudaStream_t -- and (2) the C++ API of
 * cv::cuda::* / detail::MultiBandBlender / detail::*WarperGpu.  This header restates both as plain C:
 * every entry point cites the reference interface it replaces (paths relative to the reference
 * repo; OCV = sources/modules, APP = 360_stitcher).  INTEGRATION.md shows the C++ shim a
 * maintainer adds so that timed.cpp-style callers compile unchanged.
 *
 * Conventions
 *   - ms_image has the fields of cv::cuda::PtrStepSz<T> in the reference's own order -- {data, step} (PtrStep, cuda_types.hpp:95-107) then
 *     {cols, rows} (PtrStepSz, :109-120) -- followed by the OpenCV type code: it is filled from a cv::cuda::GpuMat without copying,
 *     {m.data, m.step, m.cols, m.rows, m.type()}, and the kernels' by-value PtrStepSz arguments map onto its first four fields.
 *     All ms_image pointers are DEVICE pointers unless a parameter says "host".
 *   - ms_stream is a hipStream_t (NULL = the default stream).  Calls only enqueue work.
 *   - Every function returns MS_OK (0) or a negative ms_status; ms_last_error() returns the
 *     message of the calling thread's most recent failure (cf. sts_net_get_last_error, APP/netlib.h:74).
 *     Nothing throws across this boundary (the reference throws cv::Exception: OCV/core/include/
 *     opencv2/core/cuda/common.hpp:66-75).
 *   - There is NO CPU fallback: without a HIP device every compute entry point fails with
 *     MS_ERR_NO_DEVICE (the reference's no-CUDA build does throw_no_cuda(), OCV/cudawarping/src/remap.cpp:47).
 */
#ifndef MS_STITCH_H
#define MS_STITCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MS_API __attribute__((visibility("default")))

typedef enum ms_status {
    MS_OK = 0,
    MS_ERR_INVALID = -1,      /* bad argument (CV_Assert in the reference)          */
    MS_ERR_UNSUPPORTED = -2,  /* type/flag combination the hot path never uses      */
    MS_ERR_HIP = -3,          /* HIP runtime error (cudaSafeCall in the reference)  */
    MS_ERR_NO_DEVICE = -4,    /* no gfx950 device visible                           */
    MS_ERR_STATE = -5,        /* call order violated (e.g. stitch before calibrate) */
    MS_ERR_NOMEM = -6,
    MS_ERR_COMM = -7          /* multi-GPU transport (RCCL / host mailbox) failure or timeout: ms_dist.h */
} ms_status;
#define MS_ERR_COMM MS_ERR_COMM

/* OpenCV type codes CV_MAKETYPE(depth, cn) = depth + ((cn-1) << 3)  (OCV/core/include/opencv2/core/hal/interface.h) */
enum { MS_8UC1 = 0, MS_8UC3 = 16, MS_16SC1 = 3, MS_16SC3 = 19, MS_32FC1 = 5 };
/* cv::BorderTypes / cv::InterpolationFlags values used on the path */
enum { MS_BORDER_CONSTANT = 0, MS_BORDER_REFLECT = 2 };
enum { MS_INTER_NEAREST = 0, MS_INTER_LINEAR = 1,
       MS_INTER_LINEAR_FIXPT = 0x101 };   /* cv::remap's CPU arithmetic (1/32-px coordinates, 15-bit weight table; imgwarp.cpp:211-284, :643-850, :1203-1270) */
/* warper kinds: detail::{Plane,Cylindrical,Spherical}WarperGpu (OCV/stitching/include/opencv2/stitching/detail/warpers.hpp:435-550) */
enum { MS_PROJ_PLANE = 0, MS_PROJ_CYLINDRICAL = 1, MS_PROJ_SPHERICAL = 2 };

typedef struct ms_image {   /* PtrStepSz<T> field for field (OCV/core/include/opencv2/core/cuda_types.hpp:95-120), then the type code */
    void *data;
    size_t step;            /* row pitch in bytes */
    int cols, rows;         /* PtrStepSz order: cols first */
    int type;
} ms_image;

typedef struct ms_rect { int x, y, width, height; } ms_rect;   /* cv::Rect */
typedef void *ms_stream;                                        /* hipStream_t */

MS_API const char *ms_last_error(void);
MS_API const char *ms_version(void);
MS_API int ms_device_count(void);       /* cuda::getCudaEnabledDeviceCount (blenders.cpp:226) */

/* =============================================================================================
 * 1. Image-op entry points: one per cv::cuda:: call / device launcher on the hot path
 * ============================================================================================= */

/* cuda::remap(src, dst, xmap, ymap, INTER_LINEAR|INTER_NEAREST, borderMode, Scalar(0), stream)
 * OCV/cudawarping/src/remap.cpp:61-102 -> device::imgproc::remap_gpu<uchar3|uchar> (cuda/remap.cu:56-86).
 * BORDER_CONSTANT(0): 8UC3 linear, 8UC1 linear/nearest (the per-frame warps, timed.cpp:84-101; mask warps);
 * BORDER_REFLECT: 8UC3 linear (the seam-scale image warp, calibration.cpp:118).  dst.size == xmap.size == ymap.size.
 * MS_INTER_LINEAR_FIXPT (8UC1 / 8UC3, BORDER_CONSTANT) is the CPU cv::remap(..., INTER_LINEAR) flavour instead: what MeshWarper::createMesh
 * warps its images with (meshwarper.cpp:72) and what the reference's CPU pipeline (BASELINE configs[0]) produces, bit for bit. */
MS_API int ms_remap(const ms_image *src, const ms_image *xmap, const ms_image *ymap, ms_image *dst,
                    int interpolation, int border_type, ms_stream stream);

/* cuda::resize(src, dst, Size(), fx, fy, INTER_LINEAR, stream)  OCV/cudawarping/src/resize.cpp:57-106
 * -> device::resize<uchar3|uchar> (cuda/resize.cu:71-106).  Two call forms, as resize.cpp:72-81:
 *   fx > 0 && fy > 0  (the app's form, Size() + compose_scale, APP/timed.cpp:77): dst must be
 *                     saturate_cast<int>(cols*fx) x saturate_cast<int>(rows*fy); the kernel gets 1/fx, 1/fy;
 *   fx == 0 && fy == 0: dsize = dst size; fx = dsize.width / src.cols, fy likewise. */
MS_API int ms_resize_linear(const ms_image *src, ms_image *dst, double fx, double fy, ms_stream stream);
/* The same for n 8UC3 images of one geometry in ONE launch: stitch_online's cuda::resize of every view by compose_scale (APP/timed.cpp:75-85 -- on the
 * per-frame path with the shipped COMPOSE_MEGAPIX, defs.h:53) for all views of a frame (or of a batch of frames).  Bit-identical to n ms_resize_linear calls. */
MS_API int ms_resize_linear_batch(const ms_image *src, ms_image *dst, int n, double fx, double fy, ms_stream stream);

/* GpuMat::convertTo(dst, same type, alpha)  OCV/core/src/cuda/gpu_mat.cu:488-512 (exposure gain,
 * APP/timed.cpp:94 / GainCompensator::apply_gpu exposure_compensate.cpp:155-160).  8U any channels, in place ok. */
MS_API int ms_convert_scale_8u(const ms_image *src, ms_image *dst, double alpha, ms_stream stream);

/* GpuMat::convertTo(dst, rtype[, alpha], stream) for the depth changes the path uses:
 * 8U->16S (blenders.cpp:713), 16S->8U (APP/timed.cpp:251), 8U->32F with alpha (blenders.cpp:412). */
MS_API int ms_convert(const ms_image *src, ms_image *dst, double alpha, ms_stream stream);

/* cuda::copyMakeBorder(src, dst, top, bottom, left, right, borderType, Scalar(), stream)
 * OCV/cudaarithm/src/cuda/copy_make_border.cu:126-157.  BORDER_REFLECT for 8UC3/8UC1 (blenders.cpp:711),
 * BORDER_CONSTANT(0) for 32FC1 (blenders.cpp:420). */
MS_API int ms_copy_make_border(const ms_image *src, ms_image *dst, int top, int bottom, int left, int right,
                               int border_type, ms_stream stream);

/* cuda::pyrDown(src, dst, stream)  OCV/cudawarping/src/pyramids.cpp:66-92 -> pyrDown_gpu<short3|float>
 * (cuda/pyr_down.cu:55-188).  16SC3, 16SC1, 32FC1; dst must be ((rows+1)/2, (cols+1)/2). */
MS_API int ms_pyr_down(const ms_image *src, ms_image *dst, ms_stream stream);

/* cuda::pyrUp(src, dst, stream)  OCV/cudawarping/src/pyramids.cpp:106-132 -> pyrUp_gpu<short3>
 * (cuda/pyr_up.cu:55-157).  16SC3 / 16SC1; dst must be (2*rows, 2*cols). */
MS_API int ms_pyr_up(const ms_image *src, ms_image *dst, ms_stream stream);

/* cuda::subtract / cuda::add on CV_16S, no mask  OCV/cudaarithm/src/element_operations.cpp:182,170
 * -> SubOp1/AddOp1<short> (cuda/sub_mat.cu:59-65, cuda/add_mat.cu:59-65).  dst may alias a or b. */
MS_API int ms_subtract_16s(const ms_image *a, const ms_image *b, ms_image *dst, ms_stream stream);
MS_API int ms_add_16s(const ms_image *a, const ms_image *b, ms_image *dst, ms_stream stream);

/* device::blend::addSrcWeightGpu32F(src, src_weight, dst, dst_weight, rc)
 * OCV/stitching/src/blenders.cpp:54-55 -> cuda/multiband_blend.cu:36-60.  dst/dst_weight are the
 * already-offset ROI views (dst(rc)), rc_width x rc_height is the rect size. */
MS_API int ms_add_src_weight_32f(const ms_image *src, const ms_image *src_weight, ms_image *dst,
                                 ms_image *dst_weight, int rc_width, int rc_height, ms_stream stream);

/* device::blend::normalizeUsingWeightMapGpu32F(weight, src, width, height)
 * blenders.cpp:58-59 -> cuda/multiband_blend.cu:85-108. */
MS_API int ms_normalize_using_weight_32f(const ms_image *weight, ms_image *src, int width, int height,
                                         ms_stream stream);

/* The fixed-point flavour of the same two launchers -- MultiBandBlender(weight_type = CV_16S): weights 0..256 (8 fractional bits, blenders.cpp:414-418).
 * device::blend::addSrcWeightGpu16S / normalizeUsingWeightMapGpu16S  blenders.cpp:52-53,56-57 -> cuda/multiband_blend.cu:10-34, 62-83:
 * dst += short((src * w) >> 8), dst_weight += w;  src = short((src << 8) / w).  The app never selects this flavour (weight_type_ stays CV_32F,
 * blenders.hpp:129); provided per-op for completeness of the blender's launcher surface.  A zero weight in the normalise step is an integer division
 * by zero in the reference (undefined); here the result is 0. */
MS_API int ms_add_src_weight_16s(const ms_image *src, const ms_image *src_weight, ms_image *dst,
                                 ms_image *dst_weight, int rc_width, int rc_height, ms_stream stream);
MS_API int ms_normalize_using_weight_16s(const ms_image *weight, ms_image *src, int width, int height,
                                         ms_stream stream);

/* cuda::compare(src32F, eps, dst8U, CMP_GT) / cuda::compare(src8U, 0, dst8U, CMP_EQ)
 * OCV/cudaarithm/src/element_operations.cpp:292 -> cuda/cmp_scalar.cu:59-82 (blenders.cpp:803,808). */
MS_API int ms_compare_gt_32f(const ms_image *src, float thr, ms_image *dst, ms_stream stream);
MS_API int ms_compare_eq_8u(const ms_image *src, int val, ms_image *dst, ms_stream stream);

/* GpuMat::setTo(Scalar::all(0), mask, stream) on 16SC3  gpu_mat.cu:369-374,431 (blenders.cpp:810) */
MS_API int ms_set_zero_masked_16sc3(ms_image *img, const ms_image *mask, ms_stream stream);

/* cuda::bitwise_and 8UC1 (APP/calibration.cpp:237) and the 3x3 dilation of APP/calibration.cpp:209,232 */
MS_API int ms_bitwise_and_8u(const ms_image *a, const ms_image *b, ms_image *dst, ms_stream stream);
MS_API int ms_dilate3x3_8u(const ms_image *src, ms_image *dst, ms_stream stream);

/* device::imgproc::buildWarp{Plane,Spherical,Cylindrical}Maps(tl_u, tl_v, map_x, map_y, k_rinv, r_kinv[, t], scale, stream)
 * OCV/stitching/src/warpers_cuda.cpp:51-67 -> cuda/build_warp_maps.cu:155-216.  k_rinv, r_kinv: 9 floats
 * (HOST), t: 3 floats (HOST, plane only, may be NULL). */
MS_API int ms_build_warp_maps(int projection, int tl_u, int tl_v, ms_image *map_x, ms_image *map_y,
                              const float *k_rinv, const float *r_kinv, const float *t, float scale,
                              ms_stream stream);

/* cvtColor(src, dst, CV_YUV2BGR_NV12): the capture-side conversion the reference runs on the CPU per camera frame
 * (APP/networking.cpp:45-47, 1920x1620 NV12 -> 1920x1080 BGR, defs.h:10-17) -> YUV420sp2RGB888Invoker<0,0>
 * (OCV/imgproc/src/color.cpp).  src 8UC1 of (rows*3/2) x cols (Y plane, then interleaved UV); dst 8UC3, even size. */
MS_API int ms_nv12_to_bgr(const ms_image *src, ms_image *dst, ms_stream stream);
/* The same for n cameras of one geometry in ONE launch (the capture threads convert per camera, networking.cpp:45-47).  Bit-identical to n single calls. */
MS_API int ms_nv12_to_bgr_batch(const ms_image *src, ms_image *dst, int n, ms_stream stream);

/* cvtColor(src, dst, COLOR_BGR2YUV_I420): the encoder input of consume() (APP/timed.cpp:308-316) ->
 * RGB888toYUV420pInvoker (OCV/imgproc/src/color.cpp:9082-9160).  src 8UC3 with even width/height; dst contiguous
 * 8UC1 of (rows*3/2) x cols = planar I420.  Also what bench.py gathers across GPUs (half the bytes of BGR). */
MS_API int ms_bgr_to_i420(const ms_image *src, ms_image *dst, ms_stream stream);
/* consume()'s pixel work on the device, in ONE pass (APP/timed.cpp:251-316): the 8U panorama resized (INTER_LINEAR) to out_width x image_height --
 * image_height = (int)(out_width / cols * rows + 0.5) capped at out_height when keep_aspect_ratio (timed.cpp:256-271), else out_height --, placed in the middle of a
 * black out_width x out_height frame (from row out_height / 2 - image_height / 2, timed.cpp:283-289) and converted to the encoder's planar I420
 * (timed.cpp:308-316; the two BGR2RGB swaps before it cancel).  The reference resizes on the CPU (cv::resize's fixed-point arithmetic); this is
 * cuda::resize's (the arithmetic of ms_resize_linear), i.e. what moving the step to the GPU with the reference's own library gives: <= 1 LSB apart.
 * Equal, bit for bit, to ms_resize_linear + copy into a black frame + ms_bgr_to_i420.  dst: DEVICE 8UC1 contiguous (out_height * 3 / 2) x out_width;
 * image_height (may be NULL) receives the height used. */
MS_API int ms_consume_i420(const ms_image *pano8u, ms_image *dst, int out_width, int out_height, int keep_aspect_ratio, int *image_height, ms_stream stream);
/* cuda::cvtColor(gpu_img, gpu_img, CV_BGR2GRAY) of featurefinder::findFeatures (APP/featurefinder.cpp:34) -> RGB2GrayConvert
 * (OCV/core/include/opencv2/core/cuda/detail/color_detail.hpp:97-101, :444-447): (b * 1868 + g * 9617 + r * 4899 + 2^13) >> 14. */
MS_API int ms_bgr_to_gray(const ms_image *src, ms_image *dst, ms_stream stream);
/* The same conversion for n frames of one geometry (same size and step) in one launch: the egress of a batch of panoramas. */
MS_API int ms_bgr_to_i420_batch(const ms_image *src, ms_image *dst, int n, ms_stream stream);

/* custom_resize(GpuMat &in, GpuMat &out, Size t_size)  APP/resize.cu:30-45, APP/calibration.h:15.
 * out->rows/cols give t_size.  32FC1. */
MS_API int ms_custom_resize_32f(const ms_image *in, ms_image *out, ms_stream stream);

/* =============================================================================================
 * 2. Host geometry (integer/fp32 host code in the reference too)
 * ============================================================================================= */

/* RotationWarper::warpRoi(src_size, K, R)  OCV/stitching/include/opencv2/stitching/detail/warpers_inl.hpp:136-146
 * incl. the per-warper detectResultRoi (warpers.cpp:277-318, warpers.hpp:287-290).  K, R: 9 floats row-major. */
MS_API int ms_warp_roi(int projection, const float *K, const float *R, float scale, int src_w, int src_h,
                       ms_rect *roi);
/* detail::resultRoi(corners, sizes)  OCV/stitching/src/util.cpp:125-138 */
MS_API int ms_result_roi(int n, const ms_rect *view_rois, ms_rect *roi);

/* calibrateCameras + the scale bookkeeping of stitch_calib / warpImages (APP/calibration.cpp:28-68, :101-116, :147-181, :269-288; defs.h:51-53):
 * the fixed rig model -- view i rotated by 2 pi i / N about +y, principal point at the image centre, focal = ppx / tan(hfov / 2), aspect 1 --
 * and every scale the calibration derives from the three megapixel budgets.  Host arithmetic in the reference's types (double scales,
 * float rotation angle, K().convertTo(CV_32F)). */
#define MS_MAX_VIEWS 16
typedef struct ms_rig_params {
    int num_views;                 /* NUM_IMAGES (defs.h:37) */
    int src_width, src_height;     /* full_img_size */
    double hfov_deg;               /* 90 in the reference (calibration.cpp:31) */
    double work_megapix;           /* WORK_MEGAPIX 0.6; negative = original size (defs.h:51, calibration.cpp:269-277) */
    double seam_megapix;           /* SEAM_MEAGPIX 0.01 (defs.h:52, calibration.cpp:280) */
    double compose_megapix;        /* COMPOSE_MEGAPIX 1.4; not positive = compose at the original resolution (defs.h:53, calibration.cpp:147-150) */
} ms_rig_params;
typedef struct ms_rig {
    double work_scale, seam_scale, seam_work_aspect, compose_scale, compose_work_aspect;
    float warped_image_scale;      /* static_cast<float>(cameras[0].focal) at work scale (calibration.cpp:288) */
    float seam_warp_scale;         /* static_cast<float>(warped_image_scale * seam_work_aspect): the seam-scale warper (:101) */
    float compose_warp_scale;      /* warped_image_scale * static_cast<float>(compose_work_aspect): the compose warper (:156) = ms_config.warp_scale */
    int resize_input;              /* |compose_scale - 1| > 0.1: every frame goes through cuda::resize before the remap (timed.cpp:75, calibration.cpp:161) */
    int compose_width, compose_height;   /* cvRound(full * compose_scale) if resize_input, else the full size (:163-164) = ms_config.src_width / height */
    float K_compose[MS_MAX_VIEWS][9];    /* cameras[i].K() after *= compose_work_aspect, as CV_32F (:171-176) -> ms_set_camera */
    float K_seam[MS_MAX_VIEWS][9];       /* K at work scale as CV_32F, four entries times (float)seam_work_aspect (:108-116) -> ms_calibrate_seam */
    float R[MS_MAX_VIEWS][9];            /* Rz * Ry * Rx with only Ry non-trivial (:36-55) */
} ms_rig;
MS_API int ms_calibrate_cameras(const ms_rig_params *params, ms_rig *rig);
/* blend_width = sqrt(area(resultRoi)) * BLEND_STRENGTH / 100 and mb->setNumBands(ceil(log2(blend_width)) - 1) (calibration.cpp:183-194, defs.h:55);
 * num_bands = 0 where the reference falls back to Blender::NO (blend_width < 1). */
MS_API int ms_num_bands_rule(int pano_width, int pano_height, float blend_strength, float *blend_width, int *num_bands);

/* =============================================================================================
 * 3. The compositor context: calibration tables once, then one fused launch sequence per frame
 * ============================================================================================= */

typedef struct ms_ctx ms_ctx;

typedef struct ms_config {
    unsigned struct_size;   /* sizeof(ms_config) of the caller's header: ms_create rejects a mismatch (ABI guard)  */
    int num_views;          /* NUM_IMAGES (APP/defs.h:37)                                             */
    int src_width, src_height;   /* camera frame size (after the optional compose_scale resize)       */
    int projection;         /* MS_PROJ_*: the app ships cylindrical (APP/calibration.cpp:100,156)     */
    float warp_scale;       /* warper scale = warped_image_scale * compose_work_aspect (calibration.cpp:156) */
    int num_bands;          /* MultiBandBlender num_bands (blenders.hpp:129 default 5; calibration.cpp:193) */
    int enable_cpw;         /* enable_local (APP/defs.h:27): second remap through the mesh maps       */
    int out_width, out_height;   /* equirect canvas (0,0 = emit the pano ROI only)                     */
    int max_frames;         /* frames batched per ms_stitch call (1 = live; >1 amortises launches; <= 64) */
    int view_shards;        /* 0 / 1 = this context composites whole frames; S = 2..4: it owns one shard of the views (ms_stitch_partial) */
    int view_shard_index;   /* which shard, 0 .. S-1                                                   */
    int cpu_flavour_remap;  /* != 0: the projection warp uses cv::remap's CPU arithmetic (1/32-px coordinates, 15-bit weights): the
                             * reference's CPU pipeline of BASELINE configs[0]; needs debug_simple_kernels and no CPW; changes results BY DESIGN */
    /* developer knobs: none of them changes a result */
    int debug_simple_kernels;    /* != 0: the one-pixel-per-lane reference kernels instead of the tiled ones      */
    int warp_lds_stage;          /* 0: direct tap gathers (default); 1: source tiles staged in LDS by LDS-DMA (k_warp_a, measured slower on config 2); 2: never stage */
    int raster_tile_order;       /* != 0: work lists in raster order instead of the XCD-aware order               */
    /* Pano-column sharding (the alternative to view sharding): S = 2..16 contexts -- one per GPU -- each composite the columns
     * [ms_get_col_window) of the panorama ROI, bit-identical there to an unsharded context.  The work lists are cut down to what those columns depend on
     * (their own pixels plus <= 3 * 2^num_bands columns of halo on either side, recomputed); views that do not reach the window are never read
     * (their ms_image may be all-zero).  Output pixels outside the window are unspecified.  0 / 1 = the whole panorama. */
    int col_shards;
    int col_shard_index;         /* which shard, 0 .. S-1 */
    /* > 0 (with enable_cpw): ms_update_mask only ENQUEUES work -- the blend tables are double-buffered and the work lists are planned for masks
     * whose edges move by up to this many pixels, so a re-warped mask needs no new plan.  An update whose mesh displaces further than the margin
     * (ms_get_mesh_displacement) leaves the tables as they are.  0: ms_update_mask rebuilds tables and work lists synchronously (calibration-time call). */
    int update_mask_margin;
    int self_check;              /* 1: ms_init_blender verifies the shared-reciprocal division of the band kernels against IEEE division over every
                                  * denominator of this context's tables (a few ms; the test suite sets it); 0: no check (was reserved[0], must be 0 or 1) */
} ms_config;

MS_API int ms_create(const ms_config *cfg, ms_ctx **out);
MS_API void ms_destroy(ms_ctx *ctx);

/* cameras[i].K() / cameras[i].R as fp32 row-major 3x3 (APP/calibration.cpp:28-68,217-221) */
MS_API int ms_set_camera(ms_ctx *ctx, int view, const float *K, const float *R);
/* GainCompensator::gains()[view]  (exposure_compensate.cpp:164-170, APP/timed.cpp:94) */
MS_API int ms_set_gain(ms_ctx *ctx, int view, double gain);

/* warper->warpRoi + gpu_warper->buildMaps per view (APP/calibration.cpp:168-181,221) and
 * blender->prepare(corners, sizes) (calibration.cpp:196 -> blenders.cpp:82-85,237-295). */
MS_API int ms_build_maps(ms_ctx *ctx, ms_stream stream);

/* Compose-size blend masks (APP/calibration.cpp:224-237).  mode 0: warp(255, NEAREST) only;
 * mode 1: AND with Voronoi seams (VoronoiSeamFinder, seam_finders.cpp:85-160) computed at compose size
 * (the app computes them at seam scale and resizes up: stated simplification). */
MS_API int ms_build_masks(ms_ctx *ctx, int mode, ms_stream stream);
/* The reference's own calibration at seam scale (APP/calibration.cpp:92-135, 224-237): resize the N full frames (DEVICE 8UC3) by
 * seam_scale, warp image (LINEAR/REFLECT) and a 255-mask (NEAREST/CONSTANT) with the per-view seam intrinsics K_seam (HOST, N x 9,
 * = K at work scale times seam_work_aspect, calibration.cpp:110-116) and seam_warp_scale, estimate the exposure gains
 * (GainCompensator::feed, exposure_compensate.cpp:71-145), cut Voronoi seams, optionally dilate, resize the seam masks to the compose
 * mask size and AND them with warp(255) at compose scale.  Leaves the masks ready for ms_init_blender; gains_out (HOST, N) may be NULL.
 * The full frames may be larger than the context's source size (compose_scale < 1: the context composites the resized frames, the seam
 * images are still cut from the full ones, calibration.cpp:95); all N must have one size. */
typedef struct ms_seam_params {
    double seam_scale;        /* min(1, sqrt(SEAM_MEGAPIX*1e6 / area))  (calibration.cpp:280)            */
    float seam_warp_scale;    /* warped_image_scale * seam_work_aspect  (calibration.cpp:101)             */
    int dilate;               /* enable_local: 3x3 dilation of the seam masks (calibration.cpp:209,231)   */
    int estimate_gains;       /* also ms_set_gain() the estimated gains                                   */
} ms_seam_params;
MS_API int ms_calibrate_seam(ms_ctx *ctx, const ms_image *full_imgs, const float *K_seam, const ms_seam_params *params,
                             double *gains_out, ms_stream stream);
/* Or hand a mask over, as init_gpu(img, mask, tl) receives it (blenders.cpp:344): HOST 8UC1, view-ROI sized. */
MS_API int ms_set_mask(ms_ctx *ctx, int view, const uint8_t *mask_host, size_t step);
/* mb->init_gpu(_, mask, corner) for every view, in view order (calibration.cpp:240 -> blenders.cpp:344-461),
 * plus the frame-invariant weight sums the reference re-accumulates every frame (blenders.cpp:736, :775). */
MS_API int ms_init_blender(ms_ctx *ctx, ms_stream stream);
/* FeatherBlender (detail::FeatherBlender, blenders.cpp:139-186; the CPU blender of the reference's 2-view example): same call order as
 * ms_init_blender on a context created with num_bands = 0.  Weight maps are createWeightMap(mask, sharpness) (blenders.cpp:944-951:
 * L1 distance transform * sharpness, truncated at 1; OpenCV's default sharpness is 0.02) instead of mask/255; ms_stitch then runs
 * feed x N + blend as one single-band pass. */
MS_API int ms_init_feather(ms_ctx *ctx, float sharpness, ms_stream stream);
/* MeshWarper::interpolateMesh (meshwarper.cpp:337-354) followed by convertMeshesToMap: mesh = start + (end - start) * progress (fp32), the
 * RECALIB_INTERP branch of the recalibration thread (timed.cpp:449-457).  HOST vertex meshes, same conventions as ms_set_mesh. */
MS_API int ms_set_mesh_interp(ms_ctx *ctx, int view, const float *start_x, const float *start_y, const float *end_x, const float *end_y,
                              int N, int M, float progress, ms_stream stream);
/* Largest |x_mesh - x| / |y_mesh - y| (pixels) of the active CPW mesh of `view`, measured on the device by ms_set_mesh / ms_set_mesh_maps.
 * While it stays <= 32 the first CPW remap (timed.cpp:90-94) skips the tiles the mesh remap cannot reach; larger meshes warp whole views. */
MS_API int ms_get_mesh_displacement(ms_ctx *ctx, int view, float *out_px);
/* MultiBandBlender::update_mask (blenders.cpp:297-315; its call is commented out in the reference's main loop, timed.cpp:598-605):
 * the mask init_gpu received for `view`, remapped through the view's active CPW mesh (INTER_LINEAR, BORDER_CONSTANT 0), replaces the view's
 * blend-weight pyramid.  Needs enable_cpw, ms_init_blender and a mesh for the view.  The reference re-accumulates the weight sums each frame;
 * here they are frame-invariant tables, so the call rebuilds them (and the work lists) as ms_init_blender does, keeping meshes and gains.
 * With ms_config.update_mask_margin > 0 the call is asynchronous like ms_set_mesh: it enqueues the re-warp, the view's weight pyramid, the weight
 * sums, the result mask and the owner maps into the inactive copy of the tables on `stream` and swaps; the next ms_stitch waits for it on the GPU.
 * Safe from the recalibration thread while another thread stitches (timed.cpp:598-605 calls it right after the mesh swap).
 * ms_get_mask keeps returning the original mask; ms_set_mask / ms_build_masks / ms_calibrate_seam drop the re-warped one. */
MS_API int ms_update_mask(ms_ctx *ctx, int view, ms_stream stream);

/* MeshWarper::convertMeshesToMap for one view (APP/meshwarper.cpp:823-886): N x M vertex mesh (HOST fp32,
 * forward positions in view-ROI pixels) -> dense backward maps x_mesh/y_mesh, double-buffered; takes
 * effect at the next ms_stitch.  Thread-safe against ms_stitch (recalibration thread, APP/timed.cpp:414-463). */
MS_API int ms_set_mesh(ms_ctx *ctx, int view, const float *mesh_x, const float *mesh_y, int N, int M,
                       ms_stream stream);
/* The same for ALL views of the context in one call -- convertMeshesToMap as the reference calls it (it loops over the images, meshwarper.cpp:823-886):
 * mesh_x / mesh_y = num_views meshes of N x M back to back (HOST).  Two launches and ONE completion event per recalibration instead of two of each per view; bit-identical maps.
 * The caller's arrays are copied into pinned staging before the call returns (a ring of eight generations: the call blocks only if the caller is eight updates ahead of the GPU). */
MS_API int ms_set_meshes(ms_ctx *ctx, const float *mesh_x, const float *mesh_y, int N, int M, ms_stream stream);
/* Or supply the dense maps directly (x_mesh[i], y_mesh[i] GpuMats, APP/timed.cpp:100). DEVICE 32FC1. */
MS_API int ms_set_mesh_maps(ms_ctx *ctx, int view, const ms_image *x_mesh, const ms_image *y_mesh, ms_stream stream);

/* stitch_one (APP/timed.cpp:123-152): for each of n_frames frames, views[f*num_views + i] is the 8UC3
 * camera frame (DEVICE; full_imgs[i] after upload, timed.cpp:68).  Writes per frame:
 *   out8u[f]  8UC3 out_width x out_height equirect canvas (pano ROI placed at its spherical position;
 *             = consume()'s convertTo(CV_8U), timed.cpp:251) -- may be NULL;
 *   out16s[f] 16SC3 pano ROI (dst_roi_final sized) = blend()'s gpuOut (blenders.cpp:811) -- may be NULL.
 * No allocation, no host sync. */
MS_API int ms_stitch(ms_ctx *ctx, int n_frames, const ms_image *views, ms_image *out8u, ms_image *out16s,
                     ms_stream stream);
/* The reference's own call shape (timed.cpp:127-137): stitch_online(view) for every view, then MultiBandBlender::blend.  ms_feed records the
 * view's device image (8UC3, source size; borrowed until ms_blend), ms_blend composites the frame exactly like ms_stitch(ctx, 1, views, ...)
 * and fails with MS_ERR_STATE if a view was not fed since the previous blend.  (feed_online's per-view work is batched across views here, so
 * it runs inside ms_blend; the stream argument of ms_feed is accepted for signature parity and ignored.) */
MS_API int ms_feed(ms_ctx *ctx, int view, const ms_image *img, ms_stream stream);
MS_API int ms_blend(ms_ctx *ctx, ms_image *out8u, ms_image *out16s, ms_stream stream);

/* View sharding (BASELINE configs[4]): create every rank's context with the SAME cameras/masks and
 * ms_config.view_shards = number of shards S (<= 4), view_shard_index = this rank's shard index; shard k owns the contiguous block
 * of views [k*N/S, (k+1)*N/S).  The weighted accumulation into the dst Laplacian pyramid is a sum of int16 terms
 * (multiband_blend.cu:46-49), so each rank writes the partial sums of its views (ms_stitch_partial; entries of `views` for
 * views it does not own are ignored), the caller moves the partial buffers to the sink rank (RCCL send/recv: int16 has no
 * reduce op), and the sink adds them (wrap-around, order independent => bit-identical to one GPU), normalises, collapses and
 * writes the outputs (ms_stitch_finish).  ms_partial_bytes = size of one frame's partial buffer. */
MS_API size_t ms_partial_bytes(const ms_ctx *ctx);
MS_API int ms_stitch_partial(ms_ctx *ctx, int n_frames, const ms_image *views, void *partial_out, ms_stream stream);
MS_API int ms_stitch_finish(ms_ctx *ctx, int n_frames, const void *const *partials, int n_partials,
                            ms_image *out8u, ms_image *out16s, ms_stream stream);

/* gpu_dst_mask_ (blenders.cpp:803): frame-invariant; 8UC1 pano-ROI sized DEVICE image owned by ctx. */
MS_API int ms_get_result_mask(ms_ctx *ctx, ms_image *mask);

/* Calibration tables as a blob (the reference re-runs stitch_calib at every start, APP/timed.cpp:553).  ms_save_tables writes what the static
 * tables of a ready context derive from -- configuration, per view K, R, gain and blend mask, blender kind -- into `buf` (buf == NULL: only *bytes_out, the size
 * needed).  ms_load_tables creates a context from such a blob and rebuilds every table (same library build => bit-identical tables, e.g. on every rank of a
 * multi-GPU run); CPW meshes are run-time state and are set afterwards (ms_set_meshes).  Corrupt / foreign blobs are MS_ERR_INVALID before the device is touched: the FNV-1a
 * checksum covers the whole blob, the embedded configuration included.  A blob replays its context's configuration verbatim, so a column / view SHARD does not save
 * (MS_ERR_UNSUPPORTED): "one blob for every rank" is the blob of the unsharded context; shards are created from the same cameras / gains / masks. */
MS_API int ms_save_tables(ms_ctx *ctx, void *buf, size_t cap, size_t *bytes_out);
MS_API int ms_load_tables(const void *buf, size_t bytes, ms_ctx **out, ms_stream stream);
/* Diagnostics of the band kernels' work classification (no reference counterpart: the reference runs the general arithmetic everywhere).  The 64 x 16 pixel cells
 * of band `level` by class -- owned: one view with weight exactly 1 everywhere (no multiply, no division); exclusive (level 0 only): several views meet but every
 * pixel has one contributing view with weight exactly 1 (binary seam masks) -- the same integer arithmetic, selected by the mask bytes; general: the reference's
 * float multiply + divide.  Results are identical in every class (tests/test_compositor_gpu.py); bands without a map report every cell as general. */
MS_API int ms_get_band_cells(ms_ctx *ctx, int level, unsigned *owned, unsigned *exclusive, unsigned *general);

/* Diagnostics of the work lists (no reference counterpart: the reference launches full grids over every padded view).  build_plan keeps only the tiles some consumer
 * reads; the counts say what a launch touches, e.g. n_warp_tiles x warp_tile_w x warp_tile_h = the level-0 pixels the projection warp writes per frame (bench.py's
 * compulsory-byte figure `frac_useful` is computed from them).  struct_size as in ms_config. */
typedef struct ms_plan_stats {
    unsigned struct_size;
    int warp_tile_w, warp_tile_h, n_warp_tiles;          /* level-0 tiles of the projection warp (CPW: of the mesh remap)        */
    int n_stage1_tiles, n_stage1_reachable;              /* CPW: tiles of the first remap; those within reach of the mesh remap  */
    int down_tile_w, down_tile_h, n_down_tiles[8];       /* output tiles of the reduce from level l to l + 1 (tile kernel)       */
    int blend_tile_w, blend_tile_h, n_blend_tiles[8];    /* panorama tiles of band l (tile kernel)                               */
} ms_plan_stats;
MS_API int ms_get_plan_stats(ms_ctx *ctx, ms_plan_stats *out);

/* Which form of the remap kernels (a5: cuda::remap, cudawarping/src/cuda/remap.cu:56-86) the LAST ms_stitch* call of the context launched -- every form computes the
 * same pixels; which one runs is a function of the context and of the frames handed over.  SHARED_*: the frames of a view have one row step (and, for ALIGNED, one
 * address modulo 4), so a pixel's tap offset is built once for all frames of a call; PER_FRAME_*: they differ (e.g. per-frame ROI views of buffers of different
 * pitch, an odd byte offset) and every frame builds its own; ALIGNED / UNALIGNED: 12-byte aligned tap windows or 8-byte unaligned ones (chosen from the rig's
 * minification).  *stage1_kernel is MS_WARP_KERNEL_NONE without CPW.  Diagnostics for tests and profiles; no reference counterpart. */
enum { MS_WARP_KERNEL_NONE = 0, MS_WARP_KERNEL_SIMPLE = 1, MS_WARP_KERNEL_SHARED_ALIGNED = 2, MS_WARP_KERNEL_SHARED_UNALIGNED = 3, MS_WARP_KERNEL_PER_FRAME_ALIGNED = 4,
       MS_WARP_KERNEL_PER_FRAME_UNALIGNED = 5, MS_WARP_KERNEL_NV12 = 6, MS_WARP_KERNEL_LDS_STAGED = 7 };
MS_API int ms_get_stitch_kernels(ms_ctx *ctx, int *warp_kernel, int *stage1_kernel);

/* geometry read-back (top_/left_/bottom_/right_, x_tl_.., dst_roi_: blenders.hpp:143-175) */
typedef struct ms_view_geom {
    ms_rect roi;                        /* corner + size of the warped view (warpRoi)        */
    int top, left, bottom, right;       /* reflect border (blenders.cpp:378-381)             */
    int x_tl, y_tl, x_br, y_br;         /* padded rect inside the padded pano (blenders.cpp:425-428) */
} ms_view_geom;
typedef struct ms_pano_geom {
    int num_bands;
    ms_rect dst_roi_final, dst_roi;     /* unpadded / padded pano ROI                         */
    int canvas_x, canvas_y;             /* where pano (0,0) lands in the out8u canvas         */
} ms_pano_geom;
MS_API int ms_get_view_geom(const ms_ctx *ctx, int view, ms_view_geom *g);
MS_API int ms_get_pano_geom(const ms_ctx *ctx, ms_pano_geom *g);
/* ---- CPW mesh optimiser: MeshWarper::createMesh after feature matching (360_stitcher/meshwarper.cpp:279-301) ------------------------
 * The feature front-end (ORB, matching, RANSAC: featurefinder.cpp) stays with the caller; what crosses the boundary is what
 * createMesh hands to calcLocalTerm / calcGlobalTerm: per view, the selected matches (filterMatches, meshwarper.cpp:888-946, capped at
 * MAX_FEATURES_PER_IMAGE) as keypoint positions in WARPED-VIEW pixels. */
typedef struct ms_mesh_match {
    float x1, y1;       /* features[src].keypoints[queryIdx].pt, src = the view the list belongs to */
    float x2, y2;       /* features[dst].keypoints[trainIdx].pt (temporal lists: the same view in the previous calibration) */
    int dst;            /* matchWithDst_t::dst (ignored in temporal lists) */
} ms_mesh_match;

typedef struct ms_mesh_params {
    int mesh_cols, mesh_rows;       /* M, N: MESH_WIDTH, MESH_HEIGHT (defs.h:65-66) */
    float alphas[4];                /* ALPHAS (defs.h:69): local, global, smoothness, temporal term weights; temporal is used iff != 0 (defs.h:70) */
    int global_dist;                /* GLOBAL_DIST (defs.h:71) */
    float focal_length;             /* MeshWarper::focal_length */
    double compose_scale, work_scale;
    int wrap_around;                /* wrapAround (defs.h:25) */
    int theta_rule;                 /* 0: the reference's hard-coded 6-camera angles (meshwarper.cpp:617-629); 1: (dst - src) * 2 pi / n_views, wrapped */
    int max_iterations;             /* 0 = Eigen's default, 2 * columns */
    double tolerance;               /* 0 = Eigen's default, DBL_EPSILON */
} ms_mesh_params;

typedef struct ms_mesh_info {
    int rows, cols, nnz;            /* the system actually assembled (the reference allocates more, all-zero, rows) */
    int iterations;                 /* solver.iterations() */
    double error;                   /* solver.error() = ||A^T r|| / ||A^T b|| */
} ms_mesh_info;

MS_API int ms_mesh_default_params(ms_mesh_params *prm);      /* defs.h values, 10 x 10 mesh */
/* calcSmoothnessTerm's salience (meshwarper.cpp:497-563) of every (vertex, triangle) of one warped view (device 8UC3):
 * sal_host[(i * mesh_cols + j) * 8 + t], NaN where triangle t of vertex (j, i) leaves the mesh.  The pixel pass (masked sum / sum of squares
 * per cell crop, cv::meanStdDev under a cv::fillConvexPoly mask) runs on the device. */
MS_API int ms_mesh_saliency(const ms_image *warped_view, int mesh_cols, int mesh_rows, float *sal_host, ms_stream stream);
/* The 8 triangle masks of one mesh cell as calcSmoothnessTerm builds them (meshwarper.cpp:527-551: `Mat mask(cell_h, cell_w)` +
 * cv::fillConvexPoly of the triangle's corners), produced by the device kernel ms_mesh_saliency uses.  masks_host: 8 * cell_h * cell_w bytes
 * (0 / 255), triangle-major; counts_host (may be NULL): 8 set-pixel counts.  Inspection / test aid. */
MS_API int ms_mesh_triangle_masks(int cell_w, int cell_h, uint8_t *masks_host, unsigned *counts_host, ms_stream stream);
/* createMesh's loop + solve (meshwarper.cpp:279-301): for every view calcLocalTerm, calcGlobalTerm, calcSmoothnessTerm[, calcTemporalLocalTerm],
 * then x = LeastSquaresConjugateGradient(A).solve(b) in fp64 on the device and convertVectorToMesh.
 * warped_views[n_views]: the remapped frames `images[idx]` (device 8UC3; mesh_size = their sizes).  matches: the per-view lists back to back,
 * match_count[v] entries each (host); temporal / temporal_count likewise or NULL.  mesh_x / mesh_y: host, n_views * mesh_rows * mesh_cols
 * vertex positions in view pixels, ready for ms_set_mesh / ms_set_mesh_interp.  Synchronises `stream`. */
MS_API int ms_create_mesh(int n_views, const ms_image *warped_views, const ms_mesh_match *matches, const int *match_count,
                          const ms_mesh_match *temporal, const int *temporal_count, const ms_mesh_params *prm,
                          float *mesh_x, float *mesh_y, ms_mesh_info *info, ms_stream stream);

/* DescriptorMatcher::create("BruteForce-Hamming")->knnMatch(query, train, matches, 2) of featurefinder::matchFeatures / matchFeaturesTemporal
 * (360_stitcher/featurefinder.cpp:50-61, :117-128; BFMatcher::knnMatchImpl, features2d/src/matchers.cpp:815-880; cv::batchDistance,
 * core/src/stat.cpp:3946-4008).  query / train: DEVICE 8UC1 descriptor rows (ORB: 32 bytes; any multiple of 4 up to 64).  For query row q,
 * train_idx_host[2q], [2q+1] and distance_host[2q], [2q+1] receive the two nearest train rows in (distance, index) order -- exactly the rows
 * and tie-breaks of the reference's insertion loop; index -1 / distance INT_MAX where train has fewer than two rows.  The 0.7 ratio test and
 * findHomography stay with the caller (msshim::knnRatioMatches does the former).  Synchronises `stream`. */
MS_API int ms_knn_match_hamming2(const ms_image *query, const ms_image *train, int *train_idx_host, int *distance_host, ms_stream stream);

/* ---- feature front-end of the recalibration path ----------------------------------------------------------------------------
 * cuda::ORB::create(nfeatures, scaleFactor, nlevels)->detectAndCompute(gray, mask, keypoints, descriptors) of featurefinder::findFeatures
 * (360_stitcher/featurefinder.cpp:13-46 -> cudafeatures2d/src/orb.cpp:430-865, cuda/fast.cu, cuda/orb.cu): image / mask pyramids, FAST 9-16 with
 * score and 3 x 3 non-max suppression, per-level budgets, Harris responses, intensity-centroid angles, rBRIEF descriptors (WTA_K = 2, HARRIS_SCORE,
 * no blur: the creator's defaults).  gray: DEVICE 8UC1 (ms_bgr_to_gray); mask: DEVICE 8UC1 of the same size or NULL.
 * keypoints_host: max_keypoints x 6 floats (x, y, response, angle in degrees, octave, size), the rows of cuda::ORB's keypoint matrix;
 * descriptors: DEVICE 8UC1 max_keypoints x 32 (row i belongs to keypoint i).  Keypoint order: levels in turn, inside a level by descending Harris
 * response where the level was culled, raster order otherwise (the reference's order is left to atomics and an unstable sort).  Synchronises. */
typedef struct ms_orb_params {
    int nfeatures; float scale_factor; int nlevels;      /* ORB::create(2500, 1.2f, 8)  featurefinder.cpp:15 */
    int edge_threshold, first_level, patch_size, fast_threshold;     /* 31, 0, 31, 20 (cuda::ORB::create defaults) */
} ms_orb_params;
MS_API int ms_orb_default_params(ms_orb_params *prm);
MS_API int ms_orb_detect_and_compute(const ms_image *gray, const ms_image *mask, const ms_orb_params *prm, float *keypoints_host, int max_keypoints,
                                     ms_image *descriptors, int *n_keypoints, ms_stream stream);
/* The feature mask createMesh builds for findFeatures (360_stitcher/meshwarper.cpp:82-115): 255 inside the two column bands where neighbouring
 * views overlap (rect_a / rect_b: x0, width -- cv::rectangle clips them to the image) AND where the warped view is not black (inRange(0, 0) negated:
 * the pixels the projection did not fill).  warped_view: DEVICE 8UC3; mask: DEVICE 8UC1 of the same size. */
MS_API int ms_feature_mask(const ms_image *warped_view, int a_x0, int a_width, int b_x0, int b_width, ms_image *mask, ms_stream stream);
/* cv::findHomography(src, dst, mask, RANSAC) of featurefinder::matchFeatures (featurefinder.cpp:68-90 -> calib3d/src/fundam.cpp:319-402,
 * ptsetreg.cpp:53-290, levmarq.cpp:76-214): src_xy / dst_xy HOST, n points (x, y) each.  reproj_threshold <= 0 -> 3, max_iters <= 0 -> 2000,
 * confidence outside (0, 1) -> 0.995 (the reference's defaults).  Same cv::RNG subset sequence, subset checks, acceptance rule and adaptive
 * iteration count as the reference; all candidate 4-point models are fitted and scored on the device in one launch, the final N-point fit and the
 * 10-iteration Levenberg-Marquardt polish of the winner run on the host.  H: 9 doubles row-major (H[8] = 1); inlier_mask: n bytes (may be NULL).
 * Returns MS_OK, 1 if no model was found (H zeroed; the reference returns an empty Mat), or a negative status. */
MS_API int ms_find_homography_ransac(const float *src_xy, const float *dst_xy, int n, double reproj_threshold, int max_iters, double confidence,
                                     double *H, uint8_t *inlier_mask, int *n_inliers, ms_stream stream);

/* device-resident static tables, for parity tests: x_maps[i]/y_maps[i] (32FC1), masks (8UC1),
 * weight pyramid level (32FC1). Borrowed pointers owned by ctx. */
MS_API int ms_get_maps(const ms_ctx *ctx, int view, ms_image *xmap, ms_image *ymap);
MS_API int ms_get_mask(const ms_ctx *ctx, int view, ms_image *mask);
MS_API int ms_get_weight_level(const ms_ctx *ctx, int view, int level, ms_image *w);
MS_API int ms_get_mesh_maps(const ms_ctx *ctx, int view, ms_image *xmesh, ms_image *ymesh);

/* ms_stitch on the cameras' NV12 frames (APP/defs.h:10-17: the cameras deliver NV12; the capture threads run cvtColor(COLOR_YUV2BGR_NV12) per camera, networking.cpp:45-47):
 * views_nv12[f * num_views + i] = 8UC1, (src_height * 3 / 2) x src_width -- the Y plane followed by the interleaved UV plane -- every frame of a view with the same step.
 * The projection warp samples the planes itself and converts each bilinear tap with cvtColor's integer formula: bit-identical to ms_nv12_to_bgr_batch followed by
 * ms_stitch, without the BGR image (half the input bytes).  With CPW it is the first remap (stage 1) that samples the planes.  Frames that go through the per-frame
 * compose-scale resize first (timed.cpp:75-85) are converted with ms_nv12_to_bgr_batch, resized and stitched from BGR. */
MS_API int ms_stitch_nv12(ms_ctx *ctx, int n_frames, const ms_image *views_nv12, ms_image *out8u, ms_image *out16s, ms_stream stream);
/* Encoder-ready output: the panorama as planar I420, what consume() produces with cvtColor(BGR2YUV_I420) for the encoder (APP/timed.cpp:308-316),
 * written by the level-0 band kernel itself (no 8UC3 canvas, no conversion pass: 1.5 instead of 3 + 4.5 bytes per pixel of traffic).
 * out_i420[f]: contiguous 8UC1 image of (rows * 3 / 2) x out_width holding the canvas rows [first_row, first_row + rows) given by
 * ms_get_i420_rows (the even-aligned row span of the panorama ROI).  Only panorama pixels are written: initialise each buffer once to black
 * (Y = 16, U = V = 128; equals ms_bgr_to_i420 of a zeroed canvas).  Bit-identical to ms_stitch(out8u) + ms_bgr_to_i420 of those rows.
 * Needs the tiled band path (>= 1 band, panorama width a multiple of 8, no view sharding); MS_ERR_UNSUPPORTED otherwise. */
MS_API int ms_stitch_i420(ms_ctx *ctx, int n_frames, const ms_image *views, ms_image *out_i420, ms_stream stream);
MS_API int ms_get_i420_rows(const ms_ctx *ctx, int *first_canvas_row, int *rows);
/* Pano-column sharding: the columns [*begin, *end) of the panorama ROI (= canvas columns [*begin + canvas_x, *end + canvas_x), ms_get_pano_geom) this
 * context composites; the whole ROI for an unsharded context.  Shard boundaries are multiples of 16 columns.  Valid after ms_init_blender. */
MS_API int ms_get_col_window(const ms_ctx *ctx, int *begin, int *end);
/* Bit v set = ms_stitch reads view v (always all views of an unsharded context; a column shard reads only the views that reach its window plus
 * halo; a view shard only the views it owns): the caller need not upload the others and may pass an all-zero ms_image for them. */
MS_API int ms_get_needed_views(const ms_ctx *ctx, unsigned *mask);

/* per-kernel GPU time of the last ms_stitch_timed call (hipEvents on `stream`), for bench.py's roofline.
 * names/ms: arrays of `cap` entries; returns the number of kernels recorded. */
MS_API int ms_stitch_timed(ms_ctx *ctx, int n_frames, const ms_image *views, ms_image *out8u, ms_image *out16s,
                           ms_stream stream, int cap, const char **names, float *ms);

/* Profiling aid: tuned streaming device-to-device copy (16 B per lane, four loads in flight, non-temporal) of a known byte count.
 * Calibrates the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (tools/profile_traffic.sh) and is the measured bandwidth ceiling the
 * per-frame kernels are compared with (bench.py `ceiling`; no reference counterpart). */
MS_API int ms_calib_copy(const void *src, void *dst, size_t bytes, ms_stream stream);
/* ... and its read-only companion (the read side alone sustains more than a copy: the per-frame kernels read 3-5x what they write). */
MS_API int ms_calib_read(const void *src, size_t bytes, ms_stream stream);
/* Counter calibration on the access shapes of the per-frame kernels: every 128-byte line of `buf` is touched exactly once, so a launch moves `bytes` of HBM traffic
 * whatever the shape -- 0: one dword-aligned 12-byte read per lane at a 24-byte stride (k_warp_t's tap read), 1: one 8-byte read per lane (the band kernels' row
 * windows), 2: dword stores in 32-byte runs, four passes per line (k_warp_t's plane stores).  tools/profile_traffic.sh records what FETCH_SIZE / WRITE_SIZE report. */
MS_API int ms_calib_shape(void *buf, size_t bytes, int shape, ms_stream stream);

/* Self-test of the shared-reciprocal division used by the band kernels (normalizeUsingWeightKernel32F,
 * multiband_blend.cu:85-100 divides three channels by the same w + 1e-5): for each of the n HOST denominators,
 * all 65536 int16 numerators are divided both ways on the device; returns the number of results whose bits differ
 * from the compiler's correctly rounded a / d (expected 0), or a negative ms_status. */
MS_API int ms_selftest_divide(const float *denominators_host, int n, ms_stream stream);
/* The same comparison for EVERY float denominator in [d_lo, d_hi] (all bit patterns in between) x all 65536 int16 numerators, enumerated
 * on the device.  [1e-5, 64) -- every value a weight sum + 1e-5 of up to 16 views can take, with margin -- is 1.9e8 denominators = 1.2e13
 * quotients (tests/test_prims_gpu.py runs it: the proof by enumeration behind the bit-exactness of the normalise step). */
MS_API int ms_selftest_divide_range(float d_lo, float d_hi, unsigned long long *mismatches_out, unsigned long long *checked_out, ms_stream stream);

/* Self-test of the single-instruction saturate_cast<uchar>(float) (v_cvt_pk_u8_f32) used by the warp kernels: compares it with
 * the rint / clamp / NaN->0 definition over ALL 2^32 float bit patterns on the device; *mismatches_out must come back 0. */
MS_API int ms_selftest_cvt_u8(unsigned long long *mismatches_out, ms_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MS_STITCH_H */
