/*
 * ms_oracle.h -- CPU ORACLE for the per-frame 360-degree stitching compositor.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped library
 * (video-stitcher_amd/csrc -> libmsstitch.so) never links, includes or calls anything here.
 *
 * What it is: a plain-C restatement of the arithmetic of the reference's *CUDA* hot path
 * (ultravideo/video-stitcher: 360_stitcher/timed.cpp stitch_online/stitch_one ->
 * OpenCV-3.4-fork MultiBandBlender::feed_online/blend -> cudawarping/cudaarithm kernels).
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference; OCV = sources/modules, APP = 360_stitcher).
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - geometry (ROI detection, resultRoi, blender padding): PINNED, exact integer equality with
 *     the reference-derived known answers recorded in SURVEY.md Appendix C
 *     (tests/golden/geometry_kats.json).
 *   - pixel arithmetic (remap/pyramids/accumulate/normalise): PARITY UNPINNED by execution.
 *     The reference's CUDA path cannot run anywhere available (no nvcc / NVIDIA GPU), its
 *     vendored OpenCV cannot be compiled without its cmake-generated headers (cvconfig.h,
 *     opencv_modules.hpp, ...), and no golden image ships with its tests (opencv_extra absent).
 *     The restatement follows the kernels line by line; tolerances are the reference's own
 *     (test_remap.cpp:169, test_pyramids.cpp:80,120, test_blenders.cuda.cpp:90).
 *     Indirect anchors to executions of the real reference (tests/test_oracle_crosscheck.py): the gaps the surveyor MEASURED between the
 *     reference's CPU code and the CUDA arithmetic (pyrDown 1 / 0.17 %, pyrUp 1 / 5.1 %, remap 6 / 55 % on noise and
 *     3 / 2.1 % on a smoothed image) are reproduced between this oracle and its restatements of the CPU flavours (orc_cv_remap_linear_8u,
 *     the (x + 128) >> 8 pyramids): 1 / 0.2 %, 1 / 5 %, 6 / 61.6 %, 3 / 1.97 %; the oracle also meets the criteria of the reference's own
 *     CUDA.Remap / CUDA.Resize / pyramid / CUDA-vs-CPU blender tests against those tests' gold functions.
 *   - orc_cv_remap_linear_8u (cv::remap's CPU fixed-point arithmetic): same status.
 *
 * Floating-point conventions fixed by this oracle (the HIP kernels reproduce them bit-for-bit):
 *   - fp32 everywhere the CUDA kernels use float; compiled with -ffp-contract=off.
 *   - "acc = acc + a*b" chains in the CUDA sources are evaluated as fmaf(a, b, acc), which is
 *     what nvcc's default -fmad=true emits; every other operation is a separate IEEE op.
 *   - float -> u8/s16 "saturate_cast" = round-half-to-even + clamp (cvt.rni.sat), NaN -> 0.
 *   - C-style (short)(float) = truncate toward zero (cvt.rzi), then wrap to 16 bits.
 *
 * Image descriptor convention = cv::cuda::PtrStepSz: base pointer, row pitch in BYTES,
 * rows, cols (OCV/core/include/opencv2/core/cuda_types.hpp:95-120).
 */
#ifndef MS_ORACLE_H
#define MS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_PROJ_PLANE = 0, ORC_PROJ_CYLINDRICAL = 1, ORC_PROJ_SPHERICAL = 2 };

/* ------------------------------------------------------------------ primitives (K1..K19) */

/* K1  cuda::remap, INTER_LINEAR, BORDER_CONSTANT(0), 8UC3. */
void orc_remap_linear_8uc3(const uint8_t *src, size_t sstep, int srows, int scols,
                           const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                           uint8_t *dst, size_t dstep, int drows, int dcols);
/* K1' cuda::remap, INTER_NEAREST, BORDER_CONSTANT(0), 8UC1 (calibration masks). */
void orc_cv_remap_linear_8u(const uint8_t *src, size_t sstep, int srows, int scols, int cn,
                            const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                            uint8_t *dst, size_t dstep, int drows, int dcols);     /* cv::remap, CPU fixed-point flavour (a19) */
void orc_remap_nearest_8uc1(const uint8_t *src, size_t sstep, int srows, int scols,
                            const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                            uint8_t *dst, size_t dstep, int drows, int dcols);
/* K1'' cuda::remap, INTER_LINEAR, BORDER_CONSTANT(0), 8UC1 (update_mask). */
void orc_remap_linear_8uc1(const uint8_t *src, size_t sstep, int srows, int scols,
                           const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                           uint8_t *dst, size_t dstep, int drows, int dcols);
/* K2  cuda::resize INTER_LINEAR 8UC3 / 8UC1; fx,fy are the *inverse* scale factors passed
 * to the kernel (static_cast<float>(1.0/fx)). */
void orc_resize_linear_8u(const uint8_t *src, size_t sstep, int srows, int scols, int cn,
                          uint8_t *dst, size_t dstep, int drows, int dcols, float ifx, float ify);
/* K3  GpuMat::convertTo(same type, alpha): per-byte sat_u8(rn(alpha*v + 0)). width_bytes = cols*cn */
void orc_convert_scale_8u(const uint8_t *src, size_t sstep, uint8_t *dst, size_t dstep,
                          int rows, int width_bytes, double alpha);
/* K4  cuda::copyMakeBorder BORDER_REFLECT, elem_size bytes per pixel (3 for 8UC3). */
void orc_copy_make_border_reflect(const uint8_t *src, size_t sstep, int srows, int scols,
                                  int elem_size, uint8_t *dst, size_t dstep,
                                  int top, int bottom, int left, int right);
/* K4' cuda::copyMakeBorder BORDER_CONSTANT(0) 32FC1 (weight maps). */
void orc_copy_make_border_const_32f(const float *src, size_t sstep, int srows, int scols,
                                    float *dst, size_t dstep, int top, int bottom, int left, int right);
/* K5  convertTo 8U -> 16S (no scale). width = cols*cn elements. */
void orc_convert_8u_16s(const uint8_t *src, size_t sstep, int16_t *dst, size_t dstep, int rows, int width);
/* K17 convertTo 16S -> 8U (saturate). */
void orc_convert_16s_8u(const int16_t *src, size_t sstep, uint8_t *dst, size_t dstep, int rows, int width);
/* mask.convertTo(CV_32F, 1/255.) */
void orc_convert_8u_32f_scale(const uint8_t *src, size_t sstep, float *dst, size_t dstep,
                              int rows, int cols, double alpha);
/* K6  cuda::pyrDown 16SC3 (cn=3) / 16SC1; dst is ((rows+1)/2, (cols+1)/2). */
void orc_pyr_down_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn,
                      int16_t *dst, size_t dstep);
/* K22 cuda::pyrDown 32FC1. */
void orc_pyr_down_32f(const float *src, size_t sstep, int srows, int scols, float *dst, size_t dstep);
/* K7  cuda::pyrUp 16SC3: dst is (2*rows, 2*cols). */
void orc_pyr_up_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn,
                    int16_t *dst, size_t dstep);
/* K8/K11 cuda::subtract / cuda::add on 16S (saturating), width = cols*cn. */
void orc_sub_16s(const int16_t *a, size_t astep, const int16_t *b, size_t bstep,
                 int16_t *dst, size_t dstep, int rows, int width);
void orc_add_16s(const int16_t *a, size_t astep, const int16_t *b, size_t bstep,
                 int16_t *dst, size_t dstep, int rows, int width);
/* K9  addSrcWeightGpu32F over a rows x cols rect (dst/dst_w already offset to the rect). */
void orc_add_src_weight_32f(const int16_t *src, size_t sstep, const float *w, size_t wstep,
                            int16_t *dst, size_t dstep, float *dst_w, size_t dwstep,
                            int rows, int cols);
/* K10 normalizeUsingWeightMapGpu32F. */
void orc_normalize_32f(const float *w, size_t wstep, int16_t *src, size_t sstep, int rows, int cols);
/* the CV_16S-weight flavour of the two (multiband_blend.cu:10-34, 62-83) */
void orc_add_src_weight_16s(const int16_t *src, size_t sstep, const int16_t *w, size_t wstep,
                            int16_t *dst, size_t dstep, int16_t *dst_w, size_t dwstep, int rows, int cols);
void orc_normalize_16s(const int16_t *w, size_t wstep, int16_t *src, size_t sstep, int rows, int cols);
/* K12 compare(w > eps) -> 255/0 ; K12b compare(m == 0) -> 255/0. */
void orc_compare_gt_32f(const float *src, size_t sstep, float thr, uint8_t *dst, size_t dstep, int rows, int cols);
void orc_compare_eq_8u(const uint8_t *src, size_t sstep, uint8_t val, uint8_t *dst, size_t dstep, int rows, int cols);
/* K13 setTo(0, mask) on 16SC3. */
void orc_set_zero_masked_16sc3(int16_t *img, size_t step, const uint8_t *mask, size_t mstep, int rows, int cols);
/* bitwise_and 8U, 3x3 dilate (K20/K21). */
void orc_bitwise_and_8u(const uint8_t *a, size_t astep, const uint8_t *b, size_t bstep,
                        uint8_t *dst, size_t dstep, int rows, int cols);
void orc_dilate3x3_8u(const uint8_t *src, size_t sstep, uint8_t *dst, size_t dstep, int rows, int cols);
/* egress: cvtColor(COLOR_BGR2YUV_I420) of consume() (APP/timed.cpp:308-316); w, h even; dst = planar I420, w*h*3/2 bytes */
void orc_bgr_to_i420(const uint8_t *src, size_t sstep, int w, int h, uint8_t *dst);
/* ingest: cvtColor(COLOR_YUV2BGR_NV12) of the capture threads (APP/networking.cpp:45-47); src = (h*3/2) x w 8UC1 */
void orc_nv12_to_bgr(const uint8_t *src, size_t sstep, int w, int h, uint8_t *dst, size_t dstep);
/* K1 with BORDER_REFLECT (seam-scale image warp, calibration.cpp:118) */
void orc_remap_linear_reflect_8uc3(const uint8_t *src, size_t sstep, int srows, int scols,
                                   const float *mapx, size_t mxstep, const float *mapy, size_t mystep,
                                   uint8_t *dst, size_t dstep, int drows, int dcols);
/* K18 buildWarp{Plane,Cylindrical,Spherical}Maps: k_rinv = 9 floats, t = 3 floats (plane only). */
void orc_build_warp_maps(int proj, int tl_u, int tl_v, int rows, int cols,
                         const float *k_rinv, const float *t, float scale,
                         float *mapx, size_t mxstep, float *mapy, size_t mystep);
/* K19 APP/resize.cu custom_resize 32FC1. */
void orc_custom_resize_32f(const float *in, size_t istep, int rows, int cols,
                           float *out, size_t ostep, int ty, int tx);

/* ------------------------------------------------------------------ geometry (a14, a15, a17) */

typedef struct {
    float k[9], rinv[9], r_kinv[9], k_rinv[9], t[3];
    float scale;
} orc_projector;

/* ProjectorBase::setCameraParams  OCV/stitching/src/warpers.cpp:49-79 (K, R row-major fp32). */
void orc_set_camera_params(orc_projector *p, const float *K, const float *R, const float *T, float scale);
/* warpers_cuda.cpp:108  Mat K_Rinv = K * R.t()  (GEMM with transpose flag: double accumulate). */
void orc_k_rinv_gpu(const float *K, const float *R, float *k_rinv);
void orc_map_forward(int proj, const orc_projector *p, float x, float y, float *u, float *v);
void orc_map_backward(int proj, const orc_projector *p, float u, float v, float *x, float *y);
/* detectResultRoi as dispatched per warper type; returns tl and br (inclusive). */
void orc_detect_result_roi(int proj, const orc_projector *p, int src_w, int src_h,
                           int *tl_x, int *tl_y, int *br_x, int *br_y);
/* CPU RotationWarperBase::buildMaps (warpers_inl.hpp:65-90), maps sized (br-tl+1). */
void orc_build_maps_cpu(int proj, const orc_projector *p, int tl_x, int tl_y, int rows, int cols,
                        float *mapx, size_t mxstep, float *mapy, size_t mystep);

typedef struct { int x, y, width, height; } orc_rect;

/* detail::resultRoi  OCV/stitching/src/util.cpp:125-138 */
orc_rect orc_result_roi(int n, const int *corner_x, const int *corner_y, const int *w, const int *h);

typedef struct {
    int num_bands;            /* cropped num_bands_ (blenders.cpp:243) */
    orc_rect dst_roi_final;   /* unpadded */
    orc_rect dst_roi;         /* padded to multiple of 2^nb (blenders.cpp:249-250) */
} orc_blend_geom;

typedef struct {
    int top, left, bottom, right;     /* blenders.cpp:378-381 */
    int x_tl, y_tl, x_br, y_br;       /* blenders.cpp:425-428 (pano-relative, level 0) */
} orc_view_geom;

/* MultiBandBlender::prepare(Rect)  blenders.cpp:237-252 */
void orc_blender_prepare(orc_rect dst_roi, int actual_num_bands, orc_blend_geom *g);
/* MultiBandBlender::init_gpu geometry prologue  blenders.cpp:353-387,425-428 */
void orc_blender_view_geom(const orc_blend_geom *g, int tl_x, int tl_y, int mask_cols, int mask_rows,
                           orc_view_geom *vg);

/* ------------------------------------------------------------------ calibration-time pieces */

/* cv::distanceTransform(DIST_L1, 3)  OCV/imgproc/src/distransform.cpp:70-137 */
void orc_distance_transform_l1(const uint8_t *src, size_t sstep, int rows, int cols, float *dst, size_t dstep);
/* VoronoiSeamFinder::find(sizes, corners, masks)  OCV/stitching/src/seam_finders.cpp:71-160.
 * masks[i] is rows=h[i], cols=w[i], contiguous (step = w[i]); modified in place. */
void orc_voronoi_seams(int n, const int *corner_x, const int *corner_y, const int *w, const int *h,
                       uint8_t **masks);
/* MeshWarper::convertMeshesToMap for one view  APP/meshwarper.cpp:823-886.
 * mesh_x/mesh_y: N rows x M cols (contiguous); out maps: height x width (contiguous). */
void orc_convert_mesh_to_map(const float *mesh_x, const float *mesh_y, int N, int M,
                             int width, int height, float *map_x, float *map_y);

/* GainCompensator::feed + cv::solve  OCV/stitching/src/exposure_compensate.cpp:71-145; returns 0 if singular */
int orc_gain_compensator(int n, const int *corner_x, const int *corner_y, const int *w, const int *h,
                         const uint8_t *const *images, const uint8_t *const *masks, double *gains);

/* ------------------------------------------------------------------ whole blender (a7, a13) */

typedef struct orc_blender orc_blender;

/* MultiBandBlender(try_gpu=true, num_bands, CV_32F) + prepare(corners, sizes). */
orc_blender *orc_blender_create(int n_views, int num_bands,
                                const int *corner_x, const int *corner_y, const int *w, const int *h);
void orc_blender_destroy(orc_blender *b);
/* init_gpu(_, mask, tl): masks must be fed in view order 0..n-1 (blenders.cpp:344-461). */
void orc_blender_init_view(orc_blender *b, int view, const uint8_t *mask, size_t mstep);
/* feed_online(gpu_img 8UC3 of size w[view] x h[view])  blenders.cpp:700-749 */
void orc_blender_feed(orc_blender *b, int view, const uint8_t *img, size_t step);
/* MultiBandBlender::feed, CPU branch, for a 16SC3 image (blenders.cpp:585-696); flavour 1 */
void orc_blender_feed_cpu(orc_blender *b, int view, const int16_t *img, size_t step);
/* blend(..., gpuOut, true): out = 16SC3 dst_roi_final-sized, out_mask 8UC1 (gpu_dst_mask_);
 * clears the accumulators afterwards (blenders.cpp:758-832). */
void orc_blender_blend(orc_blender *b, int16_t *out, size_t ostep, uint8_t *out_mask, size_t mstep);
void orc_blender_get_geom(const orc_blender *b, orc_blend_geom *g);
void orc_blender_get_view_geom(const orc_blender *b, int view, orc_view_geom *vg);
/* introspection for tests: weight pyramid level of a view / accumulated dst weights / src laplace */
const float *orc_blender_weight_level(const orc_blender *b, int view, int level, int *rows, int *cols, size_t *step);
const int16_t *orc_blender_src_level(const orc_blender *b, int view, int level, int *rows, int *cols, size_t *step);

/* stitch_online (timed.cpp:56-121) for one view, then feed: remap(x_map,y_map) -> gain ->
 * [remap(x_mesh,y_mesh)] -> feed_online.  maps are h[view] x w[view] fp32 contiguous. */
void orc_stitch_online(orc_blender *b, int view, const uint8_t *src, size_t sstep, int srows, int scols,
                       const float *xmap, const float *ymap, double gain,
                       const float *xmesh, const float *ymesh, uint8_t *warped_out /* optional, w*h*3 */);

/* FeatherBlender (blenders.cpp:139-186, 944-951): BASELINE configs[0]'s CPU blender */
void orc_feather_weight_map(const uint8_t *mask, size_t mstep, int rows, int cols, float sharpness, float *w, size_t wstep);
void orc_feather_feed(const int16_t *img, size_t istep, const float *w, size_t wstep, int rows, int cols, int dx, int dy,
                      int16_t *dst, size_t dstep, float *dst_w, size_t dwstep);
void orc_feather_blend(int16_t *dst, size_t dstep, const float *dst_w, size_t dwstep, int rows, int cols, uint8_t *mask, size_t mstep);

void orc_set_num_threads(int n);

/* ---- the reference's CPU pipeline (a19): cv::pyrDown / cv::pyrUp as the CPU MultiBandBlender::feed / blend use them (ms_oracle_cpu.c) ---- */
void orc_cv_pyr_down_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn, int16_t *dst, size_t dstep);   /* (x + 128) >> 8 */
void orc_cv_pyr_down_32f(const float *src, size_t sstep, int srows, int scols, float *dst, size_t dstep);              /* FltCast<float,8>, SSE order */
void orc_cv_pyr_up_16s(const int16_t *src, size_t sstep, int srows, int scols, int cn, int16_t *dst, size_t dstep);     /* (x + 32) >> 6 */
/* flavour 0 (default): the fork's GPU branch (init_gpu / feed_online / blend(gpuOut), CUDA kernel arithmetic);
 * flavour 1: the CPU branch (feed :585-696 incl. the per-call weight pyramid, blend :832-851) with the CPU pyramids above.  Call before init_view. */
void orc_blender_set_flavour(orc_blender *b, int flavour);
/* the reference's CPU per-view stage as the surveyor's harness ran it: cv::remap (fixed-point, BORDER_CONSTANT) -> convertTo(gain)
 * -> convertTo(16S) -> MultiBandBlender::feed; blender must be flavour 1 */
void orc_stitch_online_cpu(orc_blender *b, int v, const uint8_t *src, size_t sstep, int srows, int scols,
                           const float *xmap, const float *ymap, double gain);

/* rounding / saturation helpers of the restatement, exported for the execution pin against the reference's header-only
 * cv::saturate_cast / cvRound (oracle/ref_pin/ref_pin.cpp, tests/test_ref_pin.py) */
int orc_helper_sat_u8f(float v);
int orc_helper_sat_s16f(float v);
int orc_helper_sat_s16i(int v);
int orc_helper_cv_round_f(float v);
int orc_helper_f2i_rd(float v);


/* number of static_cast<short>(float) evaluations whose argument left the int16 range since the last reset (see trunc_s16f in ms_oracle_prims.c) */
long long orc_trunc_s16_range_violations(void);
void orc_trunc_s16_range_reset(void);
#ifdef __cplusplus
}
#endif
#endif
