"""ctypes binding of libmsstitch.so (the HIP compositor) for tests and bench.py.

PyTorch is used only as plumbing: device memory (torch.uint8/int16/float32 tensors stand in for
cv::cuda::GpuMat) and streams.  There is no CPU fallback: importing works anywhere, but every
compute call raises MsError without a gfx950 device, and load() raises if the library is missing.

Names mirror the reference: cuda::remap -> remap, cuda::pyrDown -> pyr_down, ...,
MultiBandBlender::feed_online/blend -> Compositor.stitch (fused) -- see include/ms_stitch.h.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsstitch.so")

MS_8UC1, MS_8UC3, MS_16SC1, MS_16SC3, MS_32FC1 = 0, 16, 3, 19, 5
BORDER_CONSTANT, BORDER_REFLECT = 0, 2
INTER_NEAREST, INTER_LINEAR = 0, 1
INTER_LINEAR_FIXPT = 0x101      # cv::remap's CPU arithmetic
PROJ_PLANE, PROJ_CYLINDRICAL, PROJ_SPHERICAL = 0, 1, 2


class MsError(RuntimeError):
    pass


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("step", C.c_size_t), ("cols", C.c_int), ("rows", C.c_int), ("type", C.c_int)]      # PtrStepSz order


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]

    def tuple(self):
        return (self.x, self.y, self.width, self.height)


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint), ("num_views", C.c_int), ("src_width", C.c_int), ("src_height", C.c_int), ("projection", C.c_int),
                ("warp_scale", C.c_float), ("num_bands", C.c_int), ("enable_cpw", C.c_int),
                ("out_width", C.c_int), ("out_height", C.c_int), ("max_frames", C.c_int),
                ("view_shards", C.c_int), ("view_shard_index", C.c_int), ("cpu_flavour_remap", C.c_int),
                ("debug_simple_kernels", C.c_int), ("warp_lds_stage", C.c_int), ("raster_tile_order", C.c_int),
                ("col_shards", C.c_int), ("col_shard_index", C.c_int), ("update_mask_margin", C.c_int), ("self_check", C.c_int)]


class SeamParams(C.Structure):
    _fields_ = [("seam_scale", C.c_double), ("seam_warp_scale", C.c_float), ("dilate", C.c_int), ("estimate_gains", C.c_int)]


class ViewGeom(C.Structure):
    _fields_ = [("roi", Rect)] + [(n, C.c_int) for n in ("top", "left", "bottom", "right", "x_tl", "y_tl", "x_br", "y_br")]


class PanoGeom(C.Structure):
    _fields_ = [("num_bands", C.c_int), ("dst_roi_final", Rect), ("dst_roi", Rect), ("canvas_x", C.c_int), ("canvas_y", C.c_int)]


class RigParams(C.Structure):
    _fields_ = [("num_views", C.c_int), ("src_width", C.c_int), ("src_height", C.c_int), ("hfov_deg", C.c_double),
                ("work_megapix", C.c_double), ("seam_megapix", C.c_double), ("compose_megapix", C.c_double)]


class Rig(C.Structure):
    _fields_ = [("work_scale", C.c_double), ("seam_scale", C.c_double), ("seam_work_aspect", C.c_double), ("compose_scale", C.c_double),
                ("compose_work_aspect", C.c_double), ("warped_image_scale", C.c_float), ("seam_warp_scale", C.c_float),
                ("compose_warp_scale", C.c_float), ("resize_input", C.c_int), ("compose_width", C.c_int), ("compose_height", C.c_int),
                ("K_compose", (C.c_float * 9) * 16), ("K_seam", (C.c_float * 9) * 16), ("R", (C.c_float * 9) * 16)]


class PlanStats(C.Structure):
    _fields_ = [("struct_size", C.c_uint), ("warp_tile_w", C.c_int), ("warp_tile_h", C.c_int), ("n_warp_tiles", C.c_int), ("n_stage1_tiles", C.c_int),
                ("n_stage1_reachable", C.c_int), ("down_tile_w", C.c_int), ("down_tile_h", C.c_int), ("n_down_tiles", C.c_int * 8),
                ("blend_tile_w", C.c_int), ("blend_tile_h", C.c_int), ("n_blend_tiles", C.c_int * 8)]


class MeshMatch(C.Structure):
    _fields_ = [("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float), ("y2", C.c_float), ("dst", C.c_int)]


class MeshParams(C.Structure):
    _fields_ = [("mesh_cols", C.c_int), ("mesh_rows", C.c_int), ("alphas", C.c_float * 4), ("global_dist", C.c_int),
                ("focal_length", C.c_float), ("compose_scale", C.c_double), ("work_scale", C.c_double), ("wrap_around", C.c_int),
                ("theta_rule", C.c_int), ("max_iterations", C.c_int), ("tolerance", C.c_double)]


class MeshInfo(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("nnz", C.c_int), ("iterations", C.c_int), ("error", C.c_double)]


EXPORTS = [
    "ms_last_error", "ms_version", "ms_device_count", "ms_remap", "ms_resize_linear", "ms_convert_scale_8u", "ms_convert",
    "ms_copy_make_border", "ms_pyr_down", "ms_pyr_up", "ms_subtract_16s", "ms_add_16s", "ms_add_src_weight_32f",
    "ms_normalize_using_weight_32f", "ms_add_src_weight_16s", "ms_normalize_using_weight_16s", "ms_compare_gt_32f", "ms_compare_eq_8u", "ms_set_zero_masked_16sc3",
    "ms_bitwise_and_8u", "ms_dilate3x3_8u", "ms_build_warp_maps", "ms_custom_resize_32f", "ms_warp_roi", "ms_result_roi",
    "ms_calibrate_cameras", "ms_num_bands_rule", "ms_orb_default_params", "ms_orb_detect_and_compute", "ms_find_homography_ransac", "ms_feature_mask",
    "ms_create", "ms_destroy", "ms_set_camera", "ms_set_gain", "ms_build_maps", "ms_build_masks", "ms_set_mask",
    "ms_init_blender", "ms_set_mesh", "ms_set_meshes", "ms_set_mesh_maps", "ms_stitch", "ms_get_result_mask", "ms_get_band_cells", "ms_get_view_geom",
    "ms_get_pano_geom", "ms_get_maps", "ms_get_mask", "ms_get_weight_level", "ms_get_mesh_maps", "ms_stitch_timed",
    "ms_selftest_divide", "ms_selftest_divide_range", "ms_calib_copy", "ms_calib_read", "ms_mesh_triangle_masks", "ms_bgr_to_i420", "ms_calibrate_seam", "ms_nv12_to_bgr", "ms_partial_bytes", "ms_stitch_partial", "ms_stitch_finish", "ms_selftest_cvt_u8", "ms_init_feather", "ms_get_mesh_displacement", "ms_set_mesh_interp", "ms_feed", "ms_blend", "ms_update_mask",
    "ms_mesh_default_params", "ms_mesh_saliency", "ms_create_mesh", "ms_knn_match_hamming2", "ms_bgr_to_i420_batch", "ms_bgr_to_gray", "ms_stitch_i420", "ms_get_i420_rows", "ms_get_col_window", "ms_get_needed_views", "ms_consume_i420", "ms_resize_linear_batch", "ms_nv12_to_bgr_batch",
    "ms_save_tables", "ms_load_tables", "ms_calib_shape", "ms_stitch_nv12", "ms_get_plan_stats", "ms_get_stitch_kernels",
]

_lib = None


def load():
    """Load libmsstitch.so; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MsError("libmsstitch.so not built (run video-stitcher_amd/build.sh or __graft_entry__.build()); "
                          "there is no CPU fallback")
        _lib = C.CDLL(os.environ.get("MSSTITCH_LIB", LIB_PATH))       # MSSTITCH_LIB: developer knob for A/B builds of the same ABI
        _lib.ms_last_error.restype = C.c_char_p
        _lib.ms_version.restype = C.c_char_p
    return _lib


def _chk(code):
    if code < 0:
        raise MsError("msstitch error %d: %s" % (code, load().ms_last_error().decode()))
    return code


_TYPES = {}


def _torch():
    import torch
    if not _TYPES:
        _TYPES.update({(torch.uint8, 1): MS_8UC1, (torch.uint8, 3): MS_8UC3, (torch.int16, 1): MS_16SC1,
                       (torch.int16, 3): MS_16SC3, (torch.float32, 1): MS_32FC1})
    return torch


def img(t):
    """torch tensor (H,W) or (H,W,C), last dims contiguous, on the GPU -> ms_image (borrowed)."""
    torch = _torch()
    assert t.is_cuda, "msstitch operates on device memory only"
    cn = 1 if t.dim() == 2 else t.shape[2]
    if t.dim() == 3:
        assert t.stride(2) == 1 and t.stride(1) == cn
    else:
        assert t.stride(1) == 1
    return Image(t.data_ptr(), t.stride(0) * t.element_size(), t.shape[1], t.shape[0], _TYPES[(t.dtype, cn)])


def tensor_of(image, dtype=None):
    """Borrowed ms_image (device memory owned by a context) -> torch tensor COPY."""
    torch = _torch()
    dt = {MS_8UC1: (torch.uint8, 1), MS_8UC3: (torch.uint8, 3), MS_16SC1: (torch.int16, 1), MS_16SC3: (torch.int16, 3),
          MS_32FC1: (torch.float32, 1)}[image.type]
    out = torch.empty((image.rows, image.cols) + ((dt[1],) if dt[1] > 1 else ()), dtype=dt[0], device="cuda")
    row = image.cols * dt[1] * out.element_size()
    import ctypes
    hip = _hip()
    rc = hip.hipMemcpy2D(C.c_void_p(out.data_ptr()), C.c_size_t(row), C.c_void_p(image.data), C.c_size_t(image.step),
                         C.c_size_t(row), C.c_size_t(image.rows), 3)  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise MsError("hipMemcpy2D failed: %d" % rc)
    return out


_hiplib = None


def _hip():
    global _hiplib
    if _hiplib is None:
        _hiplib = C.CDLL("libamdhip64.so")
    return _hiplib


def _stream():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def device_count():
    return load().ms_device_count()


def calib_copy(src, dst):
    n = src.numel() * src.element_size()
    _chk(load().ms_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n), _stream()))


def calib_shape(buf, shape):
    """ms_calib_shape: every 128-byte line of `buf` (uint8 tensor, 128-byte aligned) touched once with access shape 0 / 1 / 2 (PMC calibration)"""
    _chk(load().ms_calib_shape(C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()), int(shape), _stream()))


def calib_read(src):
    n = src.numel() * src.element_size()
    _chk(load().ms_calib_read(C.c_void_p(src.data_ptr()), C.c_size_t(n), _stream()))


def selftest_divide_range(d_lo, d_hi):
    """(mismatches, quotients checked) of DivBy against IEEE division for every float denominator in [d_lo, d_hi] x all int16 numerators."""
    bad, n = C.c_ulonglong(1), C.c_ulonglong(0)
    _chk(load().ms_selftest_divide_range(C.c_float(d_lo), C.c_float(d_hi), C.byref(bad), C.byref(n), _stream()))
    return bad.value, n.value


def selftest_cvt_u8():
    n = C.c_ulonglong(1)
    _chk(load().ms_selftest_cvt_u8(C.byref(n), _stream()))
    return n.value


def selftest_divide(dens):
    import numpy as np
    d = np.ascontiguousarray(dens, np.float32)
    return _chk(load().ms_selftest_divide(d.ctypes.data_as(C.POINTER(C.c_float)), d.size, _stream()))


# ------------------------------------------------------------------ image ops (allocate dst like the cv::cuda API)

def _new(shape, dtype):
    return _torch().empty(shape, dtype=dtype, device="cuda")


def remap(src, xmap, ymap, interpolation=INTER_LINEAR, border_type=BORDER_CONSTANT):
    dst = _new(tuple(xmap.shape) + tuple(src.shape[2:]), src.dtype)
    _chk(load().ms_remap(C.byref(img(src)), C.byref(img(xmap)), C.byref(img(ymap)), C.byref(img(dst)), interpolation, border_type, _stream()))
    return dst


def resize_linear(src, dsize=None, fx=0.0, fy=0.0):
    import numpy as np
    rows, cols = src.shape[:2]
    if dsize is None:
        dsize = (int(np.rint(cols * fx)), int(np.rint(rows * fy)))   # saturate_cast<int>(double)
    else:
        fx = fy = 0.0
    dst = _new((dsize[1], dsize[0]) + tuple(src.shape[2:]), src.dtype)
    _chk(load().ms_resize_linear(C.byref(img(src)), C.byref(img(dst)), C.c_double(fx), C.c_double(fy), _stream()))
    return dst


def resize_linear_batch(srcs, dsize=None, fx=0.0, fy=0.0):
    """n 8UC3 images of one geometry through cuda::resize's arithmetic in one launch; returns the list of resized tensors."""
    import numpy as np
    rows, cols = srcs[0].shape[:2]
    if dsize is None:
        dsize = (int(np.rint(cols * fx)), int(np.rint(rows * fy)))
    else:
        fx = fy = 0.0
    dsts = [_new((dsize[1], dsize[0], 3), srcs[0].dtype) for _ in srcs]
    n = len(srcs)
    a = (Image * n)(*[img(t) for t in srcs]); b = (Image * n)(*[img(t) for t in dsts])
    _chk(load().ms_resize_linear_batch(a, b, n, C.c_double(fx), C.c_double(fy), _stream()))
    return dsts


def convert_scale_8u(src, alpha, inplace=False):
    dst = src if inplace else _new(src.shape, src.dtype)
    _chk(load().ms_convert_scale_8u(C.byref(img(src)), C.byref(img(dst)), C.c_double(alpha), _stream()))
    return dst


def convert(src, dtype, alpha=1.0):
    dst = _new(src.shape, dtype)
    _chk(load().ms_convert(C.byref(img(src)), C.byref(img(dst)), C.c_double(alpha), _stream()))
    return dst


def copy_make_border(src, top, bottom, left, right, border_type):
    dst = _new((src.shape[0] + top + bottom, src.shape[1] + left + right) + tuple(src.shape[2:]), src.dtype)
    _chk(load().ms_copy_make_border(C.byref(img(src)), C.byref(img(dst)), top, bottom, left, right, border_type, _stream()))
    return dst


def pyr_down(src):
    dst = _new(((src.shape[0] + 1) // 2, (src.shape[1] + 1) // 2) + tuple(src.shape[2:]), src.dtype)
    _chk(load().ms_pyr_down(C.byref(img(src)), C.byref(img(dst)), _stream()))
    return dst


def pyr_up(src):
    dst = _new((src.shape[0] * 2, src.shape[1] * 2) + tuple(src.shape[2:]), src.dtype)
    _chk(load().ms_pyr_up(C.byref(img(src)), C.byref(img(dst)), _stream()))
    return dst


def subtract(a, b, dst=None):
    dst = _new(a.shape, a.dtype) if dst is None else dst
    _chk(load().ms_subtract_16s(C.byref(img(a)), C.byref(img(b)), C.byref(img(dst)), _stream()))
    return dst


def add(a, b, dst=None):
    dst = _new(a.shape, a.dtype) if dst is None else dst
    _chk(load().ms_add_16s(C.byref(img(a)), C.byref(img(b)), C.byref(img(dst)), _stream()))
    return dst


def add_src_weight_32f(src, weight, dst_roi, dst_weight_roi):
    """dst_roi / dst_weight_roi: views (tensor slices) of the pano level, i.e. dst(rc)."""
    _chk(load().ms_add_src_weight_32f(C.byref(img(src)), C.byref(img(weight)), C.byref(img(dst_roi)), C.byref(img(dst_weight_roi)),
                                      dst_roi.shape[1], dst_roi.shape[0], _stream()))


def add_src_weight_16s(src, weight, dst_roi, dst_weight_roi):
    """addSrcWeightGpu16S: 16SC3 src, 16SC1 fixed-point weights (0..256) accumulated into dst(rc) / dst_weight(rc)."""
    _chk(load().ms_add_src_weight_16s(C.byref(img(src)), C.byref(img(weight)), C.byref(img(dst_roi)), C.byref(img(dst_weight_roi)),
                                      src.shape[1], src.shape[0], _stream()))


def normalize_using_weight_16s(weight, src):
    _chk(load().ms_normalize_using_weight_16s(C.byref(img(weight)), C.byref(img(src)), src.shape[1], src.shape[0], _stream()))


def normalize_using_weight_32f(weight, src):
    _chk(load().ms_normalize_using_weight_32f(C.byref(img(weight)), C.byref(img(src)), src.shape[1], src.shape[0], _stream()))


def compare_gt(src, thr):
    dst = _new(src.shape, _torch().uint8)
    _chk(load().ms_compare_gt_32f(C.byref(img(src)), C.c_float(thr), C.byref(img(dst)), _stream()))
    return dst


def compare_eq(src, val):
    dst = _new(src.shape, _torch().uint8)
    _chk(load().ms_compare_eq_8u(C.byref(img(src)), int(val), C.byref(img(dst)), _stream()))
    return dst


def set_zero_masked(image, mask):
    _chk(load().ms_set_zero_masked_16sc3(C.byref(img(image)), C.byref(img(mask)), _stream()))


def bitwise_and(a, b):
    dst = _new(a.shape, a.dtype)
    _chk(load().ms_bitwise_and_8u(C.byref(img(a)), C.byref(img(b)), C.byref(img(dst)), _stream()))
    return dst


def dilate3x3(src):
    dst = _new(src.shape, src.dtype)
    _chk(load().ms_dilate3x3_8u(C.byref(img(src)), C.byref(img(dst)), _stream()))
    return dst


def _fa(v, n):
    import numpy as np
    a = np.ascontiguousarray(v, np.float32).reshape(n)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def build_warp_maps(projection, tl_u, tl_v, rows, cols, k_rinv, scale, t=None):
    torch = _torch()
    mx = _new((rows, cols), torch.float32); my = _new((rows, cols), torch.float32)
    ka, kp = _fa(k_rinv, 9)
    tp = None
    if t is not None:
        ta, tp = _fa(t, 3)
    _chk(load().ms_build_warp_maps(projection, tl_u, tl_v, C.byref(img(mx)), C.byref(img(my)), kp, kp, tp, C.c_float(scale), _stream()))
    return mx, my


def nv12_to_bgr(src, dst=None):
    if dst is None:
        dst = _new((src.shape[0] * 2 // 3, src.shape[1], 3), _torch().uint8)
    _chk(load().ms_nv12_to_bgr(C.byref(img(src)), C.byref(img(dst)), _stream()))
    return dst


def nv12_to_bgr_batch(srcs, dsts=None):
    """cvtColor(YUV2BGR_NV12) of n cameras of one geometry in one launch; returns the list of BGR tensors."""
    if dsts is None:
        dsts = [_new((t.shape[0] * 2 // 3, t.shape[1], 3), _torch().uint8) for t in srcs]
    n = len(srcs)
    a = (Image * n)(*[img(t) for t in srcs]); b = (Image * n)(*[img(t) for t in dsts])
    _chk(load().ms_nv12_to_bgr_batch(a, b, n, _stream()))
    return dsts


def nv12_to_bgr_batch_prepared(srcs, dsts):
    """ms_nv12_to_bgr_batch with the descriptors marshalled once; returns a callable(stream_handle=None) (timed loops: building 192 ms_image per call costs
    more host time than the three launches take on the GPU)."""
    n = len(srcs)
    a = (Image * n)(*[img(t) for t in srcs]); b = (Image * n)(*[img(t) for t in dsts])
    fn = load().ms_nv12_to_bgr_batch

    def run(stream=None):
        _chk(fn(a, b, n, stream if stream is not None else _stream()))
    run.keep = (srcs, dsts)
    return run


def bgr_to_i420(src, dst=None):
    if dst is None:
        dst = _new((src.shape[0] * 3 // 2, src.shape[1]), _torch().uint8)
    _chk(load().ms_bgr_to_i420(C.byref(img(src)), C.byref(img(dst)), _stream()))
    return dst


def consume_i420(pano8u, out_size=(4096, 2048), keep_aspect_ratio=True):
    """consume()'s resize + black bars + BGR2YUV_I420 in one pass (timed.cpp:251-316).  Returns (I420 tensor (out_h * 3 / 2, out_w), image_height)."""
    ow, oh = out_size
    dst = _new((oh * 3 // 2, ow), _torch().uint8)
    ih = C.c_int(0)
    _chk(load().ms_consume_i420(C.byref(img(pano8u)), C.byref(img(dst)), ow, oh, 1 if keep_aspect_ratio else 0, C.byref(ih), _stream()))
    return dst, ih.value


def bgr_to_gray(src):
    dst = _new(tuple(src.shape[:2]), _torch().uint8)
    _chk(load().ms_bgr_to_gray(C.byref(img(src)), C.byref(img(dst)), _stream()))
    return dst


def resize_linear_batch_prepared(srcs, dsts, fx=0.0, fy=0.0):
    """cuda::resize of every srcs[i] into dsts[i] (8UC3, one geometry) in one launch; returns a callable(stream_handle=None) with the descriptors
    marshalled once (the per-frame compose-scale resize of stitch_online, timed.cpp:75-85, inside a timed loop)."""
    n = len(srcs)
    a = (Image * n)(*[img(t) for t in srcs]); b = (Image * n)(*[img(t) for t in dsts])
    fn = load().ms_resize_linear_batch
    cfx, cfy = C.c_double(fx), C.c_double(fy)

    def run(stream=None):
        _chk(fn(a, b, n, cfx, cfy, stream if stream is not None else _stream()))
    run.keep = (srcs, dsts)
    return run


def bgr_to_i420_batch_prepared(srcs, dsts):
    """One launch converting every srcs[i] (8UC3, same geometry) into dsts[i] (contiguous I420); returns a callable(stream_handle=None)
    with the descriptors marshalled once."""
    n = len(srcs)
    a = (Image * n)(*[img(t) for t in srcs]); b = (Image * n)(*[img(t) for t in dsts])
    fn = load().ms_bgr_to_i420_batch

    def run(stream=None):
        _chk(fn(a, b, n, stream if stream is not None else _stream()))
    run.keep = (srcs, dsts)
    return run


def custom_resize(src, tx, ty):
    dst = _new((ty, tx), src.dtype)
    _chk(load().ms_custom_resize_32f(C.byref(img(src)), C.byref(img(dst)), _stream()))
    return dst


def warp_roi(projection, K, R, scale, src_w, src_h):
    ka, kp = _fa(K, 9); ra, rp = _fa(R, 9)
    r = Rect()
    _chk(load().ms_warp_roi(projection, kp, rp, C.c_float(scale), src_w, src_h, C.byref(r)))
    return r.tuple()


def result_roi(rois):
    arr = (Rect * len(rois))(*[Rect(*r) for r in rois])
    r = Rect()
    _chk(load().ms_result_roi(len(rois), arr, C.byref(r)))
    return r.tuple()


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int), ("edge_threshold", C.c_int), ("first_level", C.c_int),
                ("patch_size", C.c_int), ("fast_threshold", C.c_int)]


def orb_detect_and_compute(gray, mask=None, nfeatures=2500, scale_factor=1.2, nlevels=8, fast_threshold=20, edge_threshold=None):
    """cuda::ORB::create(nfeatures, scaleFactor, nlevels)->detectAndCompute (featurefinder.cpp:15-40): gray / mask torch uint8 (H, W) on the device.
    Returns (keypoints (n, 6) float32 numpy: x, y, response, angle, octave, size; descriptors (n, 32) uint8 torch tensor on the device)."""
    import numpy as np
    import torch
    prm = OrbParams()
    _chk(load().ms_orb_default_params(C.byref(prm)))
    prm.nfeatures, prm.scale_factor, prm.nlevels, prm.fast_threshold = nfeatures, scale_factor, nlevels, fast_threshold
    if edge_threshold is not None:
        prm.edge_threshold = edge_threshold
    cap = nfeatures
    kp = np.zeros((cap, 6), np.float32)
    desc = torch.zeros((cap, 32), dtype=torch.uint8, device=gray.device)
    n = C.c_int(0)
    di = img(desc)
    _chk(load().ms_orb_detect_and_compute(C.byref(img(gray)), None if mask is None else C.byref(img(mask)), C.byref(prm),
                                          kp.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(di), C.byref(n), _stream()))
    return kp[:n.value].copy(), desc[:n.value]


def find_homography_ransac(src, dst, reproj_threshold=3.0, max_iters=2000, confidence=0.995):
    """cv::findHomography(src, dst, mask, RANSAC): src / dst (n, 2) float32.  Returns (H 3x3 float64 or None, inlier mask uint8 (n,))."""
    import numpy as np
    src = np.ascontiguousarray(src, np.float32); dst = np.ascontiguousarray(dst, np.float32)
    n = len(src)
    H = np.zeros(9, np.float64)
    m = np.zeros(max(n, 1), np.uint8)
    cnt = C.c_int(0)
    rc = load().ms_find_homography_ransac(src.ctypes.data_as(C.POINTER(C.c_float)), dst.ctypes.data_as(C.POINTER(C.c_float)), n, C.c_double(reproj_threshold),
                                          max_iters, C.c_double(confidence), H.ctypes.data_as(C.POINTER(C.c_double)), m.ctypes.data_as(C.POINTER(C.c_uint8)),
                                          C.byref(cnt), _stream())
    if rc < 0:
        _chk(rc)
    return (None if rc == 1 else H.reshape(3, 3)), m[:n]


# ------------------------------------------------------------------ compositor context

def knn_match_hamming2(query, train):
    """BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2): (train_idx, distance) int32 arrays of shape (nq, 2)."""
    import numpy as np
    nq = query.shape[0]
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.int32)
    ip = C.POINTER(C.c_int)
    _chk(load().ms_knn_match_hamming2(C.byref(img(query)), C.byref(img(train)), idx.ctypes.data_as(ip), dist.ctypes.data_as(ip), _stream()))
    return idx, dist


def mesh_default_params(**kw):
    """defs.h values (10 x 10 mesh, ALPHAS, GLOBAL_DIST, wrapAround); keyword arguments override fields."""
    p = MeshParams()
    _chk(load().ms_mesh_default_params(C.byref(p)))
    for k, v in kw.items():
        if k == "alphas":
            p.alphas = (C.c_float * 4)(*v)
        else:
            assert hasattr(p, k), k
            setattr(p, k, v)
    return p


def mesh_saliency(view, mesh_cols, mesh_rows):
    import numpy as np
    out = np.empty((mesh_rows, mesh_cols, 8), np.float32)
    _chk(load().ms_mesh_saliency(C.byref(img(view)), mesh_cols, mesh_rows, out.ctypes.data_as(C.POINTER(C.c_float)), _stream()))
    return out


def mesh_triangle_masks(cell_w, cell_h):
    """The 8 fillConvexPoly triangle masks of a cell_w x cell_h mesh cell, (8, cell_h, cell_w) uint8, and their set-pixel counts."""
    import numpy as np
    out = np.empty((8, cell_h, cell_w), np.uint8)
    cnt = (C.c_uint * 8)()
    _chk(load().ms_mesh_triangle_masks(int(cell_w), int(cell_h), out.ctypes.data_as(C.POINTER(C.c_ubyte)), cnt, _stream()))
    return out, list(cnt)


def _match_array(lists):
    flat = [m for l in lists for m in l]
    arr = (MeshMatch * max(1, len(flat)))()
    for k, m in enumerate(flat):
        arr[k] = MeshMatch(float(m[0]), float(m[1]), float(m[2]), float(m[3]), int(m[4]) if len(m) > 4 else 0)
    return arr, (C.c_int * len(lists))(*[len(l) for l in lists])


def create_mesh(views, matches, params, temporal=None):
    """MeshWarper::createMesh after feature matching (ms_create_mesh).  views: warped 8UC3 device tensors; matches[v]: list of
    (x1, y1, x2, y2, dst); temporal[v]: list of (x1, y1, x2, y2).  Returns mesh_x, mesh_y (n, N, M) float32 host arrays and the solver info."""
    import numpy as np
    n = len(views)
    ims = (Image * n)(*[img(v) for v in views])
    marr, mcnt = _match_array(matches)
    tarr = tcnt = None
    if temporal is not None:
        tarr, tcnt = _match_array(temporal)
    mx = np.empty((n, params.mesh_rows, params.mesh_cols), np.float32)
    my = np.empty_like(mx)
    info = MeshInfo()
    fp = C.POINTER(C.c_float)
    _chk(load().ms_create_mesh(n, ims, marr, mcnt, tarr, tcnt, C.byref(params), mx.ctypes.data_as(fp), my.ctypes.data_as(fp),
                               C.byref(info), _stream()))
    return mx, my, {k: getattr(info, k) for k, _ in MeshInfo._fields_}


def calibrate_cameras(num_views, w, h, hfov_deg=90.0, work_megapix=0.6, seam_megapix=0.01, compose_megapix=1.4):
    """ms_calibrate_cameras: calibrateCameras + stitch_calib's scale bookkeeping (APP/calibration.cpp:28-68, 101-116, 147-181, 269-288).
    Returns a dict: the scales, and per view K_compose / K_seam / R as fp32 3x3 arrays."""
    import numpy as np
    q = RigParams(num_views, w, h, hfov_deg, work_megapix, seam_megapix, compose_megapix)
    r = Rig()
    _chk(load().ms_calibrate_cameras(C.byref(q), C.byref(r)))
    out = {k: getattr(r, k) for k in ("work_scale", "seam_scale", "seam_work_aspect", "compose_scale", "compose_work_aspect", "warped_image_scale",
                                      "seam_warp_scale", "compose_warp_scale", "resize_input", "compose_width", "compose_height")}
    for k in ("K_compose", "K_seam", "R"):
        out[k] = [np.array(list(getattr(r, k)[i]), np.float32).reshape(3, 3) for i in range(num_views)]
    return out


def num_bands_rule(pano_w, pano_h, blend_strength=5.0):
    bw, nb = C.c_float(), C.c_int()
    _chk(load().ms_num_bands_rule(pano_w, pano_h, C.c_float(blend_strength), C.byref(bw), C.byref(nb)))
    return bw.value, nb.value


class Compositor:
    """stitch_calib tables once (build_maps / build_masks / init_blender), then stitch() per frame batch."""

    def __init__(self, num_views, src_size, projection, warp_scale, num_bands=5, enable_cpw=False,
                 out_size=(0, 0), max_frames=1, simple_kernels=False, lds_stage=None, shards=1, shard_index=0, cv_remap=False,
                 col_shards=1, col_shard_index=0, update_mask_margin=0):
        cfg = Config(C.sizeof(Config), num_views, src_size[0], src_size[1], projection, warp_scale, num_bands, int(enable_cpw),
                     out_size[0], out_size[1], max_frames)
        cfg.debug_simple_kernels = 1 if simple_kernels else 0   # debug: force the one-pixel-per-lane reference kernels
        cfg.warp_lds_stage = 0 if lds_stage is None else (1 if lds_stage else 2)   # True: warp source tiles staged in LDS by LDS-DMA (opt-in, measured slower); False forces the direct gathers even under MS_WARP_ASYNC=1
        cfg.view_shards = shards; cfg.view_shard_index = shard_index   # view sharding
        cfg.col_shards = col_shards; cfg.col_shard_index = col_shard_index   # pano-column sharding
        cfg.self_check = 1 if os.environ.get("MS_CHECK_DIVIDE", "0") not in ("", "0") else 0   # (this binding is test / bench infrastructure: the LIBRARY reads no environment; tests/conftest.py sets the variable)
        cfg.update_mask_margin = update_mask_margin   # > 0: ms_update_mask is enqueue-only (double-buffered tables, work lists planned with this margin)
        cfg.cpu_flavour_remap = 1 if cv_remap else 0   # cv::remap's CPU arithmetic for the projection warp (needs simple_kernels, no CPW)
        self._ctx = C.c_void_p()
        _chk(load().ms_create(C.byref(cfg), C.byref(self._ctx)))
        self.cfg = cfg
        self.n = num_views

    def save_tables(self):
        """ms_save_tables: the calibration of this (ready) context as bytes"""
        n = C.c_size_t(0)
        _chk(load().ms_save_tables(self._ctx, None, C.c_size_t(0), C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        _chk(load().ms_save_tables(self._ctx, buf, C.c_size_t(n.value), C.byref(n)))
        return bytes(buf)

    @classmethod
    def from_tables(cls, blob):
        """ms_load_tables: a ready context rebuilt from save_tables() bytes (no ms_init_blender call needed)"""
        self = cls.__new__(cls)
        self._ctx = C.c_void_p()
        b = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _chk(load().ms_load_tables(b, C.c_size_t(len(blob)), C.byref(self._ctx), _stream()))
        self.cfg = Config.from_buffer_copy(blob[48:48 + C.sizeof(Config)])      # TablesHeader: 48 bytes in front of the ms_config
        import struct
        self.n = struct.unpack_from("<i", blob, 16)[0]      # TablesHeader::n_views
        return self

    def close(self):
        if self._ctx:
            load().ms_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, view, K, R):
        ka, kp = _fa(K, 9); ra, rp = _fa(R, 9)
        _chk(load().ms_set_camera(self._ctx, view, kp, rp))

    def set_gain(self, view, gain):
        _chk(load().ms_set_gain(self._ctx, view, C.c_double(gain)))

    def build_maps(self):
        _chk(load().ms_build_maps(self._ctx, _stream()))

    def build_masks(self, mode=1):
        _chk(load().ms_build_masks(self._ctx, mode, _stream()))

    def calibrate_seam(self, full_imgs, K_seam, seam_scale, seam_warp_scale, dilate=False, estimate_gains=True):
        """stitch_calib's seam-scale pipeline; returns the estimated gains."""
        import numpy as np
        arr = (Image * self.n)(*[img(t) for t in full_imgs])
        k = np.ascontiguousarray(K_seam, np.float32).reshape(self.n * 9)
        prm = SeamParams(seam_scale, seam_warp_scale, int(dilate), int(estimate_gains))
        g = (C.c_double * self.n)()
        _chk(load().ms_calibrate_seam(self._ctx, arr, k.ctypes.data_as(C.POINTER(C.c_float)), C.byref(prm), g, _stream()))
        return list(g)

    def set_mask(self, view, mask_np):
        import numpy as np
        m = np.ascontiguousarray(mask_np, np.uint8)
        _chk(load().ms_set_mask(self._ctx, view, m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(m.strides[0])))

    def init_blender(self):
        _chk(load().ms_init_blender(self._ctx, _stream()))

    def set_mesh_interp(self, view, start, end, progress):
        """start / end: (mesh_x, mesh_y) pairs of N x M float32 host arrays."""
        import numpy as np
        a = [np.ascontiguousarray(m, np.float32) for m in (start[0], start[1], end[0], end[1])]
        n, m = a[0].shape
        fp = C.POINTER(C.c_float)
        _chk(load().ms_set_mesh_interp(self._ctx, view, a[0].ctypes.data_as(fp), a[1].ctypes.data_as(fp), a[2].ctypes.data_as(fp),
                                       a[3].ctypes.data_as(fp), n, m, C.c_float(progress), _stream()))

    def feed(self, view, image):
        _chk(load().ms_feed(self._ctx, view, C.byref(img(image)), _stream()))

    def blend(self, out8u=None, out16s=None):
        o8 = C.byref(img(out8u)) if out8u is not None else None
        o16 = C.byref(img(out16s)) if out16s is not None else None
        _chk(load().ms_blend(self._ctx, o8, o16, _stream()))

    def mesh_displacement(self, view):
        d = C.c_float(0)
        _chk(load().ms_get_mesh_displacement(self._ctx, view, C.byref(d)))
        return d.value

    def update_mask(self, view):
        _chk(load().ms_update_mask(self._ctx, view, _stream()))

    def init_feather(self, sharpness=0.02):
        _chk(load().ms_init_feather(self._ctx, C.c_float(sharpness), _stream()))

    def set_mesh(self, view, mesh_x, mesh_y):
        ax, px = _fa(mesh_x, mesh_x.size); ay, py = _fa(mesh_y, mesh_y.size)
        _chk(load().ms_set_mesh(self._ctx, view, px, py, mesh_x.shape[0], mesh_x.shape[1], _stream()))

    def set_meshes(self, meshes):
        """convertMeshesToMap for every view in one call: meshes = [(mesh_x, mesh_y)] * num_views, all N x M float32 host arrays."""
        import numpy as np
        mx = np.ascontiguousarray(np.stack([m[0] for m in meshes]), np.float32)
        my = np.ascontiguousarray(np.stack([m[1] for m in meshes]), np.float32)
        assert mx.shape[0] == self.n and mx.shape == my.shape
        fp = C.POINTER(C.c_float)
        _chk(load().ms_set_meshes(self._ctx, mx.ctypes.data_as(fp), my.ctypes.data_as(fp), mx.shape[1], mx.shape[2], _stream()))

    def set_mesh_maps(self, view, xm, ym):
        _chk(load().ms_set_mesh_maps(self._ctx, view, C.byref(img(xm)), C.byref(img(ym)), _stream()))

    def view_geom(self, view):
        g = ViewGeom()
        _chk(load().ms_get_view_geom(self._ctx, view, C.byref(g)))
        return g

    def pano_geom(self):
        g = PanoGeom()
        _chk(load().ms_get_pano_geom(self._ctx, C.byref(g)))
        return g

    def maps(self, view):
        xm, ym = Image(), Image()
        _chk(load().ms_get_maps(self._ctx, view, C.byref(xm), C.byref(ym)))
        return tensor_of(xm), tensor_of(ym)

    def mask(self, view):
        m = Image()
        _chk(load().ms_get_mask(self._ctx, view, C.byref(m)))
        return tensor_of(m)

    def weight_level(self, view, level):
        m = Image()
        _chk(load().ms_get_weight_level(self._ctx, view, level, C.byref(m)))
        return tensor_of(m)

    def mesh_maps(self, view):
        xm, ym = Image(), Image()
        _chk(load().ms_get_mesh_maps(self._ctx, view, C.byref(xm), C.byref(ym)))
        return tensor_of(xm), tensor_of(ym)

    def result_mask(self):
        m = Image()
        _chk(load().ms_get_result_mask(self._ctx, C.byref(m)))
        return tensor_of(m)

    def band_cells(self, level):
        """(owned, exclusive, general) 64 x 16 cells of a band: see ms_get_band_cells."""
        a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
        _chk(load().ms_get_band_cells(self._ctx, level, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def stitch_kernels(self):
        """(warp kernel, stage-1 kernel) of the last stitch call: ms_get_stitch_kernels, as names."""
        names = {0: "none", 1: "simple", 2: "shared_aligned", 3: "shared_unaligned", 4: "per_frame_aligned", 5: "per_frame_unaligned", 6: "nv12", 7: "lds_staged"}
        a, b = C.c_int(), C.c_int()
        _chk(load().ms_get_stitch_kernels(self._ctx, C.byref(a), C.byref(b)))
        return names[a.value], names[b.value]

    def plan_stats(self):
        """Work-list sizes of the context (ms_get_plan_stats) as a dict."""
        st = PlanStats()
        st.struct_size = C.sizeof(PlanStats)
        _chk(load().ms_get_plan_stats(self._ctx, C.byref(st)))
        return {"warp_tile": (st.warp_tile_w, st.warp_tile_h), "n_warp_tiles": st.n_warp_tiles, "n_stage1_tiles": st.n_stage1_tiles,
                "n_stage1_reachable": st.n_stage1_reachable, "down_tile": (st.down_tile_w, st.down_tile_h), "n_down_tiles": list(st.n_down_tiles),
                "blend_tile": (st.blend_tile_w, st.blend_tile_h), "n_blend_tiles": list(st.n_blend_tiles)}

    def _tables(self, frames, out8u, out16s):
        n_frames = len(frames)
        views = (Image * (n_frames * self.n))()
        k = 0
        for fr in frames:
            assert len(fr) == self.n
            for t in fr:
                views[k] = img(t) if t is not None else Image(); k += 1      # None: a view this (column / view) shard never reads
        o8 = o16 = None
        if out8u is not None:
            o8 = (Image * n_frames)(*[img(t) for t in out8u])
        if out16s is not None:
            o16 = (Image * n_frames)(*[img(t) for t in out16s])
        return n_frames, views, o8, o16

    def stitch(self, frames, out8u=None, out16s=None):
        """frames: list (per frame) of lists (per view) of uint8 HxWx3 cuda tensors."""
        n, views, o8, o16 = self._tables(frames, out8u, out16s)
        _chk(load().ms_stitch(self._ctx, n, views, o8, o16, _stream()))

    def stitch_nv12(self, frames_nv12, out8u=None, out16s=None):
        """ms_stitch_nv12: frames_nv12 = list (per frame) of lists (per view) of uint8 (H * 3 / 2) x W cuda tensors (the cameras' NV12 planes)"""
        n, views, o8, o16 = self._tables(frames_nv12, out8u, out16s)
        _chk(load().ms_stitch_nv12(self._ctx, n, views, o8, o16, _stream()))

    def prepared_nv12(self, frames_nv12, out8u=None, out16s=None):
        n, views, o8, o16 = self._tables(frames_nv12, out8u, out16s)
        fn, ctx = load().ms_stitch_nv12, self._ctx

        def run(stream=None):
            _chk(fn(ctx, n, views, o8, o16, stream if stream is not None else _stream()))
        run.keepalive = (frames_nv12, out8u, out16s, views, o8, o16)
        return run

    def prepared(self, frames, out8u=None, out16s=None):
        """Pre-marshal the descriptor tables once; returns a zero-overhead callable for timed loops."""
        n, views, o8, o16 = self._tables(frames, out8u, out16s)
        fn, ctx = load().ms_stitch, self._ctx

        def run(stream=None):
            _chk(fn(ctx, n, views, o8, o16, stream if stream is not None else _stream()))
        run.keepalive = (frames, out8u, out16s, views, o8, o16)
        return run

    def col_window(self):
        """(begin, end) pano-ROI columns this context composites (column sharding); the whole ROI otherwise."""
        a, b = C.c_int(0), C.c_int(0)
        _chk(load().ms_get_col_window(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def needed_views(self):
        """Bit mask of the views ms_stitch reads (column / view shards read a subset)."""
        m = C.c_uint(0)
        _chk(load().ms_get_needed_views(self._ctx, C.byref(m)))
        return m.value

    def i420_rows(self):
        """(first canvas row, rows) of the even-aligned span the I420 output holds."""
        a, b = C.c_int(0), C.c_int(0)
        _chk(load().ms_get_i420_rows(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def new_i420(self, n_frames=1):
        """Black I420 buffers (Y = 16, U = V = 128) of the size ms_stitch_i420 writes."""
        torch = _torch()
        _, rows = self.i420_rows()
        w = self.cfg.out_width
        outs = []
        for _ in range(n_frames):
            t = torch.full((rows * 3 // 2, w), 128, dtype=torch.uint8, device="cuda")
            t[:rows] = 16
            outs.append(t)
        return outs

    def prepared_i420(self, frames, out_i420):
        n, views, _, _ = self._tables(frames, None, None)
        oi = (Image * n)(*[img(t) for t in out_i420])
        fn, ctx = load().ms_stitch_i420, self._ctx

        def run(stream=None):
            _chk(fn(ctx, n, views, oi, stream if stream is not None else _stream()))
        run.keepalive = (frames, out_i420, views, oi)
        return run

    def stitch_i420(self, frames, out_i420):
        """The panorama as planar I420 (encoder input), written by the level-0 band kernel: no 8UC3 canvas, no conversion pass."""
        self.prepared_i420(frames, out_i420)()

    def partial_bytes(self):
        load().ms_partial_bytes.restype = C.c_size_t
        return load().ms_partial_bytes(self._ctx)

    def stitch_partial(self, frames, partial):
        """frames: per frame a list of N tensors (None for views this shard does not own); partial: int16 cuda tensor."""
        n_frames = len(frames)
        views = (Image * (n_frames * self.n))()
        k = 0
        for fr in frames:
            for t in fr:
                if t is not None:
                    views[k] = img(t)
                k += 1
        _chk(load().ms_stitch_partial(self._ctx, n_frames, views, C.c_void_p(partial.data_ptr()), _stream()))

    def stitch_finish(self, n_frames, partials, out8u=None, out16s=None):
        ptrs = (C.c_void_p * len(partials))(*[p.data_ptr() for p in partials])
        o8 = (Image * n_frames)(*[img(t) for t in out8u]) if out8u is not None else None
        o16 = (Image * n_frames)(*[img(t) for t in out16s]) if out16s is not None else None
        _chk(load().ms_stitch_finish(self._ctx, n_frames, ptrs, len(partials), o8, o16, _stream()))

    def stitch_timed(self, frames, out8u=None, out16s=None):
        n, views, o8, o16 = self._tables(frames, out8u, out16s)
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        k = _chk(load().ms_stitch_timed(self._ctx, n, views, o8, o16, _stream(), cap, names, ms))
        return [(names[i].decode(), ms[i]) for i in range(k)]
