"""ctypes/numpy front-end of the CPU oracle (libms_oracle.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product (video-stitcher_amd) never imports this module.
See oracle/ms_oracle.h for what the oracle restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libms_oracle.so")

PROJ_PLANE, PROJ_CYLINDRICAL, PROJ_SPHERICAL = 0, 1, 2


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in
            ("ms_oracle_prims.c", "ms_oracle_geom.c", "ms_oracle_blend.c", "ms_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs if os.path.exists(s))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "libms_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]

    def tuple(self):
        return (self.x, self.y, self.width, self.height)


class BlendGeom(C.Structure):
    _fields_ = [("num_bands", C.c_int), ("dst_roi_final", Rect), ("dst_roi", Rect)]


class ViewGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("top", "left", "bottom", "right", "x_tl", "y_tl", "x_br", "y_br")]

    def dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Projector(C.Structure):
    _fields_ = [("k", C.c_float * 9), ("rinv", C.c_float * 9), ("r_kinv", C.c_float * 9),
                ("k_rinv", C.c_float * 9), ("t", C.c_float * 3), ("scale", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_result_roi.restype = Rect
        _lib.orc_blender_create.restype = C.c_void_p
        _lib.orc_blender_weight_level.restype = C.c_void_p
        _lib.orc_blender_src_level.restype = C.c_void_p
        _lib.orc_trunc_s16_range_violations.restype = C.c_longlong
    return _lib


def trunc_s16_range_violations():
    """How often static_cast<short>(float) saw a value outside the int16 range (where nvcc's two possible conversions would differ) since the
    library was loaded: must stay 0 on every tested configuration (ms_oracle_prims.c: trunc_s16f)."""
    return int(lib().orc_trunc_s16_range_violations())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _st(a):
    return C.c_size_t(a.strides[0])


def _ia(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ------------------------------------------------------------------ primitives

def remap_linear_8uc3(src, mapx, mapy):
    assert src.dtype == np.uint8 and src.ndim == 3 and src.shape[2] == 3
    mapx = np.ascontiguousarray(mapx, np.float32); mapy = np.ascontiguousarray(mapy, np.float32)
    dst = np.empty(mapx.shape + (3,), np.uint8)
    lib().orc_remap_linear_8uc3(_p(src), _st(src), src.shape[0], src.shape[1], _p(mapx), _st(mapx),
                                _p(mapy), _st(mapy), _p(dst), _st(dst), dst.shape[0], dst.shape[1])
    return dst


def remap_linear_8uc1(src, mapx, mapy):
    mapx = np.ascontiguousarray(mapx, np.float32); mapy = np.ascontiguousarray(mapy, np.float32)
    dst = np.empty(mapx.shape, np.uint8)
    lib().orc_remap_linear_8uc1(_p(src), _st(src), src.shape[0], src.shape[1], _p(mapx), _st(mapx),
                                _p(mapy), _st(mapy), _p(dst), _st(dst), dst.shape[0], dst.shape[1])
    return dst


def cv_remap_linear(src, mapx, mapy):
    """cv::remap(src, dst, mapx, mapy, INTER_LINEAR) on the CPU (fixed-point coordinates and weights), 8UC1 or 8UC3, BORDER_CONSTANT 0."""
    src = np.ascontiguousarray(src, np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    mapx = np.ascontiguousarray(mapx, np.float32); mapy = np.ascontiguousarray(mapy, np.float32)
    dst = np.empty(mapx.shape + ((cn,) if src.ndim == 3 else ()), np.uint8)
    lib().orc_cv_remap_linear_8u(_p(src), _st(src), src.shape[0], src.shape[1], cn, _p(mapx), _st(mapx), _p(mapy), _st(mapy),
                                 _p(dst), _st(dst), dst.shape[0], dst.shape[1])
    return dst


def remap_linear_reflect_8uc3(src, mapx, mapy):
    mapx = np.ascontiguousarray(mapx, np.float32); mapy = np.ascontiguousarray(mapy, np.float32)
    dst = np.empty(mapx.shape + (3,), np.uint8)
    lib().orc_remap_linear_reflect_8uc3(_p(src), _st(src), src.shape[0], src.shape[1], _p(mapx), _st(mapx),
                                        _p(mapy), _st(mapy), _p(dst), _st(dst), dst.shape[0], dst.shape[1])
    return dst


def remap_nearest_8uc1(src, mapx, mapy):
    mapx = np.ascontiguousarray(mapx, np.float32); mapy = np.ascontiguousarray(mapy, np.float32)
    dst = np.empty(mapx.shape, np.uint8)
    lib().orc_remap_nearest_8uc1(_p(src), _st(src), src.shape[0], src.shape[1], _p(mapx), _st(mapx),
                                 _p(mapy), _st(mapy), _p(dst), _st(dst), dst.shape[0], dst.shape[1])
    return dst


def resize_linear_8u(src, dsize=None, fx=0.0, fy=0.0):
    """cuda::resize host logic (OCV/cudawarping/src/resize.cpp:57-106)."""
    rows, cols = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    if dsize is None:
        # saturate_cast<int>(double) = cvRound (nearest-even)
        dsize = (int(np.rint(cols * fx)), int(np.rint(rows * fy)))
    else:
        fx = dsize[0] / cols
        fy = dsize[1] / rows
    dst = np.empty((dsize[1], dsize[0]) + (() if src.ndim == 2 else (cn,)), np.uint8)
    if (dsize[1], dsize[0]) == (rows, cols):
        dst[...] = src
        return dst
    lib().orc_resize_linear_8u(_p(src), _st(src), rows, cols, cn, _p(dst), _st(dst), dsize[1], dsize[0],
                               C.c_float(np.float32(1.0 / fx)), C.c_float(np.float32(1.0 / fy)))
    return dst


def convert_scale_8u(src, alpha):
    dst = np.empty_like(src)
    wb = src.shape[1] * (1 if src.ndim == 2 else src.shape[2])
    lib().orc_convert_scale_8u(_p(src), _st(src), _p(dst), _st(dst), src.shape[0], wb, C.c_double(alpha))
    return dst


def copy_make_border_reflect(src, top, bottom, left, right):
    es = src.itemsize * (1 if src.ndim == 2 else src.shape[2])
    dst = np.empty((src.shape[0] + top + bottom, src.shape[1] + left + right) + src.shape[2:], src.dtype)
    lib().orc_copy_make_border_reflect(_p(src), _st(src), src.shape[0], src.shape[1], es, _p(dst), _st(dst),
                                       top, bottom, left, right)
    return dst


def copy_make_border_const_32f(src, top, bottom, left, right):
    dst = np.empty((src.shape[0] + top + bottom, src.shape[1] + left + right), np.float32)
    lib().orc_copy_make_border_const_32f(_p(src), _st(src), src.shape[0], src.shape[1], _p(dst), _st(dst),
                                         top, bottom, left, right)
    return dst


def convert_8u_32f_scale(src, alpha):
    dst = np.empty(src.shape, np.float32)
    lib().orc_convert_8u_32f_scale(_p(src), _st(src), _p(dst), _st(dst), src.shape[0], src.shape[1], C.c_double(alpha))
    return dst


def convert_16s_8u(src):
    dst = np.empty(src.shape, np.uint8)
    w = src.shape[1] * (1 if src.ndim == 2 else src.shape[2])
    lib().orc_convert_16s_8u(_p(src), _st(src), _p(dst), _st(dst), src.shape[0], w)
    return dst


def pyr_down_16s(src):
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty(((src.shape[0] + 1) // 2, (src.shape[1] + 1) // 2) + src.shape[2:], np.int16)
    lib().orc_pyr_down_16s(_p(src), _st(src), src.shape[0], src.shape[1], cn, _p(dst), _st(dst))
    return dst


def pyr_down_32f(src):
    dst = np.empty(((src.shape[0] + 1) // 2, (src.shape[1] + 1) // 2), np.float32)
    lib().orc_pyr_down_32f(_p(src), _st(src), src.shape[0], src.shape[1], _p(dst), _st(dst))
    return dst


def pyr_up_16s(src):
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((src.shape[0] * 2, src.shape[1] * 2) + src.shape[2:], np.int16)
    lib().orc_pyr_up_16s(_p(src), _st(src), src.shape[0], src.shape[1], cn, _p(dst), _st(dst))
    return dst


def _binop(fn, a, b):
    dst = np.empty_like(a)
    w = a.shape[1] * (1 if a.ndim == 2 else a.shape[2])
    fn(_p(a), _st(a), _p(b), _st(b), _p(dst), _st(dst), a.shape[0], w)
    return dst


def sub_16s(a, b):
    return _binop(lib().orc_sub_16s, a, b)


def add_16s(a, b):
    return _binop(lib().orc_add_16s, a, b)


def add_src_weight_32f(src, w, dst, dst_w):
    """In-place on dst (16SC3 view) and dst_w (32F view), both already cropped to the rect."""
    lib().orc_add_src_weight_32f(_p(src), _st(src), _p(w), _st(w), _p(dst), _st(dst), _p(dst_w), _st(dst_w),
                                 dst.shape[0], dst.shape[1])


def add_src_weight_16s(src, w, dst, dst_w):
    lib().orc_add_src_weight_16s(_p(src), _st(src), _p(w), _st(w), _p(dst), _st(dst), _p(dst_w), _st(dst_w), src.shape[0], src.shape[1])


def normalize_16s(w, src):
    lib().orc_normalize_16s(_p(w), _st(w), _p(src), _st(src), src.shape[0], src.shape[1])


def normalize_32f(w, src):
    lib().orc_normalize_32f(_p(w), _st(w), _p(src), _st(src), src.shape[0], src.shape[1])


def dilate3x3_8u(src):
    dst = np.empty_like(src)
    lib().orc_dilate3x3_8u(_p(src), _st(src), _p(dst), _st(dst), src.shape[0], src.shape[1])
    return dst


def nv12_to_bgr(src):
    h = src.shape[0] * 2 // 3; w = src.shape[1]
    dst = np.empty((h, w, 3), np.uint8)
    lib().orc_nv12_to_bgr(_p(src), _st(src), w, h, _p(dst), _st(dst))
    return dst


def bgr_to_i420(src):
    h, w = src.shape[:2]
    dst = np.empty((h * 3 // 2, w), np.uint8)
    lib().orc_bgr_to_i420(_p(src), _st(src), w, h, _p(dst))
    return dst


def build_warp_maps(proj, tl_u, tl_v, rows, cols, k_rinv, scale, t=(0, 0, 0)):
    k = np.ascontiguousarray(k_rinv, np.float32).reshape(9)
    tt = np.ascontiguousarray(t, np.float32).reshape(3)
    mx = np.empty((rows, cols), np.float32); my = np.empty((rows, cols), np.float32)
    lib().orc_build_warp_maps(proj, tl_u, tl_v, rows, cols, _p(k), _p(tt), C.c_float(scale),
                              _p(mx), _st(mx), _p(my), _st(my))
    return mx, my


def custom_resize_32f(src, tx, ty):
    src = np.ascontiguousarray(src, np.float32)
    out = np.empty((ty, tx), np.float32)
    lib().orc_custom_resize_32f(_p(src), _st(src), src.shape[0], src.shape[1], _p(out), _st(out), ty, tx)
    return out


# ------------------------------------------------------------------ geometry

def projector(K, R, scale, T=None):
    p = Projector()
    K = np.ascontiguousarray(K, np.float32).reshape(9); R = np.ascontiguousarray(R, np.float32).reshape(9)
    Tp = None if T is None else _p(np.ascontiguousarray(T, np.float32).reshape(3))
    lib().orc_set_camera_params(C.byref(p), _p(K), _p(R), Tp, C.c_float(scale))
    return p


def k_rinv_gpu(K, R):
    K = np.ascontiguousarray(K, np.float32).reshape(9); R = np.ascontiguousarray(R, np.float32).reshape(9)
    out = np.empty(9, np.float32)
    lib().orc_k_rinv_gpu(_p(K), _p(R), _p(out))
    return out


def detect_result_roi(proj, p, src_w, src_h):
    v = [C.c_int() for _ in range(4)]
    lib().orc_detect_result_roi(proj, C.byref(p), src_w, src_h, *[C.byref(x) for x in v])
    return tuple(x.value for x in v)  # tl_x, tl_y, br_x, br_y (inclusive)


def warp_roi(proj, K, R, scale, src_w, src_h):
    """RotationWarperBase::warpRoi (warpers_inl.hpp:136-146): Rect(tl, br + 1)."""
    tlx, tly, brx, bry = detect_result_roi(proj, projector(K, R, scale), src_w, src_h)
    return (tlx, tly, brx - tlx + 1, bry - tly + 1)


def build_maps_cpu(proj, p, tl_x, tl_y, rows, cols):
    mx = np.empty((rows, cols), np.float32); my = np.empty((rows, cols), np.float32)
    lib().orc_build_maps_cpu(proj, C.byref(p), tl_x, tl_y, rows, cols, _p(mx), _st(mx), _p(my), _st(my))
    return mx, my


def result_roi(corners, sizes):
    n = len(corners)
    r = lib().orc_result_roi(n, _ia([c[0] for c in corners]), _ia([c[1] for c in corners]),
                             _ia([s[0] for s in sizes]), _ia([s[1] for s in sizes]))
    return r.tuple()


def blender_prepare(dst_roi, num_bands):
    g = BlendGeom()
    lib().orc_blender_prepare(Rect(*dst_roi), num_bands, C.byref(g))
    return g


def blender_view_geom(g, tl, mask_size):
    vg = ViewGeom()
    lib().orc_blender_view_geom(C.byref(g), tl[0], tl[1], mask_size[0], mask_size[1], C.byref(vg))
    return vg


# ------------------------------------------------------------------ calibration-time

def distance_transform_l1(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty(src.shape, np.float32)
    lib().orc_distance_transform_l1(_p(src), _st(src), src.shape[0], src.shape[1], _p(dst), _st(dst))
    return dst


def voronoi_seams(corners, masks):
    """masks: list of contiguous uint8 (h, w) arrays, modified in place."""
    n = len(masks)
    for m in masks:
        assert m.flags.c_contiguous and m.dtype == np.uint8
    ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
    lib().orc_voronoi_seams(n, _ia([c[0] for c in corners]), _ia([c[1] for c in corners]),
                            _ia([m.shape[1] for m in masks]), _ia([m.shape[0] for m in masks]), ptrs)
    return masks


def gain_compensator(corners, images, masks):
    n = len(images)
    imgs = [np.ascontiguousarray(i, np.uint8) for i in images]; ms_ = [np.ascontiguousarray(m, np.uint8) for m in masks]
    ip = (C.c_void_p * n)(*[i.ctypes.data for i in imgs]); mp = (C.c_void_p * n)(*[m.ctypes.data for m in ms_])
    g = (C.c_double * n)()
    ok = lib().orc_gain_compensator(n, _ia([c[0] for c in corners]), _ia([c[1] for c in corners]),
                                    _ia([m.shape[1] for m in ms_]), _ia([m.shape[0] for m in ms_]), ip, mp, g)
    assert ok
    return list(g)


def convert_mesh_to_map(mesh_x, mesh_y, width, height):
    mesh_x = np.ascontiguousarray(mesh_x, np.float32); mesh_y = np.ascontiguousarray(mesh_y, np.float32)
    N, M = mesh_x.shape
    mx = np.empty((height, width), np.float32); my = np.empty((height, width), np.float32)
    lib().orc_convert_mesh_to_map(_p(mesh_x), _p(mesh_y), N, M, width, height, _p(mx), _p(my))
    return mx, my


# ------------------------------------------------------------------ blender object

def feather_weight_map(mask, sharpness=0.02):
    """createWeightMap (blenders.cpp:944-951)."""
    mask = np.ascontiguousarray(mask, np.uint8)
    w = np.empty(mask.shape, np.float32)
    lib().orc_feather_weight_map(_p(mask), _st(mask), mask.shape[0], mask.shape[1], C.c_float(sharpness), _p(w), _st(w))
    return w


def feather_blend(corners, imgs8u, masks, sharpness=0.02):
    """FeatherBlender prepare / feed x N / blend (blenders.cpp:139-186) on 8UC3 views (converted to 16S like the caller does)."""
    sizes = [(m.shape[1], m.shape[0]) for m in masks]
    roi = result_roi(corners, sizes)
    dst = np.zeros((roi[3], roi[2], 3), np.int16); dw = np.zeros((roi[3], roi[2]), np.float32)
    for (cx, cy), img, m in zip(corners, imgs8u, masks):
        w = feather_weight_map(m, sharpness)
        i16 = np.ascontiguousarray(img, np.uint8).astype(np.int16)
        lib().orc_feather_feed(_p(i16), _st(i16), _p(w), _st(w), m.shape[0], m.shape[1], cx - roi[0], cy - roi[1],
                               _p(dst), _st(dst), _p(dw), _st(dw))
    mask = np.empty((roi[3], roi[2]), np.uint8)
    lib().orc_feather_blend(_p(dst), _st(dst), _p(dw), _st(dw), roi[3], roi[2], _p(mask), _st(mask))
    return dst, mask, roi


def cv_pyr_down_16s(src):
    """cv::pyrDown on 16S (CPU): (sum + 128) >> 8, BORDER_REFLECT_101  pyramids.cpp:851-964"""
    src = np.ascontiguousarray(src, np.int16)
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty(((src.shape[0] + 1) // 2, (src.shape[1] + 1) // 2) + (() if src.ndim == 2 else (cn,)), np.int16)
    lib().orc_cv_pyr_down_16s(_p(src), _st(src), src.shape[0], src.shape[1], cn, _p(dst), _st(dst))
    return dst


def cv_pyr_down_32f(src):
    src = np.ascontiguousarray(src, np.float32)
    dst = np.empty(((src.shape[0] + 1) // 2, (src.shape[1] + 1) // 2), np.float32)
    lib().orc_cv_pyr_down_32f(_p(src), _st(src), src.shape[0], src.shape[1], _p(dst), _st(dst))
    return dst


def cv_pyr_up_16s(src):
    """cv::pyrUp on 16S (CPU) to twice the size: (sum + 32) >> 6  pyramids.cpp:976-1078"""
    src = np.ascontiguousarray(src, np.int16)
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((src.shape[0] * 2, src.shape[1] * 2) + (() if src.ndim == 2 else (cn,)), np.int16)
    lib().orc_cv_pyr_up_16s(_p(src), _st(src), src.shape[0], src.shape[1], cn, _p(dst), _st(dst))
    return dst


class Blender:
    """The fork's GPU MultiBandBlender (prepare / init_gpu / feed_online / blend(gpuOut))."""

    def __init__(self, corners, sizes, num_bands=5, cpu_flavour=False):
        """cpu_flavour: the reference's CPU branch (cv::pyrDown / pyrUp rounding, weight pyramid rebuilt per feed) instead of the GPU branch."""
        self.n = len(corners)
        self.cpu_flavour = bool(cpu_flavour)
        self.corners = [tuple(c) for c in corners]
        self.sizes = [tuple(s) for s in sizes]
        self._h = C.c_void_p(lib().orc_blender_create(
            self.n, num_bands, _ia([c[0] for c in corners]), _ia([c[1] for c in corners]),
            _ia([s[0] for s in sizes]), _ia([s[1] for s in sizes])))
        g = BlendGeom()
        lib().orc_blender_get_geom(self._h, C.byref(g))
        self.geom = g
        self.num_bands = g.num_bands
        if cpu_flavour:
            lib().orc_blender_set_flavour(self._h, 1)

    def close(self):
        if self._h:
            lib().orc_blender_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def init_view(self, v, mask):
        mask = np.ascontiguousarray(mask, np.uint8)
        assert mask.shape == (self.sizes[v][1], self.sizes[v][0])
        self.__dict__.setdefault("_masks", {})[v] = mask
        lib().orc_blender_init_view(self._h, v, _p(mask), _st(mask))

    def update_mask(self, v, xmesh, ymesh):
        """MultiBandBlender::update_mask (blenders.cpp:297-315): remap(gpu_masks_[v], x_mesh, y_mesh, INTER_LINEAR, BORDER_CONSTANT 0)
        -> convertTo 32F /255 -> copyMakeBorder -> pyrDown chain, i.e. init_gpu's weight steps on the re-warped mask."""
        warped = remap_linear_8uc1(self._masks[v], xmesh, ymesh)
        lib().orc_blender_init_view(self._h, v, _p(warped), _st(warped))
        return warped

    def view_geom(self, v):
        vg = ViewGeom()
        lib().orc_blender_get_view_geom(self._h, v, C.byref(vg))
        return vg

    def feed(self, v, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.sizes[v][1], self.sizes[v][0], 3)
        lib().orc_blender_feed(self._h, v, _p(img), _st(img))

    def feed_cpu(self, v, img16):
        """MultiBandBlender::feed, CPU branch (blenders.cpp:585-696), 16SC3 image."""
        img16 = np.ascontiguousarray(img16, np.int16)
        assert self.cpu_flavour and img16.shape == (self.sizes[v][1], self.sizes[v][0], 3)
        lib().orc_blender_feed_cpu(self._h, v, _p(img16), _st(img16))

    def stitch_online_cpu(self, v, src, xmap, ymap, gain):
        """cv::remap (CPU fixed-point) -> convertTo(gain) -> convertTo(16S) -> CPU feed: the reference's CPU per-view stage."""
        w, h = self.sizes[v]
        xmap = np.ascontiguousarray(xmap, np.float32); ymap = np.ascontiguousarray(ymap, np.float32)
        assert self.cpu_flavour and xmap.shape == (h, w) and ymap.shape == (h, w)
        lib().orc_stitch_online_cpu(self._h, v, _p(src), _st(src), src.shape[0], src.shape[1], _p(xmap), _p(ymap), C.c_double(gain))

    def stitch_online(self, v, src, xmap, ymap, gain, xmesh=None, ymesh=None, want_warped=False):
        w, h = self.sizes[v]
        xmap = np.ascontiguousarray(xmap, np.float32); ymap = np.ascontiguousarray(ymap, np.float32)
        assert xmap.shape == (h, w) and ymap.shape == (h, w)
        xm = ym = None
        if xmesh is not None:
            xm = np.ascontiguousarray(xmesh, np.float32); ym = np.ascontiguousarray(ymesh, np.float32)
        warped = np.empty((h, w, 3), np.uint8) if want_warped else None
        lib().orc_stitch_online(self._h, v, _p(src), _st(src), src.shape[0], src.shape[1], _p(xmap), _p(ymap),
                                C.c_double(gain), None if xm is None else _p(xm), None if ym is None else _p(ym),
                                None if warped is None else _p(warped))
        return warped

    def blend(self):
        fw, fh = self.geom.dst_roi_final.width, self.geom.dst_roi_final.height
        out = np.empty((fh, fw, 3), np.int16); mask = np.empty((fh, fw), np.uint8)
        lib().orc_blender_blend(self._h, _p(out), _st(out), _p(mask), _st(mask))
        return out, mask

    def _level(self, fn, v, l, dtype, cn):
        r, c, s = C.c_int(), C.c_int(), C.c_size_t()
        ptr = fn(self._h, v, l, C.byref(r), C.byref(c), C.byref(s))
        n = r.value * c.value * cn
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16 if dtype == np.int16 else C.c_float)), (n,))
        shape = (r.value, c.value) + ((cn,) if cn > 1 else ())
        return np.array(arr, dtype=dtype).reshape(shape)

    def weight_level(self, v, l):
        return self._level(lib().orc_blender_weight_level, v, l, np.float32, 1)

    def src_level(self, v, l):
        return self._level(lib().orc_blender_src_level, v, l, np.int16, 3)
