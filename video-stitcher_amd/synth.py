"""BENCH / TEST SUPPORT, not part of the product (nothing in csrc/, host/ or shim/ depends on it; it sits next to msstitch.py only so that tests/
and bench.py import it from one place): the synthetic rig, frames and CPW meshes of SURVEY.md 8(d).

Pure numpy; produces host arrays.  No oracle and no GPU code in here.
Rig model = the reference's calibrateCameras (APP/calibration.cpp:28-68): view i yaw = 2*pi*i/N about +y,
principal point at the image centre, focal = (W/2)/tan(hfov/2), aspect 1.
"""
import math

import numpy as np

CONFIGS = {
    # BASELINE.json configs[1]: 6x1080p -> 3840x1920 equirect, multiband, 5 bands
    "cfg2": dict(n=6, w=1920, h=1080, hfov_deg=90.0, out_w=3840, out_h=1920, num_bands=5),
    # configs[4]: 12x4K -> 7680x3840
    "cfg5": dict(n=12, w=3840, h=2160, hfov_deg=60.0, out_w=7680, out_h=3840, num_bands=5),
    # small rigs for oracle-speed parity tests
    "mini6": dict(n=6, w=320, h=180, hfov_deg=90.0, out_w=640, out_h=320, num_bands=3),
    "mini4": dict(n=4, w=200, h=150, hfov_deg=110.0, out_w=512, out_h=256, num_bands=4),
}


def camera(n, w, h, hfov_deg, i, yaw=None):
    """K, R (fp32 3x3) of view i: rot = static_cast<float>(2.0*PI*float(i)/N) (calibration.cpp:35)."""
    rot = np.float32(2.0 * math.pi * float(i) / n) if yaw is None else np.float32(yaw)
    c, s = math.cos(rot), math.sin(rot)   # cos(float) promoted to double, stored as float
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    f = (w / 2.0) / math.tan(math.radians(hfov_deg) / 2.0)
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]], np.float32)
    return K, R


def warp_scale(out_w):
    """scale so that the full circle is exactly out_w columns."""
    return float(np.float32(out_w / (2.0 * math.pi)))


def gains(n):
    return [1.0 + 0.02 * (i - (n - 1) / 2.0) for i in range(n)]


def frame(w, h, i, t, noise=True):
    """BGR uint8 HxWx3: clip(128 + 60 sin(2pi(x/97 + y/61 + c/3 + i/7)) + 40 checker(x//32, y//32) + U[-8,8])."""
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    phase = x / 97.0 + y / 61.0 + i / 7.0
    chk = 40.0 * (((np.arange(w)[None, :] // 32) + (np.arange(h)[:, None] // 32)) & 1)
    out = np.empty((h, w, 3), np.float64)
    for c in range(3):
        out[:, :, c] = 128.0 + 60.0 * np.sin(2.0 * math.pi * (phase + c / 3.0)) + chk
    if noise:
        rng = np.random.Generator(np.random.PCG64(1234 + 1000 * i + t))
        out += rng.uniform(-8.0, 8.0, size=out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def nv12_frame(w, h, i):
    """Synthetic NV12 camera frame (h*3/2 x w uint8): Y = the green-channel pattern of frame(noise=False), U/V = integer ramps.
    Same bytes as synth_nv12() of host/stitch_app.cpp."""
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    chk = 40.0 * (((np.arange(w)[None, :] // 32) + (np.arange(h)[:, None] // 32)) & 1)
    Y = np.clip(np.rint(128.0 + 60.0 * np.sin(2.0 * math.pi * (x / 97.0 + y / 61.0 + i / 7.0 + 1.0 / 3.0)) + chk), 0, 255).astype(np.uint8)
    uv = np.empty((h // 2, w), np.uint8)
    xs = np.arange(w // 2)[None, :]; ys = np.arange(h // 2)[:, None]
    uv[:, 0::2] = (64 + (3 * xs + 5 * i) % 128 + 0 * ys).astype(np.uint8)
    uv[:, 1::2] = (64 + (2 * ys + 7 * i) % 128 + 0 * xs).astype(np.uint8)
    return np.vstack([Y, uv])


def mesh(aw, ah, n_rows, n_cols, phase=0.0, amp=8.0):
    """Forward vertex mesh (N x M) in view-ROI pixels: identity + amp*sin(2pi u + phase)*sin(pi v)."""
    v = np.linspace(0.0, 1.0, n_rows)[:, None]
    u = np.linspace(0.0, 1.0, n_cols)[None, :]
    d = amp * np.sin(2.0 * math.pi * u + phase) * np.sin(math.pi * v)
    mx = (u * (aw - 1) + d).astype(np.float32) * np.ones((n_rows, 1), np.float32)
    my = (v * (ah - 1) + 0.5 * d).astype(np.float32) * np.ones((1, n_cols), np.float32)
    return np.ascontiguousarray(mx, np.float32), np.ascontiguousarray(my, np.float32)


def algorithmic_bytes(view_src_wh, padded_px, pano_padded_px, out_wh, warped_px=0, cpw=False):
    """SURVEY.md 8(d) contract figure B_alg (bytes per frame):
    sum_v[3 W H + 45.33 P_v] + 18 Q + 3 W_out H_out (+ 6 bytes per warped pixel for the CPW gather)."""
    n = len(padded_px)
    b = n * 3.0 * view_src_wh[0] * view_src_wh[1]
    b += (24.0 + 16.0 / 3.0 + 16.0) * float(sum(padded_px))
    b += 18.0 * pano_padded_px
    b += 3.0 * out_wh[0] * out_wh[1]
    if cpw:
        b += 6.0 * warped_px
    return b


def reference_rig(n, w, h, work_megapix=0.6, seam_megapix=0.01, compose_megapix=-1.0, hfov_deg=90.0):
    """stitch_calib's scale bookkeeping and rig model (APP/calibration.cpp:28-68, 101-116, 147-181, 269-281, defs.h:51-53).
    Returns a dict with the scales and, per view, K at compose scale, K at seam scale (fp32) and R."""
    area = float(w * h)
    work_scale = 1.0 if work_megapix < 0 else min(1.0, math.sqrt(work_megapix * 1e6 / area))
    seam_scale = min(1.0, math.sqrt(seam_megapix * 1e6 / area))
    seam_work_aspect = seam_scale / work_scale
    compose_scale = min(1.0, math.sqrt(compose_megapix * 1e6 / area)) if compose_megapix > 0 else 1.0
    compose_work_aspect = compose_scale / work_scale
    focal_tmp = 1.0 / math.tan(math.radians(hfov_deg) * 0.5)
    ppx = (w * work_scale) / 2.0
    ppy = (h * work_scale) / 2.0
    focal = focal_tmp * ppx
    warped_image_scale = float(np.float32(focal))                       # static_cast<float>(cameras[0].focal)
    out = dict(work_scale=work_scale, seam_scale=seam_scale, seam_work_aspect=seam_work_aspect, compose_scale=compose_scale,
               warped_image_scale=warped_image_scale,
               seam_warp_scale=float(np.float32(warped_image_scale * seam_work_aspect)),      # static_cast<float>(scale * swa)
               compose_warp_scale=float(np.float32(np.float32(warped_image_scale) * np.float32(compose_work_aspect))),
               K_seam=[], K_compose=[], R=[])
    swa = np.float32(seam_work_aspect)
    for i in range(n):
        _, R = camera(n, w, h, hfov_deg, i)
        Kw = np.array([[focal, 0, ppx], [0, focal, ppy], [0, 0, 1]], np.float64).astype(np.float32)   # K().convertTo(CV_32F)
        Ks = Kw.copy()
        Ks[0, 0] *= swa; Ks[0, 2] *= swa; Ks[1, 1] *= swa; Ks[1, 2] *= swa                           # calibration.cpp:112-116
        Kc = np.array([[focal * compose_work_aspect, 0, ppx * compose_work_aspect],
                       [0, focal * compose_work_aspect, ppy * compose_work_aspect], [0, 0, 1]], np.float64).astype(np.float32)
        out["K_seam"].append(Ks); out["K_compose"].append(Kc); out["R"].append(R)
    return out
