"""Independent numpy restatements of the reference kernels, used ONLY to cross-check the C oracle
(tests/test_oracle_crosscheck.py).  Deliberately written differently from oracle/*.c: vectorised, integer
arithmetic for the 16S pyramids (the CUDA fp32 sums are exact dyadics), float64 emulation of fmaf for the taps."""
import numpy as np


def r101(i, n):
    i = np.abs(i)
    return np.abs((n - 1) - np.abs((n - 1) - i)) % n


def reflect(i, n):
    last = n - 1
    hi = last - np.abs(last - i) + (i > last)
    return (np.abs(hi) - (hi < 0)) % n


def rne_shift(s, k):
    s = s.astype(np.int64)
    return (s + ((1 << (k - 1)) - 1) + ((s >> k) & 1)) >> k


def pyr_down_16s(src):
    """pyr_down.cu:55-174 as an integer 5x5 binomial with BORDER_REFLECT_101 and round-half-even."""
    h, w = src.shape[:2]
    k = np.array([1, 4, 6, 4, 1], np.int64)
    oy = np.arange((h + 1) // 2) * 2
    ox = np.arange((w + 1) // 2) * 2
    acc = 0
    s = src.astype(np.int64)
    for a in range(5):
        rows = s[r101(oy + a - 2, h)]
        for b in range(5):
            acc = acc + k[a] * k[b] * rows[:, r101(ox + b - 2, w)]
    return np.clip(rne_shift(acc, 8), -32768, 32767).astype(np.int16)


def pyr_up_16s(src):
    """pyr_up.cu:55-145: zero-insert, 5x5 binomial x4, source index min(n-1, |i|)."""
    h, w = src.shape[:2]
    s = src.astype(np.int64)
    z = np.zeros((2 * h + 4, 2 * w + 4) + src.shape[2:], np.int64)    # dst grid with 2-px apron, index = dst + 2
    ys = np.arange(-2, 2 * h + 2)
    xs = np.arange(-2, 2 * w + 2)
    ry = np.minimum(np.abs(ys >> 1), h - 1)
    rx = np.minimum(np.abs(xs >> 1), w - 1)
    full = s[ry][:, rx]
    ev_y = (ys % 2 == 0)
    ev_x = (xs % 2 == 0)
    m = ev_y[:, None] & ev_x[None, :]
    z = full * (m[..., None] if src.ndim == 3 else m)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    acc = 0
    H, W = 2 * h, 2 * w
    for a in range(5):
        for b in range(5):
            acc = acc + k[a] * k[b] * z[a:a + H, b:b + W]
    return np.clip(rne_shift(acc, 6), -32768, 32767).astype(np.int16)


def remap_linear(src, mx, my):
    """filters.hpp:90-114 + border_interpolate.hpp:698-717; fmaf emulated in float64 (products of two fp32 are exact
    in fp64; the final rounding to fp32 can double-round in rare cases -> compare with tolerance 1)."""
    h, w = src.shape[:2]
    x1 = np.floor(mx).astype(np.int64); y1 = np.floor(my).astype(np.int64)
    x2, y2 = x1 + 1, y1 + 1
    f = np.float32
    wts = [((x2.astype(f) - mx) * (y2.astype(f) - my)), ((mx - x1.astype(f)) * (y2.astype(f) - my)),
           ((x2.astype(f) - mx) * (my - y1.astype(f))), ((mx - x1.astype(f)) * (my - y1.astype(f)))]
    taps = [(y1, x1), (y1, x2), (y2, x1), (y2, x2)]
    out = np.zeros(mx.shape + (3,), np.float32)
    for (yy, xx), wt in zip(taps, wts):
        inb = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = np.where(inb[..., None], src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0).astype(np.float64)
        out = (v * wt[..., None].astype(np.float64) + out.astype(np.float64)).astype(np.float32)
    r = np.rint(out)
    return np.clip(np.nan_to_num(r, nan=0.0), 0, 255).astype(np.uint8)


def add_src_weight(src, w, dst, dst_w):
    t = np.trunc(src.astype(np.float32) * w[..., None].astype(np.float32)).astype(np.int64)
    dst[...] = ((dst.astype(np.int64) + t + 32768) % 65536 - 32768).astype(np.int16)
    dst_w += w


def normalize(w, src):
    den = (w + np.float32(1e-5)).astype(np.float32)
    src[...] = np.trunc(src.astype(np.float32) / den[..., None]).astype(np.int16)


def feather_blend_np(corners, imgs8u, masks, dt_l1, sharpness=0.02):
    """numpy restatement of FeatherBlender (blenders.cpp:139-186); dt_l1(mask) -> float32 L1 distance transform."""
    xs = [c[0] for c in corners]; ys = [c[1] for c in corners]
    x0, y0 = min(xs), min(ys)
    x1 = max(c[0] + m.shape[1] for c, m in zip(corners, masks)); y1 = max(c[1] + m.shape[0] for c, m in zip(corners, masks))
    dst = np.zeros((y1 - y0, x1 - x0, 3), np.int16); dw = np.zeros((y1 - y0, x1 - x0), np.float32)
    for (cx, cy), img, m in zip(corners, imgs8u, masks):
        w = np.minimum(dt_l1(m).astype(np.float32) * np.float32(sharpness), np.float32(1.0)).astype(np.float32)
        contrib = np.trunc(img.astype(np.float32) * w[:, :, None]).astype(np.int16)
        sl = (slice(cy - y0, cy - y0 + m.shape[0]), slice(cx - x0, cx - x0 + m.shape[1]))
        dst[sl] = (dst[sl].astype(np.int32) + contrib).astype(np.int16)
        dw[sl] = dw[sl] + w
    q = np.trunc(dst.astype(np.float32) / (dw + np.float32(1e-5))[:, :, None]).astype(np.int16)
    mask = np.where(dw > np.float32(1e-5), 255, 0).astype(np.uint8)
    q[mask == 0] = 0
    return q, mask


def cv_remap_linear_np(src, mapx, mapy):
    """Independent numpy statement of cv::remap's CPU arithmetic (INTER_LINEAR, float map pair, BORDER_CONSTANT 0, 8-bit):
    coordinates quantised to 1/32 px (round half to even), closed-form 15-bit weights (32-fy)(32-fx)*32 ..., except the all-integer entry,
    which OpenCV's table holds as {32767, 0, 0, 1} (32768 saturates to short and the fix-up adds the missing 1 to the last tap)."""
    src = np.asarray(src)
    img = src[..., None] if src.ndim == 2 else src
    h, w, cn = img.shape
    q = []
    for m in (mapx, mapy):
        r = np.rint(np.asarray(m, np.float32) * np.float32(32)).astype(np.float64)
        bad = ~np.isfinite(r) | (r >= 2.0 ** 31) | (r < -2.0 ** 31)
        q.append(np.where(bad, -2.0 ** 31, r).astype(np.int64))
    fx, fy = q[0] & 31, q[1] & 31
    sx, sy = np.clip(q[0] >> 5, -32768, 32767), np.clip(q[1] >> 5, -32768, 32767)
    wts = [(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32]
    whole = (fx == 0) & (fy == 0)
    wts[0] = np.where(whole, 32767, wts[0]); wts[3] = np.where(whole, 1, wts[3])
    acc = np.zeros(sx.shape + (cn,), np.int64)
    for t, (dx, dy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        xx, yy = sx + dx, sy + dy
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        px = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        acc += np.where(ok[..., None], px, 0) * wts[t][..., None]
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out[..., 0] if src.ndim == 2 else out


# ---- the reference's CPU pyramids (cv::pyrDown / cv::pyrUp, OCV/imgproc/src/pyramids.cpp:851-1078), numpy statements --------------------
def cv_pyr_down_16s(src):
    """pyrDown_<FixPtCast<short,8>>: integer 5x5 binomial, BORDER_REFLECT_101, (sum + 128) >> 8 (ties round UP, unlike the CUDA kernel)."""
    h, w = src.shape[:2]
    k = np.array([1, 4, 6, 4, 1], np.int64)
    oy, ox = np.arange((h + 1) // 2) * 2, np.arange((w + 1) // 2) * 2
    s = src.astype(np.int64)
    acc = 0
    for a in range(5):
        rows = s[r101(oy + a - 2, h)]
        for b in range(5):
            acc = acc + k[a] * k[b] * rows[:, r101(ox + b - 2, w)]
    return ((acc + 128) >> 8).astype(np.int16)


def cv_pyr_up_16s(src):
    """pyrUp_<FixPtCast<short,6>> to twice the size: the same taps and source indexing (left / top mirrored, right / bottom replicated:
    pyramids.cpp:1013-1031 spell out 6a + 2b and b + 7c) as the CUDA kernel, (sum + 32) >> 6."""
    h, w = src.shape[:2]
    s = src.astype(np.int64)
    ys, xs = np.arange(-2, 2 * h + 2), np.arange(-2, 2 * w + 2)
    full = s[np.minimum(np.abs(ys >> 1), h - 1)][:, np.minimum(np.abs(xs >> 1), w - 1)]
    m = (ys % 2 == 0)[:, None] & (xs % 2 == 0)[None, :]
    z = full * (m[..., None] if src.ndim == 3 else m)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    acc = 0
    for a in range(5):
        for b in range(5):
            acc = acc + k[a] * k[b] * z[a:a + 2 * h, b:b + 2 * w]
    return ((acc + 32) >> 6).astype(np.int16)


def cv_pyr_down_32f(src):
    """pyrDown_<FltCast<float,8>, PyrDownVec_32f> (x86 SSE build): fp32 throughout; horizontal ((6c + 4(b + d)) + a) + e; vertical in the SSE
    order ((r0 + r4) + (r2 + r2)) + 4((r1 + r3) + r2) for the first (width // 8) * 8 columns and ((6 r2 + 4 (r1 + r3)) + r0) + r4 for the rest;
    then * (1 / 256)."""
    f = np.float32
    h, w = src.shape
    oy, ox = np.arange((h + 1) // 2) * 2, np.arange((w + 1) // 2) * 2
    s = src.astype(f)
    col = [s[:, r101(ox + b - 2, w)] for b in range(5)]
    hrow = ((col[2] * f(6) + (col[1] + col[3]) * f(4)) + col[0]) + col[4]          # (h, dcols) fp32, evaluated op by op
    r = [hrow[r101(oy + a - 2, h)] for a in range(5)]
    vec = ((r[0] + r[4]) + (r[2] + r[2])) + ((r[1] + r[3]) + r[2]) * f(4)
    tail = ((r[2] * f(6) + (r[1] + r[3]) * f(4)) + r[0]) + r[4]
    nvec = (len(ox) // 8) * 8
    out = np.where(np.arange(len(ox))[None, :] < nvec, vec, tail).astype(f)
    return (out * f(1.0 / 256)).astype(f)
