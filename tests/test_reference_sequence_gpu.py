"""The reference's own call sequence -- stitch_online (APP/timed.cpp:56-121), MultiBandBlender::feed_online and
blend (blenders.cpp:700-832) -- replayed op by op through the per-op entry points of the C-ABI on the GPU
(cuda::remap, convertTo, copyMakeBorder, pyrDown, pyrUp, subtract, addSrcWeightGpu32F, normalizeUsingWeightMapGpu32F,
add, compare, setTo), and compared with (a) the oracle and (b) the fused ms_stitch path.  All three must agree
bit for bit: this is the drop-in claim at both granularities."""
import numpy as np
import pytest
import torch

import synth
from helpers import host, make_rig, oracle_blender_from, to_dev

pytestmark = pytest.mark.gpu


class RefSequenceBlender:
    """MultiBandBlender (GPU branch of the fork) written against msstitch's per-op API, method for method."""

    def __init__(self, ms, comp, cfg):
        self.ms, self.nb = ms, comp.pano_geom().num_bands
        pg = comp.pano_geom()
        self.final = (pg.dst_roi_final.width, pg.dst_roi_final.height)
        r, c = pg.dst_roi.height, pg.dst_roi.width
        self.dst_lap, self.dst_w = [], []
        for _ in range(self.nb + 1):                                   # prepare(): blenders.cpp:257-273
            self.dst_lap.append(torch.zeros((r, c, 3), dtype=torch.int16, device="cuda"))
            self.dst_w.append(torch.zeros((r, c), dtype=torch.float32, device="cuda"))
            r, c = (r + 1) // 2, (c + 1) // 2
        self.geom, self.wpyr = [], []
        for i in range(cfg["n"]):                                      # init_gpu(): blenders.cpp:344-461
            g = comp.view_geom(i)
            self.geom.append(g)
            w = ms.convert(comp.mask(i), torch.float32, 1.0 / 255.0)
            pyr = [ms.copy_make_border(w, g.top, g.bottom, g.left, g.right, ms.BORDER_CONSTANT)]
            for _ in range(self.nb):
                pyr.append(ms.pyr_down(pyr[-1]))
            self.wpyr.append(pyr)

    def feed_online(self, img, i):                                      # blenders.cpp:700-749
        ms, g = self.ms, self.geom[i]
        bordered = ms.copy_make_border(img, g.top, g.bottom, g.left, g.right, ms.BORDER_REFLECT)
        pyr = [ms.convert(bordered, torch.int16)]
        for _ in range(self.nb):
            pyr.append(ms.pyr_down(pyr[-1]))
        for l in range(self.nb):
            up = ms.pyr_up(pyr[l + 1])
            ms.subtract(pyr[l], up, dst=pyr[l])
        x_tl, y_tl, x_br, y_br = g.x_tl, g.y_tl, g.x_br, g.y_br
        for l in range(self.nb + 1):
            ms.add_src_weight_32f(pyr[l], self.wpyr[i][l], self.dst_lap[l][y_tl:y_br, x_tl:x_br], self.dst_w[l][y_tl:y_br, x_tl:x_br])
            x_tl //= 2; y_tl //= 2; x_br //= 2; y_br //= 2

    def blend(self):                                                    # blenders.cpp:758-832
        ms = self.ms
        for l in range(self.nb + 1):
            ms.normalize_using_weight_32f(self.dst_w[l], self.dst_lap[l])
        for l in range(self.nb, 0, -1):
            up = ms.pyr_up(self.dst_lap[l])
            ms.add(up, self.dst_lap[l - 1], dst=self.dst_lap[l - 1])
        fw, fh = self.final
        dst_mask = ms.compare_gt(self.dst_w[0][:fh, :fw], 1e-5)
        inv = ms.compare_eq(dst_mask, 0)
        roi = self.dst_lap[0][:fh, :fw]
        ms.set_zero_masked(roi, inv)
        out = roi.clone()
        for l in range(self.nb + 1):
            self.dst_lap[l].zero_(); self.dst_w[l].zero_()
        return out, dst_mask


@pytest.mark.parametrize("cpw", [False, True])
def test_reference_call_sequence(ms, cuda, oracle, cpw):
    comp, cfg, gains = make_rig(ms, "mini6", enable_cpw=cpw)
    frames = [synth.frame(cfg["w"], cfg["h"], i, 5) for i in range(cfg["n"])]
    meshes = None
    if cpw:
        meshes = []
        for i in range(cfg["n"]):
            r = comp.view_geom(i).roi
            comp.set_mesh(i, *synth.mesh(r.width, r.height, 10, 10, phase=0.4 * i, amp=4.0))
            meshes.append(comp.mesh_maps(i))
    seq = RefSequenceBlender(ms, comp, cfg)
    for t in range(2):                                                  # two frames: accumulators are cleared by blend()
        for i in range(cfg["n"]):                                       # stitch_online: timed.cpp:56-121
            full = to_dev(frames[i])
            xm, ym = comp.maps(i)
            img = ms.remap(full, xm, ym, ms.INTER_LINEAR)
            ms.convert_scale_8u(img, gains[i], inplace=True)
            warped = ms.remap(img, meshes[i][0], meshes[i][1], ms.INTER_LINEAR) if cpw else img
            seq.feed_online(warped, i)
        out_seq, mask_seq = seq.blend()
    pg = comp.pano_geom()
    out_fused = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=cuda)
    comp.stitch([[to_dev(f) for f in frames]], out16s=[out_fused])
    torch.cuda.synchronize()
    b, _ = oracle_blender_from(oracle, comp, cfg)
    for i in range(cfg["n"]):
        xm, ym = [host(t) for t in comp.maps(i)]
        b.stitch_online(i, frames[i], xm, ym, gains[i], *( [host(m) for m in meshes[i]] if cpw else [None, None]))
    ref, refmask = b.blend()
    b.close()
    assert np.array_equal(host(out_seq), ref), "per-op sequence vs oracle"
    assert np.array_equal(host(mask_seq), refmask)
    assert torch.equal(out_fused, out_seq), "fused path vs per-op sequence"
    assert np.array_equal(host(comp.result_mask()), refmask)
    # consume(): convertTo(CV_8U) of the 16S result (timed.cpp:251)
    assert np.array_equal(host(ms.convert(out_seq, torch.uint8)), oracle.convert_16s_8u(ref))
    comp.close()


def test_compose_scale_resize_path(ms, cuda, oracle):
    """|compose_scale - 1| > 0.1 branch of stitch_online (timed.cpp:75-85): cuda::resize(Size(), s, s, INTER_LINEAR)
    before the remap.  COMPOSE_MEGAPIX = 1.4 gives s = 0.82 for 1080p (APP/defs.h:53)."""
    rng = np.random.default_rng(11)
    full = rng.integers(0, 256, size=(270, 480, 3), dtype=np.uint8)
    s = 0.82
    small = ms.resize_linear(to_dev(full), fx=s, fy=s)
    ref_small = oracle.resize_linear_8u(full, fx=s, fy=s)
    assert np.array_equal(host(small), ref_small)
    assert small.shape[:2] == (int(np.rint(270 * s)), int(np.rint(480 * s)))
    K, R = synth.camera(6, small.shape[1], small.shape[0], 90.0, 1)
    roi = ms.warp_roi(ms.PROJ_CYLINDRICAL, K, R, 120.0, small.shape[1], small.shape[0])
    xm, ym = ms.build_warp_maps(ms.PROJ_CYLINDRICAL, roi[0], roi[1], roi[3], roi[2], oracle.k_rinv_gpu(K, R), 120.0)
    got = ms.remap(small, xm, ym, ms.INTER_LINEAR)
    assert np.array_equal(host(got), oracle.remap_linear_8uc3(ref_small, host(xm), host(ym)))
