"""Byte models of SURVEY.md 8(d) (algorithmic / contract bytes per launch and per frame) and the hash of the kernel sources.
Models, not measurements: everything priced on them lives under the bench line's `model` key and is called a ratio, never a fraction."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "video-stitcher_amd"),):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def kernel_bytes(comp, cfg, n_frames, cpw):
    """Per-launch algorithmic bytes, SURVEY 8(d) accounting applied to the exact level sizes:
    source read once; every Gaussian level written once (6 B/px, 16SC3) and read twice (next-level reduce,
    Laplacian+accumulate); weights read once (4 B); dst Laplacian read-modify-write per view rect (12 B);
    collapse reads level + coarser level and writes level (6 B each); output 8UC3 canvas written once."""
    nb = comp.pano_geom().num_bands
    P = []
    A = 0
    for i in range(cfg["n"]):
        g = comp.view_geom(i)
        P.append((g.roi.width + g.left + g.right) * (g.roi.height + g.top + g.bottom))
        A += g.roi.width * g.roi.height
    pg = comp.pano_geom()
    Q = pg.dst_roi.width * pg.dst_roi.height
    sumP = float(sum(P))
    kb = {}
    kb["k_warp"] = cfg["n"] * 3.0 * cfg["w"] * cfg["h"] + 6.0 * sumP
    if cpw:
        # "+ second gather (3 B read + 3 B write) x A".  The FIRST remap (timed as `k_remap_gain`: projection remap + gain into the 8UC3 stage image) reads the
        # source frames and writes 3 B per warped pixel; the SECOND (timed as `k_warp`: the mesh remap writing level 0) reads those 3 B and writes the 16SC3 level 0.  Same
        # frame total as before; round 4 priced the whole second gather against the first kernel.
        kb["k_remap_gain"] = cfg["n"] * 3.0 * cfg["w"] * cfg["h"] + 3.0 * A
        kb["k_warp"] = 3.0 * A + 6.0 * sumP
    for l in range(nb):
        kb["k_down_l%d" % l] = 6.0 * sumP / 4 ** l + 6.0 * sumP / 4 ** (l + 1)
    for l in range(nb + 1):
        b = (6.0 + 4.0 + 12.0) * sumP / 4 ** l
        if l < nb:
            b += (12.0 + 1.5) * Q / 4 ** l
        if l == 0:
            b += 3.0 * cfg["out_w"] * cfg["out_h"]
        kb["k_blend_l%d" % l] = b
    # fused coarse-level launches cover several of the per-level entries above
    kb["k_down_tail"] = sum(v for k, v in kb.items() if k.startswith("k_down_l") and int(k[8:]) >= 3)
    kb["k_blend_tail"] = sum(v for k, v in kb.items() if k.startswith("k_blend_l") and int(k[9:]) >= 3)
    return {k: v * n_frames for k, v in kb.items()}, sumP, Q, A


def csrc_sha16():
    """hash of the product's kernel sources: a PMC traffic summary collected on another state of csrc/ is flagged stale in the bench line"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "video-stitcher_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp", ".inc")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]
