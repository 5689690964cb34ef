"""Run ON THE GPU BOX: cost of the enqueue-only ms_update_mask (config 3 geometry: 6 x 1080p, CPW 40 x 40 meshes) -> stdout
host time of one call (enqueue only), GPU time of one update, and what planning the work lists with the margin costs per frame."""
import sys, time
sys.path.insert(0, "video-stitcher_amd"); sys.path.insert(0, "tests")
import torch, msstitch as ms, synth
from helpers import make_rig, to_dev

F = 16
res = {}
for margin in (0, 12):
    c, cfg, _ = make_rig(ms, "cfg2", enable_cpw=True, max_frames=F, update_mask_margin=margin)
    for i in range(cfg["n"]):
        r = c.view_geom(i).roi
        c.set_mesh(i, *synth.mesh(r.width, r.height, 40, 40, phase=0.1 * i, amp=6.0))
    pool = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, t)) for i in range(cfg["n"])] for t in range(2)]
    frames = [pool[j % 2] for j in range(F)]
    out = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
    run = c.prepared(frames, out8u=out)
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): run()
    torch.cuda.synchronize(); per_frame = (time.perf_counter() - t0) / 30 / F * 1e6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 12
    for k in range(n): c.update_mask(k % cfg["n"])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    res[margin] = (per_frame, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3)
    print("update_mask_margin %2d: stitch %.1f us/frame (F = 16, one context); ms_update_mask: %.3f ms on the host per call, %.3f ms per call until the GPU is idle"
          % (margin, per_frame, res[margin][1], res[margin][2]))
    c.close()
