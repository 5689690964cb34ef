import sys, time
sys.path.insert(0, "video-stitcher_amd"); sys.path.insert(0, "tests")
import torch, msstitch as ms, synth
from helpers import make_rig, to_dev
c, cfg, _ = make_rig(ms, "cfg2", max_frames=1)
frames = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, 0)) for i in range(cfg["n"])]]
out = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda")]
run = c.prepared(frames, out8u=out)
for _ in range(20): run()
torch.cuda.synchronize()
for n in (1, 4, 16, 64):
    ts = []
    for rep in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): run()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    ts.sort()
    print("calls in a row %3d: host enqueue %.1f us/call, to idle %.1f us/call" % (n, ts[len(ts) // 2][0], sorted(x[1] for x in ts)[len(ts) // 2]))
