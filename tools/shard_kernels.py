import sys
sys.path.insert(0, "video-stitcher_amd"); sys.path.insert(0, "tests")
import torch, msstitch as ms, synth
from helpers import make_rig, to_dev
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
F = 16 if name == "cfg2" else 4
for S, k in ((1, 0), (2, 0), (2, 1)):
    c, cfg, _ = make_rig(ms, name, max_frames=F, col_shards=S, col_shard_index=k)
    need = c.needed_views()
    pool = [[to_dev(synth.frame(cfg["w"], cfg["h"], i, t)) if (need >> i) & 1 else None for i in range(cfg["n"])] for t in range(2)]
    frames = [pool[j % 2] for j in range(F)]
    out = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
    for _ in range(3):
        c.stitch(frames, out8u=out)
    acc = {}
    for _ in range(10):
        for n, t in c.stitch_timed(frames, out8u=out):
            acc[n] = acc.get(n, 0) + t / 10
    print(name, "shard %d/%d" % (k, S), {n: round(t * 1e3, 1) for n, t in acc.items()}, "sum", round(sum(acc.values()) * 1e3, 1))
    c.close()
