#!/usr/bin/env python3
"""Time ms_resize_linear_batch on the shipped compose-scale geometry (6 x 1080p per frame -> 1578 x 887; timed.cpp:75-85) and print a checksum of the output,
so two builds (MSSTITCH_LIB=...) can be compared for speed AND bytes.   python tools/time_resize.py [frames] [scale]"""
import hashlib
import importlib
import sys

import torch

sys.path.insert(0, ".")
ms = importlib.import_module("video-stitcher_amd.msstitch")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8217
    n = frames * 6
    g = torch.Generator(device="cpu").manual_seed(7)
    base = torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, generator=g).cuda()
    srcs = [torch.roll(base, i, 0).contiguous() for i in range(n)]
    dh, dw = int(round(1080 * scale)), int(round(1920 * scale))
    dsts = [torch.empty((dh, dw, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    run = ms.resize_linear_batch_prepared(srcs, dsts)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms_call = e0.elapsed_time(e1) / reps
    nbytes = n * (1080 * 1920 * 3 + dh * dw * 3)
    h = hashlib.sha256()
    for d in dsts[:8] + dsts[-2:]:
        h.update(d.cpu().numpy().tobytes())
    print(f"resize {n} images 1920x1080 -> {dw}x{dh}: {ms_call * 1e3:.1f} us per call, {nbytes / ms_call / 1e9:.2f} TB/s (algorithmic bytes), sha256 {h.hexdigest()[:16]}")


if __name__ == "__main__":
    main()
