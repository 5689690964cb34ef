"""Synthetic input frames for the bench regions.  Two sources of the SAME content model (128 + 60 sin(2 pi (x/97 + y/61 + c/3 + i/7)) + 40 checker(32) +
U[-8, 8], clipped, 8UC3):
  * numpy_frame -- synth.frame (PCG64 noise), the bytes every earlier round benchmarked on: the 1080p pool of cfg2 / cfg3 / shipped;
  * device_frame -- the same formula evaluated with torch on the GPU (fp64, torch's Philox noise: other noise bytes, same statistics), for the 12 x 4K geometry,
    whose numpy frames cost about a second each on the host: 96 of them would be most of the bench's run time.
Whatever the source, verification (`verified`, `verified_vs_oracle`) reads the frames back from the device, so it checks exactly the bytes that were stitched."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "video-stitcher_amd") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))


def numpy_frame(w, h, i, t, dev):
    import synth
    return torch.from_numpy(synth.frame(w, h, i, t)).to(dev)


def device_frame(w, h, i, t, dev):
    x = torch.arange(w, dtype=torch.float64, device=dev)[None, :]
    y = torch.arange(h, dtype=torch.float64, device=dev)[:, None]
    phase = x / 97.0 + y / 61.0 + i / 7.0
    chk = 40.0 * (((torch.arange(w, device=dev)[None, :] // 32) + (torch.arange(h, device=dev)[:, None] // 32)) & 1).to(torch.float64)
    out = torch.empty((h, w, 3), dtype=torch.float64, device=dev)
    for c in range(3):
        out[:, :, c] = 128.0 + 60.0 * torch.sin(2.0 * math.pi * (phase + c / 3.0)) + chk
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + 1000 * i + t)
    out += torch.rand(out.shape, dtype=torch.float64, device=dev, generator=g) * 16.0 - 8.0
    return out.round_().clamp_(0, 255).to(torch.uint8)


class Pool:
    """frame sets [t][view] -> device tensor, generated once per (size, source) and shared by the regions of one bench run (cfg2, cfg3 and the shipped rig read the same cameras)."""
    _cache = {}

    @classmethod
    def get(cls, n_views, w, h, n_sets, dev, source="numpy"):
        key = (n_views, w, h, str(dev), source)
        have = cls._cache.setdefault(key, [])
        fn = numpy_frame if source == "numpy" else device_frame
        while len(have) < n_sets:
            t = len(have)
            have.append([fn(w, h, i, t, dev) for i in range(n_views)])
        return have[:n_sets]

    @classmethod
    def drop(cls, w=None, h=None):
        for key in [k for k in cls._cache if w is None or (k[1], k[2]) == (w, h)]:
            del cls._cache[key]
