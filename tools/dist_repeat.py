#!/usr/bin/env python3
"""Developer check, run ON THE GPU BOX: the same stitch_dist command lines N times; for every run whose per-frame checksums differ from the single-rank
reference, print which frames differ (a deterministic pipeline prints nothing but the counts).   python tools/dist_repeat.py [N]"""
import subprocess
import sys

APP = "video-stitcher_amd/stitch_dist"
BASE = ["--views", "6", "--size", "320x240", "--out", "1024x512", "--hfov", "80.0", "--bands", "3"]


def run(args):
    out = subprocess.run([APP] + BASE + [str(a) for a in args] + ["--frame-sums"], capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return [l.split()[2] for l in out.stderr.decode().splitlines() if l.startswith("frame ")]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    sys.path.insert(0, "video-stitcher_amd")
    import synth
    cfg = synth.CONFIGS["mini6"]
    BASE[:] = ["--views", cfg["n"], "--size", "%dx%d" % (cfg["w"], cfg["h"]), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]), "--hfov", cfg["hfov_deg"], "--bands", cfg["num_bands"]]
    BASE[:] = [str(a) for a in BASE]
    groups = [
        (["--gpus", 1, "--frames", 32, "--batch", 4, "--recalib-every", 8, "--mesh", "9x11"],
         [["--gpus", 2, "--share-gpu", "--frames", 32, "--batch", 4, "--recalib-every", 8, "--mesh", "9x11"],
          ["--gpus", 2, "--share-gpu", "--frames", 32, "--batch", 2, "--recalib-every", 8, "--mesh", "9x11"],
          ["--gpus", 4, "--share-gpu", "--frames", 32, "--batch", 2, "--recalib-every", 8, "--mesh", "9x11"]]),
        (["--gpus", 1, "--frames", 16, "--batch", 2, "--recalib-every", 8],
         [["--gpus", 4, "--share-gpu", "--col-shards", 2, "--frames", 16, "--batch", 2, "--recalib-every", 8]]),
        (["--gpus", 1, "--frames", 16, "--batch", 2],
         [["--gpus", 4, "--share-gpu", "--col-shards", 2, "--frames", 16, "--batch", 2],
          ["--gpus", 4, "--share-gpu", "--frames", 16, "--batch", 2]]),
    ]
    for ref_args, cases in groups:
        ref = run(ref_args)
        for it in range(n):
            again = run(ref_args)
            if again != ref:
                print("REFERENCE NOT REPRODUCIBLE", ref_args, [i for i, (a, b) in enumerate(zip(ref, again)) if a != b], flush=True)
        for args in cases:
            bad = 0
            for it in range(n):
                got = run(args)
                if got != ref:
                    bad += 1
                    print("MISMATCH", args, "frames", [i for i, (a, b) in enumerate(zip(ref, got)) if a != b], flush=True)
            print(" ".join(str(a) for a in args), ":", bad, "of", n, "runs differ", flush=True)


if __name__ == "__main__":
    main()
