"""The other single-GPU BASELINE configurations as SHORT regions of the default `bench.py --gpus 1` run, so that the driver's own record carries
them: cfg3 = configs[2] (CPW 40 x 40, meshes re-expanded every 60 frames inside the region), cfg5 = configs[4]'s geometry (12 x 4K -> 7680 x 3840; plus its two
single-GPU sharding variants: 2 pano-column windows and 2 view shards on the one GPU, each checked against the unsharded frame), shipped = the configuration the reference
ships (APP/defs.h:51-66, calibration.cpp:183-194, timed.cpp:75-104: cylindrical, per-frame cuda::resize inside the region, 6 bands, CPW 10 x 10).
Each entry: value, ms_per_frame, roofline {kernel, frac, traffic, traffic_stale, ...}, verified (batched = one-frame path), verified_vs_oracle (one full-size frame
bit-identical to the CPU oracle).  Same Workload class, same timed_region as the headline; only shorter."""
import time
import types

import torch

from . import pmc
from . import roofline as RL
from .cpu_baseline import oracle_check
from .frames import Pool
from .regions import Opt, Workload

PRESETS = {
    # (frames timed: cfg3 / shipped 20 x 32 x 10 = 6 400, cfg5 10 x 48 x 10 = 4 800: a quarter to half a second of GPU time each)
    "cfg3": dict(config="cfg3", passes=20, steps=10, warmup=2, distinct=8),
    "cfg5": dict(config="cfg5", passes=10, steps=10, warmup=2, distinct=8, frame_source="device"),
    "shipped": dict(config="shipped", passes=20, steps=10, warmup=2, distinct=8),
}


def run_one(name, dev, ceiling, check_oracle=True, in_run_pmc=True):
    t0 = time.perf_counter()
    opt = Opt(**PRESETS[name])
    wl = Workload(opt, 0, 1, dev, False)
    try:
        el, _ = wl.timed_region(opt.steps, opt.warmup)
        n_frames = wl.F * opt.passes * opt.steps
        ok, how = wl.verify()
        kmean, lat = wl.instrumented(6)
        pmc_run, pmc_why = pmc.measure(name, wl.Fs, wl.n_distinct, wl.frame_source, timeout_s=60.0) if in_run_pmc else (None, "--no-pmc")
        roof, frame, model = RL.blocks(wl, kmean, el, n_frames, ceiling, pmc_run, pmc_why)
        out = {"workload": wl.workload_string(), "value": round(n_frames / el, 2), "unit": "frames/s", "ms_per_frame": round(el / n_frames * 1e3, 5),
               "frames_timed": n_frames, "steps": opt.steps, "warmup": opt.warmup,
               "roofline": {k: roof[k] for k in ("kernel", "frac", "achieved", "peak", "traffic", "traffic_measured_in_this_run", "traffic_stale", "traffic_source", "mean_launch_ms",
                                                 "frac_useful", "frac_of_copy_ceiling")},
               "frame": {k: frame[k] for k in ("gpu_ms_per_frame", "hbm_bytes_per_frame", "frac_traffic", "wall_frac_traffic")},
               "kernels_ms_per_call": {k: round(v, 5) for k, v in kmean.items()}, "frames_per_call": wl.Fs,
               "verified": ok, "verified_how": how}
        if check_oracle:
            out["verified_vs_oracle"] = oracle_check(wl.cfg, wl.gains, wl.comp, wl.frames[0], wl.cpw)
    finally:
        wl.close()
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def run_cfg5_shards(dev):
    """BASELINE configs[4]'s two splits with both shards on the one GPU (the compute cost of the split; across ranks they are `--gpus 2 --col-shards 2` / `--view-shards 2`)"""
    import synth
    from . import shards
    out = {}
    for key, fn, kw in (("col_shards_2", shards.run_col_shards, dict(col_shards=2, view_shards=1)), ("view_shards_2", shards.run_view_shards, dict(view_shards=2, col_shards=1))):
        t0 = time.perf_counter()
        a = types.SimpleNamespace(config="cfg5", frames=4, steps=3, warmup=1, **kw)
        cfg = dict(synth.CONFIGS["cfg5"])
        line = fn(a, cfg, synth.gains(cfg["n"]), 0, 1, dev, False, frame_source="device")
        out[key] = {"value": line["value"], "unit": "frames/s", "equals_unsharded": line["equals_unsharded"], "unsharded_fps_same_gpu": line.get("unsharded_fps_same_gpu"),
                    "workload": line["config"]["workload"], "seconds": round(time.perf_counter() - t0, 1)}
        torch.cuda.empty_cache()
    return out


def run_all(dev, ceiling, names=("cfg3", "cfg5", "shipped"), check_oracle=True, in_run_pmc=True):
    res = {}
    for name in names:
        try:
            res[name] = run_one(name, dev, ceiling, check_oracle, in_run_pmc)
            if name == "cfg5":
                res[name].update(run_cfg5_shards(dev))
                Pool.drop(3840, 2160)
        except Exception as e:      # noqa: BLE001 -- a side region must never cost the headline line
            res[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            torch.cuda.empty_cache()
    return res
