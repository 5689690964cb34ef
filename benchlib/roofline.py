"""The measurement blocks of the bench line.

`roofline` -- the dominant kernel, PHYSICAL only: `traffic` = HBM bytes of one launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE on gfx950, calibrated on a
              1 GiB copy in the same collection), `achieved` = traffic / its mean launch duration (hipEvents on the launch stream, measured in this run), `frac` =
              achieved / 8 TB/s.  `traffic_measured_in_this_run` says whether the counters were collected by this very run (benchlib/pmc.py) or read from the committed
              summary profiles/traffic_<config>.json (then `traffic_stale` compares the hash of csrc/ the summary recorded with the sources that ran).
              `frac_useful` prices the kernel's COMPULSORY bytes, `frac_of_copy_ceiling` divides by the tuned streaming copy timed in this run.
              Nothing in this block is priced on a byte model, and nothing named frac* can exceed 1.
`frame`    -- the same for the whole frame (all kernels of one ms_stitch call; wall time of the multi-stream timed region).
`model`    -- everything priced on SURVEY 8(d)'s CONTRACT bytes (the level-materialised 16S model the reference's pass structure implies): bytes and RATIOS, named as
              such.  They exceed what the design physically moves (u8 levels, no accumulator read-modify-write, skipped zero-weight tiles), so a ratio above 1 is a
              statement about the model, not about the kernel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .model import csrc_sha16, kernel_bytes  # noqa: E402

PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def committed_traffic(config, Fs):
    """profiles/traffic_<config>.json when it describes this launch shape -> (summary, file name) or (None, None)"""
    for tname in ("traffic_%s.json" % config, "traffic_latest.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tpath):
            continue
        tj = json.load(open(tpath))
        if tj.get("config") == config and tj.get("frames_per_launch") == Fs:
            return tj, tname
    return None, None


def compulsory_bytes(wl):
    """COMPULSORY bytes of the warp-type kernels and the resize (what any implementation of that stage has to move: every source byte it samples -- at most 4 taps x 3 B per
    pixel it writes -- and the bytes it writes, over the tiles the work lists keep), per launch of Fs frames"""
    cfg, Fs = wl.cfg, wl.Fs
    ps = wl.comp.plan_stats()
    tpx = ps["warp_tile"][0] * ps["warp_tile"][1]
    warp_px, s1_px = float(ps["n_warp_tiles"] * tpx), float(ps["n_stage1_reachable"] * tpx)
    src_b = cfg["n"] * 3.0 * cfg["w"] * cfg["h"]
    useful = {}
    if wl.cpw:
        useful["k_remap_gain"] = Fs * (min(src_b, 12.0 * s1_px) + 3.0 * s1_px)
        useful["k_warp"] = Fs * (min(3.0 * s1_px, 12.0 * warp_px) + 3.0 * warp_px)
    else:
        useful["k_warp"] = Fs * (min(src_b, 12.0 * warp_px) + 3.0 * warp_px)
    if wl.resize_runs:
        useful["k_resize_batch"] = Fs * cfg["n"] * 3.0 * (wl.full_w * wl.full_h + cfg["w"] * cfg["h"])
    return useful


def blocks(wl, kmean, elapsed_s, frames_timed_per_gpu, ceiling, pmc_run=None, pmc_why=None):
    """-> (roofline, frame, model) for workload `wl` given the instrumented kernel means (ms per launch of Fs frames), the wall time of the timed region and how many
    frames one GPU stitched in it.  pmc_run: the in-run PMC summary (benchlib/pmc.py) or None."""
    synth = wl.synth
    cfg, Fs, comp = wl.cfg, wl.Fs, wl.comp
    kb, sumP, Q, A = kernel_bytes(comp, cfg, Fs, wl.cpw)
    if wl.resize_runs:
        kb["k_resize_batch"] = Fs * cfg["n"] * 3.0 * (wl.full_w * wl.full_h + cfg["w"] * cfg["h"])      # read the camera frame, write the compose-scale one
    dom = max(kmean, key=kmean.get)
    useful = compulsory_bytes(wl)
    P_list = []
    for i in range(cfg["n"]):
        g = comp.view_geom(i)
        P_list.append((g.roi.width + g.left + g.right) * (g.roi.height + g.top + g.bottom))
    b_alg_frame = synth.algorithmic_bytes((cfg["w"], cfg["h"]), P_list, Q, (cfg["out_w"], cfg["out_h"]), warped_px=A, cpw=wl.cpw)
    if wl.resize_runs:
        b_alg_frame += cfg["n"] * 3.0 * (wl.full_w * wl.full_h + cfg["w"] * cfg["h"])
    gpu_ms_call = float(sum(kmean.values()))
    src_bytes = cfg["n"] * 3.0 * cfg["w"] * cfg["h"]
    b_min_frame = src_bytes + 4.0 * (4.0 / 3.0) * float(sum(P_list)) + 3.0 * cfg["out_w"] * cfg["out_h"]
    b_ref_frame = src_bytes + 20.0 * A + 96.0 * float(sum(P_list)) + (95.0 + 9.0) * Q

    # ---- the PMC bytes: this run's own collection first, the committed summary otherwise
    traffic = traffic_call = source = stale = None
    in_run = False
    csrc_now = csrc_sha16()
    tj, tname = (pmc_run, "in-run") if pmc_run else committed_traffic(wl.opt.config, Fs)
    if tj is not None:
        if dom in tj.get("kernels", {}):
            traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
            if dom == "k_resize_batch":      # the instrumented time covers every launch of the call's resize (64 images per launch); so must the bytes
                traffic *= -(-(Fs * cfg["n"]) // 64)
        traffic_call = tj.get("hbm_bytes_per_call")
        cal = tj.get("calibration", {})
        if pmc_run:
            in_run, stale = True, False
            source = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this workload run by this bench invocation (%.0f s): FETCH_SIZE x %.2f + WRITE_SIZE x %.2f, the factors "
                      "calibrated on the 1 GiB tuned copy inside the same passes" % (tj.get("seconds", 0.0), cal.get("fetch_factor", 2.0), cal.get("write_factor", 1.0)))
        else:
            # the counters were collected on the kernels of ONE state of csrc/; the summary records a hash of those sources and the line says when they have moved on
            stale = (tj.get("csrc_sha16") != csrc_now) if tj.get("csrc_sha16") else "unknown (summary predates the source hash)"
            source = ("profiles/%s: rocprofv3 PMC passes (FETCH_SIZE x %.2f, WRITE_SIZE x %.2f, calibrated on the tuned copy in the same run) of this workload, collected %s at commit %s "
                      "(tag %s); NOT measured in this run%s" % (tname, cal.get("fetch_factor", 2.0), cal.get("write_factor", 1.0), tj.get("collected", "?"), tj.get("commit", "?"), tj.get("tag"),
                                                                (" (in-run PMC pass unavailable: %s)" % pmc_why) if pmc_why else ""))
    t_dom = kmean[dom] * 1e-3
    copy_T = ceiling.get("copy_TBps") if ceiling else None
    roof = {"bound": "hbm", "kernel": dom, "peak": PEAK_GBPS, "unit": "GB/s",
            "achieved": (round(traffic / t_dom / 1e9, 1) if traffic else None),
            "frac": (round(traffic / t_dom / (PEAK_GBPS * 1e9), 4) if traffic else None),
            "basis": "pmc-measured HBM bytes / mean launch time (hipEvents, this run) / 8 TB/s" if traffic else "no PMC bytes for this launch shape: frac is null (the byte-model ratios are under `model`)",
            "traffic": traffic, "traffic_measured_in_this_run": in_run, "traffic_source": source, "traffic_stale": stale,
            "traffic_note": "FETCH_SIZE x 2 + WRITE_SIZE = requests of the L2s to the fabric: Infinity-Cache hits are counted as HBM bytes (an upper bound on HBM traffic)",
            "mean_launch_ms": round(kmean[dom], 5),
            "useful_bytes_per_launch": (int(useful[dom]) if dom in useful else None),
            "frac_useful": (round(useful[dom] / t_dom / (PEAK_GBPS * 1e9), 4) if dom in useful else None),
            "frac_useful_note": "compulsory bytes of this kernel (source bytes it samples, capped at 4 taps x 3 B per pixel written, + the u8 bytes it writes over the planned tiles) / mean launch time / 8 TB/s",
            "frac_of_copy_ceiling": (round(traffic / t_dom / 1e12 / copy_T, 4) if (traffic and copy_T) else None)}
    wall_per_frame_s = elapsed_s / frames_timed_per_gpu
    frame = {"gpu_ms_per_frame": round(gpu_ms_call / Fs, 5), "wall_ms_per_frame": round(wall_per_frame_s * 1e3, 5),
             "hbm_bytes_per_frame": (int(traffic_call / Fs) if traffic_call else None),
             "frac_traffic": (round(traffic_call / (gpu_ms_call * 1e-3) / (PEAK_GBPS * 1e9), 4) if traffic_call else None),
             "wall_frac_traffic": (round(traffic_call / Fs / wall_per_frame_s / (PEAK_GBPS * 1e9), 4) if traffic_call else None),
             "wall_frac_of_copy_ceiling": (round(traffic_call / Fs / wall_per_frame_s / 1e12 / copy_T, 4) if (traffic_call and copy_T) else None),
             "note": "PMC-measured HBM bytes of one ms_stitch call / Fs; frac_traffic on the summed kernel time of one stream, wall_frac_traffic on the wall time of the timed region"}
    alg_dom = kb.get(dom, 0.0)
    model = {"what": "SURVEY 8(d) contract bytes (level-materialised 16S model) -- a byte MODEL priced at 8 TB/s over measured times; ratios, not roofline fractions: they exceed 1 where "
                     "the design moves fewer bytes than the model (u8 levels, no accumulator read-modify-write, skipped zero-weight tiles, only the ROI rows of the canvas rewritten)",
             "b_alg_bytes_per_frame": int(b_alg_frame), "b_min_bytes_per_frame": int(b_min_frame), "b_ref_bytes_per_frame": int(b_ref_frame),
             "kernel": dom, "kernel_alg_bytes_per_launch": int(alg_dom),
             "ratio_kernel_alg_bytes_over_peak": round(alg_dom / t_dom / (PEAK_GBPS * 1e9), 4),
             "ratio_frame_alg_bytes_over_peak": round(b_alg_frame * Fs / (gpu_ms_call * 1e-3) / (PEAK_GBPS * 1e9), 4),
             "ratio_wall_alg_bytes_over_peak": round(b_alg_frame / wall_per_frame_s / (PEAK_GBPS * 1e9), 4),
             "ratio_hbm_bytes_over_b_min": (round(traffic_call / Fs / b_min_frame, 3) if traffic_call else None)}
    return roof, frame, model


def measure_ceiling(ms, dev):
    """what a tuned streaming copy / read of 1 GiB reaches on THIS device in THIS run (csrc/compositor.hip k_calib_copy / k_calib_read)"""
    import torch
    try:
        n_c = 1 << 30
        ca = torch.empty(n_c, dtype=torch.uint8, device=dev).random_(0, 255); cb = torch.empty_like(ca)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = {"copy": 1e9, "read": 1e9}
        for it in range(7):
            for what in ("copy", "read"):
                e0.record()
                if what == "copy":
                    ms.calib_copy(ca, cb)
                else:
                    ms.calib_read(ca)
                e1.record(); e1.synchronize()
                if it >= 2:
                    best[what] = min(best[what], e0.elapsed_time(e1))
        out = {"copy_TBps": round(2.0 * n_c / (best["copy"] * 1e-3) / 1e12, 3), "read_TBps": round(n_c / (best["read"] * 1e-3) / 1e12, 3),
               "how": "best of 5 launches of the tuned 16 B/lane streaming kernels over 1 GiB (copy counts read + written bytes), hipEvents; "
                      "tools/copy_probe.hip is the 130-variant sweep they were picked from"}
        del ca, cb
        return out
    except Exception as e:      # noqa: BLE001 -- never fail the bench line on the optional ceiling
        return {"error": str(e)[:200]}
