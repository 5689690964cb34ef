#!/usr/bin/env python3
"""bench.py -- stitched frames/s of the MI355X compositor on BASELINE.json configs[1]
(6x1080p synthetic views -> 3840x1920 equirect, CPW off, 5-band multiband blend).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU; typed as is, it spawns its ranks itself)

A "step" = --passes (default 20) passes over a batch of F frames (F = --frames, default 96: 6 views x F frames, inputs already resident in HBM; 20 x 96 = 1920 frames =
64 s of 30 fps video per step), each pass issued as --streams (default 3) ms_stitch calls of F/streams = 32 frames on separate HIP streams / contexts.  Weak scaling:
every rank stitches its own F frames per step (frame-parallel, round-robin ownership); with N>1 the finished pano slabs are gathered on rank 0 through the product's
ms_dist layer (RCCL), overlapped with the next pass.  value = N*F*passes*K / max-over-ranks wall time.

This file is the command line and the ONE JSON line; the parts live in benchlib/:
  regions.py      the workload (contexts, resident frames, pass / step / timed region), `verified`, live latency, the instrumented per-kernel pass
  roofline.py     `roofline` (dominant kernel, PMC bytes / measured launch time / 8 TB/s -- physical only), `frame`, `model` (SURVEY 8(d) contract-byte ratios)
  pmc.py          HBM bytes from the hardware counters, collected inside this run (two short rocprofv3 --pmc child passes)
  cpu_baseline.py `verified_vs_oracle` (one full-size frame bit-identical to the CPU oracle) and `cpu_baseline` (the reference's CPU flavour on the host cores)
  others.py       `other_configs`: BASELINE configs[2] (cfg3), configs[4] geometry (cfg5, + its two one-GPU shardings) and the reference's shipped rig as short regions
  dist_run.py     N > 1: preamble, communicator bring-up, watchdog, the compute-only / main / full-gather regions
  shards.py       --view-shards / --col-shards
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))
# the cpu_baseline leg times an OpenMP port: bind its threads to cores (read once, when the OpenMP runtime starts -- hence before any import)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

import torch  # noqa: E402

METRIC = "stitched frames/sec, 6x1080p->4K equirect (ms/frame = 1000/value*n_gpus)"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--passes", type=int, default=None, help="passes over the F-frame batch per step (default 20: a step is 1920 frames, so that the driver's "
                    "short runs still time about a second of GPU work; 1 with --calib / profiling runs)")
    ap.add_argument("--no-live", action="store_true", help="skip the one-frame-per-call latency measurement")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive run of the C++ host pipeline (stitch_app)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--gather-every", type=int, default=0, help="N>1: the main timed region gathers the slabs of every k-th pass (default 0: the live rate of BASELINE configs[3], about 30 batches per second "
                    "and rank; 1: EVERY frame of every rank reaches the sink inside the timed region -- then `value` is the conservative every-frame figure)")
    ap.add_argument("--frames", type=int, default=None, help="frames per pass, split evenly over --streams contexts (default 96 = 3 x 32; cfg3: 32, shipped: 64 on one context; cfg5: 48 = 3 x 16; 1 = live mode; at most 64 per context)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg5", "shipped"],
                    help="cfg2 / cfg3 / cfg5 = BASELINE configs[1] / [2] / [4] geometry; shipped = the configuration the reference ships (defs.h:25-27,51-55,65-66, "
                         "calibration.cpp:100,147-194): cylindrical warper, COMPOSE_MEGAPIX 1.4 (every frame through cuda::resize INSIDE the timed region), "
                         "num_bands by the app's rule, seam-scale gains + Voronoi masks, CPW on with 10x10 meshes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frame sets cycled through the batch (SURVEY 8(d): 8 = 298 MB of source; 96 = every frame of the "
                                                             "default pass distinct, 3.6 GB: nothing of the source survives in the 256 MiB Infinity Cache between passes; 1 = cache-resident A/B)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--frame-source", default=None, choices=["numpy", "device"], help="synthetic frames from synth.frame on the host (numpy; the 1080p default) or the same content model evaluated on the GPU (the 12 x 4K default: a numpy 4K frame costs about a second)")
    ap.add_argument("--no-distinct", action="store_true", help="skip the second short timed region in which every frame of the pass is a distinct frame set (`value_nothing_cached`)")
    ap.add_argument("--no-others", action="store_true", help="skip `other_configs` (the short cfg3 / cfg5 / shipped regions the default cfg2 run at N = 1 appends)")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect the HBM byte counters inside this run (two short rocprofv3 --pmc child passes): roofline.traffic then comes from profiles/traffic_<config>.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # (the child of benchlib/pmc.py: timed region + --calib copy only)
    ap.add_argument("--egress-convert", action="store_true", help="N>1 egress as 8UC3 canvas + ms_bgr_to_i420_batch instead of ms_stitch_i420 (A/B)")
    ap.add_argument("--emulate-gather", action="store_true", help="single GPU: run the per-frame egress conversion of the N>1 path without the collective (host/GPU cost of that leg)")
    ap.add_argument("--gather-format", default="i420", choices=["i420", "bgr"],
                    help="what the sink rank receives: planar I420 of the pano rows (the encoder input of consume(), timed.cpp:308-316; half the bytes) or the 8UC3 rows themselves")
    ap.add_argument("--calib", action="store_true", help="also run 3 known-size streaming copies (PMC calibration, tools/profile_traffic.sh)")
    ap.add_argument("--view-shards", type=int, default=1,
                    help="BASELINE configs[4]: split the VIEWS of every frame over this many ranks (partial int16 accumulators sent to the "
                         "group's sink rank, which finishes the frame); world must be a multiple of it (world 1 = both shards on one GPU)")
    ap.add_argument("--col-shards", type=int, default=1,
                    help="SURVEY 8(e) pano-column sharding: C ranks per group, each compositing one window of panorama columns (+ halo) of the same frames; "
                         "--gpus 1: all C windows on this GPU (cost of the split)")
    ap.add_argument("--recalib-every", type=int, default=60,
                    help="cfg3 / shipped: re-expand new CPW meshes (ms_set_meshes) every this many frames, inside the timed region (BASELINE configs[2]: recalibrate every 60 f); 0 = never")
    ap.add_argument("--join-every", type=int, default=1, help="fork-join the streams around groups of this many passes (1 = every pass)")
    ap.add_argument("--independent-streams", action="store_true", help="do not fork-join the streams around every pass (A/B: 1 %% slower than the joined default)")
    ap.add_argument("--pad-kb", type=int, default=0, help="developer probe: allocate this many KiB of device memory before anything else (shifts the addresses of every later allocation: "
                                                         "placement sensitivity of the kernels, tools/placement_probe.sh)")
    ap.add_argument("--streams", type=int, default=None,
                    help="contexts / HIP streams a step's frames are split over (default 3 x 32 frames: the small coarse-level kernels of one batch overlap the large kernels of another: +13 %% over one stream)")
    args = ap.parse_args()
    # `other_configs` and the in-run PMC pass belong to the DEFAULT launch shape (what the driver runs); a probe that names its own shape (tools/*.sh) gets neither
    if args.frames is not None or args.streams is not None or args.calib or args.emulate_gather or args.view_shards > 1 or args.col_shards > 1 or args.gpus > 1:
        args.no_others = True
        if not args.pmc_child and (args.calib or args.emulate_gather or args.gpus > 1 or args.streams not in (None, 1)):
            args.no_pmc = True
    # ... and neither does a run that is itself being profiled (tools/*.sh wrap bench.py in rocprofv3: no profiler inside a profiler)
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY", "HSA_TOOLS_LIB")):
        args.no_pmc = True
        if not args.pmc_child:
            args.no_others = True
    if args.pmc_child:
        args.no_live = args.no_pcie = args.no_verify = args.no_cpu_baseline = args.no_distinct = args.no_others = args.no_pmc = True
    if args.view_shards > 1 or args.col_shards > 1:          # one context per shard, no stream splitting
        args.streams = 1
        if args.frames is None:
            args.frames = 4 if args.config == "cfg5" else 16
    if args.passes is None:
        args.passes = 1 if (args.calib or args.view_shards > 1 or args.col_shards > 1) else 20
    return args


def spawn_ranks(args):
    """typed as `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU, the launcher the contract names), pass their output through -- rank 0
    prints the one JSON line -- and return their exit code"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def pcie_inclusive(wl):
    """PCIe-inclusive rate: the C++ host pipeline with the reference's thread / queue graph, every source frame uploaded from pinned memory (never `value`)"""
    import subprocess
    cfg = wl.cfg
    app = os.path.join(ROOT, "video-stitcher_amd", "stitch_app")
    if not os.path.exists(app):
        return None
    try:
        cmd = [app, "--views", str(cfg["n"]), "--size", "%dx%d" % (wl.full_w, wl.full_h), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
               "--hfov", str(cfg["hfov_deg"]), "--bands", str(cfg["num_bands"]), "--frames", "1500"] + (["--cpw"] if wl.cpw else []) + (["--reference-calib"] if wl.shipped else [])

        def fps(extra):
            pr = subprocess.run(cmd + extra, capture_output=True, text=True, timeout=180)
            return json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
        pj = fps([])
        pcie = {"value": pj["frames_per_s"], "unit": "frames/s", "frames": pj["frames"],
                "how": "video-stitcher_amd/stitch_app: capture thread -> hipMemcpy2DAsync of all %d views from pinned memory per frame (double-buffered, own stream) -> "
                       "ms_stitch (1 frame) -> consume thread; %.1f MB over PCIe per frame" % (cfg["n"], cfg["n"] * cfg["w"] * cfg["h"] * 3 / 1e6)}
        # the same with the cameras' NV12 uploaded (half the bytes) and cvtColor(YUV2BGR_NV12) on the device -- a per-camera CPU step in the reference (networking.cpp:45-47)
        pcie["nv12_ingest_value"] = fps(["--nv12"])["frames_per_s"]
        if not wl.cpw and not wl.shipped:      # ... and with no conversion pass at all: the warp samples the NV12 planes (ms_stitch_nv12; contexts without CPW / per-frame resize)
            pcie["nv12_direct_value"] = fps(["--nv12-direct"])["frames_per_s"]
        return pcie
    except Exception as e:      # noqa: BLE001 -- the number is optional; never fail the bench line on it
        return {"error": str(e)[:200]}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            sys.exit(spawn_ranks(args))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU: libmsstitch has no CPU fallback"
    # MS_BENCH_SHARE_GPU=1 is a DEBUG mode for 1-GPU boxes: every rank uses cuda:0 and the process group is gloo (pano slabs
    # staged through host memory) -- it exercises the multi-rank control flow, it is not a scaling measurement.
    share = os.environ.get("MS_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo") if share else dist.init_process_group("nccl", device_id=dev)

    import msstitch as ms
    import synth
    from benchlib import cpu_baseline as CB, dist_run, others, pmc, roofline as RL, shards
    from benchlib.regions import DTYPE, Opt, Workload

    _pad = torch.empty(args.pad_kb * 1024, dtype=torch.uint8, device=dev) if args.pad_kb > 0 else None      # noqa: F841 (kept alive for the whole run)
    if args.view_shards > 1 or args.col_shards > 1:
        cfg = dict(synth.CONFIGS["cfg5" if args.config == "cfg5" else "cfg2"])
        fn = shards.run_view_shards if args.view_shards > 1 else shards.run_col_shards
        line = fn(args, cfg, synth.gains(cfg["n"]), rank, world, dev, share, frame_source=args.frame_source or ("device" if args.config == "cfg5" else "numpy"))
        if rank == 0:
            print(json.dumps(line), flush=True)
        return

    opt = Opt(config=args.config, frames=args.frames, streams=args.streams, passes=args.passes, steps=args.steps, warmup=args.warmup, distinct=args.distinct,
              recalib_every=args.recalib_every, join_every=args.join_every, independent_streams=args.independent_streams, gather_format=args.gather_format,
              egress_convert=args.egress_convert, emulate_gather=args.emulate_gather, no_gather=args.no_gather, gather_every=args.gather_every, frame_source=args.frame_source)
    wl = Workload(opt, rank, world, dev, share)
    cfg, F, S, Fs = wl.cfg, wl.F, wl.S, wl.Fs

    reg = dist_run.run_regions(args, wl, rank, world, local_rank, share, dev)
    if "fatal" in reg:      # fail LOUDLY: one JSON line with `failed` / `incomplete`, a non-zero exit code on every rank
        if rank == 0:
            print(json.dumps(reg["fatal"]), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(2)
    elapsed, n_gathered = reg["elapsed"], reg["n_gathered"]
    no_gather, full_gather = reg["no_gather"], reg["full_gather"]
    D, dist_info, state = wl.D, wl.dist_info, wl.state

    if args.calib:
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255); b = torch.empty_like(a)
        for _ in range(3):
            ms.calib_copy(a, b)
        for shape in (0, 1, 2):      # ... and the per-frame kernels' own access shapes over the same GiB (every line touched once: 1 GiB of HBM traffic each)
            for _ in range(2):
                ms.calib_shape(a, shape)
        torch.cuda.synchronize()
        del a, b

    verified, verify_note = (None, None) if args.no_verify else wl.verify()
    live = wl.live() if (not args.no_live and rank == 0) else None
    gathered_check = wl.gathered_check() if os.environ.get("MS_BENCH_CHECK_GATHERED") == "1" else None
    value_d96 = None if (args.no_distinct or args.calib) else wl.distinct_region(max(2, args.steps // 4))
    pcie = pcie_inclusive(wl) if (not args.no_pcie and rank == 0 and world == 1) else None
    kmean, lat = wl.instrumented(max(5, min(50, args.steps * args.passes)))
    rank_ceilings = dist_run.rank_copy_ceilings(ms, world, dev) if world > 1 else None
    ceiling = RL.measure_ceiling(ms, dev) if (rank == 0 and world == 1 and not args.pmc_child) else None
    pmc_run, pmc_why = (None, None)
    if rank == 0 and world == 1 and not args.no_pmc:      # the hardware counters of this workload's launch shape, collected by this very run
        pmc_run, pmc_why = pmc.measure(args.config, Fs, wl.n_distinct, wl.frame_source)

    if rank == 0:
        frames_per_step = F * args.passes
        total_frames = world * frames_per_step * args.steps
        par = "frame-parallel x%d" % world
        if wl.gather:
            par += ", gather of the %s pano rows of %s on rank 0 (%.1f MB/frame) through %s, overlapped with the next pass" % (
                args.gather_format.upper(), "EVERY frame" if state["gather_every"] == 1 else "every %d-th pass (live-rate egress)" % state["gather_every"], wl.slabs[0][0].numel() / 1e6,
                ("ms_dist / " + dist_info["transport"]) if D is not None else "torch.distributed")
        if share:
            par += " [DEBUG: ranks share one GPU, gloo]"
        roof, frame, model = RL.blocks(wl, kmean, elapsed, total_frames / world, ceiling, pmc_run, pmc_why)
        res = {
            "metric": METRIC if world == 1 else dist_run.METRIC_N,
            "value": round(total_frames / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_frame": round(elapsed / args.steps / frames_per_step * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": wl.workload_string(), "distinct_frame_sets": wl.n_distinct, "frames_per_step": frames_per_step, "frames_per_pass": F,
                       "passes_per_step": args.passes, "streams": S, "parallelism": par, "frame_source": wl.frame_source},
            "verified": verified, "verified_how": verify_note,
            "roofline": roof, "ceiling": ceiling, "frame_roofline": frame, "model": model,
            "kernels_ms_per_call": {k: round(v, 5) for k, v in kmean.items()},
            "latency": lat,
        }
        if no_gather is not None:
            res["value_no_gather"] = round(no_gather, 2)
            if state["gather_every"] == 1:
                res["value_full_gather"] = res["value"]          # (--gather-every 1: the main region IS the every-frame gather)
                res["value_definition"] = "EVERY frame of every rank stitched AND delivered to the one sink rank inside the timed region (--gather-every 1; the definition of rounds <= 4)"
            else:
                res["value_live_rate_gather"] = res["value"]     # (the main region IS the live-rate egress)
                res["value_full_gather"] = round(full_gather[0], 2) if full_gather else None
                res["value_definition"] = ("BASELINE configs[3]: every frame stitched frame-parallel; the egress of a LIVE stream (the slabs of ~30 batches per second and rank, every %d-th pass) "
                                           "gathered on rank 0 inside the timed region.  NOT every frame leaves its GPU: the every-frame-to-one-sink rate -- the definition of `value` in rounds "
                                           "<= 4, bound by one GPU's inbound xGMI links -- is `value_full_gather`; compute only is `value_no_gather`" % state["gather_every"])
            res["gather"] = {"gathered_passes": n_gathered, "of_passes": args.steps * args.passes, "every": state["gather_every"],
                             "passes_per_s_and_rank_gathered": round(n_gathered / elapsed, 1),
                             "GBps_into_sink": round(n_gathered * F * (world - 1) * wl.slabs[0][0].numel() / elapsed / 1e9, 2),
                             "full_gather_GBps_into_sink": (round(full_gather[1] * F * (world - 1) * wl.slabs[0][0].numel() / full_gather[2] / 1e9, 2) if full_gather else None)}
        if world > 1:      # what a reader of the first multi-GPU record needs, at the top level
            res["comm_nranks"] = (dist_info or {}).get("comm_nranks")
            res["transport"] = (dist_info or {}).get("transport")
            res["pci_bus_ids"] = (dist_info or {}).get("pci_bus_ids")
            res["librccl_path"] = (dist_info or {}).get("librccl_path")
            res["rank_copy_TBps"] = rank_ceilings
            res["preamble"] = reg.get("preamble")
            res["how_to_read"] = ("value = BASELINE configs[3]: every frame stitched frame-parallel, the egress of a live stream (the slabs of ~30 batches per second and rank) gathered on rank 0 "
                                  "over ms_dist inside the timed region (see value_definition); value_no_gather = compute only; value_full_gather = EVERY frame of every rank "
                                  "into the one sink at benchmark rate (bound by one GPU's inbound xGMI links, ~100 k frames/s of 3.6 MB slabs, not by the compositor)")
        if dist_info is not None:
            res["dist"] = dist_info      # what the communicator itself saw: transport, nranks (RCCL's own count), device ordinals and PCI bus ids of every rank, the librccl file
        if value_d96 is not None:
            res["value_distinct"] = value_d96
            res["value_nothing_cached"] = value_d96.get("value")      # every frame of the pass a distinct frame set: nothing of the source survives in the 256 MiB Infinity Cache
        if gathered_check is not None:
            res["gathered_frames_checked"] = gathered_check
        if live is not None:
            res["live"] = live
        if pcie is not None:
            res["pcie_inclusive_fps"] = pcie
        if world == 1 and not args.no_cpu_baseline:
            res["verified_vs_oracle"] = CB.oracle_check(cfg, wl.gains, wl.comp, wl.frames[0], wl.cpw)
            res["cpu_baseline"] = CB.cpu_baseline(cfg, wl.gains, wl.comp, frames=[synth.frame(wl.full_w, wl.full_h, i, 0) for i in range(cfg["n"])], resize=wl.resize_scale)
        if world == 1 and args.config == "cfg2" and not args.no_others:      # the other single-GPU BASELINE configurations, as short regions behind the headline one
            t_o = time.perf_counter()
            wl.close()
            res["other_configs"] = others.run_all(dev, ceiling, check_oracle=not args.no_cpu_baseline, in_run_pmc=not args.no_pmc)
            res["other_configs"]["seconds"] = round(time.perf_counter() - t_o, 1)
        print(json.dumps(res), flush=True)
    if D is not None:
        D.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
