"""N > 1 (BASELINE configs[3]: a live-rate stream, frame-parallel over G GPUs, the finished equirect frames gathered on one GPU).

Order of a multi-rank run -- everything that can fail on plumbing happens, and is PRINTED, before the first timed region:
  0. preamble (stderr): every rank's device, PCI bus id and HIP ordinal as torch.distributed sees them; two ranks on one bus id without MS_BENCH_SHARE_GPU -> the run
     fails loudly (one JSON line with `failed` / `incomplete`, exit code 2);
  1. the product's own communicator, ms_dist (csrc/dist.cpp: RCCL send / recv over xGMI; the host mailbox when the ranks share a device), under a watchdog:
     `dist: {librccl_path, rccl_version, comm_nranks, pci_bus_ids, transport}` goes to stderr; comm_nranks != N -> fails loudly;
  2. compute only (no collective): `value_no_gather`;
  3. the MAIN region = `value`: exactly K steps, every frame stitched, and the egress of a live stream gathered: the slabs of about 30 batches per second and rank
     travel to the sink (configs[3] asks for ONE 30 fps stream: this is ~100x its bytes and still leaves the links idle); `--gather-every k` (k >= 1) makes the main
     region gather every k-th pass instead (k = 1: the conservative every-frame figure IS `value`);
  4. `value_full_gather`: every frame of every rank delivered to the ONE sink at benchmark rate -- bound by that GPU's inbound xGMI links
     (7 x ~55 GB/s ~ 100 k frames/s of 3.6 MB slabs), not by the compositor; reported beside `value`, never instead of it.
A watchdog (MS_BENCH_WATCHDOG_S, default 120 s PER STAGE) turns a hanging transport into a line with `value: null`, `failed: true`, the numbers that exist in side
fields, and a NON-ZERO exit code on every rank (a hang must not look like a successful run).

torch.distributed stays for what the bench contract prescribes around the timed region (barrier, max over ranks) and to hand the communicator's id to the ranks."""
import json
import os
import sys
import threading
import time

import torch

from .regions import DTYPE

METRIC_N = "stitched frames/sec, 6x1080p->4K equirect, frame-parallel (ms/frame = 1000/value*n_gpus)"


def watchdog_seconds():
    return float(os.environ.get("MS_BENCH_WATCHDOG_S", "120"))


def fail_line(args, world, note, dist_info=None, extra=None):
    d = {"metric": METRIC_N, "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
         "config": {"workload": "%s, frame-parallel x%d" % (args.config, world)}, "failed": True, "incomplete": note, "dist": dist_info}
    if extra:
        d.update(extra)
    return d


def duplicate_bus_ids(ids):
    """True when two ranks report the same PCI bus id (= the same physical GPU); unknown ids ("?") never count"""
    known = [i for i in ids if i and i != "?"]
    return len(set(known)) != len(known)


def plumbing_verdict(pre, info, world, share):
    """the fail-loudly rules of a multi-rank run, as one pure function (tests/test_dist_frames.py): a note when the run must stop before its first timed region, else None"""
    if pre is not None and pre.get("duplicate_bus_ids") and not share:
        return "two ranks drive the same GPU (PCI bus ids %s) and MS_BENCH_SHARE_GPU is not set: not a multi-GPU run" % [r["pci_bus_id"] for r in pre["ranks"]]
    if info is not None and info.get("transport") == "rccl" and info.get("comm_nranks") != world:
        return "RCCL's communicator counts %s ranks, the launcher started %d" % (info.get("comm_nranks"), world)
    if info is not None and info.get("transport") == "rccl" and not share and duplicate_bus_ids(info.get("pci_bus_ids") or []):
        return "RCCL's own all-gather shows two ranks on one GPU (PCI bus ids %s)" % info.get("pci_bus_ids")
    return None


def preamble(rank, world, local_rank, share, dev):
    """what torch.distributed's own process group sees, before the product's communicator exists; duplicate bus ids are fatal unless the ranks share a GPU on purpose"""
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(dev)
    try:
        bus = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    except AttributeError:
        bus = "?"
    mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(dev), "pci_bus_id": bus, "visible_devices": torch.cuda.device_count(),
            "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES")}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    pre = {"world": world, "backend": dist.get_backend(), "ranks": allr, "duplicate_bus_ids": duplicate_bus_ids([r["pci_bus_id"] for r in allr]), "share_gpu_debug_mode": share}
    if rank == 0:
        print("bench preamble: " + json.dumps(pre), file=sys.stderr, flush=True)
    return pre


def bring_up(wl, rank, world, local_rank, share, dev):
    """ms_dist communicator, decided by ALL ranks together: rank 0's id (or its failure) is broadcast, and after the collective creation the ranks agree (MIN over a
    flag) on whether every one of them has a communicator -- a rank that fell back alone would wait for ever in the first gather."""
    import torch.distributed as dist
    import msdist
    # MS_BENCH_RCCL_LIB: ms_dist_set_rccl_library (tests put the loopback implementation of tests/fake_rccl.cpp there: the RCCL branch of csrc/dist.cpp
    # with N > 1 ranks on a one-GPU box); with it MS_BENCH_SHARE_GPU keeps gloo for torch.distributed but ms_dist takes the RCCL transport
    transport = msdist.HOST if share else msdist.RCCL
    if os.environ.get("MS_BENCH_RCCL_LIB"):
        msdist.set_rccl_library(os.environ["MS_BENCH_RCCL_LIB"])
        transport = msdist.RCCL
    D, info, why = None, None, None
    box = [None]
    if rank == 0:
        try:
            box = [msdist.unique_id(world, transport)]
        except Exception as e:      # noqa: BLE001
            why = str(e)[:160]
    dist.broadcast_object_list(box, src=0)
    if box[0] is not None:
        try:
            D = msdist.Dist(rank, world, box[0], device=local_rank)
            info = D.info()
        except Exception as e:      # noqa: BLE001
            D, why = None, str(e)[:160]
    flag = torch.tensor([1 if D is not None else 0], dtype=torch.int32, device="cpu" if share else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:      # the bench line must survive a transport that does not come up: fall back to torch.distributed.gather, and SAY so
        if D is not None:
            D.close()
        D, info = None, {"transport": "torch.distributed (ms_dist did not come up on every rank: %s)" % (why or "another rank failed"), "comm_nranks": dist.get_world_size(),
                         "note": "comm_nranks here is torch.distributed's own world size"}
    wl.D, wl.dist_info = D, info
    if rank == 0:
        print("dist: " + json.dumps(info), file=sys.stderr, flush=True)
    return info


class Watchdog:
    def __init__(self, args, wl, rank, world):
        self.args, self.wl, self.rank, self.world = args, wl, rank, world
        self.deadline, self.stage = None, None
        self.partial = {}
        self._t = None

    def arm(self, stage):
        self.stage, self.deadline = stage, time.time() + watchdog_seconds()
        if self._t is None:
            self._t = threading.Thread(target=self._watch, daemon=True)
            self._t.start()

    def disarm(self):
        self.deadline = None

    def _watch(self):
        while True:
            time.sleep(1.0)
            dl = self.deadline
            if dl is not None and time.time() > dl:
                if self.rank == 0:
                    note = ("the multi-GPU transport did not finish '%s' within %.0f s: no `value`; the partial numbers are in the side fields; every rank was ended by the "
                            "bench's watchdog with exit code 3" % (self.stage, watchdog_seconds()))
                    print(json.dumps(fail_line(self.args, self.world, note, self.wl.dist_info if self.wl is not None else None, dict(self.partial))), flush=True)
                os._exit(3)


def run_regions(args, wl, rank, world, local_rank, share, dev):
    """-> dict with elapsed / n_gathered of the main region and the side regions; or {"fatal": line} when the run must fail loudly"""
    import torch.distributed as dist
    F, passes = wl.F, args.passes
    out = {"no_gather": None, "full_gather": None, "live_rate": None, "preamble": None}
    if not wl.gather:
        out["elapsed"], out["n_gathered"] = wl.timed_region(args.steps, args.warmup)
        return out
    wd = Watchdog(args, wl, rank, world)
    wd.arm("preamble (torch.distributed all-gather of the ranks' devices)")
    pre = out["preamble"] = preamble(rank, world, local_rank, share, dev)
    note = plumbing_verdict(pre, None, world, share)
    if note:
        wd.disarm()
        return {"fatal": fail_line(args, world, note, None, {"preamble": pre})}
    wd.arm("communicator bring-up (ms_dist)")
    info = bring_up(wl, rank, world, local_rank, share, dev)
    note = plumbing_verdict(pre, info if wl.D is not None else None, world, share)
    if note:
        wd.disarm()
        if wl.D is not None:
            wl.D.close(); wl.D = None
        return {"fatal": fail_line(args, world, note, info, {"preamble": pre})}
    st = wl.state
    wd.arm("compute-only region")
    st["gather_on"] = False
    half = max(1, args.steps // 2)
    el_ng, _ = wl.timed_region(half, args.warmup)
    out["no_gather"] = world * F * passes * half / el_ng
    wd.partial["value_no_gather"] = round(out["no_gather"], 2)
    st["gather_on"] = True
    wd.arm("main region")
    if args.gather_every <= 0:      # the live rate: about 30 batches per second and rank
        st["gather_every"] = max(1, int(round(out["no_gather"] / world / 30.0 / F)))
    out["elapsed"], out["n_gathered"] = wl.timed_region(args.steps, args.warmup)
    main_every = st["gather_every"]
    wd.partial["value_main_region"] = round(world * F * passes * args.steps / out["elapsed"], 2)
    wd.partial["main_region_gathers_every"] = main_every
    if main_every != 1:
        wd.arm("full gather (every frame to one sink)")
        st["gather_every"] = 1
        el_fg, n_fg = wl.timed_region(half, max(1, args.warmup // 2))
        out["full_gather"] = (world * F * passes * half / el_fg, n_fg, el_fg)
        st["gather_every"] = main_every
    wd.disarm()
    return out


def rank_copy_ceilings(ms, world, dev):
    """every rank's own streaming-copy ceiling (boxes and GPUs differ by +-8 %): gathered so that the line explains an uneven scaling curve"""
    import torch.distributed as dist
    try:
        n_c = 1 << 28
        ca = torch.empty(n_c, dtype=torch.uint8, device=dev).random_(0, 255); cb = torch.empty_like(ca)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for it in range(5):
            e0.record(); ms.calib_copy(ca, cb); e1.record(); e1.synchronize()
            if it >= 1:
                best = min(best, e0.elapsed_time(e1))
        mine = round(2.0 * n_c / (best * 1e-3) / 1e12, 3)
        del ca, cb
    except Exception:      # noqa: BLE001
        mine = None
    allc = [None] * world
    dist.all_gather_object(allc, mine)
    return allc
