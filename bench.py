#!/usr/bin/env python3
"""bench.py -- stitched frames/s of the MI355X compositor on BASELINE.json configs[1]
(6x1080p synthetic views -> 3840x1920 equirect, CPW off, 5-band multiband blend).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = --passes (default 20) passes over a batch of F frames (F = --frames, default 96: 6 views x F frames, inputs already
resident in HBM; 20 x 96 = 1920 frames = 64 s of 30 fps video per step), each pass issued as --streams (default 3) ms_stitch calls of
F/streams = 32 frames on separate HIP streams / contexts.  Weak scaling: every rank stitches its own F frames per step (frame-parallel,
round-robin ownership); with N>1 the finished pano slabs are gathered on rank 0 over RCCL, overlapped with
the next step.  value = N*F*K / max-over-ranks wall time.

Also printed on the same JSON line:
  roofline     -- dominant kernel: SURVEY 8(d) algorithmic bytes per launch / mean launch duration
                  (hipEvents on the launch stream, instrumented pass right after the timed region), peak 8 TB/s
  cpu_baseline -- the CPU oracle (a port of the reference's kernel arithmetic) on a bounded sample, rank 0 / N=1
  verified     -- after the timed region, frames of every batch are re-stitched one at a time on a separate one-frame context and
                  must equal the batched outputs byte for byte; verified_vs_oracle -- one full-size frame against the CPU oracle, bit for bit
  live         -- one frame per ms_stitch call, synchronised after each call: median / p95 latency per frame (the reference's shape)
  pcie_inclusive_fps -- the C++ host pipeline (video-stitcher_amd/stitch_app: pinned H2D of all six views per frame + stitch + consume)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))
# the cpu_baseline leg times an OpenMP port: bind its threads to cores (read once, when the OpenMP runtime starts -- hence before any import)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def kernel_bytes(comp, cfg, n_frames, cpw):
    """Per-launch algorithmic bytes, SURVEY 8(d) accounting applied to the exact level sizes:
    source read once; every Gaussian level written once (6 B/px, 16SC3) and read twice (next-level reduce,
    Laplacian+accumulate); weights read once (4 B); dst Laplacian read-modify-write per view rect (12 B);
    collapse reads level + coarser level and writes level (6 B each); output 8UC3 canvas written once."""
    nb = comp.pano_geom().num_bands
    P = []
    A = 0
    for i in range(cfg["n"]):
        g = comp.view_geom(i)
        P.append((g.roi.width + g.left + g.right) * (g.roi.height + g.top + g.bottom))
        A += g.roi.width * g.roi.height
    pg = comp.pano_geom()
    Q = pg.dst_roi.width * pg.dst_roi.height
    sumP = float(sum(P))
    kb = {}
    kb["k_warp"] = cfg["n"] * 3.0 * cfg["w"] * cfg["h"] + 6.0 * sumP
    if cpw:
        # "+ second gather (3 B read + 3 B write) x A".  The FIRST remap (timed as `k_remap_gain`: projection remap + gain into the 8UC3 stage image) reads the
        # source frames and writes 3 B per warped pixel; the SECOND (timed as `k_warp`: the mesh remap writing level 0) reads those 3 B and writes the 16SC3 level 0.  Same
        # frame total as before; round 4 priced the whole second gather against the first kernel.
        kb["k_remap_gain"] = cfg["n"] * 3.0 * cfg["w"] * cfg["h"] + 3.0 * A
        kb["k_warp"] = 3.0 * A + 6.0 * sumP
    for l in range(nb):
        kb["k_down_l%d" % l] = 6.0 * sumP / 4 ** l + 6.0 * sumP / 4 ** (l + 1)
    for l in range(nb + 1):
        b = (6.0 + 4.0 + 12.0) * sumP / 4 ** l
        if l < nb:
            b += (12.0 + 1.5) * Q / 4 ** l
        if l == 0:
            b += 3.0 * cfg["out_w"] * cfg["out_h"]
        kb["k_blend_l%d" % l] = b
    # fused coarse-level launches cover several of the per-level entries above
    kb["k_down_tail"] = sum(v for k, v in kb.items() if k.startswith("k_down_l") and int(k[8:]) >= 3)
    kb["k_blend_tail"] = sum(v for k, v in kb.items() if k.startswith("k_blend_l") and int(k[9:]) >= 3)
    return {k: v * n_frames for k, v in kb.items()}, sumP, Q, A


def csrc_sha16():
    """hash of the product's kernel sources: a PMC traffic summary collected on another state of csrc/ is flagged stale in the bench line"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "video-stitcher_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp", ".inc")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def oracle_check(cfg, gains, comp, frame_dev, cpw):
    """`verified` only says the batched path equals the one-frame path of the SAME library.  This compares one full-size frame of this run's workload with
    the CPU oracle (the restatement of the reference's CUDA arithmetic, oracle/ms_oracle_*.c) fed the context's maps, masks [and meshes]: the 16SC3
    panorama and the result mask must be bit-identical (the criterion of tests/test_compositor_gpu.py::test_full_size_config2_matches_oracle)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    t0 = time.perf_counter()
    pg = comp.pano_geom()
    out16 = torch.zeros((pg.dst_roi_final.height, pg.dst_roi_final.width, 3), dtype=torch.int16, device=frame_dev[0].device)
    comp.stitch([frame_dev], out16s=[out16])
    torch.cuda.synchronize()
    rois = [comp.view_geom(i).roi.tuple() for i in range(cfg["n"])]
    b = O.Blender([r[:2] for r in rois], [r[2:] for r in rois], pg.num_bands)
    O.set_num_threads(min(32, os.cpu_count() or 1))
    for i in range(cfg["n"]):
        b.init_view(i, comp.mask(i).cpu().numpy())
    for i in range(cfg["n"]):
        xm, ym = [t.cpu().numpy() for t in comp.maps(i)]
        mesh = [t.cpu().numpy() for t in comp.mesh_maps(i)] if cpw else [None, None]
        b.stitch_online(i, frame_dev[i].cpu().numpy(), xm, ym, gains[i], mesh[0], mesh[1])
    ref16, refmask = b.blend()
    b.close()
    same = bool(np.array_equal(out16.cpu().numpy(), ref16)) and bool(np.array_equal(comp.result_mask().cpu().numpy(), refmask))
    return {"bit_identical": same, "what": "one %dx%d frame of this workload (16SC3 panorama + result mask) against the CPU oracle given the context's maps, masks%s"
            % (pg.dst_roi_final.width, pg.dst_roi_final.height, " and meshes" if cpw else ""), "seconds": round(time.perf_counter() - t0, 2)}


def _physical_cores_of_one_socket():
    """Number of physical cores of the socket CPU 0 sits on (sysfs topology of every online CPU -- NOT this thread's affinity mask: with OMP_PROC_BIND the
    OpenMP runtime has already bound the initial thread to its first place, one core); all CPUs / 2 if the topology is unreadable."""
    ncpu = os.cpu_count() or 1
    try:
        pkg0 = int(open("/sys/devices/system/cpu/cpu0/topology/physical_package_id").read())
        cores = set()
        for c in range(ncpu):
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            if os.path.exists(base) and int(open(base + "physical_package_id").read()) == pkg0:
                cores.add(int(open(base + "core_id").read()))
        return max(1, len(cores))
    except (OSError, ValueError):
        return max(1, ncpu // 2)


def cpu_baseline(cfg, gains, comp, frames=None, resize=None, budget_s=12.0):
    """Time the reference's CPU pipeline as the oracle restates it (oracle/ms_oracle_cpu.c + ms_oracle_prims.c: "port"): per view [cv::resize by
    compose_scale,] cv::remap in its fixed-point CPU arithmetic -> convertTo(gain) -> convertTo(16S) -> CPU MultiBandBlender::feed (Laplacian pyramid
    with cv::pyrDown / pyrUp's (x + 128) >> 8 / (x + 32) >> 6 rounding, weight pyramid rebuilt on every call: blenders.cpp:585-696), then blend
    (:832-851).  Threads are PINNED: OMP_PLACES=cores / OMP_PROC_BIND=close (set at the top of this file, before the OpenMP runtime starts) put thread i on
    its own physical core next to the initial thread's, so up to the core count of one socket no two threads share a core and none crosses the socket;
    the thread count is the fastest median of 5 runs among 1 / 8 / 16 / 32 / 64 (<= the socket's cores), and the spread of those 5 runs is reported."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    import synth
    ncpu, nphys = os.cpu_count() or 1, _physical_cores_of_one_socket()
    rois = [comp.view_geom(i).roi.tuple() for i in range(cfg["n"])]
    b = O.Blender([r[:2] for r in rois], [r[2:] for r in rois], comp.pano_geom().num_bands, cpu_flavour=True)
    maps = []
    for i in range(cfg["n"]):
        b.init_view(i, comp.mask(i).cpu().numpy())
        xm, ym = comp.maps(i)
        maps.append((xm.cpu().numpy(), ym.cpu().numpy()))
    if frames is None:
        frames = [synth.frame(cfg["w"], cfg["h"], i, 0) for i in range(cfg["n"])]

    def one():
        for i in range(cfg["n"]):
            f = O.resize_linear_8u(frames[i], fx=resize, fy=resize) if resize else frames[i]      # timed.cpp:75-85 on the CPU
            b.stitch_online_cpu(i, f, maps[i][0], maps[i][1], gains[i])
        b.blend()

    def timed(th, reps=5):
        O.set_num_threads(th)
        ts = []
        for _ in range(reps):
            t1 = time.perf_counter(); one(); ts.append(time.perf_counter() - t1)
        ts.sort()
        return ts[len(ts) // 2], ts[0], ts[-1]
    O.set_num_threads(1)
    one()                                   # warm-up (page-in)
    one_thread, one_lo, one_hi = timed(1, 3)
    best, cores, spread = one_thread, 1, (one_lo, one_hi)
    tried = {1: round(1.0 / one_thread, 2)}
    unstable = {}
    for th in (8, 16, 32, 64):
        if th > nphys:
            break
        med, lo, hi = timed(th, 5)
        tried[th] = round(1.0 / med, 2)
        if hi > 1.5 * lo:                   # a thread count whose five runs spread by more than 1.5x does not sustain its median (seen at 32 threads: 13 .. 37 frames/s): not a baseline
            unstable[th] = [round(1.0 / hi, 2), round(1.0 / lo, 2)]
            continue
        if med < best:
            best, cores, spread = med, th, (lo, hi)
    O.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        one(); n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    fps_all = n / el
    b.close()
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    pg = comp.pano_geom()
    return {"value": round(fps_all, 3), "unit": "frames/s", "cores": cores, "kind": "port", "cpu": model, "host_cpus": ncpu,
            "physical_cores_of_one_socket": nphys, "pinning": "OMP_PROC_BIND=%s OMP_PLACES=%s: one thread per physical core, consecutive cores of the initial thread's socket"
            % (os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES")),
            "fps_by_threads_median_of_5": tried, "thread_counts_rejected_as_unstable_min_max_fps": unstable, "spread_fps_of_the_5_runs_at_best": [round(1.0 / spread[1], 2), round(1.0 / spread[0], 2)],
            "flavour": "the reference's CPU path: %scv::remap fixed-point + CPU MultiBandBlender feed/blend ((x+128)>>8 pyramids), restated in oracle/" % ("cv::resize + " if resize else ""),
            "sample": "%d frames (%dx%dx%d -> %dx%d pano ROI, %d bands) in %.1f s with %d OpenMP threads; 1 thread: %.0f ms/frame"
                      % (n, cfg["n"], frames[0].shape[1], frames[0].shape[0], pg.dst_roi_final.width, pg.dst_roi_final.height, pg.num_bands, el, cores, one_thread * 1e3),
            "one_thread_fps": round(1.0 / one_thread, 3)}


def run_view_shards(args, cfg, gains, rank, world, dev, share):
    """SURVEY 8(e) view sharding: ranks form groups of V; rank k of a group owns views [k*N/V, (k+1)*N/V), builds the partial dst
    Laplacian pyramid of its views for F frames (ms_stitch_partial) and sends it to the group's first rank, which adds the partials,
    normalises, collapses and writes the canvases (ms_stitch_finish).  Groups are frame-parallel.  int16 has no RCCL reduction:
    point-to-point send/recv + the add inside the finish kernels."""
    import torch.distributed as dist
    import msstitch as ms
    import synth
    V, F, N = args.view_shards, args.frames, cfg["n"]
    assert world == 1 or world % V == 0, "--view-shards must divide the number of ranks"
    local = world == 1                    # both shards on this GPU, no transfer: measures the compute cost of the split
    group, k_own = (0, None) if local else (rank // V, rank % V)
    sink = group * V

    def make(shards, idx):
        c = ms.Compositor(N, (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"],
                          out_size=(cfg["out_w"], cfg["out_h"]), max_frames=F, shards=shards, shard_index=idx)
        for i in range(N):
            K, R = synth.camera(N, cfg["w"], cfg["h"], cfg["hfov_deg"], i)
            c.set_camera(i, K, R); c.set_gain(i, gains[i])
        c.build_maps(); c.build_masks(1); c.init_blender()
        return c
    mine = list(range(V)) if local else [k_own]
    comps = {k: make(V, k) for k in mine}
    pool = [[torch.from_numpy(synth.frame(cfg["w"], cfg["h"], i, t)).to(dev) if any(k * N // V <= i < (k + 1) * N // V for k in mine) else None
             for i in range(N)] for t in range(4)]
    frames = [pool[(group + j) % 4] for j in range(F)]
    c0 = comps[mine[0]]
    pel = F * c0.partial_bytes() // 2
    parts = {k: [torch.zeros(pel, dtype=torch.int16, device=dev) for _ in range(2)] for k in mine}
    is_sink = local or rank == sink
    if is_sink and not local:
        for k in range(1, V):
            parts[k] = [torch.zeros(pel, dtype=torch.int16, device=dev) for _ in range(2)]
    outs = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=dev) for _ in range(F)] if is_sink else None
    pending = [[], []]

    def xfer(t, peer, send):
        if share:      # gloo debug mode: stage through host memory
            if send:
                torch.cuda.synchronize(); dist.send(t.cpu(), peer)
            else:
                h = torch.empty(t.shape, dtype=t.dtype); dist.recv(h, peer); t.copy_(h)
            return None
        return dist.isend(t, peer) if send else dist.irecv(t, peer)

    def step(s):
        b = s & 1
        for w in pending[b]:
            w.wait()
        pending[b] = []
        for k in mine:
            comps[k].stitch_partial(frames, parts[k][b])
        if not local:
            if is_sink:
                ws = [xfer(parts[k][b], sink + k, False) for k in range(1, V)]
                for w in ws:
                    if w is not None:
                        w.wait()
            else:
                w = xfer(parts[k_own][b], sink, True)
                if w is not None:
                    pending[b].append(w)
        if is_sink:
            c0.stitch_finish(F, [parts[k][b] for k in range(V)], out8u=outs)

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
    for b in range(2):
        for w in pending[b]:
            w.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ok = None
    if is_sink:      # same frames through an unsharded context on the sink: the split (and the transfer) must not change a single byte
        full = make(1, 0)
        all_pool = [[torch.from_numpy(synth.frame(cfg["w"], cfg["h"], i, t)).to(dev) for i in range(N)] for t in range(4)]
        all_frames = [all_pool[(group + j) % 4] for j in range(F)]
        want = [torch.zeros_like(o) for o in outs]
        full.stitch(all_frames, out8u=want)
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) for a, b in zip(outs, want))
        full.close()
    if world > 1:    # every group's sink must agree
        flag = torch.tensor([1 if (ok is None or ok) else 0], dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if rank == 0:
        groups = 1 if local else world // V
        total = groups * F * args.steps
        print(json.dumps({
            "metric": "stitched frames/sec, view-sharded (%s)" % args.config, "value": round(total / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 in / int16+fp32 pyramid arithmetic",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d views -> %dx%d equirect, %d bands; views split over %d shards%s, %d frames per step per group, "
                                   "%d frame-parallel group(s); partial = %.1f MB/frame/shard"
                                   % (args.config, N, cfg["w"], cfg["h"], cfg["out_w"], cfg["out_h"], cfg["num_bands"], V,
                                      " on ONE GPU (no transfer)" if local else (" [DEBUG gloo, shared GPU]" if share else " (RCCL send/recv to the sink)"),
                                      F, groups, c0.partial_bytes() / 1e6)},
            "equals_unsharded": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_col_shards(args, cfg, gains, rank, world, dev, share):
    """SURVEY 8(e) pano-column sharding: ranks form groups of C; rank k of a group composites the panorama columns of window k (work lists cut down to the
    window plus its halo, views that do not reach it never uploaded) for the SAME F frames and sends its column slab of the canvases to the group's
    first rank.  No partial sums cross the link, only finished pixels.  Groups are frame-parallel.  world == 1: all C windows on this GPU, no transfer
    (the compute cost of the split = the recomputed halo + the coarse tails every shard runs in full)."""
    import torch.distributed as dist
    import msstitch as ms
    import synth
    Cn, F, N = args.col_shards, args.frames, cfg["n"]
    assert world == 1 or world % Cn == 0, "--col-shards must divide the number of ranks"
    local = world == 1
    group, k_own = (0, None) if local else (rank // Cn, rank % Cn)
    sink = group * Cn

    def make(shards, idx):
        c = ms.Compositor(N, (cfg["w"], cfg["h"]), ms.PROJ_SPHERICAL, synth.warp_scale(cfg["out_w"]), num_bands=cfg["num_bands"],
                          out_size=(cfg["out_w"], cfg["out_h"]), max_frames=F, col_shards=shards, col_shard_index=idx)
        for i in range(N):
            K, R = synth.camera(N, cfg["w"], cfg["h"], cfg["hfov_deg"], i)
            c.set_camera(i, K, R); c.set_gain(i, gains[i])
        c.build_maps(); c.build_masks(1); c.init_blender()
        return c
    mine = list(range(Cn)) if local else [k_own]
    comps = {k: make(Cn, k) for k in mine}
    need = {k: comps[k].needed_views() for k in mine}
    any_need = 0
    for k in mine:
        any_need |= need[k]
    pool = [[torch.from_numpy(synth.frame(cfg["w"], cfg["h"], i, t)).to(dev) if (any_need >> i) & 1 else None for i in range(N)] for t in range(4)]
    frames = {k: [[fr[i] if (need[k] >> i) & 1 else None for i in range(N)] for fr in [pool[(group + j) % 4] for j in range(F)]] for k in mine}
    pg = comps[mine[0]].pano_geom()
    r0, r1 = max(pg.canvas_y, 0), min(pg.canvas_y + pg.dst_roi_final.height, cfg["out_h"])
    is_sink = local or rank == sink
    # a shard composites whole tiles: what it writes outside its window is unspecified, so every shard has its own canvases and only the
    # window's slab is copied (local) or sent (ranks) into the sink's; the sink's own shard writes into the final canvases directly
    canvas = torch.zeros((F, cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=dev)      # one tensor: a window of all F frames moves in one copy
    canvas_k = {k: (canvas if k == mine[0] and is_sink else torch.zeros_like(canvas)) for k in mine}
    outs = [canvas[f] for f in range(F)]
    run = {k: comps[k].prepared(frames[k], out8u=[canvas_k[k][f] for f in range(F)]) for k in mine}
    # windows of every shard (the sink needs the others' to place their slabs): boundaries are a pure function of the panorama width
    fw = pg.dst_roi_final.width
    bound = lambda i: 0 if i <= 0 else (fw if i >= Cn else (i * fw // Cn) // 16 * 16)
    win = [(bound(k), bound(k + 1)) for k in range(Cn)]
    for k in mine:
        assert comps[k].col_window() == win[k]
    slab_shape = lambda k: (F, r1 - r0, win[k][1] - win[k][0], 3)
    sbuf = {k: [torch.empty(slab_shape(k), dtype=torch.uint8, device=dev) for _ in range(2)] for k in (range(1, Cn) if (is_sink and not local) else ([] if local or is_sink else [k_own]))}
    pending = [[], []]

    def xfer(t, peer, send):
        if share:      # gloo debug mode: stage through host memory
            if send:
                torch.cuda.synchronize(); dist.send(t.cpu(), peer)
            else:
                h = torch.empty(t.shape, dtype=t.dtype); dist.recv(h, peer); t.copy_(h)
            return None
        return dist.isend(t, peer) if send else dist.irecv(t, peer)

    def step(s):
        b = s & 1
        for w in pending[b]:
            w.wait()
        pending[b] = []
        for k in mine:
            run[k]()
        if local:
            for k in mine[1:]:
                cb, ce = win[k][0] + pg.canvas_x, win[k][1] + pg.canvas_x
                canvas[:, r0:r1, cb:ce].copy_(canvas_k[k][:, r0:r1, cb:ce])
            return
        if is_sink:
            ws = [xfer(sbuf[k][b], sink + k, False) for k in range(1, Cn)]
            for k, w in zip(range(1, Cn), ws):
                if w is not None:
                    w.wait()
                cb, ce = win[k][0] + pg.canvas_x, win[k][1] + pg.canvas_x
                canvas[:, r0:r1, cb:ce].copy_(sbuf[k][b])
        else:
            cb, ce = win[k_own][0] + pg.canvas_x, win[k_own][1] + pg.canvas_x
            sbuf[k_own][b].copy_(canvas_k[k_own][:, r0:r1, cb:ce])
            w = xfer(sbuf[k_own][b], sink, True)
            if w is not None:
                pending[b].append(w)

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
    for b in range(2):
        for w in pending[b]:
            w.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ok, base_fps = None, None
    if is_sink:      # same frames through an unsharded context on the sink: the split (and the transfer) must not change a single byte
        full = make(1, 0)
        all_pool = [[torch.from_numpy(synth.frame(cfg["w"], cfg["h"], i, t)).to(dev) for i in range(N)] for t in range(4)]
        all_frames = [all_pool[(group + j) % 4] for j in range(F)]
        want = [torch.zeros_like(o) for o in outs]
        frun = full.prepared(all_frames, out8u=want)
        frun(); torch.cuda.synchronize()
        ok = all(torch.equal(a[r0:r1], b[r0:r1]) for a, b in zip(outs, want))
        t1 = time.perf_counter()
        for _ in range(args.steps):
            frun()
        torch.cuda.synchronize()
        base_fps = F * args.steps / (time.perf_counter() - t1)
        full.close()
    if world > 1:    # every group's sink must agree
        flag = torch.tensor([1 if (ok is None or ok) else 0], dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if rank == 0:
        groups = 1 if local else world // Cn
        total = groups * F * args.steps
        slab_mb = [(r1 - r0) * (e - b) * 3 / 1e6 for b, e in win]
        print(json.dumps({
            "metric": "stitched frames/sec, pano-column-sharded (%s)" % args.config, "value": round(total / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if not local else "weak", "vs_baseline": None, "dtype": "u8 in / int16+fp32 pyramid arithmetic",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d views -> %dx%d equirect, %d bands; panorama columns split over %d shards%s, %d frames per step per group, "
                                   "%d frame-parallel group(s)"
                                   % (args.config, N, cfg["w"], cfg["h"], cfg["out_w"], cfg["out_h"], cfg["num_bands"], Cn,
                                      " on ONE GPU (no transfer)" if local else (" [DEBUG gloo, shared GPU]" if share else " (RCCL send/recv of column slabs to the sink)"),
                                      F, groups),
                       "windows": win, "views_read_per_shard": [bin(need[k]).count("1") for k in mine], "views": N,
                       "slab_MB_per_frame_per_shard": [round(x, 2) for x in slab_mb]},
            "unsharded_fps_same_gpu": round(base_fps, 2) if base_fps else None,
            "equals_unsharded": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--passes", type=int, default=None, help="passes over the F-frame batch per step (default 20: a step is 960 frames, so that the driver's "
                    "short runs still time about a second of GPU work; 1 with --calib / profiling runs)")
    ap.add_argument("--no-live", action="store_true", help="skip the one-frame-per-call latency measurement")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive run of the C++ host pipeline (stitch_app)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--gather-every", type=int, default=0, help="N>1: the main timed region gathers the slabs of every k-th pass (default 0: the live rate, about 30 batches per second and rank; 1: EVERY frame of every rank reaches "
                    "the sink inside the timed region -- the conservative `value`; `value_live_rate_gather` and `value_no_gather` are measured beside it)")
    ap.add_argument("--frames", type=int, default=None, help="frames per pass, split evenly over --streams contexts (default 96 = 3 x 32; cfg3 / shipped: 32 on one context; cfg5: 48 = 3 x 16; 1 = live mode)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg5", "shipped"],
                    help="cfg2 / cfg3 / cfg5 = BASELINE configs[1] / [2] / [4] geometry; shipped = the configuration the reference ships (defs.h:25-27,51-55,65-66, "
                         "calibration.cpp:100,147-194): cylindrical warper, COMPOSE_MEGAPIX 1.4 (every frame through cuda::resize INSIDE the timed region), "
                         "num_bands by the app's rule, seam-scale gains + Voronoi masks, CPW on with 10x10 meshes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frame sets cycled through the batch (SURVEY 8(d): 8 = 298 MB of source; 96 = every frame of the "
                                                             "default pass distinct, 3.6 GB: nothing of the source survives in the 256 MiB Infinity Cache between passes; 1 = cache-resident A/B)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-distinct", action="store_true", help="skip the second short timed region in which every frame of the pass is a distinct frame set (`value_distinct`)")
    ap.add_argument("--egress-convert", action="store_true", help="N>1 egress as 8UC3 canvas + ms_bgr_to_i420_batch instead of ms_stitch_i420 (A/B)")
    ap.add_argument("--emulate-gather", action="store_true", help="single GPU: run the per-frame egress conversion of the N>1 path without the collective (host/GPU cost of that leg)")
    ap.add_argument("--gather-format", default="i420", choices=["i420", "bgr"],
                    help="what the sink rank receives: planar I420 of the pano rows (the encoder input of consume(), timed.cpp:308-316; "
                         "half the bytes) or the 8UC3 rows themselves")
    ap.add_argument("--calib", action="store_true", help="also run 3 known-size streaming copies (PMC calibration, tools/profile_traffic.sh)")
    ap.add_argument("--view-shards", type=int, default=1,
                    help="BASELINE configs[4]: split the VIEWS of every frame over this many ranks (partial int16 accumulators sent to the "
                         "group's sink rank, which finishes the frame); world must be a multiple of it (world 1 = both shards on one GPU)")
    ap.add_argument("--col-shards", type=int, default=1,
                    help="SURVEY 8(e) pano-column sharding: C ranks per group, each compositing one window of panorama columns (+ halo) of the same frames; "
                         "--gpus 1: all C windows on this GPU (cost of the split)")
    ap.add_argument("--recalib-every", type=int, default=60,
                    help="cfg3 only: re-expand new CPW meshes (ms_set_mesh x views) every this many frames, inside the timed region "
                         "(BASELINE configs[2]: recalibrate every 60 f); 0 = never")
    ap.add_argument("--join-every", type=int, default=1, help="fork-join the streams around groups of this many passes (1 = every pass)")
    ap.add_argument("--independent-streams", action="store_true", help="do not fork-join the streams around every pass (A/B: 1 % slower than the joined default)")
    ap.add_argument("--pad-kb", type=int, default=0, help="developer probe: allocate this many KiB of device memory before anything else (shifts the addresses of every later allocation: "
                                                         "placement sensitivity of the kernels, tools/placement_probe.sh)")
    ap.add_argument("--streams", type=int, default=None,
                    help="contexts / HIP streams a step's frames are split over (default 3 x 16 frames: the small coarse-level kernels of one "
                         "batch overlap the large kernels of another: +13 % over one stream)")
    args = ap.parse_args()
    if args.view_shards > 1 or args.col_shards > 1:          # one context per shard, no stream splitting
        args.streams = 1
        if args.frames is None:
            args.frames = 4 if args.config == "cfg5" else 16
    if args.passes is None:
        args.passes = 1 if (args.calib or args.view_shards > 1 or args.col_shards > 1) else 20
    if args.streams is None:        # cfg3 / shipped re-expand the CPW meshes on every context: one context there, three elsewhere
        args.streams = 1 if args.config in ("cfg3", "shipped") else 3
    if args.frames is None:      # 32 frames per ms_stitch call (the ABI's per-call limit; cfg5: 16): +2 % (cfg2) to +15 % (cfg3, shipped) over 16 (8), profiles/r03_batch_sweep.txt
        args.frames = {"cfg5": 16}.get(args.config, 32) * args.streams

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # typed as `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU, the launcher the contract names), pass their
            # output through -- rank 0 prints the one JSON line -- and return their exit code
            import socket
            import subprocess
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd))
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
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import msstitch as ms
    import synth
    import dist_frames as df

    # The data path of N > 1 (the gather of the finished slabs on rank 0) goes through the product's own multi-GPU layer, ms_dist (csrc/dist.cpp: RCCL
    # send / recv over xGMI; the host mailbox when MS_BENCH_SHARE_GPU puts every rank on one device).  torch.distributed stays for what the bench contract
    # prescribes around the timed region (barrier, max over ranks) and to hand the communicator's id to the ranks.
    dd = {"D": None, "info": None}      # the ms_dist communicator comes up AFTER the compute-only region (bring_up_dist below): whatever the transport does, that number exists

    def bring_up_dist():
        import msdist
        # MS_BENCH_RCCL_LIB: ms_dist_set_rccl_library (tests put the loopback implementation of tests/fake_rccl.cpp there: the RCCL branch of csrc/dist.cpp
        # with N > 1 ranks on a one-GPU box); with it MS_BENCH_SHARE_GPU keeps gloo for torch.distributed but ms_dist takes the RCCL transport
        transport = msdist.HOST if share else msdist.RCCL
        if os.environ.get("MS_BENCH_RCCL_LIB"):
            msdist.set_rccl_library(os.environ["MS_BENCH_RCCL_LIB"])
            transport = msdist.RCCL
        # The bring-up is decided by ALL ranks together: rank 0's id (or its failure) is broadcast, and after the collective creation the ranks agree (MIN over a flag)
        # on whether every one of them has a communicator -- a rank that fell back alone would wait for ever in the first gather.
        D, dist_info, why = None, None, None
        box = [None]
        if rank == 0:
            try:
                box = [msdist.unique_id(world, transport)]
            except Exception as e:
                why = str(e)[:160]
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            try:
                D = msdist.Dist(rank, world, box[0], device=local_rank)
                dist_info = D.info()
            except Exception as e:
                D, why = None, str(e)[:160]
        flag = torch.tensor([1 if D is not None else 0], dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:      # the bench line must survive a transport that does not come up: fall back to torch.distributed.gather, and SAY so
            if D is not None:
                D.close()
            D, dist_info = None, {"transport": "torch.distributed (ms_dist did not come up on every rank: %s)" % (why or "another rank failed")}
        dd["D"], dd["info"] = D, dist_info

    _pad = torch.empty(args.pad_kb * 1024, dtype=torch.uint8, device=dev) if args.pad_kb > 0 else None      # (kept alive for the whole run)
    shipped = args.config == "shipped"
    cpw = args.config in ("cfg3", "shipped")
    cfg = dict(synth.CONFIGS["cfg5" if args.config == "cfg5" else "cfg2"])
    F = args.frames
    gains = synth.gains(cfg["n"])
    S = max(1, args.streams)
    assert F % S == 0, "--frames must be a multiple of --streams"
    full_w, full_h = cfg["w"], cfg["h"]            # what the cameras deliver; `cfg` describes what the compositor composites
    mesh_nm = (10, 10) if shipped else (40, 40)    # defs.h:65-66 / BASELINE configs[2]
    proj = ms.PROJ_SPHERICAL
    rig = None
    resize_scale = None
    if shipped:
        # stitch_calib as the reference ships it (calibration.cpp:252-311): rig + scales, compose-scale ROIs, the num_bands rule, a canvas that fits the panorama
        proj = ms.PROJ_CYLINDRICAL
        rig = ms.calibrate_cameras(cfg["n"], full_w, full_h, cfg["hfov_deg"], 0.6, 0.01, 1.4)
        cfg["w"], cfg["h"] = rig["compose_width"], rig["compose_height"]
        resize_scale = rig["compose_scale"] if rig["resize_input"] else None
        rois = [ms.warp_roi(proj, rig["K_compose"][i], rig["R"][i], rig["compose_warp_scale"], cfg["w"], cfg["h"]) for i in range(cfg["n"])]
        pr = ms.result_roi(rois)
        cfg["num_bands"] = ms.num_bands_rule(pr[2], pr[3], 5.0)[1]
        cfg["out_w"] = (2 * max(abs(pr[0]), abs(pr[0] + pr[2])) + 1) & ~1
        cfg["out_h"] = (2 * max(abs(pr[1]), abs(pr[1] + pr[3])) + 1) & ~1
    first_full = [torch.from_numpy(synth.frame(full_w, full_h, i, 0)).to(dev) for i in range(cfg["n"])] if shipped else None

    def make_comp(max_frames):
        if shipped:
            c = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), proj, rig["compose_warp_scale"], num_bands=cfg["num_bands"], enable_cpw=True,
                              out_size=(cfg["out_w"], cfg["out_h"]), max_frames=max_frames)
            for i in range(cfg["n"]):
                c.set_camera(i, rig["K_compose"][i], rig["R"][i])
            c.build_maps()
            g = c.calibrate_seam(first_full, rig["K_seam"], rig["seam_scale"], rig["seam_warp_scale"], dilate=True)      # gains + seam masks (calibration.cpp:92-135, 224-237)
            gains[:] = g
            c.init_blender()
        else:
            c = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), proj, synth.warp_scale(cfg["out_w"]),
                              num_bands=cfg["num_bands"], enable_cpw=cpw, out_size=(cfg["out_w"], cfg["out_h"]), max_frames=max_frames)
            for i in range(cfg["n"]):
                K, R = synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i)
                c.set_camera(i, K, R)
                c.set_gain(i, gains[i])
            c.build_maps(); c.build_masks(1); c.init_blender()
        if cpw:
            for i in range(cfg["n"]):
                r = c.view_geom(i).roi
                c.set_mesh(i, *synth.mesh(r.width, r.height, mesh_nm[0], mesh_nm[1], phase=0.1 * i))
        return c
    if args.view_shards > 1:
        return run_view_shards(args, cfg, gains, rank, world, dev, share)
    if args.col_shards > 1:
        return run_col_shards(args, cfg, gains, rank, world, dev, share)
    comps = [make_comp(F // S) for _ in range(S)]     # one context (own per-frame buffers) per HIP stream
    comp = comps[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]

    # synthetic input: 8 distinct frames per view, cycled; frame t of the global sequence -> rank t mod G
    n_distinct = max(1, args.distinct)
    pool = [[torch.from_numpy(synth.frame(full_w, full_h, i, t)).to(dev) for i in range(cfg["n"])] for t in range(n_distinct)]
    frames_full = [pool[(rank + j * world) % n_distinct] for j in range(F)]
    if resize_scale:      # timed.cpp:75-85: every frame of every view goes through cuda::resize(compose_scale) before the remap -- per pass, inside the timed region
        frames = [[torch.zeros((cfg["h"], cfg["w"], 3), dtype=torch.uint8, device=dev) for _ in range(cfg["n"])] for _ in range(F)]
    else:
        frames = frames_full
    pg = comp.pano_geom()
    fh = pg.dst_roi_final.height
    outs = [[torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=dev) for _ in range(F)] for _ in range(2)]
    ya, yb = max(0, pg.canvas_y & ~1), min(cfg["out_h"] & ~1, (pg.canvas_y + fh + 1) & ~1)       # even-aligned pano rows of the canvas
    i420 = args.gather_format == "i420" and cfg["out_w"] % 2 == 0
    if i420:
        slabs = [torch.zeros((F, (yb - ya) * 3 // 2, cfg["out_w"]), dtype=torch.uint8, device=dev) for _ in range(2)]
    else:
        slabs = [torch.zeros((F, fh, cfg["out_w"], 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    import ctypes
    Fs = F // S
    gather = world > 1 and not args.no_gather
    egress = gather or args.emulate_gather
    direct_i420 = i420 and egress and not args.egress_convert       # the level-0 band kernel writes the I420 slabs itself (ms_stitch_i420)
    if direct_i420:
        for b in range(2):
            slabs[b][:, :(yb - ya)] = 16; slabs[b][:, (yb - ya):] = 128          # black outside the panorama ROI, written once
        assert comp.i420_rows() == (ya, yb - ya)
        subruns = [[comps[k].prepared_i420(frames[k * Fs:(k + 1) * Fs], [slabs[b][j] for j in range(k * Fs, (k + 1) * Fs)]) for k in range(S)] for b in range(2)]
    else:
        subruns = [[comps[k].prepared(frames[k * Fs:(k + 1) * Fs], out8u=outs[b][k * Fs:(k + 1) * Fs]) for k in range(S)] for b in range(2)]
    handles = [ctypes.c_void_p(st.cuda_stream) for st in streams]
    resize_runs = None
    if resize_scale:      # one launch per context: all views of its Fs frames
        resize_runs = [ms.resize_linear_batch_prepared([t for j in range(k * Fs, (k + 1) * Fs) for t in frames_full[j]],
                                                       [t for j in range(k * Fs, (k + 1) * Fs) for t in frames[j]], resize_scale, resize_scale) for k in range(S)]

    # egress of the N>1 path (only with --egress-convert; by default the stitch writes I420 directly): the pano ROI rows of every canvas of the step -> one I420 slab each, one launch once the contexts have joined,
    # on a stream of its own (one launch per context on that context's stream measured 4 % slower)
    to_i420 = [ms.bgr_to_i420_batch_prepared([outs[b][j][ya:yb] for j in range(F)], [slabs[b][j] for j in range(F)]) for b in range(2)] if (i420 and egress and not direct_i420) else None

    egress_stream = torch.cuda.Stream(device=dev) if to_i420 else None
    egress_handle = ctypes.c_void_p(egress_stream.cuda_stream) if to_i420 else None
    egress_done = [torch.cuda.Event(), torch.cuda.Event()] if to_i420 else None
    egress_used = [False, False]

    # The S contexts are independent pipelines (own tables, own per-frame buffers, own output slots), so the passes need not be fork-joined on the caller's
    # stream.  Measured (profiles/r03_batch_sweep.txt): keeping the join is 1 % FASTER (34.24 k against 33.89 k frames/s, three alternating runs each) -- the
    # contexts stay in step and share the tables in L2 --, so the join stays the default; --independent-streams is the A/B.  An egress needs the join anyway.
    join_passes = (not args.independent_streams) or egress
    join_every = 1 if egress else max(1, args.join_every)      # fork at the first pass of a group of `join_every`, join after its last one
    join_state = {"n": 0}

    def make_run(b):
        def run():
            if S > 1:
                cur = torch.cuda.current_stream()
                first_of_group = join_state["n"] % join_every == 0
                join_state["n"] += 1
                last_of_group = join_state["n"] % join_every == 0
                for k in range(S):
                    if join_passes and first_of_group:
                        streams[k].wait_stream(cur)
                    if resize_runs:
                        resize_runs[k](handles[k])
                    subruns[b][k](handles[k])
                if join_passes and last_of_group:
                    for k in range(S):
                        cur.wait_stream(streams[k])
            else:
                if resize_runs:
                    resize_runs[0](handles[0])
                subruns[b][0](handles[0])
            if to_i420:          # on its own stream, behind this step's canvases: it overlaps the next step's kernels instead of delaying them
                egress_stream.wait_stream(torch.cuda.current_stream())
                to_i420[b](egress_handle)
                egress_done[b].record(egress_stream)
        return run
    runs = [make_run(b) for b in range(2)]
    gl = [[torch.empty_like(slabs[0]) for _ in range(world)] for _ in range(2)] if (gather and rank == 0) else [None, None]
    pending = [None, None]
    comm_stream = torch.cuda.Stream(device=dev) if gather else None      # the sends / receives overlap the next pass's kernels
    comm_done = [None, None]
    y0 = pg.canvas_y

    mesh_pool = []
    if cpw and args.recalib_every > 0:       # pre-generated meshes (the optimiser that produces them is out of scope): 4 phases, cycled
        for ph in range(4):
            mesh_pool.append([synth.mesh(comp.view_geom(i).roi.width, comp.view_geom(i).roi.height, mesh_nm[0], mesh_nm[1], phase=0.1 * i + 0.7 * (ph + 1))
                              for i in range(cfg["n"])])
    recal = {"frames": 0, "count": 0}

    state = {"pass": 0, "gather_every": max(1, args.gather_every), "gather_on": True, "last_b": 0, "gathered": 0, "last_gather_b": None}

    def step(s):
        for _ in range(args.passes):
            one_pass()

    def one_pass():
        p_idx = state["pass"]; state["pass"] += 1
        b = p_idx & 1
        state["last_b"] = b
        do_gather = gather and state["gather_on"] and (p_idx % state["gather_every"] == 0)
        if pending[b] is not None:
            pending[b].wait(); pending[b] = None
        if comm_done[b] is not None:      # the slabs of buffer b have left (or arrived): the stitch may overwrite them
            torch.cuda.current_stream().wait_event(comm_done[b]); comm_done[b] = None
        if mesh_pool:
            recal["frames"] += F
            if recal["frames"] >= args.recalib_every:
                recal["frames"] -= args.recalib_every
                for cc in comps:      # convertMeshesToMap for every view: one call, two launches (ms_set_meshes)
                    cc.set_meshes(mesh_pool[recal["count"] % 4])
                recal["count"] += 1
        if to_i420 and egress_used[b]:
            torch.cuda.current_stream().wait_event(egress_done[b])      # the canvases / slabs of buffer b are free again
        runs[b]()
        if to_i420:
            egress_used[b] = True
        if gather or args.emulate_gather:
            if not i420:         # (the I420 slabs are written by runs[b] itself)
                for j in range(F):
                    slabs[b][j].copy_(outs[b][j][y0:y0 + fh], non_blocking=True)
            if not do_gather:
                pass
            elif dd["D"] is not None:      # ms_dist: one grouped exchange, G - 1 point-to-point transfers into the sink (RCCL: enqueued; host mailbox: blocking)
                if to_i420:
                    torch.cuda.current_stream().wait_event(egress_done[b])
                comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(comm_stream):
                    dd["D"].gather_slabs(slabs[b], gl[b], sink=0)
                    comm_done[b] = torch.cuda.Event(); comm_done[b].record(comm_stream)
            elif share:
                if to_i420:
                    egress_done[b].synchronize()
                df.gather_slabs(slabs[b].cpu(), rank, world, dst=0, async_op=False)
            else:
                if to_i420:
                    torch.cuda.current_stream().wait_event(egress_done[b])      # the collective is ordered behind the caller's stream
                pending[b], _ = df.gather_slabs(slabs[b], rank, world, dst=0, async_op=True, out=gl[b])
            if do_gather:
                state["gathered"] += 1
                state["last_gather_b"] = b

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait(); pending[b] = None
            if comm_done[b] is not None:
                comm_done[b].synchronize(); comm_done[b] = None

    def timed_region(steps, warmup):
        for s_ in range(warmup):
            step(s_)
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        g0_ = state["gathered"]
        t0 = time.perf_counter()
        for s_ in range(steps):
            step(s_)
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, state["gathered"] - g0_

    # N > 1 (BASELINE configs[3]: a live-rate stream, frame-parallel over G GPUs, the finished equirect frames gathered on one GPU): three timed regions.
    #   1. compute only (no collective) -- BEFORE the ms_dist communicator exists, so this number survives whatever the transport does;
    #   2. the MAIN region = `value`: exactly K steps, every frame stitched, and the egress of a live stream: the slabs of about 30 batches per second and
    #      rank travel to the sink (configs[3] asks for ONE 30 fps stream: this is ~100x its bytes and still leaves the links idle) -- `--gather-every k`
    #      (k >= 1) makes the main region gather every k-th pass instead;
    #   3. `value_full_gather`: every frame of every rank delivered to the ONE sink at benchmark rate -- bound by that GPU's inbound xGMI links
    #      (7 x ~55 GB/s ~ 100 k frames/s of 3.6 MB slabs), not by the compositor; reported beside `value`, never instead of it.
    # A watchdog covers 2 and 3: a transport that hangs (first real multi-GPU run: RCCL has never seen N > 1 here) costs the missing numbers, not the line.
    no_gather, full_gather, dist_incomplete = None, None, None
    watchdog = {"deadline": None, "stage": None, "partial": None}
    if gather:
        state["gather_on"] = False
        el_ng, _ = timed_region(max(1, args.steps // 2), args.warmup)
        no_gather = world * F * args.passes * max(1, args.steps // 2) / el_ng
        state["gather_on"] = True
        import threading

        def minimal_line(note):
            v = watchdog["partial"] if watchdog["partial"] else (no_gather, None)
            return {"metric": "stitched frames/sec, 6x1080p->4K equirect (ms/frame = 1000/value*n_gpus)", "value": round(v[0], 2), "unit": "frames/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": (round(v[1] / args.steps * 1e3, 4) if v[1] else None), "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": "u8 in / int16+fp32 pyramid arithmetic", "data": "synthetic",
                    "config": {"workload": "%s, frame-parallel x%d" % (args.config, world), "frames_per_pass": F, "passes_per_step": args.passes},
                    "value_no_gather": round(no_gather, 2), "incomplete": note, "dist": dd["info"]}

        def watch():
            while True:
                time.sleep(1.0)
                dl = watchdog["deadline"]
                if dl is None:
                    return
                if time.time() > dl:
                    if rank == 0:
                        print(json.dumps(minimal_line("the multi-GPU transport did not finish '%s' within its deadline: `value` is %s; the process was ended by the bench's watchdog"
                                                      % (watchdog["stage"], "the main region's (live-rate egress)" if watchdog["partial"] else "the COMPUTE-ONLY rate (no frame left its GPU)"))), flush=True)
                    os._exit(0)
        watchdog["deadline"] = time.time() + float(os.environ.get("MS_BENCH_WATCHDOG_S", "300"))
        watchdog["stage"] = "communicator bring-up"
        threading.Thread(target=watch, daemon=True).start()
        bring_up_dist()
        watchdog["stage"] = "main region (live-rate egress)"
        if args.gather_every <= 0:
            state["gather_every"] = max(1, int(round(no_gather / world / 30.0 / F)))
    elapsed, n_gathered = timed_region(args.steps, args.warmup)
    if gather:
        main_every = state["gather_every"]
        watchdog["partial"] = (world * F * args.passes * args.steps / elapsed, elapsed)
        if main_every != 1:
            watchdog["deadline"] = time.time() + float(os.environ.get("MS_BENCH_WATCHDOG_S", "300"))
            watchdog["stage"] = "full gather (every frame to one sink)"
            state["gather_every"] = 1
            el_fg, n_fg = timed_region(max(1, args.steps // 2), max(1, args.warmup // 2))
            full_gather = (world * F * args.passes * max(1, args.steps // 2) / el_fg, n_fg, el_fg)
            state["gather_every"] = main_every
        watchdog["deadline"] = None
    D, dist_info = dd["D"], dd["info"]

    if args.calib:
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255); b = torch.empty_like(a)
        for _ in range(3):
            ms.calib_copy(a, b)
        for shape in (0, 1, 2):      # ... and the per-frame kernels' own access shapes over the same GiB (every line touched once: 1 GiB of HBM traffic each)
            for _ in range(2):
                ms.calib_shape(a, shape)
        torch.cuda.synchronize()
        del a, b

    # ---- verification: frames of every batch of the LAST pass re-stitched one at a time on a one-frame context (live code path:
    #      other launch configuration, band tail one band finer) must equal what the batched, multi-stream passes left in the outputs
    verified, verify_note = None, None
    live_comp = None
    if not args.no_verify or not args.no_live:
        live_comp = make_comp(1)
    if live_comp is not None and mesh_pool and recal["count"] > 0:      # the meshes the batch contexts held during the last pass
        for i in range(cfg["n"]):
            live_comp.set_mesh(i, *mesh_pool[(recal["count"] - 1) % 4][i])
    if not args.no_verify:
        b = state["last_b"]
        picks = sorted({k * Fs + j for k in range(S) for j in (0, Fs // 2, Fs - 1)})
        ok = True
        if direct_i420:
            one = live_comp.new_i420(1)
            for j in picks:
                one[0][:(yb - ya)] = 16; one[0][(yb - ya):] = 128
                live_comp.stitch_i420([frames[j]], one)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(one[0], slabs[b][j]))
        else:
            one = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=dev)]
            for j in picks:
                live_comp.stitch([frames[j]], out8u=one)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(one[0], outs[b][j]))
        verified = ok
        verify_note = "%d of the %d frames of the last pass (first / middle / last of each of the %d batches) re-stitched one frame per call: %s" % (
            len(picks), F, S, "byte-identical" if ok else "MISMATCH")
        if mesh_pool:
            verify_note += " [cfg3: the live context carries the meshes of the last recalibration]"

    # ---- live mode: one frame per ms_stitch call (the reference's shape), synchronised after every call ------------------------
    live = None
    if not args.no_live and rank == 0:
        if mesh_pool:      # same meshes as the batch contexts hold now
            for i in range(cfg["n"]):
                live_comp.set_mesh(i, *mesh_pool[(recal["count"] - 1) % 4][i])
        one = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=dev)]
        runs1 = [live_comp.prepared([frames[j % F]], out8u=one) for j in range(8)]
        st = torch.cuda.current_stream()
        h = ctypes.c_void_p(st.cuda_stream)
        for j in range(30):
            runs1[j % 8](h)
        torch.cuda.synchronize()
        lat_us = []
        for j in range(300):
            t1 = time.perf_counter()
            runs1[j % 8](h)
            torch.cuda.synchronize()
            lat_us.append((time.perf_counter() - t1) * 1e6)
        t1 = time.perf_counter()
        for j in range(300):
            runs1[j % 8](h)
        torch.cuda.synchronize()
        back_to_back = 300 / (time.perf_counter() - t1)
        live = {"us_per_frame_p50": round(float(np.percentile(lat_us, 50)), 1), "us_per_frame_p95": round(float(np.percentile(lat_us, 95)), 1),
                "frames": 300, "fps_back_to_back": round(back_to_back, 1),
                "mode": "one frame per ms_stitch call, inputs resident; latency = host call -> stream idle (host launch + %d dependent kernels)" % len(live_comp.stitch_timed([frames[0]], out8u=one))}
    if live_comp is not None:
        live_comp.close()

    # ---- MS_BENCH_CHECK_GATHERED (tests): what arrived on the sink IS what the peers stitched -- rank 0 re-stitches frames of every peer's last gathered
    #      pass (it knows their inputs: frame j of rank r is set (r + j * world) mod distinct) and compares them with the received slabs, byte for byte
    gathered_check = None
    if os.environ.get("MS_BENCH_CHECK_GATHERED") == "1" and gather and rank == 0 and direct_i420 and state["last_gather_b"] is not None and not resize_scale and not mesh_pool:
        chk_comp = make_comp(1)
        one = chk_comp.new_i420(1)
        gb = state["last_gather_b"]
        ok, n_chk = True, 0
        for r in range(1, world):
            for j in sorted({0, F // 2, F - 1}):
                one[0][:(yb - ya)] = 16; one[0][(yb - ya):] = 128
                chk_comp.stitch_i420([pool[(r + j * world) % n_distinct]], one)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(one[0], gl[gb][r][j]))
                n_chk += 1
        chk_comp.close()
        gathered_check = {"equal": ok, "frames": n_chk, "what": "frames of every peer's last gathered pass as received on the sink vs the sink's own stitch of the same inputs"}

    # ---- the same workload with EVERY frame of the pass distinct: SURVEY 8(d) prescribes 8 sets cycled = 298 MB of source against a 256 MiB
    #      Infinity Cache; here the pass's F frames are F different sets (derived on the device from the 8 base sets by a cyclic shift: other bytes at other
    #      addresses, same statistics), so no source line can be served from a cache.  A second, short timed region; `value` stays the prescribed workload.
    value_d96 = None
    if world == 1 and not resize_scale and not egress and n_distinct < F and not args.no_distinct and not args.calib:
        try:
            fr96 = [[torch.roll(pool[t % n_distinct][i], shifts=(7 * (t // n_distinct) + 1, 13 * (t // n_distinct) + 3), dims=(0, 1)) if t >= n_distinct else pool[t][i]
                     for i in range(cfg["n"])] for t in range(F)]
            sub96 = [[comps[k].prepared(fr96[k * Fs:(k + 1) * Fs], out8u=outs[b][k * Fs:(k + 1) * Fs]) for k in range(S)] for b in range(2)]
            keep_sub = [list(subruns[b]) for b in range(2)]
            for b in range(2):
                subruns[b][:] = sub96[b]
            st96 = max(2, args.steps // 4)
            el96, _ = timed_region(st96, 1)
            value_d96 = {"value": round(F * args.passes * st96 / el96, 2), "distinct_frame_sets": F, "steps": st96,
                         "source_bytes": int(F * cfg["n"] * 3 * full_w * full_h)}
            for b in range(2):
                subruns[b][:] = keep_sub[b]
            del fr96, sub96
        except Exception as e:      # optional: never fail the line on it
            value_d96 = {"error": str(e)[:200]}

    # ---- PCIe-inclusive rate: the C++ host pipeline with the reference's thread / queue graph, every source frame uploaded from pinned memory
    pcie = None
    app = os.path.join(ROOT, "video-stitcher_amd", "stitch_app")
    if not args.no_pcie and rank == 0 and world == 1 and os.path.exists(app):
        import subprocess
        try:
            cmd = [app, "--views", str(cfg["n"]), "--size", "%dx%d" % (full_w, full_h), "--out", "%dx%d" % (cfg["out_w"], cfg["out_h"]),
                   "--hfov", str(cfg["hfov_deg"]), "--bands", str(cfg["num_bands"]), "--frames", "1500"] + (["--cpw"] if cpw else []) + (["--reference-calib"] if shipped else [])
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
            pj = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
            pcie = {"value": pj["frames_per_s"], "unit": "frames/s", "frames": pj["frames"],
                    "how": "video-stitcher_amd/stitch_app: capture thread -> hipMemcpy2DAsync of all %d views from pinned memory per frame (double-buffered, own stream) -> "
                           "ms_stitch (1 frame) -> consume thread; %.1f MB over PCIe per frame" % (cfg["n"], cfg["n"] * cfg["w"] * cfg["h"] * 3 / 1e6)}
            # the same with the cameras' NV12 uploaded (half the bytes) and cvtColor(YUV2BGR_NV12) on the device -- a per-camera CPU step in the reference (networking.cpp:45-47)
            pr = subprocess.run(cmd + ["--nv12"], capture_output=True, text=True, timeout=180)
            pcie["nv12_ingest_value"] = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])["frames_per_s"]
            if not cpw and not shipped:      # ... and with no conversion pass at all: the warp samples the NV12 planes (ms_stitch_nv12; contexts without CPW / per-frame resize)
                pr = subprocess.run(cmd + ["--nv12-direct"], capture_output=True, text=True, timeout=180)
                pcie["nv12_direct_value"] = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])["frames_per_s"]
        except Exception as e:      # the number is optional; never fail the bench line on it
            pcie = {"error": str(e)[:200]}

    # ---- instrumented pass: per-kernel hipEvent durations on the launch stream ------------------
    acc = {}
    reps = max(5, min(50, args.steps * args.passes))
    for _ in range(reps):
        for name, ms_t in comp.stitch_timed(frames[:Fs], out8u=outs[0][:Fs]):
            acc.setdefault(name, []).append(ms_t)
    if resize_runs:      # the per-frame cuda::resize of the shipped configuration is part of the frame: timed the same way (events on the launch stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        hcur = ctypes.c_void_p(cur.cuda_stream)
        for _ in range(reps):
            e0.record(cur); resize_runs[0](hcur); e1.record(cur); e1.synchronize()
            acc.setdefault("k_resize_batch", []).append(e0.elapsed_time(e1))
    kmean = {k: float(np.mean(v)) for k, v in acc.items()}
    per_call = np.sum(np.array([acc[k] for k in acc]), axis=0)        # GPU ms of each instrumented ms_stitch call (sum of its kernels)
    lat = {"gpu_ms_per_call_p50": round(float(np.percentile(per_call, 50)), 5), "gpu_ms_per_call_p95": round(float(np.percentile(per_call, 95)), 5),
           "calls": int(per_call.size), "frames_per_call": Fs}
    kb, sumP, Q, A = kernel_bytes(comp, cfg, Fs, cpw)
    if resize_runs:
        kb["k_resize_batch"] = Fs * cfg["n"] * 3.0 * (full_w * full_h + cfg["w"] * cfg["h"])      # read the camera frame, write the compose-scale one
    dom = max(kmean, key=kmean.get)
    achieved = kb.get(dom, 0.0) / (kmean[dom] * 1e-3) / 1e9          # GB/s
    # COMPULSORY bytes of the warp-type kernels and the resize (what any implementation of that stage has to move: every source byte it samples -- at most 4 taps x 3 B per
    # pixel it writes -- and the bytes it writes, over the tiles the work lists keep): the third fraction of the line, so that extra traffic can never read as progress
    ps = comp.plan_stats()
    tpx = ps["warp_tile"][0] * ps["warp_tile"][1]
    warp_px, s1_px = float(ps["n_warp_tiles"] * tpx), float(ps["n_stage1_reachable"] * tpx)
    src_b = cfg["n"] * 3.0 * cfg["w"] * cfg["h"]
    useful = {}
    if cpw:
        useful["k_remap_gain"] = Fs * (min(src_b, 12.0 * s1_px) + 3.0 * s1_px)
        useful["k_warp"] = Fs * (min(3.0 * s1_px, 12.0 * warp_px) + 3.0 * warp_px)
    else:
        useful["k_warp"] = Fs * (min(src_b, 12.0 * warp_px) + 3.0 * warp_px)
    if resize_runs:
        useful["k_resize_batch"] = Fs * cfg["n"] * 3.0 * (full_w * full_h + cfg["w"] * cfg["h"])
    P_list = []
    for i in range(cfg["n"]):
        g = comp.view_geom(i)
        P_list.append((g.roi.width + g.left + g.right) * (g.roi.height + g.top + g.bottom))
    b_alg_frame = synth.algorithmic_bytes((cfg["w"], cfg["h"]), P_list, Q, (cfg["out_w"], cfg["out_h"]), warped_px=A, cpw=cpw)
    if resize_runs:
        b_alg_frame += cfg["n"] * 3.0 * (full_w * full_h + cfg["w"] * cfg["h"])
    gpu_ms_step = float(sum(kmean.values()))
    src_bytes = cfg["n"] * 3.0 * cfg["w"] * cfg["h"]
    b_min_frame = src_bytes + 4.0 * (4.0 / 3.0) * float(sum(P_list)) + 3.0 * cfg["out_w"] * cfg["out_h"]
    b_ref_frame = src_bytes + 20.0 * A + 96.0 * float(sum(P_list)) + (95.0 + 9.0) * Q

    # PMC-measured HBM bytes (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/profile_traffic.sh on the same workload, calibrated on a 1 GiB
    # copy): read from the committed summary -- counters cannot be collected inside this run -- and labelled as such
    traffic, traffic_call, traffic_src, traffic_stale = None, None, None, None
    csrc_now = csrc_sha16()
    for tname in ("traffic_%s.json" % args.config, "traffic_latest.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tpath):
            continue
        tj = json.load(open(tpath))
        if tj.get("config") == args.config and tj.get("frames_per_launch") == Fs:
            if dom in tj.get("kernels", {}):
                traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
                if dom == "k_resize_batch":      # the instrumented time covers every launch of the call's resize (64 images per launch); so must the bytes
                    traffic *= -(-(Fs * cfg["n"]) // 64)
            traffic_call = tj.get("hbm_bytes_per_call")
            # the counters were collected on the kernels of ONE state of csrc/; the summary records a hash of those sources and the line says when they have moved on
            traffic_stale = (tj.get("csrc_sha16") != csrc_now) if tj.get("csrc_sha16") else "unknown (summary predates the source hash)"
            traffic_src = "profiles/%s: rocprofv3 PMC passes (FETCH_SIZE x %.2f, WRITE_SIZE x %.2f, calibrated on the tuned copy in the same run) of this workload, " \
                          "collected %s at commit %s (tag %s); NOT measured in this run" % (tname, tj.get("calibration", {}).get("fetch_factor", 2.0), tj.get("calibration", {}).get("write_factor", 1.0),
                                                                                          tj.get("collected", "?"), tj.get("commit", "?"), tj.get("tag"))
            break

    # ---- the measured ceiling: what a tuned streaming copy / read of 1 GiB reaches on THIS device in THIS run (csrc/compositor.hip k_calib_copy / k_calib_read)
    ceiling = None
    rank_ceilings = None
    if world > 1:      # every rank's own streaming-copy ceiling (boxes and GPUs differ by +-8 %): gathered so that the line explains an uneven scaling curve
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
        except Exception:
            mine = None
        allc = [None] * world
        dist.all_gather_object(allc, mine)
        rank_ceilings = allc
    if rank == 0 and world == 1:
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
            ceiling = {"copy_TBps": round(2.0 * n_c / (best["copy"] * 1e-3) / 1e12, 3), "read_TBps": round(n_c / (best["read"] * 1e-3) / 1e12, 3),
                       "how": "best of 5 launches of the tuned 16 B/lane streaming kernels over 1 GiB (copy counts read + written bytes), hipEvents; "
                              "tools/copy_probe.hip is the 130-variant sweep they were picked from"}
            del ca, cb
        except Exception as e:      # never fail the bench line on the optional ceiling
            ceiling = {"error": str(e)[:200]}
    if rank == 0:
        frames_per_step = F * args.passes
        total_frames = world * frames_per_step * args.steps
        par = "frame-parallel x%d" % world
        if gather:
            par += ", gather of the %s pano rows of %s on rank 0 (%.1f MB/frame) through %s, overlapped with the next pass" % (
                args.gather_format.upper(), "EVERY frame" if state["gather_every"] == 1 else "every %d-th pass (live-rate egress)" % state["gather_every"], slabs[0][0].numel() / 1e6,
                ("ms_dist / " + dist_info["transport"]) if D is not None else "torch.distributed")
        if share:
            par += " [DEBUG: ranks share one GPU, gloo]"
        res = {
            "metric": "stitched frames/sec, 6x1080p->4K equirect (ms/frame = 1000/value*n_gpus)",
            "value": round(total_frames / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_frame": round(elapsed / args.steps / frames_per_step * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 in / int16+fp32 pyramid arithmetic", "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d views%s -> %dx%d %s, %d bands, CPW %s%s; a step = %d passes over a batch of "
                                   "%d frames per GPU (%d frames), each pass on %d HIP stream(s) / contexts, inputs resident in HBM"
                                   % (args.config, cfg["n"], full_w, full_h, (" resized per frame to %dx%d (compose scale %.4f)" % (cfg["w"], cfg["h"], resize_scale)) if resize_scale else "",
                                      cfg["out_w"], cfg["out_h"], "cylindrical panorama (the reference's shipped calibration: seam-scale gains + masks)" if shipped else "equirect, spherical",
                                      pg.num_bands, ("on (%dx%d mesh)" % mesh_nm) if cpw else "off",
                                      (", meshes re-expanded every %d frames" % args.recalib_every) if mesh_pool else "", args.passes, F, frames_per_step, S),
                       "distinct_frame_sets": n_distinct, "frames_per_step": frames_per_step, "frames_per_pass": F, "passes_per_step": args.passes, "streams": S, "parallelism": par},
            "verified": verified, "verified_how": verify_note,
            # `frac` is the PHYSICAL fraction: HBM bytes of the dominant kernel's launch as the PMC counters saw them / its mean launch duration measured in
            # this run / 8 TB/s.  `frac_contract` prices the same launch at SURVEY 8(d)'s ALGORITHMIC bytes (the level-materialised model: it credits bytes
            # this design does not move, so it is the larger number); without a PMC summary for this workload `frac` falls back to it and `basis` says so.
            "roofline": {"bound": "hbm", "kernel": dom, "peak": 8000.0, "unit": "GB/s",
                         "achieved": round((traffic if traffic else kb.get(dom, 0.0)) / (kmean[dom] * 1e-3) / 1e9, 1),
                         "frac": round((traffic if traffic else kb.get(dom, 0.0)) / (kmean[dom] * 1e-3) / 8e12, 4),
                         "basis": "pmc-measured HBM bytes" if traffic else "algorithmic bytes (no PMC summary for this workload)",
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "traffic_note": "FETCH_SIZE x 2 + WRITE_SIZE = requests of the L2s to the fabric: Infinity-Cache hits are counted as HBM bytes (an upper bound). Since round 4 the warp / level-0 band "
                                         "tile lists are dealt to the XCDs in chunks: 3-4 % less time for 11 % more of these bytes (125 vs 113 MB per frame) -- part of any rise of `frac` over round 3 is bytes, not speed",
                         "useful_bytes_per_launch": (int(useful[dom]) if dom in useful else None),
                         "frac_useful": (round(useful[dom] / (kmean[dom] * 1e-3) / 8e12, 4) if dom in useful else None),
                         "frac_useful_note": "compulsory bytes of this kernel (source bytes it samples, capped at 4 taps x 3 B per pixel written, + the u8 bytes it writes over the planned tiles) / mean launch time / 8 TB/s",
                         "achieved_contract": round(achieved, 1), "frac_contract": round(achieved / 8000.0, 4),
                         "alg_bytes_per_launch": int(kb.get(dom, 0)), "mean_launch_ms": round(kmean[dom], 5),
                         "frac_of_copy_ceiling": (round(traffic / (kmean[dom] * 1e-3) / 1e12 / ceiling["copy_TBps"], 4) if (traffic and ceiling and "copy_TBps" in ceiling) else None)},
            "ceiling": ceiling,
            "frame_roofline": {"alg_bytes_per_frame": int(b_alg_frame), "gpu_ms_per_frame": round(gpu_ms_step / Fs, 5),
                               "achieved_GBps": round(b_alg_frame * Fs / (gpu_ms_step * 1e-3) / 1e9, 1),
                               # NOT roofline fractions: SURVEY 8(d)'s level-materialised byte MODEL priced at 8 TB/s over the measured time; > 1 because the design moves
                               # fewer bytes than the model (u8 levels, no accumulator read-modify-write, skipped tiles).  The physical figures are the *_traffic ones.
                               "vs_contract_model": round(b_alg_frame * Fs / (gpu_ms_step * 1e-3) / 8e12, 4),
                               "wall_vs_contract_model": round(b_alg_frame * total_frames / world / elapsed / 8e12, 4),
                               # SURVEY 8(d) bounds: B_min = perfect fusion (inputs + weight pyramids + output), B_ref = the reference's own pass structure
                               "b_min_bytes_per_frame": int(b_min_frame), "b_ref_bytes_per_frame": int(b_ref_frame),
                               "hbm_bytes_per_frame": (int(traffic_call / Fs) if traffic_call else None),
                               "frac_traffic": (round(traffic_call / (gpu_ms_step * 1e-3) / 8e12, 4) if traffic_call else None),
                               "wall_frac_traffic": (round(traffic_call / Fs * total_frames / world / elapsed / 8e12, 4) if traffic_call else None),
                               "wall_frac_of_copy_ceiling": (round(traffic_call / Fs * total_frames / world / elapsed / 1e12 / ceiling["copy_TBps"], 4)
                                                             if (traffic_call and ceiling and "copy_TBps" in ceiling) else None),
                               "note": "the *_traffic fractions use PMC-measured HBM bytes and are the physical ones; *vs_contract_model price the algorithmic-byte model "
                                       "of SURVEY 8(d) and exceed 1 where the design moves fewer bytes than the model"},
            "kernels_ms_per_call": {k: round(v, 5) for k, v in kmean.items()},
            "latency": lat,
        }
        if no_gather is not None:
            res["value_no_gather"] = round(no_gather, 2)
            if state["gather_every"] == 1:
                res["value_full_gather"] = res["value"]          # (--gather-every 1: the main region IS the every-frame gather)
            else:
                res["value_live_rate_gather"] = res["value"]     # (the main region IS the live-rate egress)
                res["value_full_gather"] = round(full_gather[0], 2) if full_gather else None
            res["gather"] = {"gathered_passes": n_gathered, "of_passes": args.steps * args.passes, "every": state["gather_every"],
                             "passes_per_s_and_rank_gathered": round(n_gathered / elapsed, 1),
                             "GBps_into_sink": round(n_gathered * F * (world - 1) * slabs[0][0].numel() / elapsed / 1e9, 2),
                             "full_gather_GBps_into_sink": (round(full_gather[1] * F * (world - 1) * slabs[0][0].numel() / full_gather[2] / 1e9, 2) if full_gather else None)}
        if world > 1:      # what a reader of the first multi-GPU record needs, at the top level: compute-only and every-frame-to-one-sink rates beside `value`, what the communicator saw, every rank's copy ceiling
            res["comm_nranks"] = (dist_info or {}).get("comm_nranks")
            res["transport"] = (dist_info or {}).get("transport")
            res["pci_bus_ids"] = (dist_info or {}).get("pci_bus_ids")
            res["rank_copy_TBps"] = rank_ceilings
            res["how_to_read"] = ("value = BASELINE configs[3]: every frame stitched frame-parallel, the egress of a live stream (the slabs of ~30 batches per second and rank) gathered on rank 0 "
                                  "over ms_dist inside the timed region; value_no_gather = compute only (measured before the communicator exists); value_full_gather = EVERY frame of every rank "
                                  "into the one sink at benchmark rate (bound by one GPU's inbound xGMI links, ~100 k frames/s of 3.6 MB slabs, not by the compositor)")
        if dist_info is not None:
            res["dist"] = dist_info      # what the communicator itself saw: transport, nranks (RCCL's own count), device ordinals and PCI bus ids of every rank
        if value_d96 is not None:
            res["value_distinct"] = value_d96
        if gathered_check is not None:
            res["gathered_frames_checked"] = gathered_check
        if live is not None:
            res["live"] = live
        if pcie is not None:
            res["pcie_inclusive_fps"] = pcie
        if world == 1 and not args.no_cpu_baseline:
            res["verified_vs_oracle"] = oracle_check(cfg, gains, comp, frames[0], cpw)
            res["cpu_baseline"] = cpu_baseline(cfg, gains, comp, frames=[synth.frame(full_w, full_h, i, 0) for i in range(cfg["n"])], resize=resize_scale)
        print(json.dumps(res), flush=True)
    if D is not None:
        D.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
