"""SURVEY 8(e) view sharding and pano-column sharding as bench regions (BASELINE configs[4]): across ranks (send / recv to the group's sink) or, with one rank,
all shards on the one GPU (the compute cost of the split).  Both return the JSON line as a dict; bench.py prints it."""
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "video-stitcher_amd"),):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def run_view_shards(args, cfg, gains, rank, world, dev, share, frame_source="numpy"):
    """SURVEY 8(e) view sharding: ranks form groups of V; rank k of a group owns views [k*N/V, (k+1)*N/V), builds the partial dst
    Laplacian pyramid of its views for F frames (ms_stitch_partial) and sends it to the group's first rank, which adds the partials,
    normalises, collapses and writes the canvases (ms_stitch_finish).  Groups are frame-parallel.  int16 has no RCCL reduction:
    point-to-point send/recv + the add inside the finish kernels."""
    import torch.distributed as dist
    import msstitch as ms
    import synth
    from .frames import numpy_frame, device_frame
    fr = device_frame if frame_source == "device" else numpy_frame
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
    pool = [[fr(cfg["w"], cfg["h"], i, t, dev) if any(k * N // V <= i < (k + 1) * N // V for k in mine) else None
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
    ok, line = None, None
    if is_sink:      # same frames through an unsharded context on the sink: the split (and the transfer) must not change a single byte
        full = make(1, 0)
        all_pool = pool if all(f is not None for fs in pool for f in fs) else [[fr(cfg["w"], cfg["h"], i, t, dev) for i in range(N)] for t in range(4)]
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
        line = {
            "metric": "stitched frames/sec, view-sharded (%s)" % args.config, "value": round(total / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 in / int16+fp32 pyramid arithmetic",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d views -> %dx%d equirect, %d bands; views split over %d shards%s, %d frames per step per group, "
                                   "%d frame-parallel group(s); partial = %.1f MB/frame/shard"
                                   % (args.config, N, cfg["w"], cfg["h"], cfg["out_w"], cfg["out_h"], cfg["num_bands"], V,
                                      " on ONE GPU (no transfer)" if local else (" [DEBUG gloo, shared GPU]" if share else " (RCCL send/recv to the sink)"),
                                      F, groups, c0.partial_bytes() / 1e6)},
            "equals_unsharded": ok}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in comps.values():
        c.close()
    return line


def run_col_shards(args, cfg, gains, rank, world, dev, share, frame_source="numpy"):
    """SURVEY 8(e) pano-column sharding: ranks form groups of C; rank k of a group composites the panorama columns of window k (work lists cut down to the
    window plus its halo, views that do not reach it never uploaded) for the SAME F frames and sends its column slab of the canvases to the group's
    first rank.  No partial sums cross the link, only finished pixels.  Groups are frame-parallel.  world == 1: all C windows on this GPU, no transfer
    (the compute cost of the split = the recomputed halo + the coarse tails every shard runs in full)."""
    import torch.distributed as dist
    import msstitch as ms
    import synth
    from .frames import numpy_frame, device_frame
    fr = device_frame if frame_source == "device" else numpy_frame
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
    pool = [[fr(cfg["w"], cfg["h"], i, t, dev) if (any_need >> i) & 1 else None for i in range(N)] for t in range(4)]
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
    ok, base_fps, line = None, None, None
    if is_sink:      # same frames through an unsharded context on the sink: the split (and the transfer) must not change a single byte
        full = make(1, 0)
        all_pool = pool if all(f is not None for fs in pool for f in fs) else [[fr(cfg["w"], cfg["h"], i, t, dev) for i in range(N)] for t in range(4)]
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
        line = {
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
            "equals_unsharded": ok}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for c in comps.values():
        c.close()
    return line
