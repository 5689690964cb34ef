"""One bench workload = one BASELINE configuration on this rank's GPU: the contexts, the resident input frames, the outputs, the pass / step / timed-region
machinery, and the checks that ride on it (`verified`, live latency, the instrumented per-kernel pass).  bench.py builds one for the headline configuration
(cfg2 = BASELINE configs[1]) and benchlib/others.py builds short ones for configs[2], configs[4] and the reference's shipped rig.

Reference shapes: APP/timed.cpp:56-152 (stitch_one -> stitch_online x N -> feed_online x N -> blend), :75-104 (per-frame resize + both remaps),
APP/calibration.cpp:147-194 (compose scale, num_bands rule), APP/defs.h:51-66."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "video-stitcher_amd") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))

from .frames import Pool  # noqa: E402

DTYPE = "u8 in / int16+fp32 pyramid arithmetic"


class Opt:
    """the knobs of one workload (bench.py fills it from its command line; others.py from its short presets)"""

    def __init__(self, **kw):
        self.config = "cfg2"
        self.frames = None            # frames per pass, split evenly over `streams` contexts
        self.streams = None
        self.passes = 20
        self.steps = 50
        self.warmup = 5
        self.distinct = 8
        self.recalib_every = 60
        self.join_every = 1
        self.independent_streams = False
        self.gather_format = "i420"
        self.egress_convert = False
        self.emulate_gather = False
        self.no_gather = False
        self.gather_every = 1
        self.frame_source = None      # None: numpy for 1080p rigs, device for the 12 x 4K geometry
        self.__dict__.update(kw)
        if self.streams is None:      # cfg3 / shipped re-expand the CPW meshes on every context: one context there, three elsewhere
            self.streams = 1 if self.config in ("cfg3", "shipped") else 3
        if self.frames is None:       # frames per ms_stitch call: 32 (16 for the 12 x 4K geometry), 64 -- the ABI's limit since round 6 -- for the shipped rig, the one configuration that
            self.frames = {"cfg5": 16, "shipped": 64}.get(self.config, 32) * self.streams      # gains from it (+1.9 ... 3.2 %; profiles/r06_batch_sweep.txt, r03_batch_sweep.txt)


class Workload:
    def __init__(self, opt, rank=0, world=1, dev=None, share=False):
        import msstitch as ms
        import synth
        self.ms, self.synth = ms, synth
        self.opt, self.rank, self.world, self.share = opt, rank, world, share
        self.dev = dev if dev is not None else torch.device("cuda", torch.cuda.current_device())
        self.shipped = opt.config == "shipped"
        self.cpw = opt.config in ("cfg3", "shipped")
        cfg = self.cfg = dict(synth.CONFIGS["cfg5" if opt.config == "cfg5" else "cfg2"])
        self.F = F = opt.frames
        self.gains = synth.gains(cfg["n"])
        self.S = S = max(1, opt.streams)
        assert F % S == 0, "--frames must be a multiple of --streams"
        self.Fs = Fs = F // S
        self.full_w, self.full_h = cfg["w"], cfg["h"]            # what the cameras deliver; `cfg` describes what the compositor composites
        self.mesh_nm = (10, 10) if self.shipped else (40, 40)    # defs.h:65-66 / BASELINE configs[2]
        self.proj = ms.PROJ_SPHERICAL
        self.rig = None
        self.resize_scale = None
        dev = self.dev
        if self.shipped:
            # stitch_calib as the reference ships it (calibration.cpp:252-311): rig + scales, compose-scale ROIs, the num_bands rule, a canvas that fits the panorama
            self.proj = ms.PROJ_CYLINDRICAL
            rig = self.rig = ms.calibrate_cameras(cfg["n"], self.full_w, self.full_h, cfg["hfov_deg"], 0.6, 0.01, 1.4)
            cfg["w"], cfg["h"] = rig["compose_width"], rig["compose_height"]
            self.resize_scale = rig["compose_scale"] if rig["resize_input"] else None
            rois = [ms.warp_roi(self.proj, rig["K_compose"][i], rig["R"][i], rig["compose_warp_scale"], cfg["w"], cfg["h"]) for i in range(cfg["n"])]
            pr = ms.result_roi(rois)
            cfg["num_bands"] = ms.num_bands_rule(pr[2], pr[3], 5.0)[1]
            cfg["out_w"] = (2 * max(abs(pr[0]), abs(pr[0] + pr[2])) + 1) & ~1
            cfg["out_h"] = (2 * max(abs(pr[1]), abs(pr[1] + pr[3])) + 1) & ~1
        # synthetic input: `distinct` frame sets per view, cycled; frame t of the global sequence -> rank t mod G
        self.n_distinct = n_distinct = max(1, opt.distinct)
        source = opt.frame_source or ("device" if opt.config == "cfg5" else "numpy")
        self.frame_source = source
        self.pool = Pool.get(cfg["n"], self.full_w, self.full_h, n_distinct, dev, source)
        self.first_full = self.pool[0] if self.shipped else None
        self.comps = [self.make_comp(Fs) for _ in range(S)]      # one context (own per-frame buffers) per HIP stream
        self.comp = self.comps[0]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
        self.frames_full = [self.pool[(rank + j * world) % n_distinct] for j in range(F)]
        if self.resize_scale:      # timed.cpp:75-85: every frame of every view goes through cuda::resize(compose_scale) before the remap -- per pass, inside the timed region
            self.frames = [[torch.zeros((cfg["h"], cfg["w"], 3), dtype=torch.uint8, device=dev) for _ in range(cfg["n"])] for _ in range(F)]
        else:
            self.frames = self.frames_full
        pg = self.pg = self.comp.pano_geom()
        fh = self.fh = pg.dst_roi_final.height
        self.outs = [[torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=dev) for _ in range(F)] for _ in range(2)]
        self.ya, self.yb = max(0, pg.canvas_y & ~1), min(cfg["out_h"] & ~1, (pg.canvas_y + fh + 1) & ~1)       # even-aligned pano rows of the canvas
        ya, yb = self.ya, self.yb
        self.i420 = i420 = opt.gather_format == "i420" and cfg["out_w"] % 2 == 0
        self.gather = world > 1 and not opt.no_gather
        self.egress = egress = self.gather or opt.emulate_gather
        if egress:
            if i420:
                self.slabs = [torch.zeros((F, (yb - ya) * 3 // 2, cfg["out_w"]), dtype=torch.uint8, device=dev) for _ in range(2)]
            else:
                self.slabs = [torch.zeros((F, fh, cfg["out_w"], 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        else:
            self.slabs = None
        self.direct_i420 = direct_i420 = i420 and egress and not opt.egress_convert       # the level-0 band kernel writes the I420 slabs itself (ms_stitch_i420)
        comps, frames, slabs, outs = self.comps, self.frames, self.slabs, self.outs
        if direct_i420:
            for b in range(2):
                slabs[b][:, :(yb - ya)] = 16; slabs[b][:, (yb - ya):] = 128          # black outside the panorama ROI, written once
            assert self.comp.i420_rows() == (ya, yb - ya)
            self.subruns = [[comps[k].prepared_i420(frames[k * Fs:(k + 1) * Fs], [slabs[b][j] for j in range(k * Fs, (k + 1) * Fs)]) for k in range(S)] for b in range(2)]
        else:
            self.subruns = [[comps[k].prepared(frames[k * Fs:(k + 1) * Fs], out8u=outs[b][k * Fs:(k + 1) * Fs]) for k in range(S)] for b in range(2)]
        self.handles = [ctypes.c_void_p(st.cuda_stream) for st in self.streams]
        self.resize_runs = None
        if self.resize_scale:      # one launch per context: all views of its Fs frames
            self.resize_runs = [ms.resize_linear_batch_prepared([t for j in range(k * Fs, (k + 1) * Fs) for t in self.frames_full[j]],
                                                                [t for j in range(k * Fs, (k + 1) * Fs) for t in frames[j]], self.resize_scale, self.resize_scale) for k in range(S)]
        # egress of the N>1 path (only with --egress-convert; by default the stitch writes I420 directly): the pano ROI rows of every canvas of the step -> one I420 slab each,
        # one launch once the contexts have joined, on a stream of its own (one launch per context on that context's stream measured 4 % slower)
        self.to_i420 = [ms.bgr_to_i420_batch_prepared([outs[b][j][ya:yb] for j in range(F)], [slabs[b][j] for j in range(F)]) for b in range(2)] if (i420 and egress and not direct_i420) else None
        self.egress_stream = torch.cuda.Stream(device=dev) if self.to_i420 else None
        self.egress_handle = ctypes.c_void_p(self.egress_stream.cuda_stream) if self.to_i420 else None
        self.egress_done = [torch.cuda.Event(), torch.cuda.Event()] if self.to_i420 else None
        self.egress_used = [False, False]
        # The S contexts are independent pipelines (own tables, own per-frame buffers, own output slots), so the passes need not be fork-joined on the caller's
        # stream.  Measured (profiles/r03_batch_sweep.txt): keeping the join is 1 % FASTER (34.24 k against 33.89 k frames/s, three alternating runs each) -- the
        # contexts stay in step and share the tables in L2 --, so the join stays the default; --independent-streams is the A/B.  An egress needs the join anyway.
        self.join_passes = (not opt.independent_streams) or egress
        self.join_every = 1 if egress else max(1, opt.join_every)      # fork at the first pass of a group of `join_every`, join after its last one
        self.join_n = 0
        self.runs = [self._make_run(b) for b in range(2)]
        self.gl = [[torch.empty_like(slabs[0]) for _ in range(world)] for _ in range(2)] if (self.gather and rank == 0) else [None, None]
        self.pending = [None, None]
        self.comm_stream = torch.cuda.Stream(device=dev) if self.gather else None      # the sends / receives overlap the next pass's kernels
        self.comm_done = [None, None]
        self.mesh_pool = []
        if self.cpw and opt.recalib_every > 0:       # pre-generated meshes (the optimiser that produces them is timed elsewhere): 4 phases, cycled
            for ph in range(4):
                self.mesh_pool.append([synth.mesh(self.comp.view_geom(i).roi.width, self.comp.view_geom(i).roi.height, self.mesh_nm[0], self.mesh_nm[1], phase=0.1 * i + 0.7 * (ph + 1))
                                       for i in range(cfg["n"])])
        self.recal = {"frames": 0, "count": 0}
        self.state = {"pass": 0, "gather_every": max(1, opt.gather_every), "gather_on": True, "last_b": 0, "gathered": 0, "last_gather_b": None}
        self.D = None                 # the ms_dist communicator (benchlib/dist_run.py brings it up AFTER the compute-only region)
        self.dist_info = None
        self.live_comp = None

    # ------------------------------------------------------------------------------------------------------------------------------------------
    def make_comp(self, max_frames):
        ms, synth, cfg = self.ms, self.synth, self.cfg
        if self.shipped:
            rig = self.rig
            c = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), self.proj, rig["compose_warp_scale"], num_bands=cfg["num_bands"], enable_cpw=True,
                              out_size=(cfg["out_w"], cfg["out_h"]), max_frames=max_frames)
            for i in range(cfg["n"]):
                c.set_camera(i, rig["K_compose"][i], rig["R"][i])
            c.build_maps()
            g = c.calibrate_seam(self.first_full, rig["K_seam"], rig["seam_scale"], rig["seam_warp_scale"], dilate=True)      # gains + seam masks (calibration.cpp:92-135, 224-237)
            self.gains[:] = g
            c.init_blender()
        else:
            c = ms.Compositor(cfg["n"], (cfg["w"], cfg["h"]), self.proj, synth.warp_scale(cfg["out_w"]),
                              num_bands=cfg["num_bands"], enable_cpw=self.cpw, out_size=(cfg["out_w"], cfg["out_h"]), max_frames=max_frames)
            for i in range(cfg["n"]):
                K, R = synth.camera(cfg["n"], cfg["w"], cfg["h"], cfg["hfov_deg"], i)
                c.set_camera(i, K, R)
                c.set_gain(i, self.gains[i])
            c.build_maps(); c.build_masks(1); c.init_blender()
        if self.cpw:
            for i in range(cfg["n"]):
                r = c.view_geom(i).roi
                c.set_mesh(i, *synth.mesh(r.width, r.height, self.mesh_nm[0], self.mesh_nm[1], phase=0.1 * i))
        return c

    def _make_run(self, b):
        S, streams, handles = self.S, self.streams, self.handles

        def run():
            if S > 1:
                cur = torch.cuda.current_stream()
                first_of_group = self.join_n % self.join_every == 0
                self.join_n += 1
                last_of_group = self.join_n % self.join_every == 0
                for k in range(S):
                    if self.join_passes and first_of_group:
                        streams[k].wait_stream(cur)
                    if self.resize_runs:
                        self.resize_runs[k](handles[k])
                    self.subruns[b][k](handles[k])
                if self.join_passes and last_of_group:
                    for k in range(S):
                        cur.wait_stream(streams[k])
            else:
                if self.resize_runs:
                    self.resize_runs[0](handles[0])
                self.subruns[b][0](handles[0])
            if self.to_i420:          # on its own stream, behind this step's canvases: it overlaps the next step's kernels instead of delaying them
                self.egress_stream.wait_stream(torch.cuda.current_stream())
                self.to_i420[b](self.egress_handle)
                self.egress_done[b].record(self.egress_stream)
        return run

    def step(self):
        for _ in range(self.opt.passes):
            self.one_pass()

    def one_pass(self):
        st, F = self.state, self.F
        p_idx = st["pass"]; st["pass"] += 1
        b = p_idx & 1
        st["last_b"] = b
        do_gather = self.gather and st["gather_on"] and (p_idx % st["gather_every"] == 0)
        if self.pending[b] is not None:
            self.pending[b].wait(); self.pending[b] = None
        if self.comm_done[b] is not None:      # the slabs of buffer b have left (or arrived): the stitch may overwrite them
            torch.cuda.current_stream().wait_event(self.comm_done[b]); self.comm_done[b] = None
        if self.mesh_pool:
            self.recal["frames"] += F
            if self.recal["frames"] >= self.opt.recalib_every:
                self.recal["frames"] -= self.opt.recalib_every
                for cc in self.comps:      # convertMeshesToMap for every view: one call, two launches (ms_set_meshes)
                    cc.set_meshes(self.mesh_pool[self.recal["count"] % 4])
                self.recal["count"] += 1
        if self.to_i420 and self.egress_used[b]:
            torch.cuda.current_stream().wait_event(self.egress_done[b])      # the canvases / slabs of buffer b are free again
        self.runs[b]()
        if self.to_i420:
            self.egress_used[b] = True
        if self.egress:
            if not self.i420:         # (the I420 slabs are written by runs[b] itself)
                y0 = self.pg.canvas_y
                for j in range(F):
                    self.slabs[b][j].copy_(self.outs[b][j][y0:y0 + self.fh], non_blocking=True)
            if not do_gather:
                pass
            elif self.D is not None:      # ms_dist: one grouped exchange, G - 1 point-to-point transfers into the sink (RCCL: enqueued; host mailbox: blocking)
                if self.to_i420:
                    torch.cuda.current_stream().wait_event(self.egress_done[b])
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    self.D.gather_slabs(self.slabs[b], self.gl[b], sink=0)
                    self.comm_done[b] = torch.cuda.Event(); self.comm_done[b].record(self.comm_stream)
            else:
                import dist_frames as df
                if self.share:
                    if self.to_i420:
                        self.egress_done[b].synchronize()
                    df.gather_slabs(self.slabs[b].cpu(), self.rank, self.world, dst=0, async_op=False)
                else:
                    if self.to_i420:
                        torch.cuda.current_stream().wait_event(self.egress_done[b])      # the collective is ordered behind the caller's stream
                    self.pending[b], _ = df.gather_slabs(self.slabs[b], self.rank, self.world, dst=0, async_op=True, out=self.gl[b])
            if do_gather:
                st["gathered"] += 1
                st["last_gather_b"] = b

    def drain(self):
        for b in range(2):
            if self.pending[b] is not None:
                self.pending[b].wait(); self.pending[b] = None
            if self.comm_done[b] is not None:
                self.comm_done[b].synchronize(); self.comm_done[b] = None

    def timed_region(self, steps, warmup):
        """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + torch.cuda.synchronize() on both sides; max over ranks."""
        import torch.distributed as dist
        world = self.world
        for _ in range(warmup):
            self.step()
        self.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        g0 = self.state["gathered"]
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if self.share else self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, self.state["gathered"] - g0

    # ---- verification: frames of every batch of the LAST pass re-stitched one at a time on a one-frame context (live code path: other launch
    #      configuration, band tail one band finer) must equal what the batched, multi-stream passes left in the outputs
    def _live_context(self):
        if self.live_comp is None:
            self.live_comp = self.make_comp(1)
        if self.mesh_pool and self.recal["count"] > 0:      # the meshes the batch contexts held during the last pass
            for i in range(self.cfg["n"]):
                self.live_comp.set_mesh(i, *self.mesh_pool[(self.recal["count"] - 1) % 4][i])
        return self.live_comp

    def verify(self):
        cfg, F, S, Fs = self.cfg, self.F, self.S, self.Fs
        live_comp = self._live_context()
        b = self.state["last_b"]
        picks = sorted({k * Fs + j for k in range(S) for j in (0, Fs // 2, Fs - 1)})
        ok = True
        if self.direct_i420:
            ya, yb = self.ya, self.yb
            one = live_comp.new_i420(1)
            for j in picks:
                one[0][:(yb - ya)] = 16; one[0][(yb - ya):] = 128
                live_comp.stitch_i420([self.frames[j]], one)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(one[0], self.slabs[b][j]))
        else:
            one = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=self.dev)]
            for j in picks:
                live_comp.stitch([self.frames[j]], out8u=one)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(one[0], self.outs[b][j]))
        note = "%d of the %d frames of the last pass (first / middle / last of each of the %d batches) re-stitched one frame per call: %s" % (
            len(picks), F, S, "byte-identical" if ok else "MISMATCH")
        if self.mesh_pool:
            note += " [CPW: the live context carries the meshes of the last recalibration]"
        return ok, note

    # ---- live mode: one frame per ms_stitch call (the reference's shape), synchronised after every call
    def live(self, n=300):
        cfg = self.cfg
        live_comp = self._live_context()
        one = [torch.zeros((cfg["out_h"], cfg["out_w"], 3), dtype=torch.uint8, device=self.dev)]
        runs1 = [live_comp.prepared([self.frames[j % self.F]], out8u=one) for j in range(8)]
        st = torch.cuda.current_stream()
        h = ctypes.c_void_p(st.cuda_stream)
        for j in range(30):
            runs1[j % 8](h)
        torch.cuda.synchronize()
        lat_us = []
        for j in range(n):
            t1 = time.perf_counter()
            runs1[j % 8](h)
            torch.cuda.synchronize()
            lat_us.append((time.perf_counter() - t1) * 1e6)
        t1 = time.perf_counter()
        for j in range(n):
            runs1[j % 8](h)
        torch.cuda.synchronize()
        back_to_back = n / (time.perf_counter() - t1)
        return {"us_per_frame_p50": round(float(np.percentile(lat_us, 50)), 1), "us_per_frame_p95": round(float(np.percentile(lat_us, 95)), 1),
                "frames": n, "fps_back_to_back": round(back_to_back, 1),
                "mode": "one frame per ms_stitch call, inputs resident; latency = host call -> stream idle (host launch + %d dependent kernels)" % len(live_comp.stitch_timed([self.frames[0]], out8u=one))}

    # ---- MS_BENCH_CHECK_GATHERED (tests): what arrived on the sink IS what the peers stitched -- rank 0 re-stitches frames of every peer's last gathered
    #      pass (it knows their inputs: frame j of rank r is set (r + j * world) mod distinct) and compares them with the received slabs, byte for byte
    def gathered_check(self):
        st = self.state
        if not (self.gather and self.rank == 0 and self.direct_i420 and st["last_gather_b"] is not None and not self.resize_scale and not self.mesh_pool):
            return None
        F, world, ya, yb = self.F, self.world, self.ya, self.yb
        chk_comp = self.make_comp(1)
        one = chk_comp.new_i420(1)
        gb = st["last_gather_b"]
        ok, n_chk = True, 0
        for r in range(1, world):
            for j in sorted({0, F // 2, F - 1}):
                one[0][:(yb - ya)] = 16; one[0][(yb - ya):] = 128
                chk_comp.stitch_i420([self.pool[(r + j * world) % self.n_distinct]], one)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(one[0], self.gl[gb][r][j]))
                n_chk += 1
        chk_comp.close()
        return {"equal": ok, "frames": n_chk, "what": "frames of every peer's last gathered pass as received on the sink vs the sink's own stitch of the same inputs"}

    # ---- the same workload with EVERY frame of the pass distinct: SURVEY 8(d) prescribes 8 sets cycled = 298 MB of source against a 256 MiB
    #      Infinity Cache; here the pass's F frames are F different sets (derived on the device from the base sets by a cyclic shift: other bytes at other
    #      addresses, same statistics), so no source line can be served from a cache.  A second, short timed region; `value` stays the prescribed workload.
    def distinct_region(self, steps):
        cfg, F, S, Fs, nd = self.cfg, self.F, self.S, self.Fs, self.n_distinct
        if not (self.world == 1 and not self.resize_scale and not self.egress and nd < F):
            return None
        try:
            fr = [[torch.roll(self.pool[t % nd][i], shifts=(7 * (t // nd) + 1, 13 * (t // nd) + 3), dims=(0, 1)) if t >= nd else self.pool[t][i]
                   for i in range(cfg["n"])] for t in range(F)]
            sub = [[self.comps[k].prepared(fr[k * Fs:(k + 1) * Fs], out8u=self.outs[b][k * Fs:(k + 1) * Fs]) for k in range(S)] for b in range(2)]
            keep = [list(self.subruns[b]) for b in range(2)]
            for b in range(2):
                self.subruns[b][:] = sub[b]
            el, _ = self.timed_region(steps, 1)
            out = {"value": round(F * self.opt.passes * steps / el, 2), "distinct_frame_sets": F, "steps": steps,
                   "source_bytes": int(F * cfg["n"] * 3 * self.full_w * self.full_h)}
            for b in range(2):
                self.subruns[b][:] = keep[b]
            del fr, sub
            return out
        except Exception as e:      # optional: never fail the line on it
            return {"error": str(e)[:200]}

    # ---- instrumented pass: per-kernel hipEvent durations on the launch stream (ms_stitch_timed)
    def instrumented(self, reps):
        acc = {}
        Fs = self.Fs
        for _ in range(reps):
            for name, ms_t in self.comp.stitch_timed(self.frames[:Fs], out8u=self.outs[0][:Fs]):
                acc.setdefault(name, []).append(ms_t)
        if self.resize_runs:      # the per-frame cuda::resize of the shipped configuration is part of the frame: timed the same way (events on the launch stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            hcur = ctypes.c_void_p(cur.cuda_stream)
            for _ in range(reps):
                e0.record(cur); self.resize_runs[0](hcur); e1.record(cur); e1.synchronize()
                acc.setdefault("k_resize_batch", []).append(e0.elapsed_time(e1))
        kmean = {k: float(np.mean(v)) for k, v in acc.items()}
        per_call = np.sum(np.array([acc[k] for k in acc]), axis=0)        # GPU ms of each instrumented ms_stitch call (sum of its kernels)
        lat = {"gpu_ms_per_call_p50": round(float(np.percentile(per_call, 50)), 5), "gpu_ms_per_call_p95": round(float(np.percentile(per_call, 95)), 5),
               "calls": int(per_call.size), "frames_per_call": Fs}
        return kmean, lat

    def workload_string(self):
        cfg, opt = self.cfg, self.opt
        return ("%s: %dx%dx%d views%s -> %dx%d %s, %d bands, CPW %s%s; a step = %d passes over a batch of %d frames per GPU (%d frames), each pass on %d HIP stream(s) / contexts, "
                "inputs resident in HBM" % (opt.config, cfg["n"], self.full_w, self.full_h,
                                            (" resized per frame to %dx%d (compose scale %.4f)" % (cfg["w"], cfg["h"], self.resize_scale)) if self.resize_scale else "",
                                            cfg["out_w"], cfg["out_h"], "cylindrical panorama (the reference's shipped calibration: seam-scale gains + masks)" if self.shipped else "equirect, spherical",
                                            self.pg.num_bands, ("on (%dx%d mesh)" % self.mesh_nm) if self.cpw else "off",
                                            (", meshes re-expanded every %d frames" % opt.recalib_every) if self.mesh_pool else "", opt.passes, self.F, self.F * opt.passes, self.S))

    def close(self):
        torch.cuda.synchronize()
        if self.live_comp is not None:
            self.live_comp.close(); self.live_comp = None
        for c in self.comps:
            c.close()
        self.comps, self.comp, self.subruns, self.runs, self.resize_runs, self.to_i420 = [], None, None, None, None, None
        self.outs = self.slabs = self.frames = self.frames_full = self.pool = self.gl = None
        torch.cuda.empty_cache()
