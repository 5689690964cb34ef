"""bench.py's contract on a real GPU: one JSON line with the required keys at N = 1, and the N > 1 control flow (ownership, I420 egress,
double-buffered gather, max-over-ranks timing) exercised with two ranks sharing the one GPU over gloo (MS_BENCH_SHARE_GPU=1: a debug mode,
not a scaling measurement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


R05_TOP_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_frame", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "verified",
                "verified_how", "roofline", "ceiling", "frame_roofline", "kernels_ms_per_call", "latency", "value_distinct", "live", "pcie_inclusive_fps", "verified_vs_oracle", "cpu_baseline"}
ROOFLINE_KEYS = {"bound", "kernel", "peak", "unit", "achieved", "frac", "basis", "traffic", "traffic_measured_in_this_run", "traffic_source", "traffic_stale", "traffic_note", "mean_launch_ms",
                 "useful_bytes_per_launch", "frac_useful", "frac_useful_note", "frac_of_copy_ceiling"}


def fractions(d, path=""):
    """every (path, value) whose KEY is named like a fraction, anywhere in the line"""
    out = []
    if isinstance(d, dict):
        for k, v in d.items():
            if isinstance(v, (dict, list)):
                out += fractions(v, path + "/" + k)
            elif ("frac" in k) and isinstance(v, (int, float)) and not isinstance(v, bool):
                out.append((path + "/" + k, v))
    elif isinstance(d, list):
        for i, v in enumerate(d):
            out += fractions(v, "%s[%d]" % (path, i))
    return out


def test_default_line_is_the_drivers_record():
    """`python bench.py` AS THE DRIVER RUNS IT (N = 1; only K / W shortened): the contract keys, the headline configuration (BASELINE configs[1]) verified against the one-frame
    path and the CPU oracle, a `roofline` block that is physical only (PMC bytes -- collected by this very run -- / measured launch time / 8 TB/s; nothing named frac* above 1; the
    contract-byte model under `model`, as ratios), `cpu_baseline`, and `other_configs`: BASELINE configs[2] (cfg3), configs[4]'s geometry (cfg5, with its two one-GPU shardings)
    and the reference's shipped rig, each measured, verified and bit-identical to the oracle."""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=dict(os.environ, MS_BENCH_CPU_BUDGET_S="3"), capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1, p.stdout[-2000:]
    d = last_json(p.stdout)
    for k in KEYS:
        assert k in d, k
    # the same line keys as round 5's record (profiles/r05_bench.json), plus what round 6 adds; the contract-model figures moved from `roofline` / `frame_roofline` to `model`
    assert R05_TOP_KEYS <= set(d), R05_TOP_KEYS - set(d)
    assert {"model", "other_configs", "value_nothing_cached"} <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 1000 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["verified"] is True, d["verified_how"]          # the benchmarked batches themselves: 9 frames re-stitched one per call, byte-identical
    assert d["verified_vs_oracle"]["bit_identical"] is True
    assert d["live"]["us_per_frame_p50"] > 0 and d["live"]["us_per_frame_p95"] >= d["live"]["us_per_frame_p50"]
    assert d["pcie_inclusive_fps"]["value"] > 100 and d["pcie_inclusive_fps"].get("nv12_direct_value", 0) > 100
    assert d["config"]["frames_per_step"] == d["config"]["frames_per_pass"] * d["config"]["passes_per_step"] and d["config"]["distinct_frame_sets"] == 8 and "workload" in d["config"]
    assert d["value_nothing_cached"] == d["value_distinct"]["value"] and 0.5 * d["value"] < d["value_nothing_cached"] < 1.1 * d["value"]
    r = d["roofline"]
    assert set(r) == ROOFLINE_KEYS, set(r) ^ ROOFLINE_KEYS
    assert r["bound"] == "hbm" and r["peak"] == 8000.0
    # the PMC bytes were collected by THIS run (two rocprofv3 --pmc child passes): not a copied summary, never stale
    assert r["traffic_measured_in_this_run"] is True and r["traffic_stale"] is False and r["traffic"] > 0, r["traffic_source"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and abs(r["frac"] - r["traffic"] / (r["mean_launch_ms"] * 1e-3) / 8e12) < 1e-3
    assert 0.2 < r["frac"] < 1.0 and 0.0 < r["frac_useful"] <= r["frac"] + 1e-6 and 0.0 < r["frac_of_copy_ceiling"] < 1.2      # (the copy ceiling is a measured kernel, not a bound)
    # nothing named like a fraction exceeds 1 anywhere in the line (frac_of_copy_ceiling excepted: see above); the > 1 figures are `model` RATIOS
    bad = [(k, v) for k, v in fractions(d) if v > 1.0 and "copy_ceiling" not in k]
    assert not bad, bad
    m = d["model"]
    assert not [k for k in m if "frac" in k] and m["ratio_frame_alg_bytes_over_peak"] > 0 and m["b_alg_bytes_per_frame"] > m["b_min_bytes_per_frame"] > 0
    assert not ({"frac_contract", "achieved_contract", "alg_bytes_per_launch"} & set(r)) and not ({"vs_contract_model", "wall_vs_contract_model", "alg_bytes_per_frame"} & set(d["frame_roofline"]))
    fr = d["frame_roofline"]
    assert fr["hbm_bytes_per_frame"] > 50e6 and 0.2 < fr["wall_frac_traffic"] < 1.0
    c = d["ceiling"]      # the tuned streaming copy / read of this run: the measured ceiling the fractions are read against
    # measured spread of the tuned copy over the boxes of four rounds: 5.08 - 6.18 TB/s (profiles/r04_bench_repeats.txt, r03_copy_probe.txt), read 6.6 - 7.2:
    # 4.0 / 0.9 are the thresholds of round 2, kept (they had been loosened without a measured reason)
    assert c["copy_TBps"] > 4.0 and c["read_TBps"] > c["copy_TBps"] * 0.9
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    # ---- the other single-GPU BASELINE configurations, in the same record
    oc = d["other_configs"]
    for name in ("cfg3", "cfg5", "shipped"):
        o = oc[name]
        assert "error" not in o, o
        assert o["value"] > 1000 and o["verified"] is True and o["verified_vs_oracle"]["bit_identical"] is True, (name, o)
        assert abs(o["ms_per_frame"] - 1e3 / o["value"]) < 1e-3 * o["ms_per_frame"] + 1e-6
        ro = o["roofline"]
        assert ro["kernel"] in o["kernels_ms_per_call"] and ro["traffic_stale"] in (True, False, None)
        assert ro["frac"] is None or 0.1 < ro["frac"] < 1.0, (name, ro)
    # the bytes per frame of every configuration, collected in this run, are the ones four rounds of committed summaries show (+- 10 %): a summary that miscounts the launches of a
    # call (k_warp_tabs taken for a warp launch: 9 / 12 of cfg5's bytes; the chunked launches of a 64-frame call counted as calls: half of the shipped rig's) cannot pass
    for name, mb in (("cfg3", 158.4), ("cfg5", 431.3), ("shipped", 340.5)):
        got = oc[name]["frame"]["hbm_bytes_per_frame"] / 1e6
        assert 0.9 * mb < got < 1.1 * mb, (name, got)
    assert 0.9 * 123.0 < fr["hbm_bytes_per_frame"] / 1e6 < 1.1 * 123.0
    assert "on (40x40 mesh)" in oc["cfg3"]["workload"] and "re-expanded every 60 frames" in oc["cfg3"]["workload"] and "k_remap_gain" in oc["cfg3"]["kernels_ms_per_call"]
    assert "12x3840x2160" in oc["cfg5"]["workload"] and "7680x3840" in oc["cfg5"]["workload"]
    assert oc["cfg5"]["col_shards_2"]["equals_unsharded"] is True and oc["cfg5"]["view_shards_2"]["equals_unsharded"] is True
    sh = oc["shipped"]      # the reference's own configuration (cylindrical, COMPOSE_MEGAPIX 1.4 with the per-frame cuda::resize inside the region, num_bands by the app's rule, CPW 10 x 10)
    assert "cylindrical" in sh["workload"] and "resized per frame to 1578x887" in sh["workload"] and "k_resize_batch" in sh["kernels_ms_per_call"]
    assert "6 bands" in sh["workload"] and "10x10" in sh["workload"]
    assert wall < 240, "the default bench run took %.0f s" % wall      # (the driver's budget is 1800 s; the verdict asks for <= 120 s of it on a warm box)


def test_committed_traffic_summary_is_the_fallback():
    """--no-pmc: roofline.traffic comes from profiles/traffic_cfg2.json, says so in `traffic_source` and `traffic_measured_in_this_run`, and `traffic_stale` compares the hash of
    csrc/ the summary recorded with the sources that ran."""
    p = subprocess.run([sys.executable, "bench.py", "--frame-source", "device", "--steps", "2", "--warmup", "1", "--passes", "4", "--no-pmc", "--no-others", "--no-cpu-baseline", "--no-live", "--no-pcie", "--no-distinct"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = last_json(p.stdout)["roofline"]
    assert r["traffic_measured_in_this_run"] is False and r["traffic"] > 0 and "NOT measured in this run" in r["traffic_source"] and "collected" in r["traffic_source"]
    assert r["traffic_stale"] in (True, False) and 0.2 < r["frac"] < 1.0


FAKE_RCCL = os.path.join(ROOT, "tests", "_fake_rccl", "libfake_rccl.so")


def test_two_ranks_sharing_the_gpu_run_the_multi_rank_path():
    env = dict(os.environ, MS_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        "bench.py", "--frame-source", "device", "--gpus", "2", "--steps", "2", "--warmup", "1", "--passes", "2", "--gather-every", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "gather" in d["config"]["parallelism"] and "x2" in d["config"]["parallelism"]
    assert d["value_no_gather"] > 0 and d["gather"]["gathered_passes"] == 4 and d["verified"] is True
    assert d["value_full_gather"] == d["value"] and d["gather"]["every"] == 1 and "EVERY frame" in d["config"]["parallelism"]      # --gather-every 1: the main region IS the every-frame gather
    # the data path is the product's own ms_dist layer (host mailbox here: the ranks share the GPU); the line says what the communicator saw
    assert d["dist"]["transport"] == "host" and d["dist"]["nranks"] == 2 and "ms_dist" in d["config"]["parallelism"]
    # ... and the fields a reader of a first multi-GPU record needs, at the top level: what the communicator saw, every rank's own copy ceiling, how to read the three rates
    assert d["transport"] == "host" and len(d["pci_bus_ids"]) >= 2 and len(d["rank_copy_TBps"]) == 2 and all(x and x > 1.0 for x in d["rank_copy_TBps"]) and "value_no_gather" in d["how_to_read"]


def test_plain_invocation_spawns_ranks():
    """`python bench.py --gpus 2` AS TYPED (no launcher, no WORLD_SIZE): bench.py spawns the ranks itself, rank 0 prints the one line, exit code 0 (the
    first real multi-GPU run must not die on plumbing).  The default N > 1 line: `value` = BASELINE configs[3]'s shape (every frame stitched, a live stream's egress
    gathered), the compute-only and the every-frame-to-one-sink rates beside it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MS_BENCH_SHARE_GPU"] = "1"
    p = subprocess.run([sys.executable, "bench.py", "--frame-source", "device", "--gpus", "2", "--steps", "4", "--warmup", "1", "--passes", "4", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1, p.stdout[-2000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["verified"] is True and d["value"] > 0 and "incomplete" not in d
    assert d["value_live_rate_gather"] == d["value"] and d["value_no_gather"] > 0 and d["value_full_gather"] > 0
    assert d["gather"]["every"] >= 1 and d["gather"]["gathered_passes"] >= 1 and "configs[3]" in d["how_to_read"]


def _fake_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MS_BENCH_SHARE_GPU="1", MS_BENCH_RCCL_LIB=FAKE_RCCL, GPU_MAX_HW_QUEUES="16", **kw)
    return env


def test_scale_shaped_dry_run_and_what_is_printed_before_the_first_timed_region():
    """What the driver's SCALE run does, on one GPU: `--gpus 1` and `--gpus 2` back to back.  The two ranks share the GPU (loopback RCCL), so the
    job's compute-only rate at N = 2 must be the N = 1 rate of the same GPU (two processes time-slicing one device: within 15 %) -- a per-rank accounting error (frames counted
    twice, a rank idle) would show as a factor of two.  And before any timed region the N = 2 run has printed, on stderr, the preamble (every rank's device and bus id as
    torch.distributed sees them) and `dist: {librccl_path, rccl_version, comm_nranks, pci_bus_ids, transport}`."""
    assert os.path.isfile(FAKE_RCCL), "tests/_fake_rccl/libfake_rccl.so is built by __graft_entry__.build()"
    common = ["--steps", "3", "--warmup", "1", "--passes", "3", "--no-cpu-baseline", "--no-live", "--no-pcie", "--no-distinct"]      # (frames generated on the device: 48 numpy frames cost every rank of every run 5 s)
    p1 = subprocess.run([sys.executable, "bench.py", "--frame-source", "device", "--gpus", "1", "--no-others", "--no-pmc", "--no-verify"] + common, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    v1 = last_json(p1.stdout)["value"]
    p2 = subprocess.run([sys.executable, "bench.py", "--frame-source", "device", "--gpus", "2"] + common, cwd=ROOT, env=_fake_env(MS_BENCH_CHECK_GATHERED="1"), capture_output=True, text=True, timeout=900)
    assert p2.returncode == 0, (p2.stdout + p2.stderr)[-3000:]
    d = last_json(p2.stdout)
    assert 0.85 < d["value_no_gather"] / v1 < 1.15, (d["value_no_gather"], v1)
    # the RCCL branch of csrc/dist.cpp (ncclCommInitRank, the all-gather of rccl_attach, grouped ncclRecv on the sink / ncclSend on the peers, on the bench's communication
    # stream with its event ordering) with TWO ranks, served by tests/fake_rccl.cpp -- stream-ordered, asynchronous like NCCL, bytes through shared memory: what ARRIVED on the
    # sink is what the peers stitched.  Proves call order, grouping and stream / event ordering of the callers; RCCL over xGMI itself stays unmeasured.
    assert d["n_gpus"] == 2 and d["verified"] is True and "incomplete" not in d and "failed" not in d
    assert d["transport"] == "rccl" and d["comm_nranks"] == 2 and d["dist"]["rccl_version"] == 29999      # (29999 = the loopback library's version: not a real RCCL)
    assert d["value_full_gather"] > 0 and d["gathered_frames_checked"]["equal"] is True, d.get("gathered_frames_checked")
    assert d["value_definition"].startswith("BASELINE configs[3]") and "value_full_gather" in d["value_definition"] and d["value_live_rate_gather"] == d["value"]
    pre = [l for l in p2.stderr.splitlines() if l.startswith("bench preamble: ")]
    dl = [l for l in p2.stderr.splitlines() if l.startswith("dist: ")]
    assert len(pre) == 1 and len(dl) == 1, p2.stderr[-2000:]
    pj, dj = json.loads(pre[0][len("bench preamble: "):]), json.loads(dl[0][len("dist: "):])
    assert pj["world"] == 2 and len(pj["ranks"]) == 2 and pj["duplicate_bus_ids"] is True and pj["share_gpu_debug_mode"] is True and all(r["pci_bus_id"] != "?" for r in pj["ranks"])
    assert dj["transport"] == "rccl" and dj["comm_nranks"] == 2 and dj["rccl_version"] == 29999 and os.path.realpath(dj["librccl_path"]) == os.path.realpath(FAKE_RCCL) and len(dj["pci_bus_ids"]) == 2
    assert d["librccl_path"] == dj["librccl_path"] and d["preamble"]["world"] == 2


def test_a_communicator_of_the_wrong_size_fails_loudly():
    """RCCL's own count of the communicator (ncclCommCount) differs from the number of ranks the launcher started: the run stops BEFORE its first timed region with one JSON
    line (`value` null, `failed`, `incomplete` says why, `dist` shows what the communicator saw) and a non-zero exit code (the loopback library's fault injection)."""
    p = subprocess.run([sys.executable, "bench.py", "--frame-source", "device", "--gpus", "2", "--steps", "2", "--warmup", "1", "--passes", "2", "--no-cpu-baseline", "--no-live"], cwd=ROOT,
                       env=_fake_env(FAKE_RCCL_COUNT_DELTA="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, p.stdout[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["failed"] is True and d["value"] is None and "counts 3 ranks" in d["incomplete"] and d["dist"]["comm_nranks"] == 3 and d["n_gpus"] == 2 and "value_no_gather" not in d


def test_a_hanging_bring_up_is_ended_by_the_watchdog_with_a_failed_line():
    """A transport that hangs in ncclCommInitRank (first real N-GPU run: RCCL has never formed an N > 1 communicator here): the per-stage watchdog (120 s by default; 4 s here)
    prints the one JSON line -- `failed`, `value` null, the stage in `incomplete` -- and ends every rank with a NON-ZERO exit code (a hang used to exit 0)."""
    p = subprocess.run([sys.executable, "bench.py", "--frame-source", "device", "--gpus", "2", "--steps", "2", "--warmup", "1", "--passes", "2", "--no-cpu-baseline", "--no-live"], cwd=ROOT,
                       env=_fake_env(FAKE_RCCL_HANG_S="60", MS_BENCH_WATCHDOG_S="4"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, p.stdout[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout + p.stderr)[-3000:]
    d = json.loads(lines[0])
    assert d["failed"] is True and d["value"] is None and "communicator bring-up" in d["incomplete"] and "watchdog" in d["incomplete"]


def test_view_sharded_ranks_exchange_partials_and_match_the_unsharded_frame():
    """BASELINE configs[4]'s mechanism across two RANKS: each owns half of the views, builds the partial dst pyramid (ms_stitch_partial), the partial of
    rank 1 travels to rank 0 (send / recv), which adds, normalises, collapses (ms_stitch_finish); the sink re-stitches the same frames unsharded and
    compares.  With two GPUs: RCCL over xGMI; on a one-GPU box the two ranks share the GPU and the partials go through gloo (MS_BENCH_SHARE_GPU=1)."""
    import torch
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ) if two else dict(os.environ, MS_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29537",
                        "bench.py", "--frame-source", "device", "--gpus", "2", "--view-shards", "2", "--frames", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["equals_unsharded"] is True and d["value"] > 0
    assert ("RCCL" in d["config"]["workload"]) == two


def test_column_sharded_ranks_exchange_slabs_and_match_the_unsharded_frame():
    """SURVEY 8(e) pano-column split across two RANKS: each composites one half of the panorama's columns (its work lists and uploads cut down to that
    window plus halo), rank 1's finished column slab travels to rank 0, which assembles the canvas and compares it with an unsharded stitch of the same
    frames.  With two GPUs: RCCL; on a one-GPU box the ranks share the GPU and the slabs go through gloo."""
    import torch
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ) if two else dict(os.environ, MS_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        "bench.py", "--frame-source", "device", "--gpus", "2", "--col-shards", "2", "--frames", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["equals_unsharded"] is True and d["value"] > 0
    assert ("RCCL" in d["config"]["workload"]) == two
