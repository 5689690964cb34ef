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


def test_single_gpu_line_has_the_contract_keys():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 1000 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["verified"] is True, d["verified_how"]          # the benchmarked batches themselves: 9 frames re-stitched one per call, byte-identical
    assert d["live"]["us_per_frame_p50"] > 0 and d["live"]["us_per_frame_p95"] >= d["live"]["us_per_frame_p50"]
    assert d["pcie_inclusive_fps"]["value"] > 100
    assert d["config"]["frames_per_step"] == d["config"]["frames_per_pass"] * d["config"]["passes_per_step"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "workload" in d["config"]
    if r["traffic"]:      # `frac` is the PHYSICAL fraction (PMC bytes / measured launch time / 8 TB/s); the contract's algorithmic figure sits beside it, with provenance
        assert r["basis"].startswith("pmc") and abs(r["frac"] - r["traffic"] / (r["mean_launch_ms"] * 1e-3) / 8e12) < 1e-3 and "NOT measured in this run" in r["traffic_source"]
        assert r["frac_contract"] >= r["frac"] and "collected" in r["traffic_source"] and r["frac"] < 1.0
    else:
        assert r["basis"].startswith("algorithmic") and abs(r["frac"] - r["frac_contract"]) < 1e-6
    # False when the PMC summary was collected on these very kernel sources, True once csrc/ has moved on (the line must then say so), None only without a summary
    assert r["traffic_stale"] in (True, False) if r["traffic"] else r["traffic_stale"] is None
    assert 0.0 < r["frac_useful"] <= r["frac"] + 1e-6 or not r["traffic"]      # compulsory bytes of the kernel / time / peak: extra traffic can never read as progress
    fr = d["frame_roofline"]
    assert "vs_contract_model" in fr and "wall_vs_contract_model" in fr and "frac" not in fr and "wall_frac" not in fr      # the > 1 model figures are not called fractions
    assert d["config"]["distinct_frame_sets"] == 8 and d["pcie_inclusive_fps"].get("nv12_direct_value", 0) > 100
    c = d["ceiling"]      # the tuned streaming copy / read of this run: the measured ceiling the fractions are read against
    # measured spread of the tuned copy over the boxes of four rounds: 5.08 - 6.18 TB/s (profiles/r04_bench_repeats.txt, r03_copy_probe.txt), read 6.6 - 7.2:
    # 4.0 / 0.9 are the thresholds of round 2, kept (they had been loosened without a measured reason)
    assert c["copy_TBps"] > 4.0 and c["read_TBps"] > c["copy_TBps"] * 0.9


def test_shipped_configuration_line():
    """bench.py --config shipped: the reference's own configuration (cylindrical, COMPOSE_MEGAPIX 1.4 with the per-frame cuda::resize inside the timed region,
    num_bands by the app's rule, seam-scale gains and masks, CPW 10 x 10) as a measured, verified line; one frame of it bit-identical to the oracle."""
    p = subprocess.run([sys.executable, "bench.py", "--config", "shipped", "--steps", "2", "--warmup", "1", "--passes", "4", "--no-live", "--no-pcie"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = last_json(p.stdout)
    assert d["verified"] is True and d["value"] > 1000 and "cylindrical" in d["config"]["workload"] and "resized per frame to 1578x887" in d["config"]["workload"]
    assert "k_resize_batch" in d["kernels_ms_per_call"] and d["verified_vs_oracle"]["bit_identical"] is True
    assert "6 bands" in d["config"]["workload"] and "10x10" in d["config"]["workload"]
    assert d["cpu_baseline"]["value"] > 0 and "cv::resize" in d["cpu_baseline"]["flavour"]


FAKE_RCCL = os.path.join(ROOT, "tests", "_fake_rccl", "libfake_rccl.so")


def test_two_ranks_sharing_the_gpu_run_the_multi_rank_path():
    env = dict(os.environ, MS_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--passes", "2", "--gather-every", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
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
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--passes", "4", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1, p.stdout[-2000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["verified"] is True and d["value"] > 0 and "incomplete" not in d
    assert d["value_live_rate_gather"] == d["value"] and d["value_no_gather"] > 0 and d["value_full_gather"] > 0
    assert d["gather"]["every"] >= 1 and d["gather"]["gathered_passes"] >= 1 and "configs[3]" in d["how_to_read"]


def test_rccl_branch_with_two_ranks_over_the_loopback_library():
    """The RCCL branch of csrc/dist.cpp (ncclCommInitRank, the all-gather of rccl_attach, grouped ncclRecv on the sink / ncclSend on the peers, on the bench's
    communication stream with its event ordering) with TWO ranks: a one-GPU box cannot do that with the real library (it refuses two ranks on one device), so the
    entry points are served by tests/fake_rccl.cpp -- stream-ordered, asynchronous like NCCL, bytes through shared memory.  Proves call order, grouping and
    stream / event ordering of the callers; RCCL over xGMI itself stays unmeasured."""
    assert os.path.isfile(FAKE_RCCL), "tests/_fake_rccl/libfake_rccl.so is built by __graft_entry__.build()"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MS_BENCH_SHARE_GPU="1", MS_BENCH_RCCL_LIB=FAKE_RCCL, MS_BENCH_CHECK_GATHERED="1", GPU_MAX_HW_QUEUES="16")      # (a hardware queue per stream: see tests/test_ms_dist_gpu.py)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--passes", "3", "--no-cpu-baseline", "--no-live"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["verified"] is True and "incomplete" not in d
    assert d["transport"] == "rccl" and d["comm_nranks"] == 2 and d["dist"]["rccl_version"] == 29999      # (29999 = the loopback library's version: not a real RCCL)
    assert d["value_full_gather"] > 0 and d["gathered_frames_checked"]["equal"] is True, d.get("gathered_frames_checked")


def test_view_sharded_ranks_exchange_partials_and_match_the_unsharded_frame():
    """BASELINE configs[4]'s mechanism across two RANKS: each owns half of the views, builds the partial dst pyramid (ms_stitch_partial), the partial of
    rank 1 travels to rank 0 (send / recv), which adds, normalises, collapses (ms_stitch_finish); the sink re-stitches the same frames unsharded and
    compares.  With two GPUs: RCCL over xGMI; on a one-GPU box the two ranks share the GPU and the partials go through gloo (MS_BENCH_SHARE_GPU=1)."""
    import torch
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ) if two else dict(os.environ, MS_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29537",
                        "bench.py", "--gpus", "2", "--view-shards", "2", "--frames", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
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
                        "bench.py", "--gpus", "2", "--col-shards", "2", "--frames", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["equals_unsharded"] is True and d["value"] > 0
    assert ("RCCL" in d["config"]["workload"]) == two


def test_column_shards_on_one_gpu_config5():
    """BASELINE configs[4] geometry (12 x 4K -> 7680 x 3840), two column windows on one GPU: equal to the unsharded frame, and each shard reads a strict
    subset of the views (the ingest is what the split divides)."""
    p = subprocess.run([sys.executable, "bench.py", "--config", "cfg5", "--col-shards", "2", "--frames", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = last_json(p.stdout)
    assert d["equals_unsharded"] is True and all(v < d["config"]["views"] for v in d["config"]["views_read_per_shard"]), d
