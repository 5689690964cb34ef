"""Multi-GPU plumbing on CPU: frame round-robin ownership and the gather of finished pano slabs on the sink
rank, with torch.distributed `gloo`, world_size 2 (the GPU path uses the same code with backend nccl = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))
import dist_frames as df  # noqa: E402


def test_round_robin_ownership():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            mine = df.local_frames(0, 37, r, world)
            assert all(df.frame_owner(t, world) == r for t in mine)
            seen += mine
        assert sorted(seen) == list(range(37))
    assert df.local_frames(5, 6, 1, 4) == [5, 9]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, F, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gathered = []
        pending = []
        for s in range(steps):
            # "stitch": frame j of this rank in step s is global frame s*F*world + j*world + rank; its slab encodes that index
            slab = torch.stack([torch.full((4, 6, 3), (s * F * world + j * world + rank) % 251, dtype=torch.uint8) for j in range(F)])
            work, gl = df.gather_slabs(slab, rank, world, dst=0, async_op=True)
            pending.append((work, gl))
        for work, gl in pending:
            if work is not None:
                work.wait()
            if rank == 0:
                gathered.append(gl)
        if rank == 0:
            frames = df.reorder(gathered, world)
            q.put([int(f[0, 0, 0]) for f in frames])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gather_and_display_order_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    F, steps = 3, 4
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == [t % 251 for t in range(world * F * steps)], "frames must come back in global display order"


def test_single_rank_gather_is_identity():
    slab = torch.arange(24, dtype=torch.uint8).reshape(1, 2, 4, 3)
    work, gl = df.gather_slabs(slab, 0, 1)
    assert work is None and gl[0] is slab


def test_bench_plumbing_verdict_rules():
    """The fail-loudly rules of `bench.py --gpus N` (benchlib/dist_run.py::plumbing_verdict) as a pure function: a multi-rank run stops BEFORE its first
    timed region -- one JSON line with `failed` / `incomplete`, exit code 2 -- when two ranks drive one GPU without MS_BENCH_SHARE_GPU, when RCCL's own communicator does
    not count N ranks, or when RCCL's all-gather of the bus ids shows a duplicate; nothing else stops it."""
    import types
    sys.path.insert(0, ROOT)
    from benchlib import dist_run as DR
    ranks = lambda ids: {"ranks": [{"pci_bus_id": i} for i in ids], "duplicate_bus_ids": DR.duplicate_bus_ids(ids)}
    ok_info = {"transport": "rccl", "comm_nranks": 2, "pci_bus_ids": ["0000:05:00.0", "0000:15:00.0"]}
    assert DR.duplicate_bus_ids(["a", "b"]) is False and DR.duplicate_bus_ids(["a", "a"]) is True and DR.duplicate_bus_ids(["?", "?"]) is False
    assert DR.plumbing_verdict(ranks(["a", "b"]), ok_info, 2, False) is None
    assert "same GPU" in DR.plumbing_verdict(ranks(["a", "a"]), None, 2, False)
    assert DR.plumbing_verdict(ranks(["a", "a"]), {"transport": "host", "comm_nranks": 0}, 2, True) is None          # the declared debug mode: ranks share cuda:0 on purpose
    assert "counts 1 ranks" in DR.plumbing_verdict(ranks(["a", "b"]), dict(ok_info, comm_nranks=1), 2, False)
    assert "all-gather" in DR.plumbing_verdict(ranks(["a", "b"]), dict(ok_info, pci_bus_ids=["x", "x"]), 2, False)
    line = DR.fail_line(types.SimpleNamespace(steps=3, warmup=1, config="cfg2"), 2, "why", ok_info)
    assert line["value"] is None and line["failed"] is True and line["incomplete"] == "why" and line["n_gpus"] == 2 and line["dist"] == ok_info
    for k in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line
    assert DR.watchdog_seconds() <= 120.0 or "MS_BENCH_WATCHDOG_S" in os.environ      # a hang must leave a line inside a 1 800 s lease with three regions
