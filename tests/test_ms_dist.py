"""ms_dist (include/ms_dist.h) on CPU: the host (shared-memory mailbox) transport with host memory -- the protocol the single-GPU-box tests and
ranks that share a device use; RCCL replaces the mailbox when every rank has its own GPU (tests/test_ms_dist_gpu.py).  Host logic only: no
compute entry point is called here."""
import os
import sys
import threading

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-stitcher_amd"))


def _exchange(rank, world, idb, results):
    """what every rank runs: info, ordered point-to-point, a grouped all-to-all of multi-piece messages, broadcast, barrier, the mesh exchange"""
    import msdist
    d = msdist.Dist(rank, world, idb, device=0)
    try:
        info = d.info()
        assert info["transport"] == "host" and info["nranks"] == world and info["rank"] == rank and info["comm_nranks"] == 0
        # 1. ring: rank r sends r-stamped messages of growing size (incl. > one 1 MiB mailbox piece and an empty one) to r + 1
        nxt, prv = (rank + 1) % world, (rank - 1) % world
        for k, n in enumerate((0, 1, 4097, (1 << 20) + 13, 3 * (1 << 20) + 5)):
            out = np.full(n, (17 * rank + k) % 251, np.uint8)
            got = np.zeros(n, np.uint8)
            d.group_begin()
            d.send(out, nxt)
            d.recv(got, prv)
            d.group_end()
            assert np.array_equal(got, np.full(n, (17 * prv + k) % 251, np.uint8)), (rank, k)
        # 2. everybody to everybody in ONE group, 2.5 MiB each: single-slot channels must not deadlock
        outs = [np.full(5 * (1 << 19) + r, 10 * rank + r, np.uint8) for r in range(world)]
        ins = [np.zeros(5 * (1 << 19) + rank, np.uint8) for _ in range(world)]
        d.group_begin()
        for r in range(world):
            d.send(outs[r], r)
            d.recv(ins[r], r)
        d.group_end()
        for r in range(world):
            assert ins[r][0] == 10 * r + rank and ins[r][-1] == 10 * r + rank, (rank, r)
        # 2b. TWO sends and TWO receives per peer in one group, 2.5 MiB each (three mailbox pieces): the operations that share a channel run in posting order --
        #     the second receive must not take the first message's pieces, two multi-piece sends must not interleave
        n2 = 5 * (1 << 19)
        a_out = [np.full(n2, (3 + 2 * rank + 7 * r) % 251, np.uint8) for r in range(world)]
        b_out = [np.full(n2 + 1, (4 + 2 * rank + 7 * r) % 251, np.uint8) for r in range(world)]
        a_in = [np.zeros(n2, np.uint8) for _ in range(world)]
        b_in = [np.zeros(n2 + 1, np.uint8) for _ in range(world)]
        d.group_begin()
        for r in range(world):
            if r == rank:
                continue
            d.recv(a_in[r], r)
            d.send(a_out[r], r)
            d.send(b_out[r], r)
            d.recv(b_in[r], r)
        d.group_end()
        for r in range(world):
            if r == rank:
                continue
            assert np.all(a_in[r] == (3 + 2 * r + 7 * rank) % 251), (rank, r, "first message")
            assert np.all(b_in[r] == (4 + 2 * r + 7 * rank) % 251), (rank, r, "second message")
        # 3. broadcast from the last rank
        b = np.arange(100000, dtype=np.float32) * (1.0 if rank == world - 1 else 0.0)
        d.broadcast(b, world - 1)
        assert np.array_equal(b, np.arange(100000, dtype=np.float32))
        d.barrier()
        # 4. recalibration: no update, then one, then none -- identical on every rank
        n_views, rows, cols = 3, 5, 7
        assert d.mesh_exchange(0, None, n_views, rows, cols) is None
        rng = np.random.default_rng(5)
        mx, my = rng.random((n_views, rows, cols), dtype=np.float32), rng.random((n_views, rows, cols), dtype=np.float32)
        got = d.mesh_exchange(0, (4800, 2, mx, my) if rank == 0 else None, n_views, rows, cols)
        assert got is not None and got[0] == 4800 and got[1] == 2 and np.array_equal(got[2], mx) and np.array_equal(got[3], my)
        assert d.mesh_exchange(0, None, n_views, rows, cols) is None
        # 5. an invalid update on the root is an error on EVERY rank (it travels in the header), and the communicator stays in step
        import msstitch as ms
        bad = (1, 3, np.zeros((n_views, 1, cols), np.float32), np.zeros((n_views, 1, cols), np.float32))      # one mesh row: invalid
        with pytest.raises(ms.MsError):
            d.mesh_exchange(0, bad if rank == 0 else None, n_views, 1, cols)
        assert d.mesh_exchange(0, None, n_views, rows, cols) is None
        d.barrier()
        results[rank] = "ok"
    finally:
        d.close()


def _proc(rank, world, idb, q):
    res = {}
    try:
        _exchange(rank, world, idb, res)
        q.put((rank, res.get(rank, "fail")))
    except Exception as e:      # noqa: BLE001
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_host_transport_between_processes(world):
    import msdist
    idb = msdist.unique_id(world, msdist.HOST)
    assert msdist.id_transport(idb) == msdist.HOST
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_proc, args=(r, world, idb, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert got == {r: "ok" for r in range(world)}, got


def test_host_transport_between_threads():
    """one process, one thread per rank: how stitch_app --gpus N --share-gpu runs (ctypes releases the GIL inside the library)"""
    import msdist
    world = 2
    idb = msdist.unique_id(world, msdist.HOST)
    res, errs = {}, []

    def run(r):
        try:
            _exchange(r, world, idb, res)
        except Exception as e:      # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=180)
    assert not errs, errs
    assert res == {0: "ok", 1: "ok"}


def test_auto_transport_without_enough_devices_is_the_host_mailbox():
    import msdist
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs: AUTO picks RCCL here")
    assert msdist.id_transport(msdist.unique_id(2, msdist.AUTO)) == msdist.HOST


def test_argument_errors():
    import msdist
    import msstitch as ms
    with pytest.raises(ms.MsError):
        msdist.unique_id(0)
    with pytest.raises(ms.MsError):
        msdist.unique_id(17)
    idb = msdist.unique_id(1, msdist.HOST)
    with pytest.raises(ms.MsError):
        msdist.Dist(0, 2, idb)                 # id made for one rank
    with pytest.raises(ms.MsError):
        msdist.Dist(0, 1, b"\0" * msdist.ID_BYTES)
    d = msdist.Dist(0, 1, idb)
    try:
        with pytest.raises(ms.MsError):
            d.send(np.zeros(4, np.uint8), 1)   # no such peer
        with pytest.raises(ms.MsError):
            d.send(np.zeros(4, np.uint8), 0)   # to oneself outside a group
        a, b = np.arange(9, dtype=np.uint8), np.zeros(9, np.uint8)
        d.group_begin(); d.send(a, 0); d.recv(b, 0); d.group_end()
        assert np.array_equal(a, b)
        d.barrier()
        assert d.mesh_exchange(0, None, 1, 2, 2) is None
    finally:
        d.close()


def test_host_transport_under_tsan(tmp_path):
    """csrc/dist.cpp instrumented with ThreadSanitizer, one thread per rank (tests/dist_tsan_check.cpp): groups with two messages per peer and
    direction, rotating broadcasts, barriers and the mesh exchange -- any data race reported by TSan fails the run ("run host tests under TSan")"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "dist_tsan")
    lib = os.path.join(ROOT, "video-stitcher_amd")
    cmd = [hipcc, "-x", "hip", "--offload-host-only", "--offload-arch=gfx950", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fPIE", "-Wno-unused-result", "-Wno-option-ignored",
           os.path.join(ROOT, "tests", "dist_tsan_check.cpp"), os.path.join(lib, "csrc", "dist.cpp"), "-L" + lib, "-lmsstitch", "-Wl,-rpath," + lib, "-ldl", "-lrt", "-pthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower():
        pytest.skip("no ThreadSanitizer runtime in this toolchain: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-2000:]
    assert int(subprocess.run("nm %s | grep -c __tsan" % exe, shell=True, capture_output=True, text=True).stdout or 0) > 0, "not instrumented"
    for world, iters in ((2, 60), (3, 60), (4, 100)):
        r = subprocess.run([exe, str(world), str(iters)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
        assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (world, r.returncode, r.stderr[-3000:])
        assert ("ok: %d ranks x %d iterations" % (world, iters)) in r.stdout


def test_rccl_library_can_be_named_once_before_first_use():
    """ms_dist_set_rccl_library: the file RCCL is loaded from is fixed before the first RCCL id / communicator; a file that does not load is a clear MS_ERR_COMM at that
    first use, and naming another one afterwards is MS_ERR_STATE (process-global: checked in a fresh interpreter; no GPU needed, nothing is computed)."""
    import subprocess
    code = r'''
import sys
sys.path.insert(0, %r)
import msdist, msstitch as ms
msdist.set_rccl_library("/nonexistent/librccl.so.1")
try:
    msdist.unique_id(2, msdist.RCCL)
    print("NO ERROR")
except ms.MsError as e:
    print("first use:", e)
try:
    msdist.set_rccl_library(None)
    print("NO ERROR")
except ms.MsError as e:
    print("second call:", e)
print("host still works:", msdist.id_transport(msdist.unique_id(2, msdist.HOST)) == msdist.HOST)
''' % os.path.join(ROOT, "video-stitcher_amd")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    out = p.stdout
    assert p.returncode == 0, p.stderr[-1500:]
    assert "first use:" in out and "could not be loaded" in out, out
    assert "second call:" in out and "already been resolved" in out, out
    assert "host still works: True" in out, out


def test_one_rccl_per_process_next_to_torch():
    """With PyTorch in the process a ROCm box holds two copies of librccl (torch/lib/librccl.so and /opt/rocm/lib/librccl.so.1, same soname).  The product must resolve
    RCCL's entry points from the copy that is ALREADY mapped (RTLD_NOLOAD first), so that one process runs one RCCL; ms_dist_rccl_library_path reports which file that is
    (bench.py --gpus N prints it before its first timed region).  No GPU needed: resolving the library creates no communicator."""
    import torch  # noqa: F401  (maps torch's own librccl)
    import msdist
    path = msdist.rccl_library_path()
    if path is None:
        pytest.skip("no librccl on this machine")
    mapped = sorted({os.path.realpath(l.split()[-1]) for l in open("/proc/self/maps") if "librccl" in l})
    assert mapped == [path], (mapped, path)
