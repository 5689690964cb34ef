"""HBM traffic from the hardware counters, collected INSIDE the bench run (round 6): two short child runs of bench.py under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
and `... --pmc WRITE_SIZE` (separate passes, counters only -- never combined with a sys / hip / hsa trace), each with the known-size streaming copy of `--calib` in the
same process, summarised by tools/summarize_traffic.py::build_summary exactly as the committed profiles/traffic_*.json are (FETCH_SIZE x the factor the 1 GiB copy
calibrates -- 2.0 on gfx950, MI355X_MICROARCH.md HBM section -- + WRITE_SIZE x its factor).  The child stitches the same contexts' launch shape as the headline region
(one context, Fs frames per call).  Returns None when rocprofv3 is not there or a pass fails: the line then falls back to the committed summary and says so."""
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _counters(db, name):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, avg(value), count(*), avg(duration) from counters_collection where counter_name = ? group by kernel_name", (name,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def measure(config, frames_per_call, distinct=8, frame_source="numpy", timeout_s=150.0):
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import summarize_traffic as st
    except ImportError as e:
        return None, "tools/summarize_traffic.py: %s" % e
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="ms_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", config, "--frames", str(frames_per_call), "--streams", "1", "--steps", "3", "--warmup", "1",
             "--passes", "1", "--distinct", str(distinct), "--frame-source", "device", "--calib"]      # (frames generated on the device: the counters see addresses, not content, and 48 numpy frames cost the child 5 s)      # (the SAME number of distinct frame sets as the timed region: frames of one launch that share a set hit in L2)
    got = {}
    try:
        for ctr, tag in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            left = timeout_s - (time.perf_counter() - t0)
            if left < 10:
                return None, "out of time before the %s pass" % ctr
            p = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "-d", os.path.join(tmp, tag), "-o", tag, "--"] + child, cwd=ROOT, env=env, capture_output=True, text=True, timeout=left)
            db = os.path.join(tmp, tag, "%s_results.db" % tag)
            if p.returncode != 0 or not os.path.exists(db):
                return None, "rocprofv3 --pmc %s: rc %d %s" % (ctr, p.returncode, (p.stderr or "")[-200:])
            got[tag] = _counters(db, ctr)
        res = st.build_summary(got["fetch"], got["write"], "in-run", config, frames_per_call, ROOT)
        res["seconds"] = round(time.perf_counter() - t0, 1)
        if os.environ.get("MS_PMC_DUMP"):      # developer aid: the whole per-kernel summary of the in-run passes
            import json
            json.dump(res, open(os.environ["MS_PMC_DUMP"], "w"), indent=1, sort_keys=True)
        return res, None
    except (subprocess.TimeoutExpired, sqlite3.Error, OSError, KeyError, IndexError, ZeroDivisionError) as e:
        return None, "%s: %s" % (type(e).__name__, str(e)[:200])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
