"""bench.py's parts that need no GPU: the workload presets, the committed PMC summaries the line falls back to, the in-run PMC pass degrading to a reason instead of an
exception, and the command line (defaults of the driver's invocation; which runs get `other_configs` and the in-run counters)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workload_presets():
    from benchlib.regions import Opt
    from benchlib.others import PRESETS
    assert (Opt(config="cfg2").frames, Opt(config="cfg2").streams) == (96, 3)          # 3 contexts x 32 frames
    assert (Opt(config="cfg3").frames, Opt(config="cfg3").streams) == (32, 1)
    assert (Opt(config="shipped").frames, Opt(config="shipped").streams) == (64, 1)    # profiles/r06_batch_sweep.txt
    assert (Opt(config="cfg5").frames, Opt(config="cfg5").streams) == (48, 3)          # 3 x 16: 12 views x 16 frames = one source table
    assert Opt(config="cfg2", frames=16, streams=1).frames == 16
    assert set(PRESETS) == {"cfg3", "cfg5", "shipped"} and all(p["distinct"] == 8 for p in PRESETS.values())
    for name, p in PRESETS.items():      # every side region times at least 4 000 frames
        o = Opt(**p)
        assert o.frames * o.passes * o.steps >= 4000, name


def test_committed_traffic_summaries_describe_the_launch_shapes_the_bench_uses():
    """profiles/traffic_<config>.json is what roofline.traffic falls back to when the counters cannot be collected in the run: it must exist for every configuration, for the
    frames per call the bench uses, with a source hash and the calibration factors of the guide (FETCH_SIZE x 2, WRITE_SIZE x 1 on gfx950, within 2 %)."""
    from benchlib.regions import Opt
    from benchlib.roofline import committed_traffic
    for cfg in ("cfg2", "cfg3", "cfg5", "shipped"):
        o = Opt(config=cfg)
        tj, name = committed_traffic(cfg, o.frames // o.streams)
        assert tj is not None, "profiles/traffic_%s.json does not describe %d frames per call" % (cfg, o.frames // o.streams)
        assert tj["csrc_sha16"] and tj["hbm_bytes_per_call"] > 0 and name.startswith("traffic_")
        cal = tj["calibration"]
        assert abs(cal["fetch_factor"] - 2.0) < 0.04 and abs(cal["write_factor"] - 1.0) < 0.02
        dom = {"cfg2": "k_warp", "cfg3": "k_remap_gain", "cfg5": "k_warp", "shipped": "k_warp"}[cfg]
        assert tj["kernels"][dom]["hbm_bytes_per_launch"] > 0
    assert committed_traffic("cfg2", 7) == (None, None)


def test_in_run_pmc_degrades_to_a_reason():
    """no GPU here: the rocprofv3 child cannot run; the pass must come back as (None, why), never raise -- the line then falls back to the committed summary"""
    from benchlib import pmc
    res, why = pmc.measure("cfg2", 32, distinct=1, timeout_s=60.0)
    assert res is None and isinstance(why, str) and why


def test_command_line_decides_which_runs_carry_the_side_regions():
    code = ("import sys; sys.argv = ['bench.py'] + %r; import bench; a = bench.parse_args(); "
            "print(__import__('json').dumps({k: getattr(a, k) for k in ('no_others', 'no_pmc', 'passes', 'steps', 'warmup', 'gpus', 'gather_every')}))")

    def parse(argv, env=None):
        p = subprocess.run([sys.executable, "-c", code % (argv,)], cwd=ROOT, capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        return json.loads(p.stdout.strip().splitlines()[-1])
    d = parse(["--gpus", "1", "--steps", "20", "--warmup", "5"])      # the driver's invocation: everything on
    assert d == {"no_others": False, "no_pmc": False, "passes": 20, "steps": 20, "warmup": 5, "gpus": 1, "gather_every": 0}
    assert parse(["--frames", "16", "--streams", "1"])["no_others"] is True          # a probe that names its own launch shape
    assert parse(["--calib"]) == dict(parse(["--calib"]), no_others=True, no_pmc=True, passes=1)
    n = parse(["--gpus", "8"])
    assert n["no_others"] is True and n["no_pmc"] is True
    prof = parse([], env=dict(os.environ, LD_PRELOAD="/opt/rocm/lib/librocprofiler-sdk-tool.so.DOES_NOT_EXIST"))      # under a profiler: no profiler inside
    assert prof["no_pmc"] is True and prof["no_others"] is True
