#!/usr/bin/env python3
"""Turn the rocprofv3 databases written by tools/profile_traffic.sh into the committed evidence:
  profiles/<tag>_kernel_trace.txt   per-kernel calls/mean/min/max (kernel trace)
  profiles/<tag>_traffic.json       per-kernel HBM bytes per launch = (FETCH_SIZE*f_r + WRITE_SIZE*f_w) KB, where f_r/f_w are
                                    calibrated on k_calib_copy (a 1 GiB streaming copy in the same run), as the MI355X guide prescribes
  profiles/traffic_<config>.json    copy of the above per bench configuration (cfg2 / cfg3 / cfg5 / shipped), read by bench.py for roofline.traffic;
                                    traffic_latest.json = the cfg2 one.  MS_COMMIT (set by the caller: the GPU box has no .git) and the date are recorded."""
import json
import os
import shutil
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rocpd_stats  # noqa: E402


def counters(db, name):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, avg(value), count(*), avg(duration) from counters_collection where counter_name = ? group by kernel_name", (name,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def short(n):
    n = n.replace("void ", "").replace("ms::", "")
    return n.split("(")[0]


def csrc_sha16(root):      # same hash as bench.py's: the line flags a summary whose kernels have changed since
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "video-stitcher_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp", ".inc")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def build_summary(fetch, write, tag, cfg, frames, root):
    """fetch / write: {kernel name: (mean counter value in KB, launches, mean ns)} of the FETCH_SIZE / WRITE_SIZE passes -> the summary dict (per-kernel HBM bytes per launch,
    calibrated on the 1 GiB k_calib_copy of the same run).  Used by this script (profiles/) and by bench.py's in-run PMC pass (benchlib/pmc.py)."""
    GiB = float(1 << 30)
    cal_r = [v for k, v in fetch.items() if "k_calib_copy" in k]
    cal_w = [v for k, v in write.items() if "k_calib_copy" in k]
    f_r = GiB / (cal_r[0][0] * 1024.0) if cal_r and cal_r[0][0] > 0 else 2.0     # guide: FETCH_SIZE reads 1/2 on gfx950
    f_w = GiB / (cal_w[0][0] * 1024.0) if cal_w and cal_w[0][0] > 0 else 1.0
    import datetime
    res = {"tag": tag, "config": cfg, "frames_per_launch": int(frames), "unit": "bytes per launch",
           "collected": datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M UTC"), "commit": os.environ.get("MS_COMMIT", "?"), "csrc_sha16": csrc_sha16(root),
           "calibration": {"kernel": "k_calib_copy (1 GiB read + 1 GiB written, 16 B/lane, 4 non-temporal loads in flight: the tuned stream)",
                           "fetch_factor": f_r, "write_factor": f_w,
                           "FETCH_SIZE_KB_reported": cal_r[0][0] if cal_r else None, "WRITE_SIZE_KB_reported": cal_w[0][0] if cal_w else None,
                           "read_factor": f_r, "write_factor": f_w}, "kernels": {}}
    # the same GiB through the per-frame kernels' access shapes (k_calib_shape<0 / 1 / 2>: every 128-byte line touched once, so 1 GiB of HBM traffic each)
    labels = ["12 B / lane reads at a 24-byte stride (k_warp_t's tap read)", "8 B / lane contiguous reads (band kernels' row windows)",
              "dword stores in 32-byte runs, four passes per 128-byte line (k_warp_t's plane stores)"]
    cal = []
    for i, lab in enumerate(labels):
        fr = [v for k, v in fetch.items() if "k_calib_shape<%d>" % i in k]
        wr = [v for k, v in write.items() if "k_calib_shape<%d>" % i in k]
        fkb, wkb = (fr[0][0] if fr else None), (wr[0][0] if wr else None)
        cal.append({"shape": lab, "true_bytes": int(GiB), "FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb,
                    "fetch_factor_this_shape": (GiB / (fkb * 1024.0) if (fkb and i < 2) else None),
                    "write_factor_this_shape": (GiB / (wkb * 1024.0) if (wkb and i == 2) else None)})
    res["calibration"]["access_shapes"] = cal
    names = {"k_warp": "k_warp_t<false>", "k_remap_gain": "k_remap_gain"}
    order = sorted(set(fetch) | set(write))
    for k in order:
        if "ms::" not in k or "calib" in k or not short(k):
            continue
        fr = fetch.get(k, (0, 0, 0)); wr = write.get(k, (0, 0, 0))
        res["kernels"][short(k)] = {"fetch_KB_raw": fr[0], "write_KB_raw": wr[0], "launches": fr[1], "mean_ns": fr[2],
                                    "hbm_bytes_per_launch": int(fr[0] * 1024 * f_r + wr[0] * 1024 * f_w)}
    # bench.py names its per-level launches k_down_l<l> / k_blend_l<l>; map the dominant ones by kernel template
    alias = {}
    for k, v in res["kernels"].items():
        if k.startswith(("k_warp_t<false", "k_warp_s<false", "k_warp<false>")):
            alias["k_warp"] = v
        if k.startswith("k_blend8<true, 0>") or k.startswith("k_blend8<true>"):
            alias["k_blend_l0"] = v
        if k.startswith(("k_stage1_t", "k_stage1_s")):
            alias["k_remap_gain"] = v          # bench.py's name of the first CPW remap (timed.cpp:90-94)
        if k.startswith("k_resize_linear3"):
            alias["k_resize_batch"] = v        # bench.py's name of the per-frame cuda::resize launch (shipped configuration)
        if k.startswith(("k_warp_t<true", "k_warp_s<true")):
            alias["k_warp"] = v                # CPW contexts: the level-0 kernel is the mesh remap of the stage image
        if k.startswith("k_down_t<unsigned char>") or k.startswith("k_down_t<true>"):
            alias["k_down_l0"] = v
    # ms_stitch calls of the collection = launches of a kernel that goes out exactly ONCE per call (the level-0 reduce, the two tails, the level-0 band kernel).  The kernels that read
    # the callers' frames go out in chunks of 192 / views frames since round 6 (64 frames of six views: two launches per call): bench.py times all chunk launches of a call under one
    # name, so the aliases it reads carry the bytes of ALL of them (`launches_per_call`).
    once = [v["launches"] for k, v in res["kernels"].items() if k.startswith(("k_down_t<true>", "k_down_tail", "k_blend_tail", "k_blend8<true")) and v["launches"] > 0]
    calls = min(once) if once else max([v["launches"] for k, v in res["kernels"].items() if k.startswith(("k_warp_t<", "k_warp_s<", "k_warp_a", "k_warp<"))] or [1])
    for name in ("k_warp", "k_remap_gain"):
        if name in alias:
            per_call = max(1, int(round(alias[name]["launches"] / float(calls))))
            alias[name] = dict(alias[name], launches_per_call=per_call, hbm_bytes_per_launch=alias[name]["hbm_bytes_per_launch"] * per_call,
                               note="bytes of the %d chunk launch(es) of one ms_stitch call" % per_call)
    res["kernels"].update(alias)
    # HBM bytes of one ms_stitch call (all per-frame kernels): what bench.py's frame_roofline.frac_traffic divides by the GPU time
    per_frame = ("k_resize_linear3", "k_warp_t<", "k_warp_s<", "k_stage1_s", "k_warp_a", "k_warp<", "k_stage1_t", "k_remap_gain", "k_down_t", "k_down_tail", "k_down<", "k_blend8", "k_blend_tail",
                 "k_blend<", "k_blend_top", "k_single_band")
    # (the "<" matters: k_warp_tabs is a calibration kernel, launched once per view -- 12 times for the 12 x 4K rig, more often than a short collection launches the warp)
    steps = calls
    res["hbm_bytes_per_call"] = int(sum(v["hbm_bytes_per_launch"] * v["launches"] / steps for k, v in res["kernels"].items()
                                        if (k.startswith(per_frame) or k == "k_down") and k not in alias))
    res["calls"] = steps
    return res


def main(out, tag, cfg, frames):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    rocpd_stats.main(os.path.join(out, "trace", "trace_results.db"), os.path.join(prof, "%s_kernel_trace.txt" % tag))
    fetch = counters(os.path.join(out, "fetch", "fetch_results.db"), "FETCH_SIZE")
    write = counters(os.path.join(out, "write", "write_results.db"), "WRITE_SIZE")
    res = build_summary(fetch, write, tag, cfg, frames, root)
    path = os.path.join(prof, "%s_traffic.json" % tag)
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    shutil.copyfile(path, os.path.join(prof, "traffic_%s.json" % cfg))
    if cfg == "cfg2":
        shutil.copyfile(path, os.path.join(prof, "traffic_latest.json"))
    print(json.dumps(res["calibration"]))
    for k in ("k_warp", "k_down_l0", "k_blend_l0"):
        if k in res["kernels"]:
            print(k, res["kernels"][k])


if __name__ == "__main__":
    main(*sys.argv[1:5])
