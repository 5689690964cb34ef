#!/usr/bin/env python3
"""Per-kernel roofline report from the committed profiles: launch time (rocprofv3 kernel trace), HBM bytes (PMC, calibrated), VALU
instructions and occupancy (SQ counters), algorithmic bytes (bench.py) -> profiles/<tag>_report.md

  python tools/report.py r01"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    out = {}
    if not os.path.exists(path):
        return out
    for block in re.split(r"\n(?=\S)", open(path).read()):
        lines = block.strip().split("\n")
        d = {}
        for x in lines[1:]:
            m = re.match(r"\s+(\S+)\s+([\d.]+)\s+\(mean launch ([\d.]+)", x)
            if m:
                d[m.group(1)] = float(m.group(2)); d["_us"] = float(m.group(3))
        if d:
            out[lines[0].strip()] = d
    return out


def short(n):
    return n.replace("void ", "").replace("ms::", "").split("(")[0]


def main(tag):
    prof = os.path.join(ROOT, "profiles")
    bench = json.load(open(os.path.join(prof, "%s_bench.json" % tag)))
    traffic = json.load(open(os.path.join(prof, "%s_traffic.json" % tag)))
    ctr = counters(os.path.join(prof, "%s_counters.txt" % tag))
    F = traffic["frames_per_launch"]
    clk = 2.33e9
    rows = []
    for k, v in traffic["kernels"].items():
        if k in ("k_warp", "k_blend_l0", "k_down_l0", "k_remap_gain") or v["launches"] < 20 or "calib" in k or not k.startswith(("k_warp_t", "k_warp_s", "k_blend", "k_down", "k_stage1", "k_remap", "k_single", "k_resize_linear3")):      # (aliases, calibration-time kernels)
            continue
        c = next((d for n, d in ctr.items() if short(n) == k), {})
        waves = c.get("SQ_WAVES", 0)
        valu = c.get("SQ_INSTS_VALU", 0)
        us = v["mean_ns"] / 1e3
        busy = valu * 4 / 1024 / (us * 1e-6 * clk) if valu else None
        occ = c.get("SQ_WAVE_CYCLES", 0) * 4 / (c.get("_us", us) * 1e-6 * clk) / 1024 if c.get("SQ_WAVE_CYCLES") else None
        rows.append((us * v["launches"], k, v["launches"], us, v["hbm_bytes_per_launch"], waves, valu / waves if waves else None, busy, occ))
    rows.sort(reverse=True)
    # launches per ms_stitch call: 1 for the per-frame kernels of the compositor; the per-frame resize of the shipped configuration needs ceil(views x F / 64) launches per call
    calls = max([r[2] for r in rows if r[1].startswith(("k_warp_t", "k_warp_s", "k_stage1_t", "k_stage1_s"))] or [1])
    lines = ["# Per-kernel report (%s): %s, %d frames per launch, one context / one stream" % (tag, traffic.get("config", "cfg2"), F), "",
             "Sources: `%s_kernel_trace.txt`/`%s_traffic.json` (rocprofv3 kernel trace; FETCH_SIZE x2.0 + WRITE_SIZE x1.0, calibrated on a 1 GiB copy),"
             " `%s_counters.txt` (SQ counters), `%s_bench.json` (bench.py line).  VALU busy = VALU instructions x 4 cycles / (1024 SIMDs x launch time x 2.33 GHz) -- an UPPER bound: on gfx950 the plain 32-bit fp32 / add / and / shift-right / move forms issue in ~2.4 cycles, everything else (conversions, packed 16-bit, three-operand integer, v_perm / v_alignbyte) in ~4.2 (`r04_valu_probe.txt`).  HBM MB = FETCH_SIZE x 2 + WRITE_SIZE: requests of the L2s to the fabric, Infinity-Cache hits included." % (tag, tag, tag, tag), "",
             "| kernel | mean launch µs | µs / frame | HBM MB / launch | HBM TB/s | waves | VALU / wave | VALU busy | waves / SIMD |", "|---|---|---|---|---|---|---|---|---|"]
    for _, k, n, us, hb, waves, vpw, busy, occ in rows:
        lines.append("| `%s` | %.1f | %.2f | %.0f | %.2f | %s | %s | %s | %s |" % (
            k + (" (x %d per call)" % round(n / calls) if n > 1.5 * calls and k.startswith("k_resize") else ""), us, us * max(1, round(n / calls)) / F if k.startswith("k_resize") else us / F, hb / 1e6, hb / (us * 1e-6) / 1e12, "%d" % waves if waves else "-", "%.0f" % vpw if vpw else "-",
            "%.2f" % busy if busy is not None else "-", "%.1f" % occ if occ is not None else "-"))
    r = bench["roofline"]
    lines += ["", "bench.py: **%.0f frames/s** (%s); dominant kernel `%s`: %.0f MB of PMC-measured HBM traffic / %.1f µs = %.0f GB/s = **%.3f of 8 TB/s** (physical; %s); "
              "model ratio on SURVEY 8(d)'s algorithmic bytes (%.0f MB per launch): %.3f." % (bench["value"], bench["config"]["workload"], r["kernel"], (r["traffic"] or 0) / 1e6, r["mean_launch_ms"] * 1e3,
                                                                                   r["achieved"] or 0.0, r["frac"] or 0.0, r.get("basis", "?"),
                                                                                   (bench.get("model", {}).get("kernel_alg_bytes_per_launch") or r.get("alg_bytes_per_launch", 0)) / 1e6,
                                                                                   bench.get("model", {}).get("ratio_kernel_alg_bytes_over_peak", r.get("frac_contract", 0.0))),
              "ceiling measured in the same run: tuned copy %.2f TB/s, read %.2f TB/s." % (bench["ceiling"]["copy_TBps"], bench["ceiling"]["read_TBps"]) if bench.get("ceiling") and "copy_TBps" in bench["ceiling"] else "",
              "cpu_baseline: %.1f frames/s on %d threads of %s (%s)." % (bench["cpu_baseline"]["value"], bench["cpu_baseline"]["cores"],
                                                                          bench["cpu_baseline"].get("cpu", "?"), bench["cpu_baseline"]["kind"]) if "cpu_baseline" in bench else ""]
    out = os.path.join(prof, "%s_report.md" % tag)
    open(out, "w").write("\n".join(lines) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
