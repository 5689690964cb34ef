"""Run ON THE GPU BOX: the two intra-frame sharding schemes of SURVEY 8(e) on ONE GPU (all shards resident, no transfer: the compute cost of each split)
next to the unsharded context -> gpurun_out/<tag>_shards.json (copy into profiles/).    python tools/refresh_view_shards.py r02"""
import json, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
runs = {"cfg2_view_shards_2": ["--view-shards", "2"], "cfg5_view_shards_2": ["--config", "cfg5", "--view-shards", "2"],
        "cfg2_col_shards_2": ["--col-shards", "2"], "cfg2_col_shards_4": ["--col-shards", "4"],
        "cfg5_col_shards_2": ["--config", "cfg5", "--col-shards", "2"], "cfg5_col_shards_4": ["--config", "cfg5", "--col-shards", "4"],
        "cfg2_one_context_16": ["--frames", "16", "--streams", "1"], "cfg5_one_context_4": ["--config", "cfg5", "--frames", "4", "--streams", "1"]}
res = {}
for k, a in runs.items():
    p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-pcie", "--no-live", "--steps", "60"] + a, capture_output=True, text=True, timeout=300)
    res[k] = json.loads(p.stdout.strip().splitlines()[-1])
json.dump(res, open("gpurun_out/%s_shards.json" % tag, "w"), indent=1)
print({k: (v["value"], v.get("equals_unsharded")) for k, v in res.items()})
