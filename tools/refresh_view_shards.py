import json, subprocess, sys
runs = {"cfg2_local": ["--view-shards", "2"], "cfg5_local": ["--config", "cfg5", "--view-shards", "2"],
        "cfg2_one_context_16": ["--frames", "16", "--streams", "1"], "cfg5_one_context_4": ["--config", "cfg5", "--frames", "4", "--streams", "1"]}
res = {}
for k, a in runs.items():
    p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "60"] + a, capture_output=True, text=True, timeout=300)
    res[k] = json.loads(p.stdout.strip().splitlines()[-1])
json.dump(res, open("gpurun_out/r01_view_shards.json", "w"), indent=1)
print({k: (v["value"], v.get("equals_unsharded")) for k, v in res.items()})
