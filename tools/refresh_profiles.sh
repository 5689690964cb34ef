#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): bench.py default + the other BASELINE configurations -> gpurun_out/profiles_out/ (copy into profiles/).
set -uo pipefail
TAG=${1:-r01}
OUT=gpurun_out/refresh; mkdir -p $OUT
timeout 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
python - "$OUT" "$TAG" <<'PY'
import json, subprocess, sys
out, tag = sys.argv[1], sys.argv[2]
runs = {"cfg2_F1": ["--frames", "1", "--streams", "1", "--steps", "30", "--warmup", "3"],
        "cfg2_F4": ["--frames", "4", "--streams", "1"],
        "cfg2_F16_1stream": ["--frames", "16", "--streams", "1"],
        "cfg3": ["--config", "cfg3"],
        "cfg3_no_recalibration": ["--config", "cfg3", "--recalib-every", "0"],
        "cfg5": ["--config", "cfg5"],
        "shipped": ["--config", "shipped"],
        "shipped_no_recalibration": ["--config", "shipped", "--recalib-every", "0"],
        "cfg2_egress_i420_1gpu": ["--emulate-gather"]}
res = {}
for k, a in runs.items():
    try:
        p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + a, capture_output=True, text=True, timeout=300)
        res[k] = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:
        res[k] = {"error": str(e)}
json.dump(res, open("%s/%s_other_configs.json" % (out, tag), "w"), indent=1)
print({k: v.get("value") for k, v in res.items()})
PY
