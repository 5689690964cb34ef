#!/usr/bin/env bash
# Run ON THE GPU BOX: time k_warp (and the whole step) for each A/B build under ab/ (developer probes; results of probe builds are WRONG pixels by design)
for lib in video-stitcher_amd/libmsstitch.so ab/*.so; do
  echo -n "$lib: "
  MSSTITCH_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 60 --warmup 10 --passes 1 --no-live --no-pcie --no-verify --streams 1 --frames 16 ${AB_ARGS:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_call']; print(d['value'], {a: round(b*1e3,1) for a,b in k.items() if b > 0.05})"
done
