#!/usr/bin/env bash
# Run ON THE GPU BOX: the per-kernel times of one configuration against a pad allocation made before everything else (bench.py --pad-kb): are the kernels sensitive to where
# the buffers land?  usage: tools/placement_probe.sh <config> "<pad KiB list>"
cfg=${1:-shipped}
for pad in ${2:-0 0 1028 3100 9000 20000 0 50000 1028}; do
  echo -n "$cfg pad=${pad}KiB: "
  python bench.py --config $cfg --pad-kb $pad --no-cpu-baseline --steps 30 --warmup 5 --passes 2 --no-live --no-pcie --no-verify --no-distinct --streams 1 --frames 32 --recalib-every 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_call']; print(round(d['value']), {a: round(b*1e3,1) for a,b in k.items() if b > 0.1})"
done
