#!/usr/bin/env bash
# Run ON THE GPU BOX: per-kernel us per call (one context, 32 frames) and the default three-context frames/s for the production library and every build under ab/,
# ROUNDS times alternating (same box, interleaved).  AB_LIBS="w1 b1" restricts the builds, AB_CFGS="cfg2 cfg3 shipped" the configurations.
ROUNDS=${ROUNDS:-2}
libs="video-stitcher_amd/libmsstitch.so"
for n in ${AB_LIBS:-$(ls ab | sed 's/\.so$//')}; do libs="$libs ab/$n"; done      # an entry may carry one environment assignment for a -DMS_DEV_KNOBS build: name@MS_WARP_LDS=0
for cfg in ${AB_CFGS:-cfg2}; do
  for r in $(seq $ROUNDS); do
    for ent in $libs; do
      lib=${ent%%@*}; [ "$lib" = "${lib%.so}" ] && lib=$lib.so
      kv=""; [ "$ent" != "${ent%%@*}" ] && kv=${ent#*@}
      [ -n "$kv" ] && export "$kv"
      echo -n "[$r] $cfg $ent: "
      MSSTITCH_LIB=$PWD/$lib python bench.py --config $cfg --no-cpu-baseline --steps 40 --warmup 5 --passes 2 --no-live --no-pcie --no-verify --no-distinct --streams 1 --frames ${AB_FRAMES:-32} --recalib-every 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_call']; print(round(d['value']), {a: round(b*1e3,1) for a,b in k.items() if b > 0.03})"
      if [ "$cfg" = cfg2 ] && [ -z "${AB_NO3:-}" ]; then
        echo -n "[$r] $cfg $ent 3x32: "
        MSSTITCH_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-others --no-pmc --steps 20 --warmup 5 --no-live --no-pcie --no-verify --no-distinct 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))"
      fi
      [ -n "$kv" ] && unset "${kv%%=*}"
    done
  done
done
