#!/usr/bin/env bash
# Run ON THE GPU BOX: frames per ms_stitch call against frames/s, alternating on one box (round 6: the per-call limit went from 32 to 64 frames).
#   gpurun -- 'bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt'
ROUNDS=${ROUNDS:-2}
run() {   # config streams frames
  python bench.py --config $1 --streams $2 --frames $3 --steps ${STEPS:-10} --warmup 3 --passes ${4:-10} --no-cpu-baseline --no-live --no-pcie --no-distinct --no-others --no-pmc 2>>gpurun_out/batch_sweep.err | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2x$(( $3 / $2 )):', d['value'], d['verified'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_call'].items()})"
}
for r in $(seq $ROUNDS); do
  for w in ${SWEEP:-cfg3:1:32:20 cfg3:1:48:14 cfg3:1:64:10 shipped:1:32:20 shipped:1:48:14 shipped:1:64:10 cfg2:3:96:20 cfg2:3:144:14 cfg2:3:192:10 cfg5:3:48:10 cfg5:3:96:6}; do
    IFS=: read -r a b c d <<< "$w"; run $a $b $c $d
  done
done
