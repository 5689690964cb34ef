#!/usr/bin/env bash
# Run ON THE GPU BOX (through gpurun): SQ / TA / TCP counter passes (one --pmc group per run, kernel trace only) over a
# short bench.py run; per-kernel means are written to gpurun_out/counters_$TAG.txt (copy into profiles/ to commit).
#   gpurun -- 'bash tools/profile_counters.sh r01 cfg2 16'
set -uo pipefail
TAG=${1:-r01}; CFG=${2:-cfg2}; F=${3:-16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ctr_$TAG; mkdir -p $OUT
B="python bench.py --steps 6 --warmup 2 --passes 1 --no-live --no-pcie --no-verify --no-cpu-baseline --no-distinct --frames $F --streams 1 --config $CFG ${BENCH_EXTRA:-}"
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
         "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  case " ${PASSES:-1 2 3 6 7} " in *" $i "*) ;; *) continue;; esac     # passes 4/5 (TA/TCP groups) abort in rocprofv3 on this image
  timeout 300 rocprofv3 --kernel-trace --pmc $G -d $OUT/p$i -o p$i -- $B > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.txt
  db=$(find $OUT/p$i -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $OUT/p$i.txt 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import sys, glob, re
out, tag = sys.argv[1], sys.argv[2]
rows = {}
for f in sorted(glob.glob(out + "/p*.txt")):
    sect = False
    for l in open(f):
        if l.startswith("kernel") and "counter" in l:
            sect = True; continue
        if sect and l.strip():
            m = re.match(r"^(.{60}) (\S+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)", l)
            if m:
                rows.setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(3)), float(m.group(5)))
with open("gpurun_out/counters_%s.txt" % tag, "w") as fo:
    for k, d in rows.items():
        if not any(t in k for t in ("k_warp_t", "k_warp_s", "k_warp_a", "k_calib", "k_blend", "k_down", "k_stage1", "k_remap", "k_single", "k_resize_linear3")):
            continue
        fo.write(k + "\n")
        for c, (v, dur) in sorted(d.items()):
            fo.write("    %-40s %18.1f   (mean launch %.1f us)\n" % (c, v, dur / 1e3))
print(open("gpurun_out/counters_%s.txt" % tag).read()[:200])
PY
rm -rf $OUT/p[0-9]*/ 2>/dev/null; find $OUT -name "*.db" -delete      # keep the per-pass text summaries only (gpurun merges <= 64 MiB)
