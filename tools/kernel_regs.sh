#!/usr/bin/env bash
# Static resource usage (VGPRs, SGPRs, scratch, LDS, occupancy) of every kernel of a .hip file, from hipcc's device assembly (no GPU needed).
#   bash tools/kernel_regs.sh video-stitcher_amd/csrc/compositor.hip [name-filter]
set -euo pipefail
F=$1; FLT=${2:-}
S=/tmp/$(basename ${F%.*}).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt ${MS_EXTRA_FLAGS:-} -x hip --cuda-device-only -S "$F" -o "$S"
python3 - "$S" "$FLT" <<'PY'
import re, sys, subprocess
cur = None
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
    if m: cur = {"name": m.group(1)}; continue
    if cur is None: continue
    m = re.match(r"\s*\.amdhsa_next_free_vgpr (\d+)", line)
    if m: cur["vgpr"] = int(m.group(1))
    m = re.match(r"\s*\.amdhsa_next_free_sgpr (\d+)", line)
    if m: cur["sgpr"] = int(m.group(1))
    m = re.match(r"\s*\.amdhsa_private_segment_fixed_size (\d+)", line)
    if m: cur["scratch"] = int(m.group(1))
    m = re.match(r"\s*\.amdhsa_group_segment_fixed_size (\d+)", line)
    if m: cur["lds"] = int(m.group(1))
    m = re.match(r"\s*\.amdhsa_accum_offset (\d+)", line)
    if m: cur["accum"] = int(m.group(1))
    if ".end_amdhsa_kernel" in line: rows.append(cur); cur = None
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if sys.argv[2] and sys.argv[2] not in name: continue
    v = r.get("vgpr", 0)
    occ = min(8, 512 // max(1, (v + 7) // 8 * 8))
    print("%-70s vgpr %3d (arch %3d)  sgpr %3d  scratch %4d  lds %5d  waves/SIMD %d" % (name[:70], v, r.get("accum", 0), r.get("sgpr", 0), r.get("scratch", 0), r.get("lds", 0), occ))
PY
