#!/bin/bash
# Compile one csrc/*.hip with the kernel-resource-usage remarks and print VGPRs / spills of kernels matching $2.
# usage: tools/kres.sh inception k_mlp_pos
set -e
cd "$(dirname "$(readlink -f "$0")")/../flow-timesnet_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -c $1.hip -o $1.o ${KRES_FLAGS} 2> /tmp/$1_build.log || { grep -E "error" -A5 /tmp/$1_build.log | head -60; exit 1; }
python3 - "$1" "$2" <<'PY'
import re, sys
name, pat = sys.argv[1], sys.argv[2]
cur = None; rows = {}
for line in open(f"/tmp/{name}_build.log"):
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"\s(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1)] = int(m.group(2))
import subprocess
for k, v in rows.items():
    dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("void ", "").split("(")[0]
    if pat in dem:
        print(f"{dem:70s} V={v.get('VGPRs')} A={v.get('AGPRs')} spillV={v.get('VGPRs Spill')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')} S={v.get('TotalSGPRs')}")
PY
