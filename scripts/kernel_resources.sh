#!/bin/bash
# Prints VGPR / scratch / occupancy / LDS per kernel of every kernel file (hipcc -Rpass-analysis=kernel-resource-usage);
# cross-compiles, needs no GPU. usage: bash scripts/kernel_resources.sh > profiles/rN_kernel_resources.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for f in cm_kernels cm_kernels_v2 cm_kernels_v3 cm_kernels_v4 cm_kernels_ground; do
  echo "== $f.hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math \
    -I $ROOT/cloud_merger_amd/csrc -c $ROOT/cloud_merger_amd/csrc/$f.hip -o /tmp/k_$f.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys,subprocess
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        try: cur=subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        except Exception: cur=m.group(1)
        cur=re.sub(r"\(anonymous namespace\)::","",cur); cur=re.sub(r"^void ","",cur); cur=cur.split("(")[0][:44]; vals={}
    for key in ("VGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","LDS Size [bytes/block]","SGPRs"):
        m=re.search(re.escape(key)+r": (\d+)",l)
        if m and cur and key not in vals: vals[key]=m.group(1)
    if cur and "LDS Size [bytes/block]" in vals:
        print("%-46s vgpr %4s sgpr %4s scratch %3s waves/SIMD %2s lds %6s"%(cur,vals.get("VGPRs"),vals.get("SGPRs","?"),vals.get("ScratchSize [bytes/lane]"),vals.get("Occupancy [waves/SIMD]"),vals["LDS Size [bytes/block]"])); cur=None
'
done
