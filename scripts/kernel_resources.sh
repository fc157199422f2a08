#!/bin/bash
# Prints VGPR / scratch / occupancy / LDS per kernel (hipcc -Rpass-analysis=kernel-resource-usage).
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math \
  -I /root/repo/cloud_merger_amd/csrc -c /root/repo/cloud_merger_amd/csrc/cm_kernels.hip -o /tmp/k.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=re.sub(r"^_ZN12_GLOBAL__N_1\d+","",m.group(1))[:34]; vals={}
    for key in ("VGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","LDS Size [bytes/block]"):
        m=re.search(re.escape(key)+r": (\d+)",l)
        if m and cur: vals[key]=m.group(1)
    if cur and len(vals)==4:
        print("%-36s vgpr %4s scratch %3s waves/SIMD %2s lds %6s"%(cur,vals["VGPRs"],vals["ScratchSize [bytes/lane]"],vals["Occupancy [waves/SIMD]"],vals["LDS Size [bytes/block]"])); cur=None
'
