#!/bin/bash
# Static instruction counts (VALU / packed / SALU / LDS / VMEM) of every bucket-path kernel, from the gfx950 assembly
# (cross-compiles, needs no GPU). usage: bash scripts/isa_counts.sh [file-stem ...]   (default: cm_kernels_v2 cm_kernels_v3)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${ISA_OUT:-/tmp/isa}
mkdir -p $OUT
STEMS=${@:-cm_kernels_v2 cm_kernels_v3 cm_kernels_v4}
for f in $STEMS; do
  (cd $OUT && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-bitwise-instead-of-logical \
    -I $ROOT/cloud_merger_amd/csrc -c $ROOT/cloud_merger_amd/csrc/$f.hip -o $OUT/$f.o -save-temps=obj 2>/dev/null)
  python3 - $OUT/$f-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import re,collections,subprocess,sys
lines=open(sys.argv[1]).read().split("\n")
cur=None; stats={}
for l in lines:
    m=re.match(r"^(_Z[A-Za-z0-9_]+):",l)
    if m: cur=m.group(1); stats[cur]=collections.Counter(); continue
    if l.startswith(".Lfunc_end"): cur=None
    if cur is None: continue
    t=l.strip()
    if not l.startswith("\t") or not t or t[0] in ".;": continue
    x=t.split()[0]; c=stats[cur]
    if x.startswith("v_pk"): c["pk"]+=1
    if x in ("v_mul_lo_u32","v_mul_hi_u32","v_mad_u64_u32","v_mad_i64_i32","v_mul_hi_i32"): c["slowmul"]+=1
    if x.startswith("v_"): c["valu"]+=1
    elif x.startswith("s_cbranch"): c["br"]+=1; c["salu"]+=1
    elif x.startswith("s_"): c["salu"]+=1
    elif x.startswith("ds_"): c["lds"]+=1
    elif x.split("_")[0] in ("global","buffer","flat","scratch"): c["vmem"]+=1
for k,c in stats.items():
    dn=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip()
    dn=dn.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:44]
    print(f"{dn:46s} valu {c['valu']:5d} (pk {c['pk']:3d}, quarter-rate mul {c['slowmul']:3d}) salu {c['salu']:5d} (branches {c['br']:4d}) lds {c['lds']:4d} vmem {c['vmem']:4d}")
PY
done
