#!/bin/bash
# SQ counters per kernel for one cfg2 frame sequence (scripts/dbg_stage.py); two --pmc passes of <= 8 counters.
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf gpurun_out/${TAG}_p$i
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/${TAG}_p$i -- python3 scripts/dbg_stage.py > gpurun_out/${TAG}_p$i.log 2>&1 || { tail -5 gpurun_out/${TAG}_p$i.log; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, re, sys, collections
tag=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for i in (1,2):
    f=glob.glob(f"gpurun_out/{tag}_p{i}/**/*counter_collection.csv", recursive=True)[0]
    seen=set()
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k[234]?_\w+)", r["Kernel_Name"])
        if not m: continue
        k=m.group(1)
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if i==1 and r["Counter_Name"]=="SQ_WAVE_CYCLES": cnt[k]+=1
for k,v in acc.items():
    n=max(cnt[k],1)
    print(k, "dispatches", n)
    for c in sorted(v): print("   %-24s %14.0f per dispatch" % (c, v[c]/n))
PY
