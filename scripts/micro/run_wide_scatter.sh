#!/bin/bash
# usage: scripts/micro/run_wide_scatter.sh [TAG]   (on the GPU box, from the repository root)
TAG=${1:-ws}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BIN=scripts/micro/wide_scatter
timeout -k 10 200 $BIN > gpurun_out/${TAG}_times.txt 2>&1 || { tail -5 gpurun_out/${TAG}_times.txt; exit 1; }
cat gpurun_out/${TAG}_times.txt
for C in WRITE_SIZE FETCH_SIZE; do
  rm -rf gpurun_out/${TAG}_$C
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/${TAG}_$C -- $BIN > gpurun_out/${TAG}_$C.txt 2>&1 || { tail -5 gpurun_out/${TAG}_$C.txt; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != c: continue
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        res[k][c] = (sum(v) / len(v) * 1024 / 1e6, len(v))
with open(f"gpurun_out/{tag}_pmc.txt", "w") as o:
    for k in sorted(res):
        line = "%-60s " % k[:60] + "  ".join("%s %8.1f MB/launch (%d)" % (c, *res[k][c]) for c in res[k])
        print(line); o.write(line + "\n")
PY
