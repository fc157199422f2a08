#!/bin/bash
# usage: kstats_run.sh TAG [bench args...] -> rocprofv3 --kernel-trace --stats of bench.py with one frame in flight; per-kernel table
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/${TAG}_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e --inflight 1 "$@" > gpurun_out/${TAG}_prof_bench.json 2> gpurun_out/${TAG}_prof.err || { tail -5 gpurun_out/${TAG}_prof.err; exit 1; }
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_inflight1_kernel_stats.csv
python3 - <<PY
import json,csv,re
d=json.load(open("gpurun_out/${TAG}_prof_bench.json"))
print("ms/step %.4f alone t_dev_ms %.4f M=%d"%(d["ms_per_step"], d["roofline"]["one_frame_alone"]["t_device_ms"], d["config"]["voxels_out"]))
rows=list(csv.DictReader(open("gpurun_out/${TAG}_inflight1_kernel_stats.csv")))
for r in rows[:12]:
    m=re.search(r"(k[234g]?_\w+(<[\w, ]+>)?|__amd\w+)", r["Name"]); nm=m.group(1) if m else r["Name"][:30]
    print("%-40s calls %5s avg_us %8.2f"%(nm, r["Calls"], float(r["AverageNs"])/1e3))
PY
