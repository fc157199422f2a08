#!/bin/bash
# usage: gpu_cycle.sh TAG  -> tests + bench + rocprof stats into gpurun_out/TAG_*
TAG=$1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/${TAG}_pytest.log | head -20; exit 1; }
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
timeout -k 10 300 python bench.py --min-pts 0 --no-cpu-baseline --no-e2e > gpurun_out/${TAG}_bench_mp0.json 2>> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof3 -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > gpurun_out/${TAG}_prof3_bench.json 2> gpurun_out/${TAG}_prof.err
f=$(find gpurun_out/${TAG}_prof3 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_default_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --inflight 1 > gpurun_out/${TAG}_prof_bench.json 2> gpurun_out/${TAG}_prof.err
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_kernel_stats.csv; cp "$f" gpurun_out/${TAG}_inflight1_kernel_stats.csv
python3 - <<PY
import json,csv
for n in ("bench","bench_mp0"):
    d=json.load(open("gpurun_out/${TAG}_%s.json"%n))
    print(n, "ms/step %.4f  Gpts/s %.2f  frac %.4f  alone: t_dev_ms %.4f frac %.4f M=%d"%(d["ms_per_step"], d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["one_frame_alone"]["t_device_ms"], d["roofline"]["one_frame_alone"]["frac"], d["config"]["voxels_out"]), d.get("parity"))
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:14]:
    import re
    m=re.search(r"(k[23]?_\w+(<[\w, ]+>)?|__amd\w+)", r["Name"]); nm=m.group(1) if m else r["Name"][:30]
    print("%-28s calls %5s avg_us %8.2f total_us_per_frame %8.2f"%(nm, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/max(1,int(r["Calls"]))))
PY
