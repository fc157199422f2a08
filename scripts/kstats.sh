#!/bin/bash
# per-kernel rocprofv3 stats for one frame alone (inflight 1); usage: kstats.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_prof -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --inflight 1 "$@" > gpurun_out/ks.json 2> gpurun_out/ks.err
f=$(find gpurun_out/ks_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,re,json
d=json.load(open("gpurun_out/ks.json")); print("ms/step %.4f alone t_dev %.4f"%(d["ms_per_step"], d["roofline"]["one_frame_alone"]["t_device_ms"]))
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"(k2?_\w+(<[\w, ]+>)?)", r["Name"])
    if m and "probe" not in m.group(1) and "setup" not in m.group(1): print("%-26s calls %4s avg_us %7.2f"%(m.group(1), r["Calls"], float(r["AverageNs"])/1e3))
PY
