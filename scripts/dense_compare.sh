#!/bin/bash
# usage: dense_compare.sh TAG — cfg3's dense variant (13.7 M records per frame) both ways in one GPU call (boxes differ by +-15 %):
# the default route (one quantile pass over shared bins, cm_device.h cm_quant_sub_shift) and CM_QUANT_SUB=0 (three fixed-grid
# passes), three frames in flight and one; rocprofv3 --kernel-trace --stats of the default route; the same on a moving stream.
T=${1:-dense}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
B="--config 3 --dense --no-cpu-baseline --no-e2e"
timeout -k 10 300 python3 bench.py $B > gpurun_out/${T}_dense_shared_bins.json 2> gpurun_out/${T}_dense.err || { tail -5 gpurun_out/${T}_dense.err; exit 1; }
CM_QUANT_SUB=0 timeout -k 10 300 python3 bench.py $B > gpurun_out/${T}_dense_fixed_grid.json 2>> gpurun_out/${T}_dense.err || exit 1
timeout -k 10 300 python3 bench.py $B --inflight 1 > gpurun_out/${T}_dense_shared_bins_inflight1.json 2>> gpurun_out/${T}_dense.err || exit 1
CM_QUANT_SUB=0 timeout -k 10 300 python3 bench.py $B --inflight 1 > gpurun_out/${T}_dense_fixed_grid_inflight1.json 2>> gpurun_out/${T}_dense.err || exit 1
rm -rf gpurun_out/${T}_dense_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_dense_prof -- python3 bench.py $B --inflight 1 --steps 30 --warmup 5 > /dev/null 2>> gpurun_out/${T}_dense.err || exit 1
cp $(find gpurun_out/${T}_dense_prof -name '*kernel_stats.csv' | head -1) gpurun_out/${T}_dense_shared_bins_kernel_stats.csv
rm -rf gpurun_out/${T}_dense_prof
python3 - $T <<'PY'
import json, sys, csv
t = sys.argv[1]
for n in ("shared_bins", "fixed_grid", "shared_bins_inflight1", "fixed_grid_inflight1"):
    d = json.loads(open(f"gpurun_out/{t}_dense_{n}.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("%-24s ms/step %.4f  alone %.4f  quantile %s redone %s  parity %s" % (n, d["ms_per_step"], c.get("latency_one_frame_ms", {}).get("host_enqueue_to_result", 0),
          c.get("quantile_frames"), c.get("redone_frames"), d.get("parity", c.get("parity"))))
for r in list(csv.DictReader(open(f"gpurun_out/{t}_dense_shared_bins_kernel_stats.csv")))[:8]:
    print("  %-56s calls %5s avg %8.1f us" % (r["Name"].split("(anonymous namespace)::")[-1][:56], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
# the same on a moving stream (six fresh draws of the scene in turn): how often the quantiles of one frame fail the next
B="$B --moving"
timeout -k 10 400 python3 bench.py $B > gpurun_out/${T}_dense_moving_shared_bins.json 2>> gpurun_out/${T}_dense.err || exit 1
CM_QUANT_SUB=0 timeout -k 10 400 python3 bench.py $B > gpurun_out/${T}_dense_moving_fixed_grid.json 2>> gpurun_out/${T}_dense.err || exit 1
python3 - $T <<'PY'
import json, sys
t = sys.argv[1]
for n in ("moving_shared_bins", "moving_fixed_grid"):
    d = json.loads(open(f"gpurun_out/{t}_dense_{n}.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("%-24s ms/step %.4f  alone %.4f  quantile %s redone %s" % (n, d["ms_per_step"], c.get("latency_one_frame_ms", {}).get("host_enqueue_to_result", 0),
          c.get("quantile_frames"), c.get("redone_frames")))
PY
