#!/bin/bash
# usage: profiles_extra.sh TAG — what profiles_run.sh does not collect: frames in flight 1 / 2, the stream without box misses, the leaf
# sweep, cfg5 (dense variant) at full size on one GPU, the live-node script, the fixed-grid passes alone for comparison.
TAG=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-e2e"
for v in "--inflight 1" "--inflight 2" "--jump-every 0"; do
  n=$(echo $v | tr -d ' -')
  timeout -k 10 200 python3 bench.py --steps 500 --warmup 50 $B $v > gpurun_out/${TAG}_bench_$n.json 2>/dev/null
done
CM_QUANT=0 timeout -k 10 200 python3 bench.py --steps 500 --warmup 50 $B > gpurun_out/${TAG}_bench_fixedgrid.json 2>/dev/null
timeout -k 10 200 python3 scripts/leaf_sweep.py > gpurun_out/${TAG}_leaf_sweep.txt 2>&1
timeout -k 10 500 python3 bench.py --config 5 --dense --steps 5 --warmup 2 --min-pts 2 > gpurun_out/${TAG}_cfg5_dense_one_gpu.json 2> gpurun_out/${TAG}_cfg5_dense.err || tail -3 gpurun_out/${TAG}_cfg5_dense.err
timeout -k 10 400 python3 bench.py --config 5 --steps 5 --warmup 2 --min-pts 2 > gpurun_out/${TAG}_cfg5_one_gpu.json 2> gpurun_out/${TAG}_cfg5.err || tail -3 gpurun_out/${TAG}_cfg5.err
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f)); a = d["roofline"]["one_frame_alone"]
        print("%-40s ms/step %.4f frac %.4f alone %.4f ms redone %s quantile %s" % (f.split("/")[-1], d["ms_per_step"], d["roofline"]["frac"], a["t_device_ms"], d["config"].get("redone_frames"), d["config"].get("quantile_frames")))
    except Exception as e:
        print(f, "unreadable", e)
for f in sorted(glob.glob("gpurun_out/${TAG}_cfg5*_one_gpu.json")):
    try:
        d = json.load(open(f)); c = d["config"]
        print("%-40s ms/step %.3f voxels_out %d table entries %d kept %s steps %s" % (f.split("/")[-1], d["ms_per_step"], c["voxels_out"], c["table_entries_this_rank"], c.get("points_kept_this_rank"), c["step_ms_worst_rank"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
bash scripts/live_node.sh ${TAG} | tail -28
