#!/bin/bash
# usage: profiles_run.sh TAG [ROUND]   (on the GPU box, from the repository root) — everything profiles/ keeps for a round:
# kernel stats (rocprofv3 --kernel-trace --stats) of bench.py with one and with three frames in flight, PMC traffic
# (separate --pmc passes) for cfg2 (min 2 / min 0), cfg3 and its dense variant, SQ counters, the bench lines themselves.
TAG=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-e2e"
run_stats() {  # name, bench args...
  local N=$1; shift
  rm -rf gpurun_out/${TAG}_$N
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_$N -- python3 bench.py --steps 100 --warmup 10 $B "$@" > gpurun_out/${TAG}_${N}_bench.json 2> gpurun_out/${TAG}_$N.err || { tail -5 gpurun_out/${TAG}_$N.err; return 1; }
  cp "$(find gpurun_out/${TAG}_$N -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_${N}_kernel_stats.csv
}
run_stats cfg2_inflight1 --inflight 1 && run_stats cfg2_inflight3 && run_stats cfg2_minpts0_inflight1 --inflight 1 --min-pts 0 && \
run_stats cfg3_inflight1 --config 3 --inflight 1 && run_stats cfg3_dense_inflight1 --config 3 --dense --inflight 1 || exit 1
bash scripts/pmc_traffic.sh ${TAG}_pmc_cfg2_minpts2 && MP=0 bash scripts/pmc_traffic.sh ${TAG}_pmc_cfg2_minpts0 && \
BENCH_ARGS="--config 3" bash scripts/pmc_traffic.sh ${TAG}_pmc_cfg3_minpts2 && \
BENCH_ARGS="--config 3 --dense" bash scripts/pmc_traffic.sh ${TAG}_pmc_cfg3_dense_minpts2 || exit 1
# (bench.py quotes roofline.traffic from profiles/ when the file's source hash is this build's: put the fresh files there — on
# this box's copy of the tree — before the bench lines below are produced; keep_profiles.sh does the same in the build container)
R=${2:-r3}
cp gpurun_out/${TAG}_pmc_cfg2_minpts2_traffic.json profiles/${R}_pmc_traffic_cfg2_minpts2_bucket.json
cp gpurun_out/${TAG}_pmc_cfg2_minpts0_traffic.json profiles/${R}_pmc_traffic_cfg2_minpts0_bucket.json
cp gpurun_out/${TAG}_pmc_cfg3_minpts2_traffic.json profiles/${R}_pmc_traffic_cfg3_minpts2_bucket.json
cp gpurun_out/${TAG}_pmc_cfg3_dense_minpts2_traffic.json profiles/${R}_pmc_traffic_cfg3_dense_minpts2_bucket.json
bash scripts/pmc_sq.sh ${TAG}_sq > gpurun_out/${TAG}_sq_counters.txt 2>&1
timeout -k 10 400 python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
timeout -k 10 300 python3 bench.py --static $B > gpurun_out/${TAG}_bench_static.json 2>/dev/null
timeout -k 10 300 python3 bench.py --min-pts 0 $B > gpurun_out/${TAG}_bench_minpts0.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config 3 $B > gpurun_out/${TAG}_bench_cfg3.json 2>/dev/null
timeout -k 10 300 python3 bench.py --config 3 --dense > gpurun_out/${TAG}_bench_cfg3_dense.json 2>/dev/null
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_steps20_warmup5.json 2>/dev/null
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f)); a = d["roofline"]["one_frame_alone"]
        print("%-44s ms/step %.4f frac %.4f alone %.4f ms  M=%d redone %s" % (f.split("/")[-1], d["ms_per_step"], d["roofline"]["frac"], a["t_device_ms"], d["config"]["voxels_out"], d["config"].get("redone_frames")), d.get("parity"))
    except Exception as e:
        print(f, "unreadable", e)
PY
