#!/bin/bash
# Quick optimisation loop on the GPU box: the quantile tests, per-kernel stats of one frame alone (rocprofv3), the moving bench.
# usage: quick.sh TAG
TAG=$1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_quantile.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/${TAG}_pytest.log | head -20; exit 1; }
bash scripts/kstats_run.sh ${TAG} || exit 1
timeout -k 10 200 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-e2e > gpurun_out/${TAG}_bench_moving.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_moving.json"))
print("moving ms/step %.4f frac %.4f alone %.4f ms redone %s"%(d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["one_frame_alone"]["t_device_ms"],d["config"].get("redone_frames")))
PY
