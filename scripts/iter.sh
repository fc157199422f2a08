#!/bin/bash
# One optimisation iteration on the GPU box: a quick parity subset (both paths' core cases), per-kernel stats of one frame
# alone, and the pipelined bench (static and moving). usage: iter.sh TAG [full]   (full: the whole -m gpu suite first)
TAG=$1
mkdir -p gpurun_out
if [ "$2" = "full" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_pytest.log
else
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg1 or cfg2 or ragged or wire or dense or crop or misrank or predicted" > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/${TAG}_pytest.log
fi
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/${TAG}_pytest.log | head -20; exit 1; }
bash scripts/kstats_run.sh ${TAG} || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e --static > gpurun_out/${TAG}_bench_static.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e > gpurun_out/${TAG}_bench_moving.json 2>> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e --config 3 > gpurun_out/${TAG}_bench_cfg3.json 2>> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
for n in ("static","moving","cfg3"):
    d=json.load(open("gpurun_out/${TAG}_bench_%s.json"%n))
    print("%-7s ms/step %.4f frac %.4f alone %.4f ms redone %s parity %s"%(n,d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["one_frame_alone"]["t_device_ms"],d["config"].get("redone_frames"),d.get("parity")))
PY
