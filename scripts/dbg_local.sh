#!/bin/bash
# timing experiments on k2_local (CM_DBG bits: 1 one sort pass, 8 no sort pass, 2 stop before the reduce, 4 no look-back)
for d in 0 1 8 2 4 10 15; do
  CM_DBG=$d timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --inflight 1 > gpurun_out/dbg_$d.json 2> gpurun_out/dbg_$d.err || { tail -3 gpurun_out/dbg_$d.err; }
  python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/dbg_$d.json"))
    k={x["name"]:x["avg_us"] for x in d["roofline"]["one_frame_alone"]["kernels"]}
    print("dbg $d", {n:round(v,1) for n,v in k.items()}, "ms/step", round(d["ms_per_step"],4))
except Exception as e: print("dbg $d failed", e)
PY
done
