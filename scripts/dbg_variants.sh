#!/bin/bash
# local-finish geometry sweep: alone (inflight 1) and three frames in flight
for v in 0 1 2; do for inf in 1 3; do
  CM_LOCAL_VARIANT=$v timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --inflight $inf --min-pts ${MP:-2} > gpurun_out/var_${v}_$inf.json 2> gpurun_out/var_${v}_$inf.err || { tail -3 gpurun_out/var_${v}_$inf.err; }
  python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/var_${v}_$inf.json"))
    k={x["name"]:x["avg_us"] for x in d["roofline"]["one_frame_alone"]["kernels"]}
    print("variant $v inflight $inf", {n:round(v,1) for n,v in k.items()}, "ms/step", round(d["ms_per_step"],4), "alone", round(d["roofline"]["one_frame_alone"]["t_device_ms"],4))
except Exception as e: print("variant $v failed", e)
PY
done; done
