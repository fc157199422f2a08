#!/bin/bash
for e in "" "CM_DBG_NOGRP=1"; do
  env $e CM_LOCAL_VARIANT=1 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --inflight 1 > gpurun_out/nogrp.json 2> gpurun_out/nogrp.err || tail -3 gpurun_out/nogrp.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/nogrp.json"))
k={x["name"]:x["avg_us"] for x in d["roofline"]["one_frame_alone"]["kernels"]}
print("$e", {n:round(v,1) for n,v in k.items()})
PY
done
