#!/bin/bash
# bucket path vs classic path, 1..4 frames in flight (cfg2)
for inf in 1 2 3 4; do
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline --inflight $inf --min-pts ${MP:-2} > gpurun_out/inf_$inf.json 2>/dev/null
  CM_PATH=classic python bench.py --steps 40 --warmup 4 --no-cpu-baseline --inflight $inf --min-pts ${MP:-2} > gpurun_out/infc_$inf.json 2>/dev/null
  python3 -c "
import json
a=json.load(open('gpurun_out/inf_$inf.json')); b=json.load(open('gpurun_out/infc_$inf.json'))
print('inflight $inf  bucket %.4f ms  classic %.4f ms' % (a['ms_per_step'], b['ms_per_step']))
"
done
