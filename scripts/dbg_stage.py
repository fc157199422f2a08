"""Stage times of one cfg2 frame (CM_FLAG_PROFILE), whatever its status: for timing experiments."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth
sensors, params = synth.config2(min_pts=2)
with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_PROFILE) as cm:
    acc = {}
    for it in range(6):
        cm.submit_all(sensors)
        res = cm.merge_voxelize(params)
        if it >= 2:
            for n, ms in cm.stage_times():
                acc.setdefault(n, []).append(ms * 1e3)
    print("status", res.status, "n_out", res.n_out, {n: round(float(np.mean(v)), 1) for n, v in acc.items()})
