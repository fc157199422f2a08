"""Stage times (CM_FLAG_PROFILE) of an outlier-filtered frame inside a crop box: cfg2's clouds, a box that
keeps most of them, r = 0.15 m. Run once per path: CM_PATH=classic python scripts/outlier_stage_times.py."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth

radius = float(sys.argv[1]) if len(sys.argv) > 1 else 0.15
sensors, params = synth.config2(min_pts=2)
params.crop_min, params.crop_max = (-60.0, -60.0, -6.0), (60.0, 60.0, 8.0)
params.outlier_radius, params.outlier_min_neighbors = radius, 1
with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_PROFILE) as cm:
    acc, tot = {}, []
    for it in range(8):
        cm.submit_all(sensors)
        res = cm.merge_voxelize(params)
        if it >= 3:
            tot.append(res.device_ms * 1e3)
            for n, ms in cm.stage_times():
                acc.setdefault(n, []).append(ms * 1e3)
    print("path", os.environ.get("CM_PATH", "auto"), "flags", res.path_flags, "status", res.status, "n_in", res.n_in,
          "n_merged", res.n_merged, "n_out", res.n_out, "frame_us", round(float(np.mean(tot)), 1))
    print({n: round(float(np.sum(v)) / len(tot), 1) for n, v in acc.items()})
