"""cfg2's clouds at several voxel sizes on one context each: path flags, global passes and device time per frame — how the
bucket path adapts when voxels hold many points (coarse grids: first frames handed back, then more global passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth
for leaf in (0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0, 5.0):
    sensors, params = synth.config2(min_pts=0)
    params.leaf = (leaf,) * 3
    with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_PROFILE) as cm:
        rows = []
        for it in range(8):
            cm.submit_all(sensors)
            r = cm.merge_voxelize(params)
            rows.append((r.path_flags, r.sort_passes, round(r.device_ms * 1e3)))
        print(f"leaf {leaf}: status {r.status} n_out {r.n_out} key_bits {r.key_bits} (flags, passes, us) per frame: {rows}")
        acc = {}
        for n_, ms in cm.stage_times():                       # (a stage launched several times: launches x mean)
            acc.setdefault(n_, []).append(ms * 1e3)
        print("      last frame:", {n_: (f"{len(v)} x {sum(v) / len(v):.1f}" if len(v) > 1 else round(v[0], 1)) for n_, v in acc.items()})
