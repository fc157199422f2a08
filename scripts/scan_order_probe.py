"""cfg2 with the points of every cloud in scan order (sorted by elevation ring, then azimuth — what a spinning lidar
delivers) instead of the random permutation of synth.ground_scene: neighbours in memory are neighbours in space, so the
lanes of a wave meet on the same counters. Stage times of both, one frame alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth


def scan_order(sensors):
    out = []
    for s in sensors:
        a = s.data.copy()
        r = np.hypot(a["x"], a["y"])
        ring = np.floor(np.degrees(np.arctan2(a["z"], r)) / 0.4).astype(np.int64)        # 0.4 degree rings
        az = np.arctan2(a["y"], a["x"])
        order = np.lexsort((az, ring))
        s2 = type(s)(**{**s.__dict__, "data": np.ascontiguousarray(a[order])})
        out.append(s2)
    return out


for min_pts in (2, 0):
    sensors, params = synth.config2(min_pts=min_pts)
    for name, ss in (("random order", sensors), ("scan order", scan_order(sensors))):
        with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_PROFILE) as cm:
            ts = []
            for it in range(8):
                cm.submit_all(ss)
                r = cm.merge_voxelize(params)
                ts.append(round(r.device_ms * 1e3))
            acc = {}
            for n_, ms in cm.stage_times():
                acc.setdefault(n_, []).append(ms * 1e3)
            print(f"min_pts {min_pts} {name:13s}: status {r.status} n_out {r.n_out} flags {r.path_flags} us/frame {ts[2:]}")
            print("      ", {n_: (f"{len(v)} x {sum(v) / len(v):.1f}" if len(v) > 1 else round(v[0], 1)) for n_, v in acc.items()})
