"""Stage times of a frame with zone-wise ground removal (6 sensors, the reference's slabs) beside the oracle's CPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi
from cloud_merger_amd.types import MergeParams, xyzi_cloud
from oracle import oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_ground import FRONT, GP, ROI, scene, expected
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(5)
REAR = [(30.0, 30.0, 2.0), (4.0, 26.0, 1.5), (-4.0, 8.0, 0.3), (-15.0, 11.0, 0.5)]
zones = [FRONT, FRONT, REAR, REAR, [(20.0, 40.0, 1.0), (-15.0, 35.0, -1.0)],
         [(34.0, 26.0, 1.5), (24.0, 10.0, 1.2), (14.0, 10.0, 0.8), (4.0, 10.0, 0.5)]]
sensors = [xyzi_cloud(scene(rng, n), rng.uniform(0, 255, n)) for _ in range(6)]
params = MergeParams(leaf=(0.1,) * 3, min_points_per_voxel=2, **ROI)
t0 = time.time(); want_ng, want_g, _ = expected(sensors, zones, params, GP); t_cpu = time.time() - t0
with capi.CloudMerger(max_points_total=6 * n, max_sensors=6, flags=capi.FLAG_PROFILE) as cm:
    cm.set_ground_removal(capi.make_ground_params(zones))
    acc = {}
    for it in range(8):
        cm.submit_all(sensors)
        res = cm.merge_voxelize(params)
        if it >= 2:
            for name, ms in cm.stage_times():
                acc.setdefault(name, []).append(ms * 1e3)
    planes = cm.ground_planes()
print("points", 6 * n, "no-ground", res.n_merged, "(oracle", len(want_ng), ") voxels", res.n_out, "device_ms %.3f" % res.device_ms)
print({k: round(float(np.mean(v)), 1) for k, v in acc.items()})
print("oracle per-sensor stage (transform + crop + slabs + RANSAC, python-composed): %.1f ms" % (t_cpu * 1e3))
print("iterations per slab:", [planes[s * 8 + k].iterations for s in range(6) for k in range(len(zones[s])) if planes[s * 8 + k].band_points])
