"""Where a cfg4-sized frame's host-side time goes (4 sensors x 120 k points from host memory, result to host memory)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth, replay_data
from cloud_merger_amd.types import MergeParams, xyzi_cloud
sensors = []
for s in range(4):
    xyz, inten = synth.velodyne_frame(0, s)
    sensors.append(xyzi_cloud(xyz, inten))
params = MergeParams(leaf=(0.05,) * 3, min_points_per_voxel=2)
n_total = sum(s.n for s in sensors)
cp = capi.make_params(params)
with capi.CloudMerger(max_points_total=n_total * 2, max_sensors=4) as cm:
    for k, s in enumerate(sensors): cm.set_transform(k, s.q_xyzw, s.t_xyz)
    T = {"submit": 0.0, "enqueue": 0.0, "wait": 0.0, "copy16": 0.0, "copy32": 0.0}
    N = 200
    for it in range(N + 20):
        t0 = time.perf_counter()
        for k, s in enumerate(sensors): cm.submit(k, s)
        t1 = time.perf_counter()
        cm.merge_voxelize_async(cp)
        t2 = time.perf_counter()
        res = cm.wait()
        t3 = time.perf_counter()
        out = cm.result(res.n_out)
        t4 = time.perf_counter()
        out32 = cm.result(res.n_out, 32)
        t5 = time.perf_counter()
        if it >= 20:
            T["submit"] += t1 - t0; T["enqueue"] += t2 - t1; T["wait"] += t3 - t2; T["copy16"] += t4 - t3; T["copy32"] += t5 - t4
    print("points", n_total, "voxels", res.n_out, {k: round(1e6 * v / N, 1) for k, v in T.items()}, "us per frame")
