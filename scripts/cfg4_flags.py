"""Path flags, key widths and merge times of cfg4-like frames (Velodyne-style sweeps), with and without the ROI."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from cloud_merger_amd import capi, synth
from cloud_merger_amd.types import MergeParams, xyzi_cloud
from cloud_merger_amd.replay_data import sensor_poses
POSES = sensor_poses(4)
def frame(f):
    out=[]
    for s in range(4):
        xyz, inten = synth.velodyne_frame(f, s)
        c = xyzi_cloud(xyz, inten)
        c.q_xyzw, c.t_xyz = POSES[s]
        out.append(c)
    return out
for name, params in (("5cm", MergeParams(leaf=(0.05,)*3, min_points_per_voxel=2)),
                     ("roi10cm", MergeParams(leaf=(0.1,)*3, min_points_per_voxel=2, crop_min=(-15,-5,-0.5), crop_max=(60,5,3)))):
    with capi.CloudMerger(max_points_total=600_000, max_sensors=4) as cm:
        fl=[]; t=[]
        for f in range(12):
            s = frame(f)
            cm.submit_all(s)
            t0=time.perf_counter(); res = cm.merge_voxelize(params); t.append(time.perf_counter()-t0)
            fl.append((res.path_flags, res.sort_passes, res.key_bits, int(res.n_merged), int(res.n_out)))
        print(name, fl[:4], fl[-1], "merge ms", [round(x*1e3,3) for x in t[-4:]])
