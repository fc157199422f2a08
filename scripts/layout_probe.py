"""Stage times of a cfg2 frame per wire layout of the clouds (16-byte XYZI, the 32-byte pcl::PointXYZI image, Velodyne's
unaligned 22-byte step): what the ingest paths of the streaming kernels cost."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from cloud_merger_amd import capi, synth
for layout in ("xyzi16", "pcl32", "velo22"):
    sensors, params = synth.config2(min_pts=2, layout=layout)
    with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_PROFILE) as cm:
        ts = []
        for it in range(8):
            cm.submit_all(sensors)
            r = cm.merge_voxelize(params)
            ts.append(round(r.device_ms * 1e3))
        acc = {}
        for n_, ms in cm.stage_times():
            acc.setdefault(n_, []).append(ms * 1e3)
        print(layout, "us/frame", ts[3:], {n_: (f"{len(v)} x {sum(v)/len(v):.1f}" if len(v) > 1 else round(v[0], 1)) for n_, v in acc.items()})
