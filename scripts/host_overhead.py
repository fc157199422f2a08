"""Host time of the per-frame calls (4 x cm_submit_cloud_device + cm_merge_voxelize_async, then cm_wait) for cfg2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cloud_merger_amd import capi, synth
sensors, params = synth.config2(min_pts=2)
dev = torch.device("cuda", 0)
bufs = [torch.from_numpy(np.ascontiguousarray(s.data).view(np.uint8).reshape(-1)).to(dev) for s in sensors]
cp = capi.make_params(params)
with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4) as cm:
    for k, s in enumerate(sensors):
        cm.set_transform(k, s.q_xyzw, s.t_xyz)
    te, tw = [], []
    for it in range(60):
        t0 = time.perf_counter()
        for k, s in enumerate(sensors):
            cm.submit_device(k, bufs[k].data_ptr(), s.n)
        cm.merge_voxelize_async(cp)
        t1 = time.perf_counter()
        cm.wait()
        t2 = time.perf_counter()
        te.append(t1 - t0); tw.append(t2 - t1)
    print("enqueue (4 submits + async merge) median %.1f us; wait median %.1f us" % (1e6 * np.median(te[10:]), 1e6 * np.median(tw[10:])))
