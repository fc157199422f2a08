"""Where a k3_local tile's time goes, as thread 0 of every workgroup sees it; usage: phase_times3.py [min_pts [leaf]] (build with CM_PHASE_TIMING=1 python -m cloud_merger_amd.build --force; rebuild without it afterwards)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth
sensors, params = synth.config2(min_pts=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
if len(sys.argv) > 2:
    params.leaf = (float(sys.argv[2]),) * 3                # (coarser grids: longer voxels)
L = capi.load()
buf = (ctypes.c_ulonglong * (16 * 4096))()
with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_PROFILE) as cm:
    for it in range(6):
        cm.submit_all(sensors)
        res = cm.merge_voxelize(params)
        L.cm_debug_phases3(buf, 1)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
    v = a.sum(axis=0)
    n = float((a[:, 1] > 0).sum())
    names = ["start+load+keys", "a/ext", "sort", "heads+masks+sums of the block", "voxels past the block", "jobs (long voxels)"]
    tot = v[:len(names)].sum()
    print("tiles", int(n), "ticks(10ns)/tile", round(tot / n))
    for k, nm in enumerate(names):
        print(f"{nm:28s} {v[k] / n:9.0f} ticks  {100 * v[k] / tot:5.1f} %")
    print({n_: round(ms * 1e3, 1) for n_, ms in cm.stage_times()})
