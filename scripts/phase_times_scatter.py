"""Where a k2_scatter tile's time goes (CM_PHASE_TIMING=1 build). The last scatter of the frame overwrites the first one's
stamps: argv[1] = number of global passes to look at is fixed by the frame (cfg2: stamps are those of pass 2; run with
CM_FINISH unset)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth
sensors, params = synth.config2(min_pts=2)
L = capi.load()
buf = (ctypes.c_ulonglong * (16 * 4096))()
with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_PROFILE) as cm:
    for it in range(6):
        cm.submit_all(sensors)
        res = cm.merge_voxelize(params)
        L.cm_debug_phases(buf, 1)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
    v = a.sum(axis=0)
    n = float((a[:, 1] > 0).sum())
    names = ["load(+transform+keys)", "before + totals scan", "clear + rank", "digit scan + bases", "stage/write round 1", "round 2"]
    for off, what in ((0, "first pass (raw points)"), (8, "later pass (records)")):
        tot = v[off:off + 6].sum()
        print(what, "tiles", int(n), "ticks(10ns)/tile", round(tot / n))
        for k, nm in enumerate(names):
            print(f"{nm:28s} {v[off + k] / n:9.0f} ticks  {100 * v[off + k] / tot:5.1f} %")
    print({n_: round(ms * 1e3, 1) for n_, ms in cm.stage_times()})
