"""Path flags per frame of a cfg2 stream on one context (debugging aid: which frames take the quantile passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloud_merger_amd import capi, synth
n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
frames = [synth.config2_stream(k, n_per_sensor=n_per, min_pts=2) for k in range(6)]
with capi.CloudMerger(max_points_total=4 * n_per, max_sensors=4) as cm:
    out = []
    for i in range(24):
        sensors, params = frames[i % 6]
        cm.submit_all(sensors)
        r = cm.merge_voxelize(params)
        out.append((r.path_flags, r.sort_passes, r.n_out))
    print(out)
