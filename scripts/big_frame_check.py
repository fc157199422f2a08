"""One 32 M-point frame (4 x 8 M, cfg2 statistics) through both paths against the oracle: sizes of cfg5's order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloud_merger_amd import capi, synth
from oracle import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
sensors, params = synth.config2(n_per_sensor=n, min_pts=2)
t0 = time.time(); st, _, ref, rep = oracle.merge_voxelize(sensors, params, threads=8, stable=True, want_merged=False); t_cpu = time.time() - t0
for path in ("auto", "classic"):
    os.environ["CM_PATH"] = path
    with capi.CloudMerger(max_points_total=4 * n, max_sensors=4, flags=capi.FLAG_OCCUPANCY | capi.FLAG_PROFILE) as cm:
        for it in range(3):
            cm.submit_all(sensors)
            res = cm.merge_voxelize(params)
        out = cm.result(res.n_out); cells, counts = cm.cells(res.n_out)
    ok = res.n_out == rep.n_out and np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts)
    g = np.stack([out["x"], out["y"], out["z"]], 1); r = np.stack([ref["x"], ref["y"], ref["z"]], 1)
    print(path, "n_in", res.n_in, "n_out", res.n_out, "flags", res.path_flags, "passes", res.sort_passes, "device_ms %.3f" % res.device_ms,
          "occupancy", ok, "max|d| %.2e" % (np.abs(g.astype(np.float64) - r).max() if ok else -1), "cpu_s %.1f" % t_cpu)
