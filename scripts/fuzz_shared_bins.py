"""Differential run of the quantile pass ABOVE 2048 buckets (shared bins: cm_device.h cm_quant_sub_shift, k3_local<SUB>) on one
persistent context: cfg3's dense scene, every frame a different uniform subsample of each sensor's 2 M points (0.1 ... 2 M: the
frame moves between 2048 buckets, two to a bin and four to a bin), poses drifting in the plane, now and then a jump in z (the
index is z-major: the buckets no longer fit, the frame is handed back) or a tiny frame; min_points_per_voxel and downsample_all_data random, now and then a sensor without points. Every frame
against the oracle (occupancy bit-exact, centroids as tests/util.py).
usage: python scripts/fuzz_shared_bins.py SECONDS [SEED]   -> gpurun_out/fuzz_shared_bins_SEED.log"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cloud_merger_amd import capi, synth
from cloud_merger_amd.types import SensorCloud
from tests.test_quantile import frame_against_oracle, QUANTILE, REDONE, BUCKET

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(9090 + seed)
base, params = synth.config3_dense(min_pts=2)
n_cap = sum(s.n for s in base)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"fuzz_shared_bins_{seed}.log"), "w")


def quat_mul(a, b):
    x1, y1, z1, w1 = a
    x2, y2, z2, w2 = b
    return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])


stats = {"frames": 0, "quantile": 0, "redone": 0, "shift": {0: 0, 1: 0, 2: 0}}
drift, zoff, size = np.zeros(3), 0.0, 1.0
t_end = time.time() + budget
with capi.CloudMerger(max_points_total=n_cap, max_sensors=len(base), flags=capi.FLAG_OCCUPANCY) as cm:
    while time.time() < t_end:
        f = stats["frames"]
        u = rng.random()
        if u < 0.08:
            zoff += float(rng.choice([-0.02, 0.013, 0.031]))                    # a jump across voxel layers
        if u > 0.92:
            size = float(rng.choice([0.05, 0.35, 0.5, 1.0]))                    # another frame size from here on
        elif rng.random() < 0.5:
            size = float(np.clip(size * rng.uniform(0.85, 1.18), 0.05, 1.0))
        drift[:2] += rng.normal(0.0, 0.006, 2)
        yaw = float(rng.normal(0.0, 0.001))
        dq = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
        params.min_points_per_voxel = int(rng.choice([0, 1, 2, 2, 3]))
        params.downsample_all_data = bool(rng.random() < 0.8)
        absent = int(rng.integers(0, len(base))) if rng.random() < 0.1 else -1     # now and then a sensor delivers an empty cloud
        sensors = []
        for i, sc in enumerate(base):
            k = 0 if i == absent else int(sc.n * size * rng.uniform(0.9, 1.0))
            sensors.append(SensorCloud(data=sc.data[:k], n=k, q_xyzw=quat_mul(dq, np.asarray(sc.q_xyzw)),
                                       t_xyz=np.asarray(sc.t_xyz) + drift + np.array([0.0, 0.0, zoff]),
                                       point_step=sc.point_step, off_x=sc.off_x, off_y=sc.off_y, off_z=sc.off_z, off_i=sc.off_i))
        res, rep = frame_against_oracle(cm, sensors, params, n_cap)
        stats["frames"] += 1
        stats["quantile"] += 1 if res.path_flags & QUANTILE else 0
        stats["redone"] += 1 if res.path_flags & REDONE else 0
        if res.path_flags & QUANTILE:
            nb = -(-rep.n_merged // 1920)
            stats["shift"][0 if rep.n_merged <= 2048 * 2600 else 1 if nb <= 4096 else 2] += 1     # (by this frame's size: about the route it took)
        log.write(f"frame {f}: n_in {res.n_in} kept {rep.n_merged} out {rep.n_out} flags {res.path_flags} passes {res.sort_passes} "
                  f"min_pts {params.min_points_per_voxel} z {zoff:+.3f}\n"); log.flush()
print("fuzz_shared_bins", stats)
