"""Differential run of the quantile passes at cfg2's scale on one persistent context: what a sensor stream can do to a context
whose grid stays put. A pool of cfg2 stream frames (fresh draws, jittered poses); every frame takes a random one and changes
it: a uniform subsample of 2 ... 100 % of every sensor's points (sizes jump by up to 50x between frames), a sensor without
points, NaNs in a non-dense cloud, a shift in z by a fraction of a voxel (the index is z-major: buckets overflow, hand-backs), a
frame that reaches further out (box miss), min_points_per_voxel / downsample_all_data at random, now and then a crop box or
another leaf for a few frames (the grid changes: fixed-grid passes, new splitters). Every frame against the oracle (occupancy
bit-exact, centroids as tests/util.py).
usage: python scripts/fuzz_quantile_stream.py SECONDS [SEED [N_PER_SENSOR]]   -> gpurun_out/fuzz_quantile_stream_SEED.log"""
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
nps = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
rng = np.random.default_rng(7070 + seed)
POOL = 4
pool = [synth.config2_stream(k + 11 * seed, n_per_sensor=nps, min_pts=2)[0] for k in range(POOL)]
wide = synth.config2_stream(500 + seed, n_per_sensor=nps, min_pts=2, wide=True)[0]
params = synth.config2(n_per_sensor=8, min_pts=2)[1]
n_cap = 4 * nps
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"fuzz_quantile_stream_{seed}.log"), "w")

stats = {"frames": 0, "quantile": 0, "redone": 0, "bucket": 0, "general": 0}
size, zoff, special, special_left = 1.0, 0.0, None, 0
t_end = time.time() + budget
with capi.CloudMerger(max_points_total=n_cap, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
    while time.time() < t_end:
        f = stats["frames"]
        u = rng.random()
        if u < 0.15:
            size = float(rng.choice([0.02, 0.05, 0.2, 0.5, 1.0]))               # another frame size from here on
        elif u < 0.6:
            size = float(np.clip(size * rng.uniform(0.8, 1.25), 0.02, 1.0))
        if rng.random() < 0.06:
            zoff += float(rng.choice([-0.05, 0.02, 0.031]))
        if special_left == 0 and rng.random() < 0.05:                           # a few frames on another grid
            special, special_left = str(rng.choice(["crop", "leaf"])), int(rng.integers(1, 4))
        src = wide if rng.random() < 0.03 else pool[int(rng.integers(0, POOL))]
        absent = int(rng.integers(0, 4)) if rng.random() < 0.08 else -1
        nan_in = int(rng.integers(0, 4)) if rng.random() < 0.1 else -1
        sensors = []
        for i, sc in enumerate(src):
            k = 0 if i == absent else max(1, int(sc.n * size * rng.uniform(0.85, 1.0)))
            data, dense = sc.data[:k], True
            if i == nan_in and k > 100:
                data = data.copy()
                data["x"][rng.integers(0, k, max(1, k // 200))] = np.nan
                dense = False
            sensors.append(SensorCloud(data=data, n=k, q_xyzw=sc.q_xyzw, t_xyz=np.asarray(sc.t_xyz) + np.array([0.0, 0.0, zoff]),
                                       point_step=sc.point_step, off_x=sc.off_x, off_y=sc.off_y, off_z=sc.off_z, off_i=sc.off_i, is_dense=dense))
        params.min_points_per_voxel = int(rng.choice([0, 1, 2, 2, 3]))
        params.downsample_all_data = bool(rng.random() < 0.8)
        params.leaf = (0.05,) * 3
        params.crop_min = params.crop_max = None
        if any(not s.is_dense for s in sensors):                                # (PCL needs a filter in front of non-dense clouds: the crop box is it)
            params.crop_min, params.crop_max = (-30.0, -30.0, -10.0), (30.0, 30.0, 10.0)
        if special_left:
            special_left -= 1
            if special == "crop":
                params.crop_min, params.crop_max = (-9.0, -7.0, -2.5), (8.0, 9.0, 3.0)
            else:
                params.leaf = (0.08, 0.08, 0.08)
        res, rep = frame_against_oracle(cm, sensors, params, n_cap)
        stats["frames"] += 1
        stats["quantile"] += 1 if res.path_flags & QUANTILE else 0
        stats["redone"] += 1 if res.path_flags & REDONE else 0
        stats["bucket" if res.path_flags & BUCKET else "general"] += 1
        log.write(f"frame {f}: n_in {res.n_in} kept {rep.n_merged} out {rep.n_out} flags {res.path_flags} passes {res.sort_passes} "
                  f"min_pts {params.min_points_per_voxel} z {zoff:+.3f} crop {params.crop_min is not None} leaf {params.leaf[0]}\n"); log.flush()
print("fuzz_quantile_stream", stats)
