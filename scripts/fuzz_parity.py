"""Long differential run on ONE persistent context: random frames (1-6 sensors, mixed layouts, clusters, NaNs, crop
boxes, outlier filter, sizes up to a few hundred thousand points) one after the other, each compared with the oracle —
exercises what carries over between frames (predicted box, extra global passes, hand-backs, back-offs), which the
per-scenario contexts of tests/test_gpu_parity.py::test_randomized_differential do not.
usage: python scripts/fuzz_parity.py SECONDS [SEED0 [SPIKE_PROBABILITY [CLUSTER_PROBABILITY]]]   (CM_PATH=classic for the general path);
progress in gpurun_out/. Spikes (thousands of points in one voxel) overflow the bucket path's tiles: with many of them
the context soon stays on the general path, so run once with SPIKE_PROBABILITY 0 for the bucket path itself."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cloud_merger_amd import capi, synth
from cloud_merger_amd.types import MergeParams, SensorCloud
from oracle import oracle
from test_gpu_parity import run_gpu, same_bits, xyzi_of, BUCKET, REDONE
from util import assert_centroids_close_or_exact

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p_spike = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
p_cluster = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5    # clouds with half their points in tight clusters (voxels of thousands of points)
BIG = os.environ.get("FUZZ_BIG") == "1"        # frames of up to 13 M points: the multi-group paths (> 8 M points), ~1 s of oracle each
CAP = 13_000_000 if BIG else 600_000
SIZES = [0, 5000, 300_000, 1_200_000, 3_000_000] if BIG else [0, 1, 7, 300, 5000, 9000, 40_000, 90_000]


def scenario(rng, frame):
    n_sensors = int(rng.integers(1, 5 if BIG else 7))
    layouts = ["xyzi16", "pcl32", "velo22", "xyz12"]
    # the scale drifts slowly and jumps sometimes: predicted boxes hold for a while, then miss
    scale = float(rng.choice([0.5, 3.0, 20.0])) if frame % 7 == 0 else scenario.scale * float(rng.uniform(0.9, 1.15))
    scenario.scale = scale
    sensors = []
    for _ in range(n_sensors):
        n = int(rng.choice(SIZES))
        xyz = rng.uniform(-scale, scale, (n, 3)).astype(np.float32)
        if n and rng.random() < p_cluster:
            xyz[: n // 2] = (rng.integers(-3, 4, (n // 2, 3)) * (scale / 4) + rng.normal(0, scale / 200, (n // 2, 3))).astype(np.float32)
        if n > 5000 and rng.random() < p_spike:               # a spike: thousands of points in one voxel (no-return points at the origin)
            xyz[: int(rng.integers(3000, 7000))] = rng.normal(0, 1e-4, 3).astype(np.float32)
        dense = True
        if n and rng.random() < 0.3:
            xyz[rng.integers(0, n, max(1, n // 50))] = np.nan
            dense = False
        data, lay = synth.pack(xyz, rng.uniform(0, 255, n).astype(np.float32), str(rng.choice(layouts)))
        q = synth.random_quaternion(rng) if rng.random() < 0.7 else np.array([0.0, 0.0, 0.0, 1.0])
        sensors.append(SensorCloud(data=data, n=n, q_xyzw=q, t_xyz=rng.uniform(-1, 1, 3), is_dense=dense, **lay))
    leaf = float(rng.choice([0.02, 0.1, 0.37, 1.0])) * max(scale / 3.0, 0.2)
    if BIG and p_cluster == 0.0:                               # voxels of about 0.5 / 3 / 20 points: the bucket path's territory
        density = max(1.0, sum(s.n for s in sensors)) / (2.0 * scale) ** 3
        leaf = float((float(rng.choice([0.5, 3.0, 20.0])) / density) ** (1.0 / 3.0))
    p = MergeParams(leaf=(leaf, leaf * float(rng.choice([1.0, 1.5])), leaf), min_points_per_voxel=int(rng.choice([0, 1, 2, 3])),
                    downsample_all_data=bool(rng.random() < 0.8))
    if rng.random() < 0.5 or any(not s.is_dense for s in sensors):
        c = scale * float(rng.choice([0.4, 0.9, 1.5]))
        p.crop_min, p.crop_max = (-c, -c, -c * 0.8), (c, c * 0.7, c)
    if rng.random() < 0.3:
        p.outlier_radius, p.outlier_min_neighbors = leaf * float(rng.choice([0.8, 2.0])), int(rng.choice([1, 2]))
        if BIG:                                                # keep the oracle's neighbour counting finite: about five neighbours per point
            density = max(1.0, sum(s.n for s in sensors)) / (2.0 * scale) ** 3
            p.outlier_radius = float((5.0 / (4.19 * density)) ** (1.0 / 3.0))
    return sensors, p


scenario.scale = 3.0
rng = np.random.default_rng(424242 + seed0)
tag = os.environ.get("CM_PATH", "auto")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"fuzz_{tag}_{seed0}_{p_spike}.log"), "w")
t_end, frame, stats = time.time() + budget, 0, {"bucket": 0, "redone": 0, "general": 0, "errors": 0}
with capi.CloudMerger(max_points_total=CAP, max_sensors=6, flags=capi.FLAG_OCCUPANCY) as cm:
    while time.time() < t_end:
        sensors, params = scenario(rng, frame)
        for k in range(len(sensors), 6):                   # sensors this frame does not use must not ride along stale
            cm.clear(k)
        verbose = os.environ.get("FUZZ_VERBOSE") == "1"
        t_f = time.time()
        if verbose:
            log.write(f"frame {frame}: {[s.n for s in sensors]} leaf {params.leaf} crop {params.crop_min} outlier {params.outlier_radius} min_pts {params.min_points_per_voxel}\n"); log.flush()
        st, merged, out, rep = oracle.merge_voxelize(sensors, params, threads=4, stable=True)
        if verbose:
            log.write(f"  oracle {time.time() - t_f:.1f} s, status {st}, n_out {rep.n_out}\n"); log.flush()
        try:
            g = run_gpu(sensors, params, cm=cm)
            if verbose:
                log.write(f"  gpu done {time.time() - t_f:.1f} s flags {g['res'].path_flags}\n"); log.flush()
        except capi.CloudMergeError as e:
            # device-detected capacity errors of the outlier grid have no oracle counterpart
            stats["errors"] += 1
            log.write(f"frame {frame}: {e}\n"); log.flush()
            frame += 1
            continue
        r = g["res"]
        ctx = f"frame {frame} flags {r.path_flags} n_in {r.n_in}"
        assert r.status == st, (ctx, capi.status_string(r.status), st)
        assert same_bits(g["merged"], xyzi_of(merged)), ctx
        if st == oracle.OK:
            assert r.n_out == rep.n_out, ctx
            assert np.array_equal(g["cells"], rep.cells) and np.array_equal(g["counts"], rep.counts), ctx
            d_gpu, d_orc = assert_centroids_close_or_exact(g["out"], xyzi_of(out), rep.counts, rep.cells, merged, params.leaf,
                                                           sequential=bool(r.path_flags & BUCKET) and not (r.path_flags & 32))   # 32: k3_local (long voxels in tree order)
            stats["max_dev_gpu"] = max(stats.get("max_dev_gpu", 0.0), d_gpu)
            stats["max_dev_oracle"] = max(stats.get("max_dev_oracle", 0.0), d_orc)
            if r.path_flags & BUCKET:
                small = rep.counts <= (17 if r.path_flags & 32 else 1 << 30)   # tests/util.py: SEQ_EXACT_MAX
                assert same_bits(g["out"][small], xyzi_of(out)[small]), ctx
        elif st == oracle.GRID_OVERFLOW:
            assert same_bits(g["out"], xyzi_of(out)), ctx
        stats["redone" if r.path_flags & REDONE else "bucket" if r.path_flags & BUCKET else "general"] += 1
        frame += 1
        if frame % (2 if BIG else 25) == 0:
            log.write(f"{frame} frames ok {stats}\n"); log.flush()
print(tag, "frames", frame, stats)
