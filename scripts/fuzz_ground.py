"""Long differential run of the zone-wise ground removal on ONE persistent context: random scenes (tilted ground,
objects, degenerate slabs: empty, 1-3 points, collinear, identical points), random slab tables and RANSAC parameters,
1-6 sensors, with and without the per-slab outlier filter — each frame against the oracle's composition
(tests/test_ground.py::expected). usage: python scripts/fuzz_ground.py SECONDS [SEED0]; progress in gpurun_out/."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cloud_merger_amd import capi, synth
from cloud_merger_amd.types import MergeParams, xyzi_cloud
from oracle import oracle
from tests.util import assert_centroids_close_or_exact, same_bits, xyzi_of
from tests.test_ground import ROI, expected, run, a4, scene

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(777 + seed0)
CAP = 400_000


def cloud(rng):
    kind = rng.choice(["scene", "scene", "scene", "wide", "wide", "flat", "line", "same", "tiny", "empty", "steep"])
    n = int(rng.choice([300, 3000, 20_000, 60_000]))
    if kind == "scene":
        xyz = scene(rng, n, tilt=float(rng.uniform(-0.03, 0.03)), ground_sigma=float(rng.choice([0.0, 0.01, 0.05])),
                    obj_frac=float(rng.choice([0.0, 0.25, 0.9])))
    elif kind == "wide":                                   # a 360-degree scan: the ROI keeps a small part (survivor packing, CM_PATH_PACKED)
        xyz = scene(rng, n, tilt=float(rng.uniform(-0.01, 0.01)))
        xyz[:, 0] = rng.uniform(-120, 120, n).astype(np.float32)
        xyz[:, 1] = rng.uniform(-60, 60, n).astype(np.float32)
    elif kind == "flat":                                   # exact plane z = 0: every sample fits every point
        xyz = np.stack([rng.uniform(-15, 60, n), rng.uniform(-5, 5, n), np.zeros(n)], 1).astype(np.float32)
    elif kind == "line":                                   # collinear points: no sample gives a plane
        t = rng.uniform(-15, 60, n)
        xyz = np.stack([t, 0.1 * t - 1, 0.01 * t], 1).astype(np.float32)
    elif kind == "same":
        xyz = np.tile(np.array([[5.0, 1.0, 0.1]], np.float32), (n, 1))
    elif kind == "tiny":
        xyz = scene(rng, int(rng.integers(1, 6)))
    elif kind == "steep":                                  # a wall in the band: the "ground" plane is vertical
        xyz = np.stack([rng.uniform(10, 10.05, n), rng.uniform(-5, 5, n), rng.uniform(-0.4, 2.5, n)], 1).astype(np.float32)
    else:
        xyz = np.zeros((0, 3), np.float32)
    return xyzi_cloud(xyz, rng.uniform(0, 255, len(xyz)).astype(np.float32))


def slabs(rng):
    k = int(rng.integers(1, 9))
    edges = np.sort(rng.uniform(-15, 60, k + 1))
    z = [(float(edges[i]), float(edges[i + 1] - edges[i]) * float(rng.choice([1.0, 1.0, 0.5, 1.5])),   # gaps and overlaps
          float(rng.choice([-1.0, 0.3, 0.5, 1.5, 2.5]))) for i in range(k)]
    rng.shuffle(z)
    return z


os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"fuzz_ground_{os.environ.get('CM_PATH', 'auto')}_{seed0}.log"), "w")
t_end, frame, n_planes, n_found, n_packed = time.time() + budget, 0, 0, 0, 0
with capi.CloudMerger(max_points_total=CAP, max_sensors=6, flags=capi.FLAG_OCCUPANCY) as cm:
    while time.time() < t_end:
        n_s = int(rng.integers(1, 7))
        sensors = [cloud(rng) for _ in range(n_s)]
        for k in range(n_s, 6):
            cm.clear(k)
        zones = [slabs(rng) for _ in range(n_s)]
        gp = dict(max_iterations=int(rng.choice([1, 10, 100, 1000])), threshold=float(rng.choice([0.01, 0.1, 0.3])),
                  probability=float(rng.choice([0.5, 0.99])), optimize=bool(rng.random() < 0.7),
                  z_keep_max=float(rng.choice([3.0, 1.0])), seed=int(rng.integers(0, 2**31)))
        if rng.random() < 0.3:
            gp["outlier_radius"], gp["outlier_min_neighbors"] = float(rng.choice([0.15, 0.5])), int(rng.choice([1, 3]))
        leaf = float(rng.choice([0.05, 0.1, 0.5]))
        params = MergeParams(leaf=(leaf,) * 3, min_points_per_voxel=int(rng.choice([0, 2])), **ROI)
        want_ng, want_g, want_planes = expected(sensors, zones, params, gp)
        g = run(sensors, zones, params, gp, cm=cm)
        ctx = f"frame {frame} flags {g['res'].path_flags} gp {gp}"
        assert same_bits(a4(g["merged"]), xyzi_of(want_ng)), ("no-ground cloud", ctx)
        assert same_bits(a4(g["ground"]), xyzi_of(want_g)), ("ground cloud", ctx)
        for s, pls in enumerate(want_planes):
            for k, pl in enumerate(pls):
                got = g["planes"][s * 8 + k]
                if pl is None:
                    assert got.band_points == 0 and got.found == 0, ctx
                    continue
                n_planes += 1
                n_found += int(pl.found)
                assert got.found == pl.found and got.inliers == pl.n_inliers and got.iterations == pl.iterations, (ctx, s, k, (got.found, got.inliers, got.iterations, got.band_points, list(got.plane)), (pl.found, pl.n_inliers, pl.iterations, list(pl.plane)), g['res'].status, g['res'].n_in, [c.n for c in sensors], zones[s])
                if pl.found:
                    assert np.abs(np.array(got.plane) - np.array(pl.plane)).max() <= 1e-6, (ctx, s, k)
        st, vox, rep = oracle.voxelgrid(want_ng, params.leaf, params.min_points_per_voxel, stable=True)
        assert g["res"].status == st and g["res"].n_out == len(vox) and g["res"].n_merged == len(want_ng), ctx
        if st == oracle.OK:
            assert np.array_equal(g["cells"], rep.cells) and np.array_equal(g["counts"], rep.counts), ctx
            assert_centroids_close_or_exact(a4(g["out"]), xyzi_of(vox), rep.counts, rep.cells, want_ng, params.leaf,
                                            sequential=bool(g["res"].path_flags & 2) and not (g["res"].path_flags & 32))
        n_packed += int(bool(g["res"].path_flags & 16))
        frame += 1
        if frame % 10 == 0:
            log.write(f"{frame} frames ok, {n_planes} slabs with band points, {n_found} planes found\n"); log.flush()
print("ground fuzz: frames", frame, "slabs", n_planes, "planes found", n_found, "frames with packed survivors", n_packed)
