"""Long differential run of the fused-cloud exchange on one GPU: 1-4 persistent contexts act as ranks, every frame's
sensors are dealt to them (fused.shard_sensors), each makes its partial table (cm_merge_partial, grid from the crop box
or from the all-reduced local bounds), rank 0 merges the tables (cm_merge_tables) — against the oracle on the whole
frame. usage: python scripts/fuzz_fused.py SECONDS [SEED0]   (CM_PATH=classic for the general path)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cloud_merger_amd import capi, fused, synth
from cloud_merger_amd.types import MergeParams, SensorCloud
from oracle import oracle
from util import assert_centroids_close_or_exact, xyzi_of

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(9090 + seed0)
CAP, MAXS = 400_000, 8


def scenario(rng):
    n_sensors = int(rng.integers(1, MAXS + 1))
    scale = float(rng.choice([0.5, 3.0, 20.0]))
    sensors = []
    for _ in range(n_sensors):
        n = int(rng.choice([0, 1, 7, 300, 5000, 20_000, 45_000]))
        xyz = rng.uniform(-scale, scale, (n, 3)).astype(np.float32)
        if n and rng.random() < 0.5:
            xyz[: n // 2] = (rng.integers(-3, 4, (n // 2, 3)) * (scale / 4) + rng.normal(0, scale / 200, (n // 2, 3))).astype(np.float32)
        dense = True
        if n and rng.random() < 0.3:
            xyz[rng.integers(0, n, max(1, n // 50))] = np.nan
            dense = False
        data, lay = synth.pack(xyz, rng.uniform(0, 255, n).astype(np.float32), str(rng.choice(["xyzi16", "pcl32", "velo22", "xyz12"])))
        q = synth.random_quaternion(rng) if rng.random() < 0.7 else np.array([0.0, 0.0, 0.0, 1.0])
        sensors.append(SensorCloud(data=data, n=n, q_xyzw=q, t_xyz=rng.uniform(-1, 1, 3), is_dense=dense, **lay))
    leaf = float(rng.choice([0.02, 0.1, 0.37, 1.0])) * max(scale / 3.0, 0.2)
    p = MergeParams(leaf=(leaf, leaf * float(rng.choice([1.0, 1.5])), leaf), min_points_per_voxel=int(rng.choice([0, 1, 2, 3])))
    if rng.random() < 0.5 or any(not s.is_dense for s in sensors):
        c = scale * float(rng.choice([0.4, 0.9, 1.5]))
        p.crop_min, p.crop_max = (-c, -c, -c * 0.8), (c, c * 0.7, c)
    return sensors, p


os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"fuzz_fused_{os.environ.get('CM_PATH', 'auto')}_{seed0}.log"), "w")
cms = [capi.CloudMerger(max_points_total=CAP, max_sensors=MAXS, flags=capi.FLAG_OCCUPANCY) for _ in range(4)]
t_end, frame, stats = time.time() + budget, 0, {"ok": 0, "empty": 0, "overflow": 0, "bucket_tables": 0}
while time.time() < t_end:
    sensors, params = scenario(rng)
    world = int(rng.integers(1, 5))
    st, merged, ref, rep = oracle.merge_voxelize(sensors, params, threads=4, stable=True)
    for r in range(world):
        mine = fused.shard_sensors(len(sensors), r, world)
        for k in range(MAXS):
            cms[r].clear(k)
        for k, s in enumerate(mine):
            cms[r].set_transform(k, sensors[s].q_xyzw, sensors[s].t_xyz)
            cms[r].submit(k, sensors[s])
    ranks = [r for r in range(world) if fused.shard_sensors(len(sensors), r, world)]
    ctx = f"frame {frame} world {world} sensors {[s.n for s in sensors]} crop {params.crop_min is not None}"
    bounds = None
    if params.crop_min is None:
        lb = [cms[r].local_bounds(params) for r in ranks]
        lb = [b for b in lb if b[2]]
        if lb:
            bounds = np.concatenate([np.min([b[0] for b in lb], axis=0), np.max([b[1] for b in lb], axis=0)]).astype(np.float32)
    if params.crop_min is None and bounds is None:
        assert st == oracle.EMPTY_INPUT, ctx               # no rank has a valid point: nothing to exchange
        stats["empty"] += 1
        frame += 1
        for r in ranks:                                    # consume the clouds
            cms[r].merge_voxelize(params)
        continue
    parts, overflow = [], False
    for r in ranks:
        try:
            res = cms[r].merge_partial(params, bounds)
        except capi.CloudMergeError as e:
            assert e.status == capi.GRID_OVERFLOW or "grid" in str(e).lower(), (ctx, str(e))
            overflow = True
            break
        if res.status == capi.GRID_OVERFLOW:
            overflow = True
            break
        assert res.status in (capi.OK, capi.EMPTY_INPUT), (ctx, res.status)
        stats["bucket_tables"] += 1 if res.path_flags & 2 else 0
        parts.append(cms[r].partial_device() if res.status == capi.OK else (0, 0))
    if overflow:
        assert st == oracle.GRID_OVERFLOW, (ctx, st)
        stats["overflow"] += 1
        frame += 1
        continue
    parts = [p for p in parts if p[1]]
    if not parts:
        assert st == oracle.EMPTY_INPUT, ctx
        stats["empty"] += 1
        frame += 1
        continue
    res = cms[0].merge_tables([p[0] for p in parts], [p[1] for p in parts], params)
    assert st == oracle.OK and res.status == capi.OK, (ctx, st, res.status)
    assert res.n_out == rep.n_out, (ctx, res.n_out, rep.n_out)
    if res.n_out:
        out = cms[0].result(res.n_out)
        cells, counts = cms[0].cells(res.n_out)
        if not (np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts)):
            bad = np.nonzero(np.any(cells != rep.cells, axis=1) | (counts != rep.counts))[0]
            raise AssertionError((ctx, "differing voxels", len(bad), "of", len(counts), "first", int(bad[0]), cells[bad[0]].tolist(), rep.cells[bad[0]].tolist(),
                                  int(counts[bad[0]]), int(rep.counts[bad[0]]), "bounds", None if bounds is None else bounds.tolist(),
                                  "res min_b", list(res.min_b), list(res.div_b), "oracle", list(rep.min_b), list(rep.div_b), "leaf", params.leaf, "minpts", params.min_points_per_voxel))
        got = np.stack([out["x"], out["y"], out["z"], out["intensity"]], axis=1)
        assert_centroids_close_or_exact(got, xyzi_of(ref), rep.counts, rep.cells, merged, params.leaf, sequential=False)
    stats["ok"] += 1
    frame += 1
    if frame % 25 == 0:
        log.write(f"{frame} frames {stats}\n"); log.flush()
for cm in cms:
    cm.close()
print("fused fuzz:", os.environ.get("CM_PATH", "auto"), "frames", frame, stats)
