"""Differential run with SEVERAL contexts busy at once on one GPU (the shape of bench.py's frames in flight and of the
replay driver's shards): each round every context gets its own random frame, all are enqueued (cm_merge_voxelize_async on
the contexts' own streams) before any is waited for, then each result is compared with the oracle. What it is after:
anything one context's kernels can do to another's (tickets and look-back with foreign workgroups on the CUs, shared
state). usage: python scripts/fuzz_concurrent.py SECONDS [SEED0 [N_CONTEXTS]]   (CM_PATH=classic for the general path)."""
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
from util import assert_centroids_close_or_exact, same_bits, xyzi_of

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(31337 + seed0)
CAP, NS = 1_300_000, 4


def scenario(scale):
    n_sensors = int(rng.integers(1, NS + 1))
    sensors = []
    for _ in range(n_sensors):
        n = int(rng.choice([300, 9000, 90_000, 300_000]))
        xyz = rng.uniform(-scale, scale, (n, 3)).astype(np.float32)
        data, lay = synth.pack(xyz, rng.uniform(0, 255, n).astype(np.float32), str(rng.choice(["xyzi16", "pcl32"])))
        q = synth.random_quaternion(rng) if rng.random() < 0.7 else np.array([0.0, 0.0, 0.0, 1.0])
        sensors.append(SensorCloud(data=data, n=n, q_xyzw=q, t_xyz=rng.uniform(-1, 1, 3), is_dense=True, **lay))
    density = sum(s.n for s in sensors) / (2.0 * scale) ** 3
    leaf = float((float(rng.choice([0.5, 3.0, 20.0])) / density) ** (1.0 / 3.0))
    p = MergeParams(leaf=(leaf,) * 3, min_points_per_voxel=int(rng.choice([0, 2])))
    if rng.random() < 0.5:
        c = scale * float(rng.choice([0.5, 0.9, 1.5]))
        p.crop_min, p.crop_max = (-c, -c, -c * 0.8), (c, c * 0.7, c)
        if rng.random() < 0.3:
            p.outlier_radius, p.outlier_min_neighbors = float((5.0 / (4.19 * density)) ** (1.0 / 3.0)), 1
    return sensors, p


os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"fuzz_concurrent_{os.environ.get('CM_PATH', 'auto')}_{seed0}.log"), "w")
cms = [capi.CloudMerger(max_points_total=CAP, max_sensors=NS, flags=capi.FLAG_OCCUPANCY) for _ in range(n_ctx)]
scales = [float(rng.choice([3.0, 20.0])) for _ in range(n_ctx)]
t_end, rounds, stats = time.time() + budget, 0, dict(frames=0, bucket=0, redone=0)
while time.time() < t_end:
    jobs = []
    for i, cm in enumerate(cms):
        if rng.random() < 0.1:
            scales[i] = float(rng.choice([3.0, 20.0]))          # a jump: the predicted box misses
        sensors, p = scenario(scales[i] * float(rng.uniform(0.95, 1.05)))
        for k in range(NS):
            cm.clear(k)
        cm.submit_all(sensors)
        jobs.append((cm, sensors, p))
    for cm, sensors, p in jobs:                                  # everything in flight before the first wait
        assert cm.merge_voxelize_async(capi.make_params(p)) == capi.OK
    for i, (cm, sensors, p) in enumerate(jobs):
        res = cm.wait()
        st, merged, out, rep = oracle.merge_voxelize(sensors, p, threads=4, stable=True)
        ctx = (rounds, i, [s.n for s in sensors], res.path_flags)
        assert res.status == st, (ctx, res.status, st)
        if st == oracle.OK:
            assert res.n_out == rep.n_out and res.n_merged == rep.n_merged, (ctx, res.n_out, rep.n_out)
            o = cm.result(res.n_out)
            got = np.stack([o[f] for f in ("x", "y", "z", "intensity")], 1)
            cells, counts = cm.cells(res.n_out)
            assert np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts), ctx
            assert_centroids_close_or_exact(got, xyzi_of(out), rep.counts, rep.cells, merged, p.leaf, sequential=bool(res.path_flags & 2) and not (res.path_flags & 32))
            if res.path_flags & 2:
                small = rep.counts <= (17 if res.path_flags & 32 else 1 << 30)     # tests/util.py: SEQ_EXACT_MAX
                assert same_bits(got[small], xyzi_of(out)[small]), ctx
        stats["frames"] += 1
        stats["bucket"] += int(bool(res.path_flags & 2))
        stats["redone"] += int(bool(res.path_flags & 8))
    rounds += 1
    if rounds % 5 == 0:
        log.write(f"{rounds} rounds {stats}\n"); log.flush()
for cm in cms:
    cm.close()
print("concurrent fuzz:", os.environ.get("CM_PATH", "auto"), "contexts", n_ctx, "rounds", rounds, stats)
