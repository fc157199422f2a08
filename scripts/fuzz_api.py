"""Long differential run of API SEQUENCES on one context against a small model of the slots plus the oracle: submits
(host buffers, device buffers), skipped submits under the first-wins policy, clears, new transforms, and merges (blocking
or async + wait) whose parameters change from call to call — crop on/off, outlier filter on/off, zone-wise ground
removal switched on/off with new slab tables, required-sensor masks (CM_NOT_READY), result copies in both layouts.
What it is after: state that leaks from one call into the next (masks, predicted boxes, flags, stale planes).
usage: python scripts/fuzz_api.py SECONDS [SEED0]   (CM_PATH=classic for the general path)."""
import dataclasses
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch                                   # device-resident submits; torch first: one HIP runtime per process
from cloud_merger_amd import capi, synth
from cloud_merger_amd.types import MergeParams, SensorCloud
from oracle import oracle
from util import assert_centroids_close_or_exact, same_bits, xyzi_of

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(5150 + seed0)
NS, CAP = 4, 200_000
latest_wins = bool(rng.random() < 0.5)
GP = dict(max_iterations=50, threshold=0.2, probability=0.99, optimize=True, z_keep_max=3.0, seed=7)


def new_cloud():
    n = int(rng.choice([0, 1, 50, 2000, 12_000, 40_000]))
    xyz = np.stack([rng.uniform(-15, 60, n), rng.uniform(-5, 5, n), rng.uniform(-0.4, 2.9, n)], 1).astype(np.float32)
    if n and rng.random() < 0.5:                                      # a ground sheet for the RANSAC slabs
        g = slice(0, n // 2)
        xyz[g, 2] = (-0.05 + 0.01 * xyz[g, 0] + 0.03 * rng.standard_normal(n // 2)).astype(np.float32)
    dense = True
    if n and rng.random() < 0.2:
        xyz[rng.integers(0, n, max(1, n // 50))] = np.nan
        dense = False
    data, lay = synth.pack(xyz, rng.uniform(0, 255, n).astype(np.float32), str(rng.choice(["xyzi16", "pcl32", "velo22", "xyz12"])))
    return SensorCloud(data=data, n=n, q_xyzw=np.array([0.0, 0.0, 0.0, 1.0]), t_xyz=np.zeros(3), is_dense=dense, **lay)


def new_pose():
    return synth.yaw_quaternion(float(rng.uniform(-0.05, 0.05))), rng.uniform(-0.5, 0.5, 3)


def new_zones():
    out = []
    for _ in range(NS):
        k = int(rng.integers(0, 5))
        edges = np.sort(rng.uniform(-15, 60, k + 1))
        out.append([(float(edges[i]), float(edges[i + 1] - edges[i]), float(rng.choice([-1.0, 0.3, 1.0, 2.0]))) for i in range(k)])
    return out


slots = [dict(cloud=None, fresh=False, q=np.array([0.0, 0.0, 0.0, 1.0]), t=np.zeros(3), keep=None) for _ in range(NS)]
zones = None
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"fuzz_api_{os.environ.get('CM_PATH', 'auto')}_{seed0}.log"), "w")
stats = dict(submits=0, skipped=0, merges=0, not_ready=0, ground=0, outlier=0, empty=0, redone=0, bucket=0)
t_end, step = time.time() + budget, 0
flags = capi.FLAG_OCCUPANCY | (capi.FLAG_LATEST_WINS if latest_wins else 0)
with capi.CloudMerger(max_points_total=CAP, max_sensors=NS, flags=flags) as cm:
    while time.time() < t_end:
        step += 1
        op = rng.choice(["submit", "submit", "submit", "submit_dev", "clear", "pose", "ground", "merge", "merge", "merge"])
        if op in ("submit", "submit_dev"):
            k = int(rng.integers(0, NS))
            c = new_cloud()
            if op == "submit":
                st, keep = cm.submit(k, c), None
            else:
                keep = torch.from_numpy(np.ascontiguousarray(c.data).view(np.uint8).reshape(-1).copy()).cuda() if c.n else torch.zeros(16, dtype=torch.uint8).cuda()
                torch.cuda.synchronize()
                st = cm.submit_device(k, keep.data_ptr(), c.n, c.point_step, c.off_x, c.off_y, c.off_z, c.off_i)
            want_skip = slots[k]["fresh"] and not latest_wins
            assert (st == capi.SKIPPED) == want_skip, (step, op, k, st, want_skip)
            stats["submits"] += 1
            stats["skipped"] += int(want_skip)
            if not want_skip:
                slots[k].update(cloud=c, fresh=True, keep=keep)
        elif op == "clear":
            k = int(rng.integers(0, NS))
            cm.clear(k)
            slots[k].update(cloud=None, fresh=False, keep=None)
        elif op == "pose":
            k = int(rng.integers(0, NS))
            q, t = new_pose()
            cm.set_transform(k, q, t)
            slots[k].update(q=q, t=t)
        elif op == "ground":
            zones = new_zones() if rng.random() < 0.6 else None
            cm.set_ground_removal(None if zones is None else capi.make_ground_params(
                zones, GP["max_iterations"], GP["threshold"], GP["probability"], GP["optimize"], GP["z_keep_max"], GP["seed"]))
        else:
            leaf = float(rng.choice([0.05, 0.1, 0.4]))
            p = MergeParams(leaf=(leaf,) * 3, min_points_per_voxel=int(rng.choice([0, 2])))
            any_nan = any(s["cloud"] is not None and not s["cloud"].is_dense for s in slots)
            if rng.random() < 0.6 or any_nan or zones is not None:
                p.crop_min, p.crop_max = (-15.0, -5.0, -0.5), (60.0, 5.0, 3.0)
            if zones is None and rng.random() < 0.3:
                p.outlier_radius, p.outlier_min_neighbors = float(rng.choice([0.15, 0.6])), int(rng.choice([1, 2]))
            have = [k for k in range(NS) if slots[k]["cloud"] is not None]
            mask = int(rng.integers(0, 1 << NS)) if rng.random() < 0.4 else 0
            p.required_sensor_mask = mask
            required = mask if mask else sum(1 << k for k in have)
            fresh = sum(1 << k for k in range(NS) if slots[k]["fresh"])
            want_not_ready = (not have) or bool(required & ~fresh)
            if rng.random() < 0.5:
                res = cm.merge_voxelize(p)
            else:
                st = cm.merge_voxelize_async(capi.make_params(p))
                if st == capi.NOT_READY:
                    res = None
                else:
                    assert st == capi.OK, (step, st)
                    res = cm.wait()
            got_not_ready = res is None or res.status == capi.NOT_READY
            assert got_not_ready == want_not_ready, (step, "gate", have, mask, fresh, got_not_ready)
            stats["merges"] += 1
            if want_not_ready:
                stats["not_ready"] += 1
                continue
            for s in slots:
                s["fresh"] = False
            # what the frame must be: every slot holding a cloud, in slot order, with the slot's current pose
            sensors = [dataclasses.replace(slots[k]["cloud"], q_xyzw=slots[k]["q"], t_xyz=slots[k]["t"]) for k in have]
            ctx = (step, "frame", have, [c.n for c in sensors], "crop", p.crop_min is not None, "outlier", p.outlier_radius,
                   "ground", zones is not None, "flags", res.path_flags)
            n_total = max(1, sum(c.n for c in sensors))
            if zones is not None:
                stats["ground"] += 1
                # the oracle's composition numbers sensors by list position: hand it the tables in that order, and
                # make the sample generator's zone keys the slots' (ground_split takes the sensor number)
                want_ng, want_g = [], []
                for k, c in zip(have, sensors):
                    # this sensor's cloud after ingest + transform + ROI crop, by the oracle's own path
                    cp = oracle.merge_voxelize([c], MergeParams(leaf=(1.0,) * 3, crop_min=p.crop_min, crop_max=p.crop_max), stable=True)[1]
                    keepm, gr, _ = oracle.ground_split(cp, zones[k], k, GP)
                    want_ng.append(cp[keepm]); want_g.append(cp[gr])
                merged = np.concatenate(want_ng)
                st_o, out, rep = oracle.voxelgrid(merged, p.leaf, p.min_points_per_voxel, stable=True)
                gg = cm.ground(n_total)
                assert same_bits(np.stack([gg[f] for f in ("x", "y", "z", "intensity")], 1), xyzi_of(np.concatenate(want_g))), (ctx, "ground cloud")
            else:
                st_o, merged, out, rep = oracle.merge_voxelize(sensors, p, threads=4, stable=True)
                stats["outlier"] += int(p.outlier_radius is not None)
            assert res.status == st_o, (ctx, res.status, st_o)
            mg = cm.merged(n_total)
            assert same_bits(np.stack([mg[f] for f in ("x", "y", "z", "intensity")], 1), xyzi_of(merged)), (ctx, "merged cloud")
            if st_o == oracle.OK:
                n_out = len(out) if zones is not None else rep.n_out
                assert res.n_out == n_out, (ctx, res.n_out, n_out)
                o16 = cm.result(res.n_out)
                got = np.stack([o16[f] for f in ("x", "y", "z", "intensity")], 1)
                cells, counts = cm.cells(res.n_out)
                assert np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts), (ctx, "occupancy")
                assert_centroids_close_or_exact(got, xyzi_of(out), rep.counts, rep.cells, merged, p.leaf, sequential=bool(res.path_flags & 2) and not (res.path_flags & 32))
                if rng.random() < 0.3:                                   # the 32-byte pcl::PointXYZI image of the same result
                    o32 = cm.result(res.n_out, point_step=32)
                    assert same_bits(o32[:, [0, 1, 2, 4]], got) and np.all(o32[:, 3] == 1.0) and not o32[:, 5:].any(), (ctx, "pcl32 image")
            else:
                stats["empty"] += 1
            stats["redone"] += int(bool(res.path_flags & 8))
            stats["bucket"] += int(bool(res.path_flags & 2))
        if step % 200 == 0:
            log.write(f"{step} steps {stats}\n"); log.flush()
print("api fuzz:", os.environ.get("CM_PATH", "auto"), "latest_wins", latest_wins, "steps", step, stats)
