"""Single fused cloud across GPUs: the table exchange on CPU with gloo
(world_size 2), and on the GPU two contexts standing in for two ranks."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from cloud_merger_amd import fused, synth
from cloud_merger_amd.capi import ENTRY_DTYPE
from cloud_merger_amd.types import MergeParams
from oracle import np_oracle, oracle
from tests.util import assert_centroids_close, xyzi_of


def crop_grid(params):
    inv = np.float32(1) / np.asarray(params.leaf, dtype=np.float32)
    lo = np.floor(np.asarray(params.crop_min, np.float32) * inv).astype(np.int64)
    hi = np.floor(np.asarray(params.crop_max, np.float32) * inv).astype(np.int64)
    return lo, hi - lo + 1


def cpu_partial_table(sensors, params):
    """What cm_merge_partial produces, restated with numpy (crop-box grid)."""
    xyz, inten = np_oracle.merge(sensors, params)
    lo, div = crop_grid(params)
    c = np_oracle.cells(xyz, params.leaf) - lo
    key = (c[:, 0] + c[:, 1] * div[0] + c[:, 2] * div[0] * div[1]).astype(np.uint32)
    order = np.argsort(key, kind="stable")
    key, vals = key[order], np.concatenate([xyz, inten[:, None]], axis=1)[order]
    head = np.flatnonzero(np.r_[True, key[1:] != key[:-1]]) if len(key) else np.zeros(0, np.int64)
    t = np.zeros(len(head), dtype=ENTRY_DTYPE)
    if len(head):
        sums = np.add.reduceat(vals.astype(np.float64), head, axis=0).astype(np.float32)
        t["key"], t["count"] = key[head], np.diff(np.r_[head, len(key)])
        t["sx"], t["sy"], t["sz"], t["si"] = sums[:, 0], sums[:, 1], sums[:, 2], sums[:, 3]
    return t


def decode_cells(keys, params):
    lo, div = crop_grid(params)
    k = keys.astype(np.int64)
    return np.stack([k % div[0] + lo[0], (k // div[0]) % div[1] + lo[1], k // (div[0] * div[1]) + lo[2]], axis=1)


def test_numpy_table_merge_matches_oracle():
    sensors, params = synth.config3(n_per_sensor=30_000, n_sensors=6, min_pts=2, leaf=0.1)
    tables = [cpu_partial_table([sensors[s] for s in fused.shard_sensors(6, r, 3)], params) for r in range(3)]
    keys, cnt, cent = fused.merge_tables_numpy(tables, params.min_points_per_voxel)
    st, _, ref, rep = oracle.merge_voxelize(sensors, params, stable=True)
    assert st == oracle.OK and len(keys) == rep.n_out
    assert np.array_equal(decode_cells(keys, params), rep.cells) and np.array_equal(cnt, rep.counts)
    assert_centroids_close(cent, xyzi_of(ref))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_rank(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        sensors, params = synth.config3(n_per_sensor=20_000, n_sensors=4, min_pts=2, leaf=0.1)
        mine = [sensors[s] for s in fused.shard_sensors(len(sensors), rank, world)]
        # bounds exchange (what ranks do when no crop box fixes the grid)
        xyz, _ = np_oracle.merge(mine, params)
        b = fused.allreduce_bounds(dist, xyz.min(axis=0), xyz.max(axis=0), len(xyz))
        allxyz, _ = np_oracle.merge(sensors, params)
        assert np.array_equal(b[:3], allxyz.min(axis=0)) and np.array_equal(b[3:], allxyz.max(axis=0))
        # table exchange
        t = cpu_partial_table(mine, params)
        tt = torch.from_numpy(t.view(np.int32).reshape(-1, 8).copy())
        gathered, counts = fused.allgather_tables(dist, tt, len(t), world)
        tables = [gathered[r][:counts[r]].numpy().copy().view(ENTRY_DTYPE).reshape(-1) for r in range(world)]
        assert counts[rank] == len(t) and np.array_equal(tables[rank], t)
        keys, cnt, cent = fused.merge_tables_numpy(tables, params.min_points_per_voxel)
        st, _, ref, rep = oracle.merge_voxelize(sensors, params, stable=True)
        assert len(keys) == rep.n_out and np.array_equal(decode_cells(keys, params), rep.cells)
        assert np.array_equal(cnt, rep.counts)
        assert_centroids_close(cent, xyzi_of(ref))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:                      # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_table_exchange_gloo_world2():
    ctx = mp.get_context("spawn")
    q, port, world = ctx.Queue(), _free_port(), 2
    procs = [ctx.Process(target=_gloo_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---- GPU: two contexts as two ranks -----------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("path", ["auto", "classic"])
@pytest.mark.parametrize("crop", [True, False])
def test_two_rank_fused_cloud_on_one_gpu(crop, path, monkeypatch):
    monkeypatch.setenv("CM_PATH", path)         # auto: the ranks' partial tables come from the bucket path
    from cloud_merger_amd import capi
    if crop:
        sensors, params = synth.config3(n_per_sensor=150_000, n_sensors=6, min_pts=2, leaf=0.05)
    else:
        sensors, params = synth.config2(n_per_sensor=100_000, n_sensors=4, min_pts=2)
    world = 2
    cms, parts = [], []
    n_total = sum(s.n for s in sensors)
    for r in range(world):
        cm = capi.CloudMerger(max_points_total=n_total, max_sensors=len(sensors), flags=capi.FLAG_OCCUPANCY)
        for k, s in enumerate(fused.shard_sensors(len(sensors), r, world)):
            cm.set_transform(k, sensors[s].q_xyzw, sensors[s].t_xyz)
            cm.submit(k, sensors[s])
        cms.append(cm)
    bounds = None
    if not crop:
        lb = [cm.local_bounds(params) for cm in cms]
        bounds = np.concatenate([np.min([b[0] for b in lb], axis=0), np.max([b[1] for b in lb], axis=0)])
    for cm in cms:
        res = cm.merge_partial(params, bounds)
        assert res.status == capi.OK
        assert bool(res.path_flags & 2) == (path == "auto" and bool(res.path_flags & 1)) and res.path_flags & 4 == 0
        parts.append(cm.partial_device())
    # sanity of one table against the numpy restatement (crop grid only)
    if crop:
        t0 = cms[0].partial(parts[0][1])
        ref0 = cpu_partial_table([sensors[s] for s in fused.shard_sensors(len(sensors), 0, world)], params)
        assert np.array_equal(t0["key"], ref0["key"]) and np.array_equal(t0["count"], ref0["count"])
    res = cms[0].merge_tables([p[0] for p in parts], [p[1] for p in parts], params)
    out = cms[0].result(res.n_out)
    cells, counts = cms[0].cells(res.n_out)
    st, _, ref, rep = oracle.merge_voxelize(sensors, params, stable=True)
    assert res.status == capi.OK and res.n_out == rep.n_out
    assert np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts)
    got = np.stack([out["x"], out["y"], out["z"], out["intensity"]], axis=1)
    assert_centroids_close(got, xyzi_of(ref))
    for cm in cms:
        cm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("min_pts", [0, 2])
def test_cfg5_shape_contexts_as_ranks(world, min_pts):
    """BASELINE.json configs[4] at reduced size: 16 sensors, 1 cm voxels, crop x[-15,45] y[-5,5] z[-0.5,3] (a 31-bit
    index: 6001 x 1001 x 351 cells), the sensors dealt to 2 or 4 ranks (contexts on one GPU), cm_merge_partial ->
    tables -> cm_merge_tables against the oracle on all 16 sensors. The points are drawn denser than cfg5's (a 12 m
    scene) so that voxels are shared between ranks and the deferred threshold matters."""
    from cloud_merger_amd import capi
    from cloud_merger_amd.types import SensorCloud
    n_sensors, nps = 16, 30_000
    per_rank = [synth.config5_shard(r, world, n_per_sensor=nps, n_sensors=n_sensors, min_pts=min_pts) for r in range(world)]
    params = per_rank[0][1]
    # squeeze every cloud into a 12 m x 4 m patch in front of the vehicle so that the 1 cm voxels collide
    for sens, _ in per_rank:
        for s in sens:
            s.data["x"] = s.data["x"] * np.float32(0.15); s.data["y"] = s.data["y"] * np.float32(0.05)
            s.data["z"] = np.round(s.data["z"] * np.float32(20)) / np.float32(20)
            s.data["x"] = np.round(s.data["x"] * np.float32(20)) / np.float32(20)
    allsens = [None] * n_sensors
    for r in range(world):
        for k, s in enumerate(fused.shard_sensors(n_sensors, r, world)):
            allsens[s] = per_rank[r][0][k]
    cms, parts = [], []
    try:
        for r in range(world):
            cm = capi.CloudMerger(max_points_total=n_sensors * nps, max_sensors=n_sensors, flags=capi.FLAG_OCCUPANCY)
            for k, sc in enumerate(per_rank[r][0]):
                cm.set_transform(k, sc.q_xyzw, sc.t_xyz)
                cm.submit(k, sc)
            cms.append(cm)
            res = cm.merge_partial(params, None)
            assert res.status == capi.OK and res.key_bits == 31
            parts.append(cm.partial_device())
        shared = sum(p[1] for p in parts)
        res = cms[world - 1].merge_tables([p[0] for p in parts], [p[1] for p in parts], params)
        out = cms[world - 1].result(res.n_out)
        cells, counts = cms[world - 1].cells(res.n_out)
        st, _, ref, rep = oracle.merge_voxelize(allsens, params, threads=4, stable=True)
        assert res.status == st == capi.OK and res.n_out == rep.n_out > 0
        assert shared > res.n_merged, "the case must have voxels that live on several ranks"
        assert np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts)
        got = np.stack([out["x"], out["y"], out["z"], out["intensity"]], axis=1)
        assert_centroids_close(got, xyzi_of(ref))
    finally:
        for cm in cms:
            cm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("crop", [True, False])
def test_rank_with_an_empty_share_still_merges_on_the_shared_grid(crop):
    """Rank 0's only sensor delivered an empty cloud: its partial table is empty, but cm_merge_tables on it must
    report and decode cells on the grid all ranks share (the crop box, or the all-reduced bounds)."""
    from cloud_merger_amd import capi
    from cloud_merger_amd.types import xyzi_cloud
    rng = np.random.default_rng(3)
    full = xyzi_cloud(rng.uniform(-4, 4, (30_000, 3)).astype(np.float32), rng.uniform(0, 9, 30_000).astype(np.float32))
    empty = xyzi_cloud(np.zeros((0, 3), np.float32), np.zeros(0, np.float32))
    sensors = [empty, full]
    from cloud_merger_amd.types import MergeParams
    params = MergeParams(leaf=(0.1, 0.15, 0.1), min_points_per_voxel=2)
    if crop:
        params.crop_min, params.crop_max = (-3.0, -3.5, -2.0), (3.5, 3.0, 2.5)
    cms = [capi.CloudMerger(max_points_total=40_000, max_sensors=2, flags=capi.FLAG_OCCUPANCY) for _ in range(2)]
    try:
        for r, cm in enumerate(cms):
            cm.set_transform(0, sensors[r].q_xyzw, sensors[r].t_xyz)
            cm.submit(0, sensors[r])
        bounds = None
        if not crop:
            lb = [b for b in (cm.local_bounds(params) for cm in cms) if b[2]]
            assert len(lb) == 1
            bounds = np.concatenate([lb[0][0], lb[0][1]])
        r0 = cms[0].merge_partial(params, bounds)
        r1 = cms[1].merge_partial(params, bounds)
        assert r0.status == capi.EMPTY_INPUT and r1.status == capi.OK
        assert list(r0.min_b) == list(r1.min_b) and list(r0.div_b) == list(r1.div_b)
        ptr, n = cms[1].partial_device()
        res = cms[0].merge_tables([ptr], [n], params)
        cells, counts = cms[0].cells(res.n_out)
        st, _, ref, rep = oracle.merge_voxelize(sensors, params, stable=True)
        assert res.status == capi.OK and res.n_out == rep.n_out > 0
        assert np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts)
    finally:
        for cm in cms:
            cm.close()


_WORLD1_SCRIPT = r"""
import sys
import torch                                   # torch first: one HIP runtime per process (its bundled one)
sys.path.insert(0, sys.argv[1])
from cloud_merger_amd import capi, fused, synth
sensors, params = synth.config3(n_per_sensor=80_000, n_sensors=4, min_pts=2, leaf=0.05)
n_total = sum(s.n for s in sensors)
with capi.CloudMerger(max_points_total=n_total, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
    cm.submit_all(sensors)
    a = cm.merge_voxelize(params)
    plain = cm.result(a.n_out)
    cm.submit_all(sensors)
    b = fused.fused_cloud(cm, params, None, 0, 1, torch.device("cuda", 0))
    viaf = cm.result(b.n_out)
assert a.status == 0 and b.status == 0 and a.n_out == b.n_out > 0, (a.n_out, b.n_out)
assert plain.tobytes() == viaf.tobytes()
print("ok", a.n_out)
"""


@pytest.mark.gpu
def test_fused_cloud_world1_equals_plain_path():
    """fused.fused_cloud with torch tensors as exchange buffers (own process: a process that uses
    torch and the library must import torch first so both share one HIP runtime)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _WORLD1_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


@pytest.mark.gpu
def test_cfg5_runner_two_processes_gloo_on_one_gpu():
    """`bench.py --config 5` as the driver would launch it, rehearsed on one GPU: two ranks (processes) under
    torch.distributed.run, --backend gloo --single-device, reduced size, --check: every rank runs cm_merge_partial on
    its 8 sensors, the REAL tables are all-gathered (on host copies here; RCCL on a node) and merged with
    cm_merge_tables; rank 0 compares the fused cloud with the oracle on all 16 sensors and exits non-zero on a
    difference."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CM_PATH", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--config", "5", "--backend", "gloo",
           "--single-device", "--points-per-sensor", "150000", "--steps", "3", "--warmup", "1", "--check", "--min-pts", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["points_per_frame"] == 16 * 150000
    assert out["parity"]["occupancy_bit_exact"] and out["parity"]["max_abs_dxyz_m"] <= 1e-4
    assert out["config"]["gathered_entries"] >= out["config"]["voxels_out"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_cfg5_dense_variant_contexts_as_ranks(world):
    """`bench.py --config 5 --dense` at reduced size: 16 sensors drawn inside the crop box on a shared
    road surface, a 10 cm leaf so that the reduced clouds are as dense per voxel as the full ones at 1 cm; the sensors
    dealt to 2 or 4 ranks (contexts on one GPU): most points survive the crop, most voxels live on several ranks, and
    min_points_per_voxel = 2 is met across ranks only — cm_merge_partial -> cm_merge_tables against the oracle."""
    from cloud_merger_amd import capi
    n_sensors, nps = 16, 40_000
    per_rank = [synth.config5_dense_shard(r, world, n_per_sensor=nps, n_sensors=n_sensors, min_pts=2, leaf=0.1) for r in range(world)]
    params = per_rank[0][1]
    allsens = [None] * n_sensors
    for r in range(world):
        for k, s in enumerate(fused.shard_sensors(n_sensors, r, world)):
            allsens[s] = per_rank[r][0][k]
    cms, parts, kept = [], [], 0
    try:
        for r in range(world):
            cm = capi.CloudMerger(max_points_total=n_sensors * nps, max_sensors=n_sensors, flags=capi.FLAG_OCCUPANCY)
            for k, sc in enumerate(per_rank[r][0]):
                cm.set_transform(k, sc.q_xyzw, sc.t_xyz)
                cm.submit(k, sc)
            cms.append(cm)
            res = cm.merge_partial(params, None)
            assert res.status == capi.OK
            kept += res.n_merged
            parts.append(cm.partial_device())
        assert kept >= 0.8 * n_sensors * nps, "the dense variant keeps most of its points"
        res = cms[0].merge_tables([p[0] for p in parts], [p[1] for p in parts], params)
        out = cms[0].result(res.n_out)
        cells, counts = cms[0].cells(res.n_out)
        st, _, ref, rep = oracle.merge_voxelize(allsens, params, threads=4, stable=True)
        assert res.status == st == capi.OK and res.n_out == rep.n_out > 20_000
        assert sum(p[1] for p in parts) > 1.3 * res.n_merged, "many voxels live on several ranks"
        assert np.array_equal(cells, rep.cells) and np.array_equal(counts, rep.counts)
        got = np.stack([out["x"], out["y"], out["z"], out["intensity"]], axis=1)
        assert_centroids_close(got, xyzi_of(ref))
    finally:
        for cm in cms:
            cm.close()


@pytest.mark.gpu
def test_cfg5_dense_runner_two_processes_gloo_on_one_gpu():
    """`bench.py --config 5 --dense` under torch.distributed.run, two ranks on one GPU (gloo rehearsal), --check."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CM_PATH", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--config", "5", "--dense",
           "--backend", "gloo", "--single-device", "--points-per-sensor", "100000", "--steps", "3", "--warmup", "1", "--check",
           "--min-pts", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["parity"]["occupancy_bit_exact"] and out["parity"]["max_abs_dxyz_m"] <= 1e-4
    assert out["config"]["voxels_out"] > 0 and out["config"]["gathered_entries"] > out["config"]["voxels_out"]


@pytest.mark.gpu
def test_frame_sharded_bench_two_processes_gloo_on_one_gpu():
    """the N > 1 launch of the headline bench as the driver does it — `python -m torch.distributed.run
    --nproc-per-node 2 bench.py --gpus 2 ...` — rehearsed on one GPU (--single-device --backend gloo, reduced clouds): every
    rank runs its own frame stream (frames are independent units: no data-path collective), barrier + max-over-ranks
    timing, rank 0 prints the contract line. Checks that the N > 1 path starts, finishes and reports whole-job figures."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CM_PATH", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--single-device", "--points-per-sensor", "100000", "--steps", "12", "--warmup", "3", "--stream-frames", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["scaling"] == "weak"
    assert out["config"]["points_per_frame"] == 400_000 and out["config"]["voxels_out"] > 0
    # whole-job value: both ranks' frames over the slower rank's time
    assert abs(out["value"] - 2 * 12 * 400_000 / (out["ms_per_step"] * 1e-3 * 12)) <= 1e-6 * out["value"]
    assert "REHEARSAL" in out["config"]["workload"]


def _splitmix_clouds(n_sensors, n_points, seed=5001):
    """The generator of cloud_merger_amd/host/fused_main.cpp, vectorised: sensor s draws 4 values per point from
    splitmix64(seed + s) — x, y, z uniform in the cfg5 crop box widened by 10 %, intensity in [0, 255)."""
    from cloud_merger_amd.types import xyzi_cloud
    cmin, cmax = np.array([-15.0, -5.0, -0.5], np.float32), np.array([45.0, 5.0, 3.0], np.float32)
    clouds = []
    for s in range(n_sensors):
        st = np.uint64(seed + s) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, 4 * n_points + 1, dtype=np.uint64)
        z = st
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        u = ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(n_points, 4)
        ext = cmax - cmin
        xyz = (cmin - np.float32(0.05) * ext) + (np.float32(1.1) * ext) * u[:, :3]          # fp32 arithmetic, as in the tool
        clouds.append(xyzi_cloud(xyz.astype(np.float32), (np.float32(255.0) * u[:, 3]).astype(np.float32)))
    return clouds


@pytest.mark.gpu
def test_cpp_fused_tool_world1_matches_the_oracle(tmp_path):
    """cloudmerge_fused (C++: C-ABI + RCCL called directly, one process per GPU) with one rank: ncclCommInitRank through the
    file rendezvous, both all-gathers, cm_merge_partial -> cm_merge_tables; its fused cloud against the oracle on the same
    sensors (regenerated here with the tool's splitmix64 stream). More ranks need more GPUs: the driver's node."""
    import json
    import subprocess
    from cloud_merger_amd import build as cm_build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cm_build.build()
    subprocess.run(["make", "-C", os.path.join(root, "cloud_merger_amd", "host"), "-s"], check=True)
    n_sensors, n_points, leaf, min_pts = 16, 40_000, 0.05, 2
    r = subprocess.run([os.path.join(root, "cloud_merger_amd", "host", "bin", "cloudmerge_fused"), "--rank", "0", "--world", "1",
                        "--rendezvous", str(tmp_path / "id"), "--sensors", str(n_sensors), "--points", str(n_points),
                        "--leaf", str(leaf), "--min-pts", str(min_pts), "--steps", "3"],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    sensors = _splitmix_clouds(n_sensors, n_points)
    params = MergeParams(leaf=(leaf,) * 3, min_points_per_voxel=min_pts, crop_min=(-15.0, -5.0, -0.5), crop_max=(45.0, 5.0, 3.0))
    st, _, ref, rep = oracle.merge_voxelize(sensors, params, threads=4, stable=True)
    assert st == oracle.OK and got["status"] == 0 and got["voxels_out"] == rep.n_out > 0
    want = float((ref["x"].astype(np.float64) + 2.0 * ref["y"] + 3.0 * ref["z"] + 5.0 * ref["intensity"]).sum())
    assert abs(got["checksum"] - want) <= 1e-4 * rep.n_out * 10, (got["checksum"], want)
