"""The C++ host shell (cloud_merger_amd/host): CPU-side unit tests of the PointCloud2 / PCD / node
configuration code, and on the GPU a replay of a synthesised multi-sensor .pcd sequence through
CloudMergerNode compared with the oracle frame by frame."""
import json
import os
import subprocess

import numpy as np
import pytest

from cloud_merger_amd import build as cm_build
from cloud_merger_amd import replay_data
from cloud_merger_amd.types import MergeParams, xyzi_cloud

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cloud_merger_amd", "host")


@pytest.fixture(scope="module")
def host_bins():
    cm_build.build()
    subprocess.run(["make", "-C", HOST, "-s"], check=True)
    return os.path.join(HOST, "bin")


def test_host_unit_tests_cpu(host_bins, tmp_path):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU box: covered by the gpu variant")
    r = subprocess.run([os.path.join(host_bins, "host_tests"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_python_pcd_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    xyz, inten = rng.normal(size=(100, 3)).astype(np.float32), rng.uniform(0, 255, 100).astype(np.float32)
    p = str(tmp_path / "a.pcd")
    replay_data.write_pcd(p, xyz, inten)
    data, names = replay_data.read_pcd(p)
    assert names == ["x", "y", "z", "intensity"]
    assert np.array_equal(data[:, :3], xyz) and np.array_equal(data[:, 3], inten)


@pytest.mark.gpu
def test_host_unit_tests_gpu(host_bins, tmp_path):
    r = subprocess.run([os.path.join(host_bins, "host_tests"), str(tmp_path), "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [False, True, "pipeline", "threads+pipeline", "defer", "defer+pin", "threads+defer+pin", "threads+defer+pin+async"])
def test_replay_sequence_matches_oracle(host_bins, tmp_path, threads):
    """threads: subscriber threads beside the loop thread; pipeline: NodeConfig::pipelined_publish — frame n - 1 is
    published (cm_result_publish_async into a registered message buffer) while frame n computes; defer: NodeConfig::deferred_wait on
    top of it — frame n is only waited for during tick n + 1, beside that tick's host-to-device copies; pin: the clouds' host
    buffers registered for DMA (cm_host_register); async: the subscriber threads' submits do not wait for their copies
    (cm_submit_cloud_async); same clouds every way."""
    flags = {False: [], True: ["--threads"], "pipeline": ["--pipeline"], "threads+pipeline": ["--threads", "--pipeline"],
             "defer": ["--defer"], "defer+pin": ["--defer", "--pin"], "threads+defer+pin": ["--threads", "--defer", "--pin"],
             "threads+defer+pin+async": ["--threads", "--defer", "--pin", "--async"]}[threads]
    from oracle import oracle
    from tests.util import assert_centroids_close, xyzi_of

    seq, out = str(tmp_path / "seq"), str(tmp_path / "out")
    os.makedirs(out)
    frames, sensors = 6, 4
    poses = replay_data.write_sequence(seq, frames=frames, sensors=sensors, rings=16, azimuths=900)
    crop = ["-15", "-5", "-0.5", "60", "5", "3"]
    r = subprocess.run([os.path.join(host_bins, "cloudmerge_replay"), "--dir", seq, "--sensors", str(sensors),
                        "--frames", str(frames), "--leaf", "0.1", "--min-pts", "2", "--crop", *crop, "--out", out]
                       + flags,      # (--threads: subscriber threads beside the loop thread: AsyncSpinner(6), :513)
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    assert stats["frames"] == frames and stats["points_in"] == frames * sensors * 16 * 900
    params = MergeParams(leaf=(0.1,) * 3, min_points_per_voxel=2, crop_min=(-15.0, -5.0, -0.5), crop_max=(60.0, 5.0, 3.0))
    total = 0
    for f in range(frames):
        clouds = []
        for s in range(sensors):
            d, _ = replay_data.read_pcd(os.path.join(seq, f"frame_{f:04d}_sensor_{s}.pcd"))
            clouds.append(xyzi_cloud(d[:, :3], d[:, 3], q_xyzw=poses[s][0], t_xyz=poses[s][1], is_dense=False))
        st, _, ref, rep = oracle.merge_voxelize(clouds, params, stable=True)
        got, names = replay_data.read_pcd(os.path.join(out, f"voxel_{f:04d}.pcd"))
        assert st == oracle.OK and names == ["x", "y", "z", "intensity"]
        assert len(got) == rep.n_out
        assert_centroids_close(got, xyzi_of(ref))
        total += len(got)
    assert stats["voxels_out"] == total


@pytest.mark.gpu
def test_replay_frame_sharding(host_bins, tmp_path):
    """Two shards over the same sequence publish exactly the frames a single process does."""
    seq = str(tmp_path / "seq")
    replay_data.write_sequence(seq, frames=6, sensors=2, rings=8, azimuths=600)
    def run(shard):
        r = subprocess.run([os.path.join(host_bins, "cloudmerge_replay"), "--dir", seq, "--sensors", "2",
                            "--frames", "6", "--leaf", "0.2", "--min-pts", "0", "--shard", shard],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        return json.loads(r.stdout.strip().splitlines()[-1])
    whole, a, b = run("0/1"), run("0/2"), run("1/2")
    assert a["frames"] + b["frames"] == whole["frames"] == 6
    assert a["voxels_out"] + b["voxels_out"] == whole["voxels_out"]
    assert a["points_in"] + b["points_in"] == whole["points_in"]
