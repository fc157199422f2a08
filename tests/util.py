"""Shared helpers for the test-suite: fixture loading and result comparison."""
import json
import os

import numpy as np

from cloud_merger_amd.types import MergeParams, SensorCloud, XYZI_DTYPE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)["cases"]


def bits_to_xyzi(rows):
    """list of [x,y,z,i] uint32 bit patterns -> XYZI structured array"""
    a = np.zeros(len(rows), dtype=XYZI_DTYPE)
    if rows:
        raw = np.array(rows, dtype=np.uint32).view(np.float32)
        a["x"], a["y"], a["z"], a["intensity"] = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]
    return a


def case_inputs(case):
    sensors = []
    for s in case["sensors"]:
        a = bits_to_xyzi(s["points"])
        sensors.append(SensorCloud(data=a, n=len(a), q_xyzw=s["q"], t_xyz=s["t"],
                                   is_dense=s.get("is_dense", True)))
    crop = case.get("crop")
    params = MergeParams(leaf=(case["leaf"],) * 3, min_points_per_voxel=case["min_pts"],
                         downsample_all_data=case.get("downsample_all", True),
                         crop_min=crop["min"] if crop else None, crop_max=crop["max"] if crop else None)
    return sensors, params


def xyzi_of(points32):
    """oracle 32-B PointXYZI records -> (n,4) float32 x,y,z,intensity"""
    return np.stack([points32["x"], points32["y"], points32["z"], points32["intensity"]], axis=1)


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def assert_centroids_close(got, want, tol=1e-4):
    """north_star tolerance: centroid xyz within 1e-4 m; intensity within 1e-4*max(1,|I|)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if len(got) == 0:
        return
    dxyz = np.abs(got[:, :3] - want[:, :3]).max()
    assert dxyz <= tol, f"centroid xyz differs by {dxyz}"
    di = np.abs(got[:, 3] - want[:, 3]) / np.maximum(1.0, np.abs(want[:, 3]))
    assert di.max() <= tol, f"intensity differs by {di.max()} (relative)"


def assert_centroids_close_or_exact(got, want, counts, cells, merged, leaf, sequential, big=1000):
    """assert_centroids_close for voxels of up to `big` points. Above that the oracle's (PCL's) sequential fp32 sum
    itself drifts by ~1e-4 m when the addends are nearly identical (one-sided rounding), so a path that adds in a
    tree (sequential=False) is held to the exact fp64 mean of the voxel's points instead; a path that adds one
    after the other like the oracle (sequential=True) is compared with the oracle everywhere. Returns
    (max deviation of `got` from the exact mean, same for `want`) over the big voxels, or (0, 0)."""
    from oracle import oracle
    got = np.asarray(got)
    want = np.asarray(want)
    is_big = np.asarray(counts) > big
    if sequential or not is_big.any():
        assert_centroids_close(got, want)
        return 0.0, 0.0
    assert_centroids_close(got[~is_big], want[~is_big])
    pc = oracle.voxel_cells(merged, leaf).astype(np.int64)
    bc = np.asarray(cells)[is_big].astype(np.int64)
    lo = np.minimum(pc.min(axis=0), bc.min(axis=0))
    ext = np.maximum(pc.max(axis=0), bc.max(axis=0)) - lo + 1

    def lin(c):
        c = c - lo
        return (c[:, 2] * ext[1] + c[:, 1]) * ext[0] + c[:, 0]
    big_keys = lin(bc)                                        # ascending: the voxels come in linear-index order
    pk = lin(pc)
    pos = np.searchsorted(big_keys, pk)
    pos[pos == len(big_keys)] = 0
    member = big_keys[pos] == pk
    m64 = xyzi_of(merged).astype(np.float64)[member][:, :3]
    idx = pos[member]
    n_in = np.bincount(idx, minlength=len(big_keys)).astype(np.float64)
    exact = np.stack([np.bincount(idx, weights=m64[:, a], minlength=len(big_keys)) for a in range(3)], axis=1) / n_in[:, None]
    d_got = float(np.abs(got[is_big][:, :3].astype(np.float64) - exact).max())
    d_want = float(np.abs(want[is_big][:, :3].astype(np.float64) - exact).max())
    assert d_got <= 1e-4, f"centroid of a voxel of more than {big} points is {d_got} m off the exact mean (oracle: {d_want})"
    return d_got, d_want


# k3_local (the bucket path's finish) adds a voxel's points one after the other — pcl's own order, bit for bit — as long as
# the voxel's run of sorted positions ends within 16 positions of its owner thread's block: always for voxels of up to
# 17 points. Longer runs may be finished by a wave that adds 64 points per step in a fixed tree order (deterministic,
# within the north-star tolerance, not bit-identical to the sequential sum).
SEQ_EXACT_MAX = 17


def assert_bucket_centroids(got, want, counts, cells=None, merged=None, leaf=None):
    """Bucket path with the k3_local finish: bit-exact for voxels of up to SEQ_EXACT_MAX points, assert_centroids_close
    (or, with the merged cloud at hand, the exact-mean rule for voxels of more than 1000 points) for the rest."""
    got, want, counts = np.asarray(got), np.asarray(want), np.asarray(counts)
    small = counts <= SEQ_EXACT_MAX
    assert same_bits(got[small], want[small]), "voxels of up to 17 points: one summation order, bit for bit"
    if (~small).any():
        if merged is not None:
            assert_centroids_close_or_exact(got, want, counts, cells, merged, leaf, sequential=False)
        else:
            assert_centroids_close(got[~small], want[~small])
