"""The oracle's ground-plane restatement (orc_ransac_plane) on cases with answers known in advance, and the
slab/band bookkeeping of oracle.ground_split. CPU only."""
import numpy as np

from oracle import oracle


def test_exact_plane_is_recovered():
    rng = np.random.default_rng(3)
    xy = rng.uniform(-10, 10, (5000, 2))
    z = 0.25 + 0.05 * xy[:, 0] - 0.02 * xy[:, 1]                  # exact plane, fp32-rounded coordinates
    pts = oracle.make_points(np.column_stack([xy, z]).astype(np.float32))
    res, inl = oracle.ransac_plane(pts, threshold=0.05)
    assert res.found == 1 and inl.all() and res.n_inliers == 5000
    assert res.iterations == 1                                       # w = 1: PCL's loop stops after the first model
    n = np.array(res.plane[:3]) * np.sign(res.plane[2])
    d = res.plane[3] * np.sign(res.plane[2])
    assert abs(n[0] / n[2] + 0.05) < 1e-5 and abs(n[1] / n[2] - 0.02) < 1e-5 and abs(d / n[2] + 0.25) < 1e-5
    assert abs(np.linalg.norm(n) - 1) < 1e-6


def test_outliers_and_refit_against_numpy_least_squares():
    rng = np.random.default_rng(4)
    n = 4000
    xy = rng.uniform(0, 30, (n, 2))
    z = 0.01 * xy[:, 0] + 0.02 * rng.standard_normal(n)
    obj = np.column_stack([rng.uniform(0, 30, (800, 2)), rng.uniform(0.6, 2.0, 800)])
    xyz = np.concatenate([np.column_stack([xy, z]), obj]).astype(np.float32)
    res, inl = oracle.ransac_plane(oracle.make_points(xyz), threshold=0.3)
    assert res.found == 1 and inl[:n].all() and not inl[n:].any()
    p = xyz[inl].astype(np.float64)
    c = p.mean(0)
    w, v = np.linalg.eigh(np.cov((p - c).T, bias=True))
    nn = v[:, 0] * np.sign(v[2, 0])
    got = np.array(res.plane[:3]) * np.sign(res.plane[2])
    assert np.abs(got - nn).max() < 1e-6 and abs(res.plane[3] * np.sign(res.plane[2]) + nn @ c) < 1e-6
    # same call, same answer; another seed, another sample sequence (possibly the same plane)
    res2, inl2 = oracle.ransac_plane(oracle.make_points(xyz), threshold=0.3)
    assert list(res2.plane) == list(res.plane) and np.array_equal(inl, inl2)


def test_no_model_cases():
    two = oracle.make_points(np.array([[0, 0, 0], [1, 1, 0]], np.float32))
    res, inl = oracle.ransac_plane(two)
    assert res.found == 0 and not inl.any()
    line = oracle.make_points(np.column_stack([np.arange(50), np.zeros(50), np.zeros(50)]).astype(np.float32))
    res, inl = oracle.ransac_plane(line, max_iterations=20)
    assert res.found == 0 and res.iterations == 0 and not inl.any()    # every sample collinear: all skipped


def test_ground_split_bookkeeping():
    xyz = np.array([[5, 0, 0.0], [5, 0, 0.4], [5, 0, 0.505], [5, 0, 0.52], [5, 0, 2.9], [5, 0, 3.1],    # slab 0: band / gap / above
                    [10, 0, 0.1],                                                                      # border: first slab only
                    [15, 0, 2.0], [25, 0, 0.0], [40, 0, 0.0]], np.float32)                             # keep-all slab; uncovered x
    pts = oracle.make_points(xyz)
    zones = [(0.0, 10.0, 0.5), (10.0, 10.0, -1.0)]
    gp = dict(max_iterations=10, threshold=0.3, probability=0.99, optimize=True, z_keep_max=3.0, seed=1)
    keep, ground, planes = oracle.ground_split(pts, zones, 0, gp)
    band = [0, 1, 6]                                   # z in [-0.5, 0.5] of slab 0 (the border point included)
    assert planes[1] is None and planes[0].found == 1
    assert ground[band].all()                          # three points define their plane exactly: all inliers
    assert list(np.nonzero(keep)[0]) == [3, 4, 7]      # above the band (0.51 .. 3.0], and the keep-all slab
    assert not keep[[2, 5, 8, 9]].any() and not ground[[2, 5, 8, 9]].any()   # gap, too high, no slab
