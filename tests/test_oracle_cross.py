"""Cross-checks the two independent CPU restatements (C++ and numpy) on seeded synthetic frames
of the BASELINE configurations, scaled so the whole CPU suite stays within minutes."""
import numpy as np
import pytest

from cloud_merger_amd import synth
from oracle import np_oracle, oracle
from tests.util import assert_centroids_close, same_bits, xyzi_of


def _cross(sensors, params, sequential):
    st, merged, out, rep = oracle.merge_voxelize(sensors, params, threads=1, stable=True)
    nst, xyz, inten, cnt, cell = np_oracle.merge_voxelize(sensors, params, sequential=sequential)
    assert st == nst == oracle.OK
    mx, mi = np_oracle.merge(sensors, params)
    if getattr(params, "outlier_radius", None):
        k = np_oracle.radius_outlier_mask(mx, params.outlier_radius, params.outlier_min_neighbors)
        mx, mi = mx[k], mi[k]
    assert same_bits(xyzi_of(merged), np.concatenate([mx, mi[:, None]], axis=1))   # transform/crop/concat(/outliers)
    assert rep.n_out == len(xyz)                                                     # occupancy count
    assert np.array_equal(rep.cells, cell) and np.array_equal(rep.counts, cnt)      # occupancy + order
    got = xyzi_of(out)
    want = np.concatenate([xyz, inten[:, None]], axis=1)
    if sequential:
        assert same_bits(got, want)
    else:
        assert_centroids_close(got, want)
    return rep


def test_config1_plumbing():
    sensors, params = synth.config1(n_per_sensor=100_000, min_pts=0)
    rep = _cross(sensors, params, sequential=False)
    assert rep.n_in == 200_000 and rep.n_merged == 200_000


@pytest.mark.parametrize("min_pts", [0, 2])
def test_config2_scaled(min_pts):
    sensors, params = synth.config2(n_per_sensor=50_000, min_pts=min_pts)
    _cross(sensors, params, sequential=False)


def test_config2_sequential_sums_bit_exact():
    sensors, params = synth.config2(n_per_sensor=4_000, min_pts=0)
    params.leaf = (0.5, 0.5, 0.5)        # several points per voxel so summation order matters
    _cross(sensors, params, sequential=True)


def test_config3_scaled_crop():
    sensors, params = synth.config3(n_per_sensor=100_000, n_sensors=8, min_pts=2, leaf=0.05)
    rep = _cross(sensors, params, sequential=False)
    assert 0 < rep.n_merged < rep.n_in


@pytest.mark.parametrize("layout", ["pcl32", "velo22", "xyz12"])
def test_wire_layouts_agree(layout):
    base, params = synth.config2(n_per_sensor=10_000, min_pts=0, layout="xyzi16")
    other, _ = synth.config2(n_per_sensor=10_000, min_pts=0, layout=layout)
    _, m0, o0, _ = oracle.merge_voxelize(base, params)
    _, m1, o1, _ = oracle.merge_voxelize(other, params)
    assert np.array_equal(m0["x"], m1["x"]) and np.array_equal(m0["z"], m1["z"])
    if layout != "xyz12":
        assert o0.tobytes() == o1.tobytes()
    else:
        assert np.all(o1["intensity"] == 0) and np.array_equal(o0["x"], o1["x"])


def test_unstable_vs_stable_within_tolerance():
    sensors, params = synth.config2(n_per_sensor=20_000, min_pts=0)
    params.leaf = (0.4, 0.4, 0.4)
    _, _, a, ra = oracle.merge_voxelize(sensors, params, stable=False)
    _, _, b, rb = oracle.merge_voxelize(sensors, params, stable=True)
    assert np.array_equal(ra.cells, rb.cells) and np.array_equal(ra.counts, rb.counts)
    assert_centroids_close(xyzi_of(a), xyzi_of(b))


def test_permutation_invariance_of_occupancy():
    sensors, params = synth.config2(n_per_sensor=10_000, n_sensors=2, min_pts=2)
    _, _, a, ra = oracle.merge_voxelize(sensors, params, stable=True)
    rng = np.random.default_rng(7)
    for s in sensors:
        s.data = s.data[rng.permutation(s.n)]
    _, _, b, rb = oracle.merge_voxelize(sensors, params, stable=True)
    assert np.array_equal(ra.cells, rb.cells) and np.array_equal(ra.counts, rb.counts)
    assert_centroids_close(xyzi_of(a), xyzi_of(b))


# ---- radius outlier removal ---------------------------------------------
def test_outlier_removal_known_answers():
    # a pair 0.10 apart, a pair exactly 0.15f apart on x (strict '<': not neighbours), a loner, a NaN
    r = np.float32(0.15)
    xyz = np.array([[0, 0, 0], [0.1, 0, 0], [5, 0, 0], [5 + float(r), 0, 0], [9, 9, 9], [np.nan, 0, 0]], np.float32)
    pts = oracle.make_points(xyz, np.arange(6))
    out, mask = oracle.radius_outlier_removal(pts, float(r), 1)
    d = np.float32(xyz[3, 0] - xyz[2, 0])
    boundary_is_neighbour = bool(np.float32(d * d) < np.float32(float(r) * float(r)))
    assert mask.tolist() == [True, True, boundary_is_neighbour, boundary_is_neighbour, False, False]
    assert out["intensity"].tolist() == [float(i) for i in np.flatnonzero(mask)]
    # min_neighbors = 2 needs two other points
    tri = oracle.make_points(np.array([[0, 0, 0], [0.05, 0, 0], [0, 0.05, 0], [1, 1, 1], [1.05, 1, 1]], np.float32))
    _, m2 = oracle.radius_outlier_removal(tri, 0.15, 2)
    assert m2.tolist() == [True, True, True, False, False]


@pytest.mark.parametrize("radius,min_nb", [(0.15, 1), (0.1, 1), (0.3, 3)])
def test_outlier_removal_cpp_vs_scipy(radius, min_nb):
    sensors, params = synth.config2(n_per_sensor=15_000, min_pts=0)
    xyz, inten = np_oracle.merge(sensors, params)
    _, mask = oracle.radius_outlier_removal(oracle.make_points(xyz, inten), radius, min_nb)
    assert np.array_equal(mask, np_oracle.radius_outlier_mask(xyz, radius, min_nb))
    assert 0 < mask.sum() < len(mask)


def test_pipeline_with_outlier_removal_cross():
    sensors, params = synth.config3(n_per_sensor=60_000, n_sensors=4, min_pts=2, leaf=0.1)
    params.outlier_radius, params.outlier_min_neighbors = 0.15, 1
    rep = _cross(sensors, params, sequential=False)
    params.outlier_radius = None
    st, _, _, rep0 = oracle.merge_voxelize(sensors, params)
    assert rep.n_merged < rep0.n_merged
