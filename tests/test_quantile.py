"""GPU parity of the quantile passes (cm_kernels_v4.hip): ONE global pass into buckets cut at the quantiles of the
previous frame's sorted records, then one finish workgroup per bucket — against the CPU oracle, through the C-ABI.

The path needs a predecessor: the first frame of a context (and every frame whose grid changed) runs the fixed-grid
passes and leaves the splitters; from the second frame on cm_result.path_flags carries CM_PATH_QUANTILE. Same bars as
tests/test_gpu_parity.py: merged cloud and occupancy bit-exact, centroids bit-exact for voxels of up to 17 points
(tests/util.py) and within 1e-4 m beyond."""
import numpy as np
import pytest

from cloud_merger_amd import capi, synth
from cloud_merger_amd.types import MergeParams, SensorCloud, xyzi_cloud
from oracle import oracle
from tests.util import assert_bucket_centroids, assert_centroids_close_or_exact, same_bits, xyzi_of

pytestmark = pytest.mark.gpu

BUCKET, PREDICTED, REDONE, SPLIT, QUANTILE = 2, 4, 8, 32, 64


def xyzi4(a):
    return np.stack([a["x"], a["y"], a["z"], a["intensity"]], axis=1)


def frame_against_oracle(cm, sensors, params, n_cap):
    """One frame on a persistent context, compared with the oracle like tests/test_gpu_parity.py::check_against_oracle."""
    st, merged, out, rep = oracle.merge_voxelize(sensors, params, threads=4, stable=True)
    cm.submit_all(sensors)
    res = cm.merge_voxelize(params)
    assert res.status == st
    assert res.n_in == rep.n_in
    got_merged = xyzi4(cm.merged(n_cap))
    assert same_bits(got_merged, xyzi_of(merged)), "merged cloud (transform + crop + concat) must be bit-exact"
    if st != oracle.OK:
        return res, rep
    got = xyzi4(cm.result(res.n_out))
    cells, counts = cm.cells(res.n_out)
    assert res.n_merged == rep.n_merged
    assert res.n_out == rep.n_out, "occupancy: number of kept voxels"
    assert np.array_equal(cells, rep.cells), "occupancy: kept cells and their order"
    assert np.array_equal(counts, rep.counts), "occupancy: points per voxel"
    if res.path_flags & BUCKET:          # (a frame handed back twice ends on the general path: tree-order sums)
        assert res.path_flags & SPLIT
        assert_bucket_centroids(got, xyzi_of(out), rep.counts, rep.cells, merged, params.leaf)
    else:
        assert_centroids_close_or_exact(got, xyzi_of(out), rep.counts, rep.cells, merged, params.leaf, sequential=False)
    if not res.bounds_from_crop:
        assert list(res.min_b) == list(rep.min_b) and list(res.div_b) == list(rep.div_b)
    return res, rep


def needs_lds_rank(res):
    if not res.path_flags & 1:
        pytest.skip("the device probe did not find lane-ordered LDS adds: no bucket path on this device")


@pytest.mark.parametrize("min_pts", [0, 2])
def test_stream_of_frames_takes_one_global_pass(min_pts):
    n_per = 150_000
    with capi.CloudMerger(max_points_total=4 * n_per, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
        flags = []
        for k in range(4):
            sensors, params = synth.config2_stream(k, n_per_sensor=n_per, min_pts=min_pts)
            res, rep = frame_against_oracle(cm, sensors, params, 4 * n_per)
            needs_lds_rank(res)
            flags.append(res.path_flags)
            assert res.path_flags & BUCKET
        assert not flags[0] & QUANTILE, "the first frame has no predecessor"
        assert all(f & QUANTILE for f in flags[1:]), flags
        assert not any(f & REDONE for f in flags[1:]), flags
        assert res.sort_passes == 1


def test_full_size_stream():
    """cfg2 itself: 4 x 1 M points, 5 cm, min 2 points per voxel — 2048 buckets of about 1950 records. Six consecutive
    frames of the moving stream on ONE context (bench.py deals them to three): every frame equals the oracle; the poses
    jitter by a couple of milliradians from frame to frame, and one of these transitions (4 -> 5) tilts the ground layer far
    enough across the 5 cm voxel layers — the index is z-major — that a bucket outgrows the finish's usual workgroup: that
    frame is handed back and redone with the fixed-grid passes (CM_PATH_REDONE), the others take the one quantile pass."""
    with capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
        flags = []
        for k in range(12):
            sensors, params = synth.config2_stream(k % 6, min_pts=2)
            res, rep = frame_against_oracle(cm, sensors, params, 4_000_000)
            needs_lds_rank(res)
            flags.append(res.path_flags)
        assert not flags[0] & QUANTILE and all(f & QUANTILE for f in flags[1:3]) and not any(f & REDONE for f in flags[:3]), flags
        assert all((f & QUANTILE) or (f & REDONE) for f in flags[1:]), flags    # every later frame tried the quantile pass
        # A hand-back arms the large finish shape (1024 threads, room for 8064 records) for the next frames: the second time
        # round, the same transition is a few slower workgroups, not a handed-back frame.
        if any(f & REDONE for f in flags[1:6]):
            assert all(f & QUANTILE and not f & REDONE for f in flags[6:]), flags


def test_changed_scene_is_handed_back_and_redone():
    """The splitters are a prediction. A frame whose points sit where the last frame had few overfills a bucket: k4_colscan
    notices, the frame is redone with the fixed-grid passes (CM_PATH_REDONE, same result), the quantile passes rest and
    come back with fresh splitters."""
    n_per = 150_000
    with capi.CloudMerger(max_points_total=4 * n_per, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
        sensors, params = synth.config2_stream(0, n_per_sensor=n_per, min_pts=2)
        res, _ = frame_against_oracle(cm, sensors, params, 4 * n_per)
        needs_lds_rank(res)
        # same poses, same bounds (the corner points keep the predicted box), but nearly everything inside one cubic metre
        squeezed = []
        for s in sensors:
            a = s.data.copy()
            rng = np.random.default_rng(7)
            keep = rng.random(s.n) < 0.02
            for f in ("x", "y", "z"):
                a[f] = np.where(keep, a[f], (a[f] * np.float32(0.03)).astype(np.float32))
            squeezed.append(SensorCloud(data=a, n=s.n, q_xyzw=s.q_xyzw, t_xyz=s.t_xyz, point_step=s.point_step,
                                        off_x=s.off_x, off_y=s.off_y, off_z=s.off_z, off_i=s.off_i))
        res, _ = frame_against_oracle(cm, squeezed, params, 4 * n_per)
        assert res.path_flags & REDONE and not res.path_flags & QUANTILE
        seen_quant = False
        for k in range(1, 14):
            sensors, params = synth.config2_stream(k, n_per_sensor=n_per, min_pts=2)
            res, _ = frame_against_oracle(cm, sensors, params, 4 * n_per)
            seen_quant = seen_quant or bool(res.path_flags & QUANTILE)
        assert seen_quant, "the quantile passes come back after their rest"


def test_crop_box_frames_use_the_quantile_pass_too():
    n_per = 200_000
    with capi.CloudMerger(max_points_total=4 * n_per, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
        flags = []
        for k in range(3):
            sensors, params = synth.config2_stream(k, n_per_sensor=n_per, min_pts=2)
            params = MergeParams(leaf=params.leaf, min_points_per_voxel=2, crop_min=(-21.0, -22.0, -23.0), crop_max=(22.0, 21.0, 20.0))
            res, rep = frame_against_oracle(cm, sensors, params, 4 * n_per)
            needs_lds_rank(res)
            flags.append(res.path_flags)
        assert flags[1] & QUANTILE and flags[2] & QUANTILE, flags


def test_switch_off(monkeypatch):
    monkeypatch.setenv("CM_QUANT", "0")
    n_per = 100_000
    with capi.CloudMerger(max_points_total=4 * n_per, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
        for k in range(2):
            sensors, params = synth.config2_stream(k, n_per_sensor=n_per, min_pts=2)
            res, _ = frame_against_oracle(cm, sensors, params, 4 * n_per)
            assert not res.path_flags & QUANTILE


def test_frame_of_a_twentieth_of_its_predecessors_points():
    """The number of buckets comes from the LAST frame's size: a frame with far fewer points than its predecessor (same grid,
    so it still takes the quantile pass) has fewer 1024-slot groups than buckets. Found by scripts/fuzz_shared_bins.py: the
    per-bucket bookkeeping behind the last slot group was not zeroed (the kept voxels' group totals kept the last frame's)."""
    sensors, params = synth.config2(n_per_sensor=1_000_000, min_pts=2)
    small = [SensorCloud(data=sc.data[:50_000], n=50_000, q_xyzw=sc.q_xyzw, t_xyz=sc.t_xyz, point_step=sc.point_step,
                         off_x=sc.off_x, off_y=sc.off_y, off_z=sc.off_z, off_i=sc.off_i) for sc in sensors]
    n = sum(s.n for s in sensors)
    with capi.CloudMerger(max_points_total=n, max_sensors=len(sensors), flags=capi.FLAG_OCCUPANCY) as cm:
        seen = []
        for fr in (sensors, sensors, small, small, sensors, small):
            res, rep = frame_against_oracle(cm, fr, params, n)
            needs_lds_rank(res)
            seen.append(bool(res.path_flags & QUANTILE))
        assert seen[1] and seen[2], seen          # (the small frame right behind the large one is the case)


@pytest.mark.parametrize("n_per_sensor,shift,ballot", [(1_000_000, 1, False), (2_000_000, 2, False), (1_000_000, 1, True)])
def test_big_frame_takes_one_pass_over_shared_bins(n_per_sensor, shift, ballot, monkeypatch):
    """cfg3's dense variant (8 x 1 M / 2 M points, 86 % inside the ROI: 6.9 M / 13.7 M records): more buckets of 1920 records
    than the pass has bins, so 2 / 4 neighbouring buckets share a bin and every finish workgroup picks its bucket's records
    out of the bin by their index (k3_local<SUB>, cm_device.h cm_quant_sub_shift) — still ONE global pass where the fixed
    grid takes three."""
    assert shift == (1 if n_per_sensor == 1_000_000 else 2)
    if ballot:
        monkeypatch.setenv("CM_LDS_RANK", "0")           # (ranks by ballots in every kernel: what a device without lane-ordered LDS adds runs)
    sensors, params = synth.config3_dense(n_per_sensor=n_per_sensor, min_pts=2)
    n = sum(s.n for s in sensors)
    with capi.CloudMerger(max_points_total=n, max_sensors=len(sensors), flags=capi.FLAG_OCCUPANCY) as cm:
        seen = []
        for k in range(4):
            res, rep = frame_against_oracle(cm, sensors, params, n)
            if not ballot:
                needs_lds_rank(res)
            seen.append((res.sort_passes, bool(res.path_flags & QUANTILE), bool(res.path_flags & REDONE)))
        assert rep.n_merged > 2048 * 2600 * (1 if shift == 1 else 2), "more records than 2048 buckets hold"
        # (the fixed-grid frames in front leave the splitters — the very first may overflow a bucket and be redone with a pass
        # more; from the third frame on at the latest: one pass)
        assert not seen[0][1], seen
        assert seen[2] == (1, True, False) and seen[3] == (1, True, False), seen


def test_shared_bins_on_a_drifting_scene():
    """The same with the sensors' poses drifting from frame to frame (a centimetre in the plane and a fraction of a degree of yaw: a vehicle): the buckets no
    longer hold what the last frame's quantiles promised — a frame is either fine (every bucket still fits its workgroup) or
    handed back and redone, and right either way."""
    sensors, params = synth.config3_dense(n_per_sensor=1_000_000, min_pts=2)
    n = sum(s.n for s in sensors)
    with capi.CloudMerger(max_points_total=n, max_sensors=len(sensors), flags=capi.FLAG_OCCUPANCY) as cm:
        flags = []
        for k in range(5):
            moved = []
            for i, sc in enumerate(sensors):
                yaw = 0.002 * k * (1 if i % 2 else -1)
                dq = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
                x1, y1, z1, w1 = dq
                x2, y2, z2, w2 = sc.q_xyzw
                q = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                              w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
                moved.append(SensorCloud(data=sc.data, n=sc.n, q_xyzw=q, t_xyz=np.asarray(sc.t_xyz) + np.array([0.011, 0.007, 0.0]) * k,
                                         point_step=sc.point_step, off_x=sc.off_x, off_y=sc.off_y, off_z=sc.off_z, off_i=sc.off_i))
            res, rep = frame_against_oracle(cm, moved, params, n)
            needs_lds_rank(res)
            flags.append(res.path_flags)
        assert any(f & QUANTILE for f in flags[1:]), flags


@pytest.mark.parametrize("min_pts", [0, 2])
def test_ballot_ranked_bucket_path(min_pts, monkeypatch):
    """a device whose LDS-order probe fails (forced here: CM_LDS_RANK=0) does not lose the bucket path — its
    kernels (k2_scatter, k4_scatter, k3_local's LDS sort) rank by ballots instead of returning LDS adds (cm_common.hpp
    wave_rank_ballot): same records in the same stable order, hence the same results; CM_PATH_LDS_RANK is clear in path_flags."""
    monkeypatch.setenv("CM_LDS_RANK", "0")
    n_per = 150_000
    with capi.CloudMerger(max_points_total=4 * n_per, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
        flags = []
        for k in range(3):
            sensors, params = synth.config2_stream(k, n_per_sensor=n_per, min_pts=min_pts)
            res, rep = frame_against_oracle(cm, sensors, params, 4 * n_per)
            flags.append(res.path_flags)
        assert all(f & BUCKET and not f & 1 for f in flags), flags
        assert flags[1] & QUANTILE and flags[2] & QUANTILE and not any(f & REDONE for f in flags), flags
    # a crop box that drops most points: the packed / sparse first scatter, ballot-ranked
    sensors, params = synth.config3(n_per_sensor=300_000, min_pts=min_pts)
    n = sum(s.n for s in sensors)
    with capi.CloudMerger(max_points_total=n, max_sensors=len(sensors), flags=capi.FLAG_OCCUPANCY) as cm:
        for k in range(3):
            res, rep = frame_against_oracle(cm, sensors, params, n)
            assert res.path_flags & BUCKET and not res.path_flags & 1


def test_what_changes_between_frames():
    """The splitters belong to a grid and a scene. Between frames on one context: a sensor drops out (fewer records: the
    splitters still apply), min_points_per_voxel changes (same grid: they apply), the leaf changes (another grid: the fixed-grid
    passes run and leave new ones), a frame of empty clouds in between (the splitters survive it). Every frame against the oracle."""
    n_per = 120_000
    with capi.CloudMerger(max_points_total=4 * n_per, max_sensors=4, flags=capi.FLAG_OCCUPANCY) as cm:
        def frame(k, sensors_used=4, **over):
            sensors, params = synth.config2_stream(k, n_per_sensor=n_per, min_pts=2)
            for name, v in over.items():
                setattr(params, name, v)
            for s in range(sensors_used, 4):
                cm.clear(s)
            res, rep = frame_against_oracle(cm, sensors[:sensors_used], params, 4 * n_per)
            needs_lds_rank(res)
            return res.path_flags
        f0 = frame(0)
        f1 = frame(1)
        f2 = frame(2, sensors_used=3)                              # one sensor silent: three quarters of the records
        f3 = frame(3, min_points_per_voxel=0)
        f4 = frame(4, leaf=(0.08, 0.08, 0.08))                     # another grid
        f5 = frame(5, leaf=(0.08, 0.08, 0.08))
        assert not f0 & QUANTILE and f1 & QUANTILE, (f0, f1, f2, f3)
        assert (f2 & QUANTILE) or (f2 & REDONE), f2                # (tried; three quarters of the points in the same buckets: fits)
        assert (f3 & QUANTILE) or (f3 & REDONE), f3                # (tried; the fourth sensor's points come back: a bucket may overflow)
        assert not f4 & QUANTILE and f5 & QUANTILE, (f4, f5)
        # a frame of empty clouds, then the stream goes on
        empty = [SensorCloud(data=np.zeros(0, dtype=s.data.dtype), n=0, q_xyzw=s.q_xyzw, t_xyz=s.t_xyz) for s in synth.config2_stream(0, n_per_sensor=8)[0]]
        cm.submit_all(empty)
        res = cm.merge_voxelize(synth.config2_stream(0, n_per_sensor=8, min_pts=2)[1])
        assert res.status == capi.EMPTY_INPUT
        f6 = frame(0, leaf=(0.08, 0.08, 0.08))
        assert f6 & BUCKET
