"""Independent numpy restatement of the same path as oracle/cm_oracle.cpp.  TEST INFRASTRUCTURE ONLY.

Written separately from the C++ oracle (array formulation, stable argsort) so the two can check
each other: occupancy, counts, order and the fp32 transform/crop must agree bit-for-bit; centroid
sums agree bit-for-bit in `sequential` mode with the C++ oracle's stable_ties mode.
Semantics: SURVEY.md Appendix A (PCL 1.8.1 / pcl_ros / Eigen 3.3, recalled — parity unpinned).
Call sites restated: pc_preprocessing_main.cpp:322 (transform), :20-40 (getROI), :137-142
(concatenate), :171-176 (VoxelGrid).
"""
import numpy as np

F = np.float32
INT32_MAX = 2**31 - 1
OK, EMPTY_INPUT, GRID_OVERFLOW = 0, 1, 2


def quat_to_matrix(q_xyzw, t_xyz):
    x, y, z, w = (F(v) for v in q_xyzw)          # double -> float per component
    two = F(2)
    tx, ty, tz = two * x, two * y, two * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = F(1)
    m = np.empty((3, 4), dtype=F)
    m[0, 0], m[0, 1], m[0, 2] = one - (tyy + tzz), txy - twz, txz + twy
    m[1, 0], m[1, 1], m[1, 2] = txy + twz, one - (txx + tzz), tyz - twx
    m[2, 0], m[2, 1], m[2, 2] = txz - twy, tyz + twx, one - (txx + tyy)
    m[:, 3] = [F(t_xyz[0]), F(t_xyz[1]), F(t_xyz[2])]
    return m


def ingest(s):
    """PointCloud2 payload -> (xyz (n,3) f32, intensity (n,) f32)."""
    raw = np.ascontiguousarray(s.data).view(np.uint8).reshape(-1)[: s.n * s.point_step]
    raw = raw.reshape(s.n, s.point_step)

    def field(off):
        return np.ascontiguousarray(raw[:, off:off + 4]).view("<f4").reshape(-1).copy()

    xyz = np.stack([field(s.off_x), field(s.off_y), field(s.off_z)], axis=1)
    inten = field(s.off_i) if s.off_i is not None else np.zeros(s.n, dtype=F)
    return xyz, inten


def transform(xyz, m, is_dense=True):
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    with np.errstate(all="ignore"):
        out = np.stack([((m[r, 0] * x + m[r, 1] * y) + m[r, 2] * z) + m[r, 3] for r in range(3)], axis=1)
    if not is_dense:
        bad = ~np.isfinite(xyz).all(axis=1)
        out[bad] = xyz[bad]
    return out.astype(F)


def crop_mask(xyz, mn, mx):
    mn = np.asarray(mn, dtype=F)
    mx = np.asarray(mx, dtype=F)
    with np.errstate(invalid="ignore"):
        return np.isfinite(xyz).all(axis=1) & (xyz >= mn).all(axis=1) & (xyz <= mx).all(axis=1)


def cells(xyz, leaf):
    inv = F(1) / np.asarray(leaf, dtype=F)
    return np.floor(xyz * inv).astype(np.int64)


def voxelgrid(xyz, inten, leaf, min_pts=0, downsample_all=True, sequential=False):
    """Returns (status, out_xyz, out_intensity, counts, cell_ijk_abs). Finite input assumed."""
    n = len(xyz)
    if n == 0:
        z = np.zeros((0, 3), dtype=F)
        return EMPTY_INPUT, z, np.zeros(0, dtype=F), np.zeros(0, dtype=np.int64), np.zeros((0, 3), dtype=np.int64)
    inv = F(1) / np.asarray(leaf, dtype=F)
    min_p, max_p = xyz.min(axis=0), xyz.max(axis=0)
    d = ((max_p - min_p) * inv).astype(np.int64) + 1       # fp32 product, truncation
    if int(d[0]) * int(d[1]) * int(d[2]) > INT32_MAX:
        return GRID_OVERFLOW, xyz.copy(), inten.copy(), np.ones(n, dtype=np.int64), cells(xyz, leaf)
    min_b = np.floor(min_p * inv).astype(np.int64)
    max_b = np.floor(max_p * inv).astype(np.int64)
    div_b = max_b - min_b + 1
    rel = (np.floor(xyz * inv) - min_b.astype(F)).astype(np.int64)
    key = rel[:, 0] + rel[:, 1] * div_b[0] + rel[:, 2] * div_b[0] * div_b[1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    head = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    cnt = np.diff(np.r_[head, n])
    keep = cnt >= min_pts
    vals = np.concatenate([xyz, inten[:, None]], axis=1)[order]
    if sequential:                                           # fp32 running sums, run order
        sums = np.zeros((len(head), 4), dtype=F)
        for r, (h, c) in enumerate(zip(head, cnt)):
            acc = np.zeros(4, dtype=F)
            for k in range(h, h + c):
                acc = acc + vals[k]
            sums[r] = acc
        mean = (sums / cnt[:, None].astype(F)).astype(F)
    else:                                                    # fp64 means (ground truth)
        sums = np.add.reduceat(vals.astype(np.float64), head, axis=0)
        mean = (sums / cnt[:, None]).astype(F)
    abs_cell = (rel[order][head] + min_b)[keep]
    out_i = mean[keep, 3] if downsample_all else np.zeros(int(keep.sum()), dtype=F)
    return OK, mean[keep, :3], out_i, cnt[keep], abs_cell


def merge(sensors, params):
    """transform + crop + concatenate. Returns merged (xyz, intensity)."""
    xs, ins = [], []
    for s in sensors:
        xyz, inten = ingest(s)
        xyz = transform(xyz, quat_to_matrix(s.q_xyzw, s.t_xyz), s.is_dense)
        if params.crop_min is not None:
            k = crop_mask(xyz, params.crop_min, params.crop_max)
            xyz, inten = xyz[k], inten[k]
        xs.append(xyz)
        ins.append(inten)
    return np.concatenate(xs), np.concatenate(ins)


def radius_outlier_mask(xyz, radius, min_neighbors=1):
    """Independent restatement with scipy's kd-tree for candidates (float64, slightly enlarged radius) and
    the fp32 test ((dx*dx + dy*dy) + dz*dz) < float(r*r) on the candidates."""
    from scipy.spatial import cKDTree
    r2 = F(float(radius) * float(radius))
    ok = np.isfinite(xyz).all(axis=1)
    idx = np.flatnonzero(ok)
    keep = np.zeros(len(xyz), dtype=bool)
    if len(idx) == 0:
        return keep
    pts = xyz[idx]
    tree = cKDTree(pts.astype(np.float64))
    cand = tree.query_ball_point(pts.astype(np.float64), float(radius) * 1.001)
    for a, lst in enumerate(cand):
        q = pts[np.asarray(lst, dtype=np.int64)]
        d = pts[a] - q
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        keep[idx[a]] = int((d2 < r2).sum()) > min_neighbors        # k includes the point itself
    return keep


def merge_voxelize(sensors, params, sequential=False):
    xyz, inten = merge(sensors, params)
    if getattr(params, "outlier_radius", None):
        k = radius_outlier_mask(xyz, params.outlier_radius, params.outlier_min_neighbors)
        xyz, inten = xyz[k], inten[k]
    return voxelgrid(xyz, inten, params.leaf, params.min_points_per_voxel,
                     params.downsample_all_data, sequential)
