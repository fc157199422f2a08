"""ctypes front-end of the CPU oracle (oracle/libcm_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/cm_oracle.h for provenance: a restatement of PCL 1.8.1 / pcl_ros semantics, parity
unpinned by the reference, pinned by tests/golden known-answer cases).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcm_oracle.so")

POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"),
                        ("intensity", "<f4"), ("_unused", "<f4", (3,))])
assert POINT_DTYPE.itemsize == 32

NO_FIELD = 0xFFFFFFFF
OK, EMPTY_INPUT, GRID_OVERFLOW, BAD_ARG = 0, 1, 2, -1


class _Sensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_uint32), ("point_step", C.c_uint32),
                ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32),
                ("off_i", C.c_uint32), ("q_xyzw", C.c_double * 4), ("t_xyz", C.c_double * 3),
                ("is_dense", C.c_int32), ("_pad", C.c_int32)]


class _Params(C.Structure):
    _fields_ = [("leaf", C.c_float * 3), ("min_points_per_voxel", C.c_uint32),
                ("downsample_all_data", C.c_int32), ("crop_enable", C.c_int32),
                ("crop_min", C.c_float * 3), ("crop_max", C.c_float * 3),
                ("outlier_enable", C.c_int32), ("outlier_radius", C.c_float), ("outlier_min_neighbors", C.c_uint32)]


class PlaneResult(C.Structure):
    _fields_ = [("plane", C.c_float * 4), ("n_inliers", C.c_uint32), ("best_hypothesis", C.c_uint32),
                ("iterations", C.c_uint32), ("found", C.c_int32)]


class Report(C.Structure):
    _fields_ = [("status", C.c_int32), ("threads_used", C.c_int32),
                ("n_in", C.c_uint64), ("n_merged", C.c_uint64), ("n_out", C.c_uint64),
                ("min_b", C.c_int32 * 3), ("max_b", C.c_int32 * 3), ("div_b", C.c_int32 * 3),
                ("min_p", C.c_float * 3), ("max_p", C.c_float * 3),
                ("t_ingest_s", C.c_double), ("t_transform_crop_s", C.c_double),
                ("t_concat_s", C.c_double), ("t_voxel_s", C.c_double), ("t_total_s", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("cm_oracle.cpp", "cm_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "libcm_oracle.so"], check=True,
                   stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_quat_to_matrix.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float)]
        L.orc_quat_to_matrix.restype = None
        L.orc_transform.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_int, C.c_void_p]
        L.orc_transform.restype = None
        L.orc_crop.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]
        L.orc_crop.restype = C.c_size_t
        L.orc_radius_outlier_removal.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_radius_outlier_removal.restype = C.c_size_t
        L.orc_ransac_plane.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_uint64,
                                       C.c_uint32, C.POINTER(PlaneResult), C.c_void_p]
        L.orc_ransac_plane.restype = None
        L.orc_voxel_cells.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_void_p]
        L.orc_voxel_cells.restype = None
        L.orc_voxelgrid.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_uint32, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(Report),
                                    C.c_void_p, C.c_void_p]
        L.orc_voxelgrid.restype = C.c_int
        L.orc_merge_voxelize.argtypes = [C.POINTER(_Sensor), C.c_int, C.POINTER(_Params), C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.POINTER(Report), C.c_void_p, C.c_void_p]
        L.orc_merge_voxelize.restype = C.c_int
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def quat_to_matrix(q_xyzw, t_xyz):
    m = (C.c_float * 12)()
    lib().orc_quat_to_matrix((C.c_double * 4)(*q_xyzw), (C.c_double * 3)(*t_xyz), m)
    return np.array(m, dtype=np.float32).reshape(3, 4)


def make_points(xyz, intensity=None):
    """(n,3) float32 [+ (n,) intensity] -> 32-B PointXYZI records."""
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    p = np.zeros(len(xyz), dtype=POINT_DTYPE)
    p["x"], p["y"], p["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    p["pad"] = 1.0
    if intensity is not None:
        p["intensity"] = np.asarray(intensity, dtype=np.float32)
    return p


def transform(points, m, is_dense=True):
    points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
    out = np.empty_like(points)
    mm = (C.c_float * 12)(*np.asarray(m, dtype=np.float32).reshape(-1))
    lib().orc_transform(points.ctypes.data, len(points), mm, int(is_dense), out.ctypes.data)
    return out


def crop(points, mn, mx):
    points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
    out = np.empty_like(points)
    k = lib().orc_crop(points.ctypes.data, len(points), _f3(mn), _f3(mx), out.ctypes.data)
    return out[:k].copy()


def radius_outlier_removal(points, radius, min_neighbors=1):
    """Returns (survivors, keep mask)."""
    points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
    out = np.empty_like(points)
    mask = np.zeros(len(points), dtype=np.uint8)
    k = lib().orc_radius_outlier_removal(points.ctypes.data, len(points), float(radius), int(min_neighbors),
                                         out.ctypes.data, mask.ctypes.data)
    return out[:k].copy(), mask.astype(bool)


def ransac_plane(points, max_iterations=1000, threshold=0.3, probability=0.99, optimize=True, seed=12345, zone_key=0):
    """Ground plane of one slab's band points (orc_ransac_plane). Returns (PlaneResult, inlier mask)."""
    points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
    res = PlaneResult()
    mask = np.zeros(len(points), dtype=np.uint8)
    lib().orc_ransac_plane(points.ctypes.data, len(points), int(max_iterations), float(threshold), float(probability),
                           int(optimize), int(seed), int(zone_key), C.byref(res), mask.ctypes.data)
    return res, mask.astype(bool)


def ground_split(points, zones, sensor, gp):
    """The per-sensor stage of the live node for one transformed + cropped cloud (proceedFront / proceedRear
    and the top / Livox callbacks, pc_preprocessing_main.cpp:228-312, :436-497), composed from the pieces above:
    for every slab (x_min, x_length, z_max_ground) in order — a point on a border goes to the first slab only —
    band / above-band split (removeGround :81-91), plane on the band, inliers = ground. gp: dict with
    max_iterations, threshold, probability, optimize, z_keep_max, seed, and optionally outlier_radius /
    outlier_min_neighbors: removeGround's outlierRemoval(no_ground_cloud_ptr) (:119) on the band points of the
    slab that are not ground. Returns (keep mask, ground mask, planes) over `points` (no-ground = points[keep],
    ground = points[ground])."""
    x, z = points["x"], points["z"]
    taken = np.zeros(len(points), dtype=bool)
    keep = np.zeros(len(points), dtype=bool)
    ground = np.zeros(len(points), dtype=bool)
    planes = []
    for k, (x0, ln, zm) in enumerate(zones):
        x0 = np.float32(x0); x1 = np.float32(x0 + np.float32(ln)); zm = np.float32(zm)
        inz = ~taken & ~((x < x0) | (x > x1))
        taken |= inz
        if zm < 0:
            keep |= inz
            planes.append(None)
            continue
        band = inz & ~((z < -zm) | (z > zm))
        zlo = np.float32(np.float64(zm) + 0.01)
        above = inz & ~band & ~((z < zlo) | (z > np.float32(gp["z_keep_max"])))
        keep |= above
        idx = np.nonzero(band)[0]
        res, inl = ransac_plane(points[idx], gp["max_iterations"], gp["threshold"], gp["probability"], gp["optimize"],
                                gp["seed"], sensor * 8 + k)
        ground[idx[inl]] = True
        rest = idx[~inl]
        if gp.get("outlier_radius", 0) and len(rest):
            _, ok = radius_outlier_removal(points[rest], gp["outlier_radius"], gp.get("outlier_min_neighbors", 1))
            rest = rest[ok]
        keep[rest] = True
        planes.append(res)
    return keep, ground, planes


def voxel_cells(points, leaf):
    points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
    ijk = np.empty((len(points), 3), dtype=np.int32)
    lib().orc_voxel_cells(points.ctypes.data, len(points), _f3(leaf), ijk.ctypes.data)
    return ijk


def voxelgrid(points, leaf, min_pts=0, downsample_all=True, is_dense=True, stable=False):
    points = np.ascontiguousarray(points, dtype=POINT_DTYPE)
    out = np.empty(max(len(points), 1), dtype=POINT_DTYPE)
    n_out = C.c_size_t(0)
    rep = Report()
    cells = np.zeros((len(out), 3), dtype=np.int32)
    counts = np.zeros(len(out), dtype=np.uint32)
    st = lib().orc_voxelgrid(points.ctypes.data, len(points), _f3(leaf), int(min_pts), int(downsample_all),
                             int(is_dense), int(stable), out.ctypes.data, C.byref(n_out), C.byref(rep),
                             cells.ctypes.data, counts.ctypes.data)
    rep.cells, rep.counts = cells[:n_out.value].copy(), counts[:n_out.value].copy()
    return st, out[:n_out.value].copy(), rep


def merge_voxelize(sensors, params, threads=1, stable=False, want_merged=True):
    """sensors: objects with .data (contiguous ndarray), .n, .point_step, .off_x/.off_y/.off_z/.off_i
    (None => absent), .q_xyzw, .t_xyz, .is_dense.  params: object with .leaf, .min_points_per_voxel,
    .downsample_all_data, .crop_min/.crop_max (None => crop disabled).
    Returns (status, merged or None, out, Report)."""
    arr = (_Sensor * len(sensors))()
    keep = []
    n_in = 0
    for k, s in enumerate(sensors):
        d = np.ascontiguousarray(s.data)
        keep.append(d)
        arr[k].data = d.ctypes.data
        arr[k].n = int(s.n)
        arr[k].point_step = int(s.point_step)
        arr[k].off_x, arr[k].off_y, arr[k].off_z = int(s.off_x), int(s.off_y), int(s.off_z)
        arr[k].off_i = NO_FIELD if s.off_i is None else int(s.off_i)
        arr[k].q_xyzw = (C.c_double * 4)(*[float(v) for v in s.q_xyzw])
        arr[k].t_xyz = (C.c_double * 3)(*[float(v) for v in s.t_xyz])
        arr[k].is_dense = int(bool(s.is_dense))
        n_in += int(s.n)
    p = _Params()
    p.leaf = _f3(params.leaf)
    p.min_points_per_voxel = int(params.min_points_per_voxel)
    p.downsample_all_data = int(bool(params.downsample_all_data))
    if params.crop_min is not None:
        p.crop_enable = 1
        p.crop_min, p.crop_max = _f3(params.crop_min), _f3(params.crop_max)
    if getattr(params, "outlier_radius", None):
        p.outlier_enable = 1
        p.outlier_radius = float(params.outlier_radius)
        p.outlier_min_neighbors = int(params.outlier_min_neighbors)
    merged = np.empty(max(n_in, 1), dtype=POINT_DTYPE) if want_merged else None
    out = np.empty(max(n_in, 1), dtype=POINT_DTYPE)
    rep = Report()
    cells = np.zeros((len(out), 3), dtype=np.int32)
    counts = np.zeros(len(out), dtype=np.uint32)
    st = lib().orc_merge_voxelize(arr, len(sensors), C.byref(p), int(threads), int(stable),
                                  merged.ctypes.data if want_merged else None, out.ctypes.data, C.byref(rep),
                                  cells.ctypes.data, counts.ctypes.data)
    if st == BAD_ARG:
        raise ValueError("orc_merge_voxelize: bad argument")
    m = merged[:rep.n_merged].copy() if want_merged else None
    # occupancy of the kept voxels (only meaningful for status OK): absolute cells and point counts
    rep.cells = cells[:rep.n_out].copy() if st == OK else None
    rep.counts = counts[:rep.n_out].copy() if st == OK else None
    return st, m, out[:rep.n_out].copy(), rep
