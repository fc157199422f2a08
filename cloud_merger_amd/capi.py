"""ctypes binding of include/cloudmerge.h (libcloudmerge_hip.so).

This is plumbing for tests and bench.py: every call goes straight through the C-ABI. The library
is loaded from the in-tree build (cloud_merger_amd/lib/); a missing library is an error — there is
no Python or CPU fallback for the path.
"""
import ctypes as C
import os

import numpy as np

from .types import MergeParams, SensorCloud, XYZI_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libcloudmerge_hip.so")

MAX_SENSORS = 16
NO_FIELD = 0xFFFFFFFF
MAX_STAGES = 48

OK, EMPTY_INPUT, GRID_OVERFLOW, NOT_READY, SKIPPED = 0, 1, 2, 3, 4
BAD_ARG, HIP_ERROR, NO_DEVICE, CAPACITY, INTERNAL = -1, -2, -3, -4, -5
FLAG_PROFILE, FLAG_LATEST_WINS, FLAG_OCCUPANCY = 0x1, 0x2, 0x4
# cm_result.path_flags (CM_PATH_*)
PATH_LDS_RANK, PATH_BUCKET, PATH_PREDICTED, PATH_REDONE, PATH_PACKED, PATH_SPLIT, PATH_QUANTILE = 1, 2, 4, 8, 16, 32, 64

# Every symbol include/cloudmerge.h declares (tests/test_capi_symbols.py checks both directions).
SYMBOLS = [
    "cm_create", "cm_destroy", "cm_set_stream", "cm_set_sensor_transform", "cm_set_sensor_matrix",
    "cm_get_sensor_matrix", "cm_submit_cloud", "cm_submit_cloud_device", "cm_clear_sensor",
    "cm_merge_voxelize", "cm_merge_voxelize_async", "cm_wait", "cm_result_copy", "cm_result_device",
    "cm_result_copy_cells", "cm_merged_copy", "cm_get_stage_times", "cm_status_string",
    "cm_last_error", "cm_version", "cm_host_alloc", "cm_host_free",
    "cm_local_bounds", "cm_merge_partial", "cm_partial_device", "cm_partial_copy", "cm_merge_tables",
    "cm_set_ground_removal", "cm_ground_copy", "cm_ground_planes",
    "cm_submit_cloud_async", "cm_result_copy_async", "cm_sync", "cm_get_frame_stats",
    "cm_result_publish_async", "cm_publish_wait", "cm_host_register", "cm_host_unregister",
]
MAX_ZONES = 8

# cm_partial_entry (32 bytes)
ENTRY_DTYPE = np.dtype([("key", "<u4"), ("count", "<u4"), ("sx", "<f4"), ("sy", "<f4"), ("sz", "<f4"),
                        ("si", "<f4"), ("_pad", "<u4", (2,))])
assert ENTRY_DTYPE.itemsize == 32


class Limits(C.Structure):
    _fields_ = [("max_sensors", C.c_uint32), ("flags", C.c_uint32), ("max_points_total", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("leaf", C.c_float * 3), ("min_points_per_voxel", C.c_uint32),
                ("downsample_all_data", C.c_int32), ("crop_enable", C.c_int32),
                ("crop_min", C.c_float * 3), ("crop_max", C.c_float * 3),
                ("required_sensor_mask", C.c_uint32), ("outlier_enable", C.c_int32),
                ("outlier_radius", C.c_float), ("outlier_min_neighbors", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_sensors", C.c_uint32), ("n_in", C.c_uint64),
                ("n_merged", C.c_uint64), ("n_out", C.c_uint64),
                ("min_b", C.c_int32 * 3), ("max_b", C.c_int32 * 3), ("div_b", C.c_int32 * 3),
                ("min_p", C.c_float * 3), ("max_p", C.c_float * 3),
                ("bounds_from_crop", C.c_uint32), ("key_bits", C.c_uint32), ("sort_passes", C.c_uint32),
                ("path_flags", C.c_uint32), ("device_ms", C.c_float)]


class Zone(C.Structure):
    _fields_ = [("x_min", C.c_float), ("x_length", C.c_float), ("z_max_ground", C.c_float)]


class GroundParams(C.Structure):
    _fields_ = [("max_iterations", C.c_uint32), ("distance_threshold", C.c_float), ("probability", C.c_float),
                ("optimize_coefficients", C.c_int32), ("z_keep_max", C.c_float), ("outlier_radius", C.c_float),
                ("outlier_min_neighbors", C.c_uint32), ("_pad", C.c_uint32), ("seed", C.c_uint64),
                ("n_zones", C.c_uint32 * MAX_SENSORS), ("zones", (Zone * MAX_ZONES) * MAX_SENSORS)]


class GroundPlane(C.Structure):
    _fields_ = [("plane", C.c_float * 4), ("band_points", C.c_uint32), ("inliers", C.c_uint32),
                ("iterations", C.c_uint32), ("found", C.c_int32)]


def make_ground_params(zones_per_sensor, max_iterations=1000, distance_threshold=0.3, probability=0.99,
                       optimize=True, z_keep_max=3.0, seed=12345, outlier_radius=0.0, outlier_min_neighbors=1):
    """zones_per_sensor: list (one entry per sensor) of lists of (x_min, x_length, z_max_ground); a negative
    z_max_ground keeps the slab whole. Defaults: the reference's Parameter.h:35-42."""
    g = GroundParams()
    g.max_iterations, g.distance_threshold, g.probability = max_iterations, distance_threshold, probability
    g.optimize_coefficients, g.z_keep_max, g.seed = int(optimize), z_keep_max, seed
    g.outlier_radius, g.outlier_min_neighbors = outlier_radius, outlier_min_neighbors
    for s, zs in enumerate(zones_per_sensor):
        g.n_zones[s] = len(zs)
        for k, (x0, ln, zm) in enumerate(zs):
            g.zones[s][k] = Zone(x0, ln, zm)
    return g


class FrameStats(C.Structure):
    _fields_ = [("n_sensors", C.c_uint32), ("_pad", C.c_uint32), ("sensor", C.c_uint32 * MAX_SENSORS),
                ("n_in", C.c_uint32 * MAX_SENSORS), ("n_kept", C.c_uint32 * MAX_SENSORS), ("fresh", C.c_uint32 * MAX_SENSORS),
                ("generation", C.c_uint64 * MAX_SENSORS), ("bytes_h2d", C.c_uint64 * MAX_SENSORS), ("bytes_h2d_total", C.c_uint64), ("bytes_d2h_total", C.c_uint64),
                ("bytes_algorithmic", C.c_uint64)]


class StageTimes(C.Structure):
    _fields_ = [("n_stages", C.c_uint32), ("_pad", C.c_uint32),
                ("name", (C.c_char * 24) * MAX_STAGES), ("ms", C.c_float * MAX_STAGES)]


class CloudMergeError(RuntimeError):
    def __init__(self, status, what=""):
        super().__init__(f"{status_string(status)} ({status}) {what}".strip())
        self.status = status


_lib = None


def load():
    """Load the in-tree HIP library; raises if it has not been built.

    A process that also uses PyTorch must `import torch` BEFORE this is called: torch bundles its
    own libamdhip64/libhsa-runtime64, and the first HIP runtime loaded serves the whole process
    (same soname). Loaded the other way round torch finds no GPU."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("CM_LIB_VARIANT") == "testhooks":      # the test build (python -m cloud_merger_amd.build --test-hooks)
        path = os.path.join(HERE, "lib", "libcloudmerge_hip_testhooks.so")
    if not os.path.exists(path):
        raise OSError(f"{path} is missing: build it with `python -m cloud_merger_amd.build` "
                      "(the path has no fallback implementation)")
    L = C.CDLL(path)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.cm_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Limits)]
    L.cm_destroy.argtypes = [vp]
    L.cm_set_stream.argtypes = [vp, vp]
    L.cm_set_sensor_transform.argtypes = [vp, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.cm_set_sensor_matrix.argtypes = [vp, u32, C.POINTER(C.c_float)]
    L.cm_get_sensor_matrix.argtypes = [vp, u32, C.POINTER(C.c_float)]
    L.cm_submit_cloud.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, u32]
    L.cm_submit_cloud_device.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, u32]
    L.cm_result_publish_async.argtypes = [vp, vp, u64, u32]
    L.cm_publish_wait.argtypes = [vp]
    L.cm_host_register.argtypes = [vp, C.c_size_t]
    L.cm_host_unregister.argtypes = [vp]
    L.cm_submit_cloud_async.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, u32]
    L.cm_result_copy_async.argtypes = [vp, vp, u64]
    L.cm_sync.argtypes = [vp]
    L.cm_get_frame_stats.argtypes = [vp, C.POINTER(FrameStats)]
    L.cm_clear_sensor.argtypes = [vp, u32]
    L.cm_merge_voxelize.argtypes = [vp, C.POINTER(Params), C.POINTER(Result)]
    L.cm_merge_voxelize_async.argtypes = [vp, C.POINTER(Params)]
    L.cm_wait.argtypes = [vp, C.POINTER(Result)]
    L.cm_result_copy.argtypes = [vp, vp, u64, u32]
    L.cm_result_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.cm_result_copy_cells.argtypes = [vp, vp, vp, u64]
    L.cm_merged_copy.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.cm_get_stage_times.argtypes = [vp, C.POINTER(StageTimes)]
    L.cm_status_string.argtypes = [C.c_int]
    L.cm_status_string.restype = C.c_char_p
    L.cm_last_error.argtypes = [vp]
    L.cm_last_error.restype = C.c_char_p
    L.cm_version.argtypes = []
    L.cm_local_bounds.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(u64)]
    L.cm_merge_partial.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_float), C.POINTER(Result)]
    L.cm_partial_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.cm_partial_copy.argtypes = [vp, vp, u64]
    L.cm_merge_tables.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), u32, C.POINTER(Params), C.POINTER(Result)]
    L.cm_set_ground_removal.argtypes = [vp, C.POINTER(GroundParams)]
    L.cm_ground_copy.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.cm_ground_planes.argtypes = [vp, C.POINTER(GroundPlane), u32]
    L.cm_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.cm_host_free.argtypes = [vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("cm_status_string", "cm_last_error"):
            fn.restype = C.c_int
    _lib = L
    return L


def status_string(status):
    return load().cm_status_string(int(status)).decode()


def make_params(p: MergeParams) -> Params:
    cp = Params()
    cp.leaf = (C.c_float * 3)(*[float(v) for v in p.leaf])
    cp.min_points_per_voxel = int(p.min_points_per_voxel)
    cp.downsample_all_data = int(bool(p.downsample_all_data))
    if p.crop_min is not None:
        cp.crop_enable = 1
        cp.crop_min = (C.c_float * 3)(*[float(v) for v in p.crop_min])
        cp.crop_max = (C.c_float * 3)(*[float(v) for v in p.crop_max])
    cp.required_sensor_mask = int(p.required_sensor_mask)
    if getattr(p, "outlier_radius", None):
        cp.outlier_enable = 1
        cp.outlier_radius = float(p.outlier_radius)
        cp.outlier_min_neighbors = int(p.outlier_min_neighbors)
    return cp


def pinned_array(nbytes):
    """uint8 numpy view of nbytes of pinned host memory (cm_host_alloc); keep the returned holder alive
    and call holder.free() when done."""
    L = load()
    ptr = C.c_void_p()
    st = L.cm_host_alloc(C.byref(ptr), int(nbytes))
    if st != OK:
        raise CloudMergeError(st, "cm_host_alloc")
    buf = (C.c_uint8 * int(nbytes)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=np.uint8)

    class _Holder:
        def __init__(self):
            self.array, self.ptr = arr, ptr

        def free(self):
            if self.ptr:
                L.cm_host_free(self.ptr)
                self.ptr = None

    return _Holder()


class CloudMerger:
    """Thin object wrapper over a cm_ctx. Mirrors the calling pattern of the reference node:
    set the static transforms once (:556-561), submit one cloud per sensor (callbacks :318-508),
    then merge_voxelize() (main loop :574-577) and fetch the voxel cloud (:215-219)."""

    def __init__(self, max_points_total, max_sensors=MAX_SENSORS, device=0, flags=0):
        self._lib = load()
        self._ctx = C.c_void_p()
        lim = Limits(int(max_sensors), int(flags), int(max_points_total))
        st = self._lib.cm_create(C.byref(self._ctx), int(device), C.byref(lim))
        if st != OK:
            self._ctx = None
            raise CloudMergeError(st, "cm_create")
        self.flags = flags
        self.max_points_total = int(max_points_total)
        self._keep = {}

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.cm_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st, what, ok=(OK,)):
        if st not in ok:
            raise CloudMergeError(st, f"{what}: {self._lib.cm_last_error(self._ctx).decode()}")
        return st

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.cm_set_stream(self._ctx, C.c_void_p(hip_stream_ptr)), "cm_set_stream")

    def set_transform(self, sensor, q_xyzw, t_xyz):
        q = (C.c_double * 4)(*[float(v) for v in q_xyzw])
        t = (C.c_double * 3)(*[float(v) for v in t_xyz])
        self._check(self._lib.cm_set_sensor_transform(self._ctx, sensor, q, t), "cm_set_sensor_transform")

    def set_matrix(self, sensor, m):
        mm = (C.c_float * 12)(*np.asarray(m, dtype=np.float32).reshape(-1))
        self._check(self._lib.cm_set_sensor_matrix(self._ctx, sensor, mm), "cm_set_sensor_matrix")

    def get_matrix(self, sensor):
        mm = (C.c_float * 12)()
        self._check(self._lib.cm_get_sensor_matrix(self._ctx, sensor, mm), "cm_get_sensor_matrix")
        return np.array(mm, dtype=np.float32).reshape(3, 4)

    def submit(self, sensor, cloud: SensorCloud):
        data = np.ascontiguousarray(cloud.data)
        off_i = NO_FIELD if cloud.off_i is None else cloud.off_i
        return self._check(self._lib.cm_submit_cloud(self._ctx, sensor, data.ctypes.data, cloud.n, cloud.point_step,
                                                     cloud.off_x, cloud.off_y, cloud.off_z, off_i), "cm_submit_cloud",
                           ok=(OK, SKIPPED))

    def submit_async(self, sensor, cloud: SensorCloud, host_ptr=None):
        """cm_submit_cloud_async: the payload (cloud.data, or host_ptr: e.g. pinned memory) must stay alive and unchanged
        until the frame that consumes it has been waited for."""
        ptr = host_ptr if host_ptr is not None else np.ascontiguousarray(cloud.data).ctypes.data
        off_i = NO_FIELD if cloud.off_i is None else cloud.off_i
        return self._check(self._lib.cm_submit_cloud_async(self._ctx, sensor, C.c_void_p(ptr), cloud.n, cloud.point_step,
                                                           cloud.off_x, cloud.off_y, cloud.off_z, off_i), "cm_submit_cloud_async",
                           ok=(OK, SKIPPED))

    def result_async(self, host_ptr, capacity):
        self._check(self._lib.cm_result_copy_async(self._ctx, C.c_void_p(host_ptr), int(capacity)), "cm_result_copy_async")

    def sync(self):
        self._check(self._lib.cm_sync(self._ctx), "cm_sync")

    def publish_async(self, host_ptr, capacity, step_out=16):
        """Copy-out of the last waited-for frame on the context's publish stream (cm_result_publish_async)."""
        self._check(self._lib.cm_result_publish_async(self._ctx, C.c_void_p(host_ptr), int(capacity), int(step_out)),
                    "cm_result_publish_async")

    def publish_wait(self):
        self._check(self._lib.cm_publish_wait(self._ctx), "cm_publish_wait")

    def frame_stats(self):
        fs = FrameStats()
        self._check(self._lib.cm_get_frame_stats(self._ctx, C.byref(fs)), "cm_get_frame_stats")
        k = fs.n_sensors
        return {"sensor": list(fs.sensor[:k]), "n_in": list(fs.n_in[:k]), "n_kept": list(fs.n_kept[:k]), "fresh": list(fs.fresh[:k]),
                "generation": list(fs.generation[:k]), "bytes_h2d": list(fs.bytes_h2d[:k]), "bytes_h2d_total": fs.bytes_h2d_total, "bytes_d2h_total": fs.bytes_d2h_total,
                "bytes_algorithmic": fs.bytes_algorithmic}

    def submit_device(self, sensor, dev_ptr, n, point_step=16, off_x=0, off_y=4, off_z=8, off_i=12):
        off_i = NO_FIELD if off_i is None else off_i
        return self._check(self._lib.cm_submit_cloud_device(self._ctx, sensor, C.c_void_p(dev_ptr), n, point_step,
                                                            off_x, off_y, off_z, off_i), "cm_submit_cloud_device",
                           ok=(OK, SKIPPED))

    def clear(self, sensor):
        self._check(self._lib.cm_clear_sensor(self._ctx, sensor), "cm_clear_sensor")

    def submit_all(self, sensors):
        for s, cloud in enumerate(sensors):
            self.set_transform(s, cloud.q_xyzw, cloud.t_xyz)
            self.submit(s, cloud)

    def merge_voxelize(self, params: MergeParams) -> Result:
        res = Result()
        st = self._lib.cm_merge_voxelize(self._ctx, C.byref(make_params(params)), C.byref(res))
        self._check(st, "cm_merge_voxelize", ok=(OK, EMPTY_INPUT, GRID_OVERFLOW, NOT_READY))
        return res

    def merge_voxelize_async(self, cparams: Params):
        return self._check(self._lib.cm_merge_voxelize_async(self._ctx, C.byref(cparams)),
                           "cm_merge_voxelize_async", ok=(OK, NOT_READY))

    def wait(self) -> Result:
        res = Result()
        self._check(self._lib.cm_wait(self._ctx, C.byref(res)), "cm_wait", ok=(OK, EMPTY_INPUT, GRID_OVERFLOW))
        return res

    def result(self, n_out, point_step=16):
        """(n_out,) structured XYZI array (step 16) or (n_out, 8) float32 PointXYZI images (step 32)."""
        n_out = int(n_out)
        if point_step == 16:
            out = np.zeros(n_out, dtype=XYZI_DTYPE)
        else:
            out = np.zeros((n_out, 8), dtype=np.float32)
        self._check(self._lib.cm_result_copy(self._ctx, out.ctypes.data if n_out else None, n_out, point_step),
                    "cm_result_copy")
        return out

    def result_device(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.cm_result_device(self._ctx, C.byref(ptr), C.byref(n)), "cm_result_device")
        return ptr.value, n.value

    def cells(self, n_out):
        n_out = int(n_out)
        ijk = np.zeros((n_out, 3), dtype=np.int32)
        cnt = np.zeros(n_out, dtype=np.uint32)
        self._check(self._lib.cm_result_copy_cells(self._ctx, ijk.ctypes.data, cnt.ctypes.data, n_out),
                    "cm_result_copy_cells")
        return ijk, cnt

    def merged(self, capacity):
        out = np.zeros(int(capacity), dtype=XYZI_DTYPE)
        n = C.c_uint64()
        self._check(self._lib.cm_merged_copy(self._ctx, out.ctypes.data, int(capacity), C.byref(n)), "cm_merged_copy")
        return out[: n.value].copy()

    # ---- fused cloud across GPUs ----
    def set_ground_removal(self, gparams):
        """gparams: GroundParams (make_ground_params) or None to switch the stage off."""
        self._check(self._lib.cm_set_ground_removal(self._ctx, C.byref(gparams) if gparams is not None else None),
                    "cm_set_ground_removal")

    def ground(self, capacity):
        out = np.zeros(int(capacity), dtype=XYZI_DTYPE)
        n = C.c_uint64()
        self._check(self._lib.cm_ground_copy(self._ctx, out.ctypes.data, int(capacity), C.byref(n)), "cm_ground_copy")
        return out[:n.value]

    def ground_planes(self):
        arr = (GroundPlane * (MAX_SENSORS * MAX_ZONES))()
        self._check(self._lib.cm_ground_planes(self._ctx, arr, MAX_SENSORS * MAX_ZONES), "cm_ground_planes")
        return arr

    def local_bounds(self, params: MergeParams):
        mn, mx, n = (C.c_float * 3)(), (C.c_float * 3)(), C.c_uint64()
        self._check(self._lib.cm_local_bounds(self._ctx, C.byref(make_params(params)), mn, mx, C.byref(n)),
                    "cm_local_bounds")
        return np.array(mn, dtype=np.float32), np.array(mx, dtype=np.float32), n.value

    def merge_partial(self, params: MergeParams, global_min_max=None) -> Result:
        res = Result()
        b = None if global_min_max is None else (C.c_float * 6)(*[float(v) for v in global_min_max])
        st = self._lib.cm_merge_partial(self._ctx, C.byref(make_params(params)), b, C.byref(res))
        self._check(st, "cm_merge_partial", ok=(OK, EMPTY_INPUT, NOT_READY))
        return res

    def partial_device(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.cm_partial_device(self._ctx, C.byref(ptr), C.byref(n)), "cm_partial_device")
        return ptr.value, n.value

    def partial(self, n_entries):
        out = np.zeros(int(n_entries), dtype=ENTRY_DTYPE)
        self._check(self._lib.cm_partial_copy(self._ctx, out.ctypes.data if n_entries else None, int(n_entries)),
                    "cm_partial_copy")
        return out

    def partial_to_device(self, dst_ptr, capacity):
        self._check(self._lib.cm_partial_copy(self._ctx, C.c_void_p(dst_ptr), int(capacity)), "cm_partial_copy")

    def merge_tables(self, table_ptrs, counts, params: MergeParams) -> Result:
        n = len(table_ptrs)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in table_ptrs])
        cnts = (C.c_uint64 * n)(*[int(v) for v in counts])
        res = Result()
        st = self._lib.cm_merge_tables(self._ctx, ptrs, cnts, n, C.byref(make_params(params)), C.byref(res))
        self._check(st, "cm_merge_tables", ok=(OK, EMPTY_INPUT))
        return res

    def stage_times(self):
        t = StageTimes()
        self._check(self._lib.cm_get_stage_times(self._ctx, C.byref(t)), "cm_get_stage_times")
        return [(t.name[i].value.decode(), float(t.ms[i])) for i in range(t.n_stages)]
