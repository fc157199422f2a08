"""Plain data carriers shared by the ctypes binding, the synthetic generators, bench and tests."""
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

# Reference defaults: pcl_preprocessing/src/Parameter.h:27-35.
REF_VOXEL_SIZE = 0.1
REF_POINTS_PER_VOXEL = 2
REF_ROI_MIN = (-15.0, -5.0, -0.5)          # x: -roi_mid, y: -roi_width/2, z: roi_z_min
REF_ROI_MAX = (60.0, 5.0, 3.0)             # x: roi_length-roi_mid, y: roi_width/2, z: roi_z_max
REF_OUTLIER_RADIUS = 0.15                  # Parameter.h:23 (my_cloud_fusion/Parameter.h:15 uses 0.1)
REF_OUTLIER_MIN_NEIGHBORS = 1              # Parameter.h:24

# Compact device/wire layout used by the synthetic inputs: x,y,z,intensity float32.
XYZI_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4")])


@dataclass
class SensorCloud:
    """One sensor_msgs/PointCloud2 payload and the sensor's cached static transform
    (reference: subscriber pc_preprocessing_main.cpp:520-525, lookupTransform :556-561)."""
    data: np.ndarray                      # contiguous bytes, n * point_step
    n: int
    point_step: int = 16
    off_x: int = 0
    off_y: int = 4
    off_z: int = 8
    off_i: Optional[int] = 12             # None: no intensity field (treated as 0)
    q_xyzw: Sequence[float] = (0.0, 0.0, 0.0, 1.0)
    t_xyz: Sequence[float] = (0.0, 0.0, 0.0)
    is_dense: bool = True


@dataclass
class MergeParams:
    """Runtime form of the reference's compile-time constants (Parameter.h:27-35) and the
    VoxelGrid settings at pc_preprocessing_main.cpp:173-175."""
    leaf: Sequence[float] = (REF_VOXEL_SIZE,) * 3
    min_points_per_voxel: int = REF_POINTS_PER_VOXEL
    downsample_all_data: bool = True
    crop_min: Optional[Sequence[float]] = None
    crop_max: Optional[Sequence[float]] = None
    required_sensor_mask: int = 0          # 0: every submitted sensor is required
    # RadiusOutlierRemoval on the fused cloud before VoxelGrid (CloudFusionNode.h:74-85); None = off.
    outlier_radius: Optional[float] = None
    outlier_min_neighbors: int = 1


def xyzi_cloud(xyz, intensity=None, **kw) -> SensorCloud:
    """Pack (n,3) [+ (n,)] float32 into the 16-byte XYZI layout."""
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    a = np.zeros(len(xyz), dtype=XYZI_DTYPE)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if intensity is not None:
        a["intensity"] = np.asarray(intensity, dtype=np.float32)
    return SensorCloud(data=a, n=len(a), point_step=16, off_x=0, off_y=4, off_z=8, off_i=12, **kw)
