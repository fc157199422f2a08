/*
 * cloudmerge.h — C-ABI of the MI355X merge → voxel-grid library (libcloudmerge_hip.so).
 *
 * Drop-in boundary for ONE path of timspilak/cloud_merger: per-frame rigid transform of every
 * sensor cloud into the common frame, optional AABB crop, concatenation across sensors and the
 * PCL-VoxelGrid downsample.  The reference has no FFI layer for this path; its seam is the set of
 * free functions and third-party calls cited per entry point below (all file:line references are
 * into /root/reference/pcl_preprocessing/src/).  INTEGRATION.md shows the binding a maintainer of
 * the reference node adds around these calls.
 *
 * Conventions: plain pointers and sizes only; every function returns a cm_status (never throws);
 * the caller owns all host buffers; results live in the context until the next merge.
 * Threading: cm_submit_cloud* may be called concurrently for DIFFERENT sensor slots (the
 * reference's six subscriber threads, pc_preprocessing_main.cpp:513-525); cm_merge_voxelize* /
 * cm_wait / cm_result_* from one consumer thread (the reference's 10 Hz main loop, :549-584).
 * There is no CPU fallback: cm_create fails with CM_NO_DEVICE / CM_HIP_ERROR without a gfx950 GPU.
 */
#ifndef CLOUDMERGE_H
#define CLOUDMERGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define CM_API __attribute__((visibility("default")))
#else
#define CM_API
#endif

#define CM_VERSION 100            /* 0.1.0 */
#define CM_MAX_SENSORS 16
#define CM_NO_FIELD 0xFFFFFFFFu   /* off_i: the cloud has no intensity field (treated as 0) */

typedef struct cm_ctx cm_ctx;

typedef enum cm_status {
    CM_OK = 0,
    CM_EMPTY_INPUT = 1,    /* no point survived: PCL VoxelGrid returns width = height = 0 */
    CM_GRID_OVERFLOW = 2,  /* PCL's int32 index guard tripped: output = merged input, unvoxelised */
    CM_NOT_READY = 3,      /* a required sensor has no fresh cloud: the reference skips the tick (:134,:575) */
    CM_SKIPPED = 4,        /* cm_submit_cloud*: this sensor already holds an unconsumed cloud and the policy is
                              "first since the last fuse wins" (:330): the new one was dropped */
    CM_BAD_ARG = -1,
    CM_HIP_ERROR = -2,
    CM_NO_DEVICE = -3,
    CM_CAPACITY = -4,      /* more points than cm_limits allows, or the outlier stage's radius grid does not fit */
    CM_INTERNAL = -5
} cm_status;

/* cm_limits.flags */
#define CM_FLAG_PROFILE        0x1u  /* record a HIP event pair around every kernel (cm_get_stage_times) */
#define CM_FLAG_LATEST_WINS    0x2u  /* a newer cloud replaces an unconsumed one; default is the
                                        reference's "first cloud since the last fuse wins" (:330,:356,...) */
#define CM_FLAG_OCCUPANCY      0x4u  /* also keep (voxel index, point count) per output voxel for
                                        cm_result_copy_cells (+8 B of HBM writes per voxel) */

typedef struct cm_limits {
    uint32_t max_sensors;          /* 1..CM_MAX_SENSORS */
    uint32_t flags;
    uint64_t max_points_total;     /* per frame, summed over sensors (< 2^30) */
} cm_limits;

/* Runtime form of the reference's compile-time constants (Parameter.h:27-35) and of the
 * VoxelGrid settings at pc_preprocessing_main.cpp:173-175. */
typedef struct cm_params {
    float leaf[3];                 /* setLeafSize(v,v,v) :173; Parameter.h:28 */
    uint32_t min_points_per_voxel; /* setMinimumPointsNumberPerVoxel :175; Parameter.h:27 */
    int32_t downsample_all_data;   /* setDownsampleAllData(true) :174 */
    int32_t crop_enable;           /* getROI :20-40 */
    float crop_min[3];             /* x,y,z closed interval; Parameter.h:31-35 */
    float crop_max[3];
    uint32_t required_sensor_mask; /* bit s: sensor s must be fresh (:134); 0 = all submitted */
    /* pcl::RadiusOutlierRemoval on the fused cloud before VoxelGrid (my_cloud_fusion/src/
     * CloudFusionNode.h:74-85, called at cloud_fusion_node.cpp:72; live node outlierRemoval :184-192;
     * SURVEY.md §8f rank 2). A point stays iff more than outlier_min_neighbors points (itself
     * included) lie within outlier_radius (fp32 squared distance < float(r*r)). */
    int32_t outlier_enable;
    float outlier_radius;          /* setRadiusSearch; Parameter.h:23 (0.15), my_cloud_fusion Parameter.h:15 (0.1) */
    uint32_t outlier_min_neighbors;/* setMinNeighborsInRadius; Parameter.h:24 (1) */
} cm_params;

typedef struct cm_result {
    int32_t status;                /* cm_status of the frame */
    uint32_t n_sensors;            /* sensors that contributed */
    uint64_t n_in;                 /* points submitted */
    uint64_t n_merged;             /* after transform + crop (+ non-finite drop) */
    uint64_t n_out;                /* voxels written (or n_merged on CM_GRID_OVERFLOW) */
    int32_t min_b[3], max_b[3], div_b[3];   /* PCL's min_b_/max_b_/div_b_ (crop-box grid when
                                               bounds_from_crop) */
    float min_p[3], max_p[3];      /* getMinMax3D of the merged cloud (unset when bounds_from_crop) */
    uint32_t bounds_from_crop;     /* 1: grid origin taken from the crop box (same occupancy and
                                      order; the data min/max pass was skipped) */
    uint32_t key_bits;             /* bits of the linear voxel index */
    uint32_t sort_passes;          /* 8-bit radix passes over the whole frame that were run */
    uint32_t path_flags;           /* CM_PATH_* bits: how the frame was computed (same results either way) */
    float device_ms;               /* first kernel start -> last kernel end (CM_FLAG_PROFILE) */
} cm_result;

#define CM_PATH_LDS_RANK 1u    /* radix ranking by lane-ordered LDS adds (device probe at cm_create passed);
                                  otherwise ballot matching */
#define CM_PATH_BUCKET 2u      /* bucket path: point records sorted by the high index bits in sort_passes
                                  passes, the rest finished per bucket inside LDS (needs a box before the
                                  first point is read: the crop box or a predicted one) */
#define CM_PATH_PREDICTED 4u   /* ... the box was the previous frame's bounds plus a margin; every point was
                                  checked against it, min_b/max_b/div_b/min_p/max_p are the cloud's own */
#define CM_PATH_PACKED 16u    /* bucket path, crop box that dropped most of the last frame's points: the survivors' records were
                                  packed while counting, so the raw clouds were read once instead of twice */
#define CM_PATH_SPLIT 32u      /* bucket path, finish by k3_local + k3_compact (tiles stage their centroids, a second launch
                                  packs them: no look-back between tiles); otherwise k2_local. Summation order: a voxel of up
                                  to 17 points is added one point after the other in (sensor, point) order — pcl::VoxelGrid's
                                  own sum, bit for bit given that tie order; a longer one may be finished 64 points per step in
                                  a fixed tree order (deterministic, within 1e-4 m of the one-after-the-other sum, closer to the
                                  exact mean). CM_FINISH=v2 (k2_local) adds every voxel one after the other. */
#define CM_PATH_QUANTILE 64u   /* bucket path, ONE global pass (sort_passes == 1) into buckets cut at the quantiles of the previous
                                  frame's sorted records (same grid), one finish workgroup per bucket; bucket sizes are verified
                                  on the device, a frame whose buckets outgrew the finish is redone with the fixed-grid passes
                                  (CM_PATH_REDONE). The points of a voxel are added in the same (sensor, point) order: same results */
#define CM_PATH_REDONE 8u      /* the bucket path gave the frame back and it was computed a second time inside cm_wait: after
                                  a point outside the predicted box on the bucket path again, in a box around the bounds the
                                  first attempt measured (CM_PATH_BUCKET | CM_PATH_PREDICTED stay set); after a bucket too large
                                  for LDS, or more survivors of the crop than the last frame promised, on the general path */

#define CM_MAX_STAGES 48
typedef struct cm_stage_times {
    uint32_t n_stages;
    uint32_t _pad;
    char name[CM_MAX_STAGES][24];  /* kernel name */
    float ms[CM_MAX_STAGES];       /* duration of that launch in the last profiled frame */
} cm_stage_times;

/* ---- lifetime ---------------------------------------------------------------------------- */
/* Allocates the context, its stream and HBM work buffers on `device`. */
CM_API int cm_create(cm_ctx** out, int device, const cm_limits* limits);
CM_API int cm_destroy(cm_ctx* ctx);
/* Run on a caller-owned hipStream_t (NULL: back to the context's own stream). */
CM_API int cm_set_stream(cm_ctx* ctx, void* hip_stream);

/* ---- static transforms: replaces tf::Transform(stf.getRotation(), stf.getOrigin()) (:320,:346,
 * :371,:397,:424,:463) and the tf->Eigen conversion inside pcl_ros::transformPointCloud --------- */
/* q/t are what tf::Transform::getRotation()/getOrigin() return (doubles). Converted on the host
 * exactly as Eigen::Quaternionf::toRotationMatrix does in fp32. */
CM_API int cm_set_sensor_transform(cm_ctx* ctx, uint32_t sensor, const double q_xyzw[4], const double t_xyz[3]);
/* Row-major 3x4 fp32 [R|t], used verbatim. */
CM_API int cm_set_sensor_matrix(cm_ctx* ctx, uint32_t sensor, const float m[12]);
CM_API int cm_get_sensor_matrix(cm_ctx* ctx, uint32_t sensor, float m[12]);

/* ---- ingest: replaces the subscriber callbacks' deserialise + transformPointCloud + getROI
 * (:318-337 and siblings); the arithmetic itself runs inside cm_merge_voxelize ---------------- */
/* Copies a sensor_msgs/PointCloud2 payload (n * point_step bytes, FLOAT32 fields at the given
 * byte offsets) into the sensor's HBM slot. Returns after the caller's buffer may be reused.
 * Never waits for a merge: every slot has two HBM buffers, the frame enqueued last (and its by-products:
 * cm_merged_copy, cm_ground_copy) reads one, submits fill the other — the reference's callbacks run
 * beside its 10 Hz loop on AsyncSpinner(6) (:513, :318-337, :549-584). Callable from one thread per sensor. */
CM_API int cm_submit_cloud(cm_ctx* ctx, uint32_t sensor, const void* host_data, uint32_t n,
                    uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_i);
/* The same without waiting for the copy either: the H2D transfer is enqueued on the slot's own stream and
 * the frame that consumes the cloud waits for it ON THE DEVICE. `host_data` must stay valid and unchanged
 * until that frame has been enqueued and cm_wait (or cm_sync) has returned; memory from cm_host_alloc makes
 * the transfer a true DMA that overlaps the previous frame's kernels. */
CM_API int cm_submit_cloud_async(cm_ctx* ctx, uint32_t sensor, const void* host_data, uint32_t n,
                    uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_i);
/* Zero-copy variant: `dev_data` is already resident in HBM and stays valid until the merge that
 * consumes it has completed. */
CM_API int cm_submit_cloud_device(cm_ctx* ctx, uint32_t sensor, const void* dev_data, uint32_t n,
                           uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_i);
/* Forget a sensor's cloud (fresh or stale). */
CM_API int cm_clear_sensor(cm_ctx* ctx, uint32_t sensor);

/* ---- the path: replaces fusePointclouds (:131-160) + voxelgrid (:168-177) -------------------- */
/* Synchronous: enqueue, wait, fill `res`. Returns res->status. */
CM_API int cm_merge_voxelize(cm_ctx* ctx, const cm_params* p, cm_result* res);
/* Enqueue only (no host wait); pair with cm_wait. Returns CM_OK / CM_NOT_READY / error. */
CM_API int cm_merge_voxelize_async(cm_ctx* ctx, const cm_params* p);
CM_API int cm_wait(cm_ctx* ctx, cm_result* res);

/* ---- results: replaces pcl::toROSMsg of the voxel cloud (:215-216) --------------------------- */
/* Copies the n_out output points to host memory: point_step_out 16 (x,y,z,intensity) or 32 (the
 * pcl::PointXYZI image pcl::toROSMsg puts on the wire: x,y,z,1.0f,intensity,0,0,0). */
CM_API int cm_result_copy(cm_ctx* ctx, void* host_dst, uint64_t capacity_points, uint32_t point_step_out);
/* The same (16-byte records only) without waiting: the copy is enqueued behind the frame on the context's
 * stream; `host_dst` (cm_host_alloc memory for a true DMA) is complete when cm_sync returns. The next frame
 * may be enqueued right away (stream order keeps it off the result until the copy has read it). */
CM_API int cm_result_copy_async(cm_ctx* ctx, void* host_dst, uint64_t capacity_points);
/* Pipelined publish (the loop body of the reference publishes every tick, pc_preprocessing_main.cpp:199-220, :574-577): the
 * copy-out of the LAST WAITED-FOR frame — 16-byte records or the 32-byte pcl::PointXYZI images — is enqueued on a stream of
 * its own, and the next frame may be enqueued right away: its kernels run BESIDE the copy (the result buffers exist twice;
 * a frame only waits, on the device, for the copy that read the buffers it is about to write, i.e. the one of two frames
 * ago). `host_dst` — cm_host_alloc memory, or any buffer made DMA-able with cm_host_register — is complete when
 * cm_publish_wait (or cm_sync) returns. Call between cm_wait and the next cm_merge_voxelize_async. */
CM_API int cm_result_publish_async(cm_ctx* ctx, void* host_dst, uint64_t capacity_points, uint32_t point_step_out);
CM_API int cm_publish_wait(cm_ctx* ctx);
/* Waits for everything enqueued on the context's streams. */
CM_API int cm_sync(cm_ctx* ctx);
/* Device pointer of the compact 16-byte result records (valid until the next merge). */
CM_API int cm_result_device(cm_ctx* ctx, const void** dev_ptr, uint64_t* n_points);
/* Occupancy of the last CM_OK frame (needs CM_FLAG_OCCUPANCY): absolute voxel cell (i,j,k) and
 * point count of every output voxel, in output order. Either pointer may be NULL. */
CM_API int cm_result_copy_cells(cm_ctx* ctx, int32_t* ijk_host, uint32_t* counts_host, uint64_t capacity_voxels);
/* The merged (transformed + cropped + concatenated) cloud of the last frame as 16-byte
 * x,y,z,intensity records in sensor order — the reference's fused cloud (:137-142). */
CM_API int cm_merged_copy(cm_ctx* ctx, void* host_dst, uint64_t capacity_points, uint64_t* n_points);

/* ---- diagnostics --------------------------------------------------------------------------- */
/* Per-frame figures of the last frame that was waited for — what the reference logs per callback
 * (ROS_INFO of the cloud sizes, :334,:360,:386,:413,:452,:505) plus the bytes that moved. */
typedef struct cm_frame_stats {
    uint32_t n_sensors;                    /* sensors fused into the frame, in fuse order */
    uint32_t _pad;
    uint32_t sensor[CM_MAX_SENSORS];       /* the caller's sensor number */
    uint32_t n_in[CM_MAX_SENSORS];         /* points submitted */
    uint32_t n_kept[CM_MAX_SENSORS];       /* ... that were finite, inside the crop box and passed the pre-stages' masks,
                                              i.e. entered the voxel grid */
    uint32_t fresh[CM_MAX_SENSORS];        /* 1: a cloud submitted since the previous frame, 0: a stale one rode along (:141) */
    uint64_t generation[CM_MAX_SENSORS];   /* which accepted submit of that sensor the frame read (1 = its first; CM_SKIPPED ones
                                              do not count): a caller that counts its accepted submits knows exactly what was consumed */
    uint64_t bytes_h2d[CM_MAX_SENSORS];    /* payload bytes copied host -> HBM for this frame (0: device submit or stale) */
    uint64_t bytes_h2d_total, bytes_d2h_total;   /* d2h: result / merged / ground copies since the frame was enqueued */
    uint64_t bytes_algorithmic;            /* 16 B x points in + 16 B x voxels out */
} cm_frame_stats;
CM_API int cm_get_frame_stats(cm_ctx* ctx, cm_frame_stats* out);
CM_API int cm_get_stage_times(cm_ctx* ctx, cm_stage_times* out);
CM_API const char* cm_status_string(int status);
CM_API const char* cm_last_error(cm_ctx* ctx);
CM_API int cm_version(void);

/* ---- multi-GPU single fused cloud (nothing like it exists in the reference) ---
 * One process per GPU. Each rank voxelises ITS sensors into a partial table of per-voxel sums
 * (thresholding deferred: a voxel may hold one point on each of two GPUs), the host all-gathers the
 * tables (RCCL over xGMI), and the merge re-sorts the concatenated entries by voxel index, adds
 * them in rank order (deterministic), applies min_points_per_voxel and divides.
 * All ranks must index the SAME grid: the crop box fixes it when it fits PCL's int32 index;
 * otherwise pass the bounds of the whole fused cloud (cm_local_bounds on every rank, min/max
 * all-reduced by the host) as global_min_max = {min x,y,z, max x,y,z}. */
typedef struct cm_partial_entry {   /* 32 bytes */
    uint32_t key;                   /* linear voxel index in the shared grid (PCL order) */
    uint32_t count;
    float sx, sy, sz, si;           /* fp32 sums, stable point order */
    uint32_t _pad[2];
} cm_partial_entry;
/* fp32 min/max of this rank's transformed (+cropped) points; does not consume the clouds. */
CM_API int cm_local_bounds(cm_ctx* ctx, const cm_params* p, float min_xyz[3], float max_xyz[3], uint64_t* n_valid);
/* Like cm_merge_voxelize but stops before thresholding/division. res->n_out = table entries. */
CM_API int cm_merge_partial(cm_ctx* ctx, const cm_params* p, const float* global_min_max, cm_result* res);
CM_API int cm_partial_device(cm_ctx* ctx, const void** dev_entries, uint64_t* n_entries);
/* dst may be host or device memory (e.g. the send buffer of the all-gather). */
CM_API int cm_partial_copy(cm_ctx* ctx, cm_partial_entry* dst, uint64_t capacity);
/* Merge n_tables tables (16-byte aligned device pointers, rank order) into the context's result
 * buffer (cm_result_copy / cm_result_copy_cells read it; cells need the grid of the ranks'
 * cm_merge_partial results). Synchronous. res->n_merged = distinct voxels, res->n_out = kept. */
CM_API int cm_merge_tables(cm_ctx* ctx, const void* const* dev_tables, const uint64_t* n_entries,
                           uint32_t n_tables, const cm_params* p, cm_result* res);

/* ---- zone-wise ground removal before the fuse -------------------------------
 * What the live node does to every sensor's cloud between getROI and the fuse: proceedFront / proceedRear
 * (pc_preprocessing_main.cpp:228-312), the top-middle and Livox callbacks (:436-497) — the cropped cloud is
 * cut into x-slabs (getCloudPart :49-59), and in each slab removeGround (:71-122) takes the points of a z band,
 * fits one plane to them (RANSAC, Parameter.h:38-42) and calls its inliers ground; the rest of the band and the
 * part above it (up to z_keep_max) are "no ground". With this enabled the voxel grid (and cm_merged_copy) sees
 * the fused no-ground cloud (:137-142) and cm_ground_copy returns the fused ground cloud (:144-149).
 * Differences from the reference, all documented in DESIGN.md §10: RANSAC samples come from a counter-based
 * generator, not boost::mt19937 (planes agree statistically, not draw for draw); a point on the border of two
 * slabs goes to the first one only (the reference's closed intervals put it in both); points keep sensor order
 * (the reference concatenates slab by slab). The radius outlier filter that removeGround applies to the band's
 * non-ground points (:119) runs when outlier_radius > 0, among the points of the same slab, like there. */
#define CM_MAX_ZONES 8
typedef struct cm_zone {
    float x_min, x_length;         /* getCloudPart(cloud, part, length, deviation): x in [x_min, x_min + x_length] */
    float z_max_ground;            /* removeGround(.., -z, z, ..): band z in [-z, z]; negative: no ground removal,
                                      the slab is kept whole (top-middle's rear part, :440-441) */
} cm_zone;
typedef struct cm_ground_params {
    uint32_t max_iterations;       /* Parameter.h:38 (1000) */
    float distance_threshold;      /* :40 (0.3 m) */
    float probability;             /* :41 (0.99) */
    int32_t optimize_coefficients; /* :95 (true) */
    float z_keep_max;              /* roi_z_max (:35): the part above a band reaches from z_max_ground + 0.01 up to here (:91) */
    float outlier_radius;          /* > 0: RadiusOutlierRemoval on every band's non-ground points (:119, Parameter.h:23) */
    uint32_t outlier_min_neighbors;/* Parameter.h:24 */
    uint32_t _pad;
    uint64_t seed;                 /* of the sample generator */
    uint32_t n_zones[CM_MAX_SENSORS];
    cm_zone zones[CM_MAX_SENSORS][CM_MAX_ZONES];   /* in the order the reference processes them */
} cm_ground_params;
typedef struct cm_ground_plane {
    float plane[4];                /* a x + b y + c z + d = 0 */
    uint32_t band_points, inliers, iterations;
    int32_t found;
} cm_ground_plane;
/* NULL switches the stage off. Takes effect with the next cm_merge_voxelize; not combined with
 * cm_params.outlier_enable or cm_merge_partial. */
CM_API int cm_set_ground_removal(cm_ctx* ctx, const cm_ground_params* g);
/* Fused ground cloud of the last frame, 16-byte x,y,z,intensity records in (sensor, point) order. */
CM_API int cm_ground_copy(cm_ctx* ctx, void* host_dst, uint64_t capacity_points, uint64_t* n_points);
/* Planes of the last frame, indexed [sensor * CM_MAX_ZONES + zone]; capacity in entries. */
CM_API int cm_ground_planes(cm_ctx* ctx, cm_ground_plane* planes, uint32_t capacity);

/* ---- host memory helpers (pinned staging for PointCloud2 payloads) --------------------------- */
CM_API int cm_host_alloc(void** ptr, size_t bytes);
CM_API int cm_host_free(void* ptr);
/* Makes memory the caller already owns (a message's payload vector) DMA-able for the asynchronous copies, and undoes it. */
CM_API int cm_host_register(void* ptr, size_t bytes);
CM_API int cm_host_unregister(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDMERGE_H */
