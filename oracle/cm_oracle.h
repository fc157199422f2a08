/*
 * cm_oracle.h — CPU oracle for the merge → voxel-grid hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch CPU restatement of the arithmetic the reference node performs per frame
 * (reference: pcl_preprocessing/src/pc_preprocessing_main.cpp:318-337 transform, :20-40 getROI,
 * :131-160 fusePointclouds, :168-177 voxelgrid).  The arithmetic itself lives in third-party,
 * un-vendored libraries (PCL 1.8.1, pcl_ros 1.7, tf 1.12, Eigen 3.3 — the versions implied by the
 * reference's ROS Melodic target, none pinned by a lock file, none present in the build container),
 * so it is restated here from their published algorithms as recorded in SURVEY.md Appendix A.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path and its
 * own code cannot be compiled here (needs PCL/ROS/Eigen/Boost).  The oracle is pinned only by the
 * hand-derived known-answer cases under tests/golden/  and by an independent
 * numpy restatement (oracle/np_oracle.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libcloudmerge_hip.so) never links or calls it.
 */
#ifndef CM_ORACLE_H
#define CM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::PointXYZI memory image: 32 bytes. */
typedef struct orc_point {
    float x, y, z, pad;      /* pad == 1.0f for every constructed point */
    float intensity;
    float _unused[3];
} orc_point;

#define ORC_NO_FIELD 0xFFFFFFFFu

/* One incoming sensor_msgs/PointCloud2 payload plus its cached static transform
 * (reference: subscriber type pc_preprocessing_main.cpp:520-525, transforms :556-561). */
typedef struct orc_sensor {
    const void* data;        /* n * point_step bytes */
    uint32_t n;
    uint32_t point_step;
    uint32_t off_x, off_y, off_z;
    uint32_t off_i;          /* ORC_NO_FIELD => intensity 0 */
    double q_xyzw[4];        /* what tf::Transform::getRotation() hands pcl_ros */
    double t_xyz[3];         /* tf::Transform::getOrigin() */
    int32_t is_dense;        /* PointCloud::is_dense of the deserialised message */
    int32_t _pad;
} orc_sensor;

typedef struct orc_params {
    float leaf[3];                   /* VoxelGrid::setLeafSize (:173) */
    uint32_t min_points_per_voxel;   /* :175 */
    int32_t downsample_all_data;     /* :174 */
    int32_t crop_enable;             /* getROI (:20-40) */
    float crop_min[3];               /* x,y,z inclusive */
    float crop_max[3];
    int32_t outlier_enable;          /* RadiusOutlierRemoval on the fused cloud (CloudFusionNode.h:74-85,
                                        cloud_fusion_node.cpp:72; live node :184-192) */
    float outlier_radius;            /* setRadiusSearch (Parameter.h:23 / my_cloud_fusion Parameter.h:15) */
    uint32_t outlier_min_neighbors;  /* setMinNeighborsInRadius (Parameter.h:24) */
} orc_params;

enum {
    ORC_OK = 0,
    ORC_EMPTY_INPUT = 1,     /* VoxelGrid on empty cloud: width = height = 0 */
    ORC_GRID_OVERFLOW = 2,   /* PCL int32 index guard: output = input unchanged */
    ORC_BAD_ARG = -1
};

typedef struct orc_report {
    int32_t status;
    int32_t threads_used;
    uint64_t n_in;           /* sum of sensor n */
    uint64_t n_merged;       /* after transform + crop + concat */
    uint64_t n_out;
    int32_t min_b[3], max_b[3], div_b[3];
    float min_p[3], max_p[3];
    double t_ingest_s;       /* wire -> PointXYZI (+ the reference's by-value callback copy) */
    double t_transform_crop_s; /* wall time of the (optionally threaded) per-sensor stage */
    double t_concat_s;
    double t_voxel_s;
    double t_total_s;
} orc_report;

/* tf quaternion/origin (double) -> Eigen::Affine3f rows, row-major 3x4. */
void orc_quat_to_matrix(const double q_xyzw[4], const double t_xyz[3], float m[12]);

/* pcl_ros serializer: field-mapped copy into 32-B points. */
void orc_ingest(const orc_sensor* s, orc_point* out);

/* pcl::transformPointCloud scalar form (A.1 step 3, A.2). in may equal out. */
void orc_transform(const orc_point* in, size_t n, const float m[12], int is_dense, orc_point* out);

/* getROI: PassThrough z, then y, then x (A.3). Returns survivors written to out (order kept). */
size_t orc_crop(const orc_point* in, size_t n, const float mn[3], const float mx[3], orc_point* out);

/* pcl::VoxelGrid<PointXYZI>::applyFilter (A.4). out must hold n points (overflow passes input
 * through). stable_ties != 0 replaces std::sort by std::stable_sort (ties in input order) — not
 * what PCL does, offered so sums can be compared in a defined order. */
int orc_voxelgrid(const orc_point* in, size_t n, const float leaf[3], uint32_t min_pts,
                  int downsample_all, int is_dense, int stable_ties,
                  orc_point* out, size_t* n_out, orc_report* rep,
                  int32_t* out_cells /* optional, 3 per voxel: absolute i,j,k */,
                  uint32_t* out_counts /* optional, points per kept voxel */);

/* pcl::RadiusOutlierRemoval, keep_organized = false (PCL 1.8.1
 * filters/impl/radius_outlier_removal.hpp, recalled): a point stays iff k > min_neighbors, where k
 * counts the points of the cloud (itself included) with fp32 squared distance
 * ((dx*dx + dy*dy) + dz*dz) < float(double(r)*double(r)) — FLANN's L2_Simple and its strict
 * RadiusResultSet test (assumption: not verifiable here). Non-finite points never stay.
 * Survivors keep their order. Returns the number written to out. */
size_t orc_radius_outlier_removal(const orc_point* in, size_t n, float radius, uint32_t min_neighbors,
                                  orc_point* out, uint8_t* keep_mask /* optional, n */);

/* Ground plane of one zone, as removeGround finds it (pc_preprocessing_main.cpp:93-104): pcl::SACSegmentation
 * with SACMODEL_PLANE, SAC_RANSAC, setMaxIterations, setDistanceThreshold, setProbability,
 * setOptimizeCoefficients(true); setAxis / setEpsAngle are set there but SACMODEL_PLANE ignores them.
 * PCL 1.8.1's RandomSampleConsensus::computeModel and SampleConsensusModelPlane, recalled,
 * restated with these choices where PCL's behaviour depends on things that cannot be reproduced here:
 *   - samples: hypothesis j of a zone takes three distinct points chosen by splitmix64(seed, zone_key, j)
 *     (PCL draws them from boost::mt19937 seeded 12345 through a running shuffle: same distribution,
 *     another sequence) — results agree with PCL's statistically, not draw for draw;
 *   - PCL's loop is kept: a hypothesis replaces the best one only with strictly more inliers, a collinear
 *     sample is skipped without counting as an iteration, and the loop ends once
 *     (1 - w^3)^iterations <= 1 - probability for the best inlier ratio w (PCL: iterations >= log(1-p)/log(1-w^3)),
 *     or after max_iterations; the power is built by repeated multiplication so that it is the same
 *     number on every machine;
 *   - plane of a sample and point-to-plane distance in fp32, PCL's operation order; inlier: |d| < threshold;
 *   - optimizeModelCoefficients (more than 3 inliers): mean and covariance of the inliers, plane normal =
 *     eigenvector of the smallest eigenvalue, d = -n . mean; sums in fp64 in a fixed blocked order (chunks of 8192
 *     points; in a chunk element i goes to partial i mod 256, the partials are added pairwise: 128, 64, ... 1; the
 *     chunk sums are added one after the other) and a fixed-sweep Jacobi
 *     iteration instead of PCL's fp32 running sums and closed-form eigen33, then the inliers are selected again.
 * found = 0 (fewer than 3 points or no valid sample): no plane, no inliers, like PCL's "could not estimate". */
typedef struct orc_plane_result {
    float plane[4];          /* a, b, c, d with a*x + b*y + c*z + d = 0, |(a,b,c)| = 1 */
    uint32_t n_inliers;
    uint32_t best_hypothesis;
    uint32_t iterations;     /* hypotheses PCL's loop looked at */
    int32_t found;
} orc_plane_result;
void orc_ransac_plane(const orc_point* pts, size_t n, uint32_t max_iterations, float threshold, float probability,
                      int optimize, uint64_t seed, uint32_t zone_key, orc_plane_result* res,
                      uint8_t* inlier_mask /* n */);

/* Absolute voxel cell of each point: floor(fl32(p * inv_leaf)) per axis (A.4 step 5). */
void orc_voxel_cells(const orc_point* in, size_t n, const float leaf[3], int32_t* ijk /* 3n */);

/* Whole path. merged_out (optional, capacity n_in) receives the transformed+cropped+concatenated
 * cloud; out (capacity n_in) the voxelised cloud. threads: 1 = fully serial; k>1 = one thread per
 * sensor up to k for ingest+transform+crop (mirrors AsyncSpinner(6), :513); concat and VoxelGrid
 * are always single-threaded like the reference's main loop (:574-577). */
int orc_merge_voxelize(const orc_sensor* sensors, int n_sensors, const orc_params* p,
                       int threads, int stable_ties,
                       orc_point* merged_out, orc_point* out, orc_report* rep,
                       int32_t* out_cells, uint32_t* out_counts);

#ifdef __cplusplus
}
#endif
#endif
