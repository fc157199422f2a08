// cm_device.h — structures shared by the host-side C-ABI (cm_api.cpp) and the gfx950 kernels.
#pragma once
#include <stdint.h>

#define CM_DEV_MAX_SENSORS 16

// Tiling of the padded point index space: every sensor starts at a multiple of CM_TILE, so a tile
// never straddles two sensors (wave-uniform transform) and the key kernel can emit the first radix
// histogram per tile.
#define CM_TILE 4096          // points per workgroup in the streaming and radix kernels
#define CM_BLOCK 256          // threads per workgroup (4 wave64)
#define CM_ITEMS 16           // CM_TILE / CM_BLOCK
#define CM_WAVES 4
#define CM_RADIX_BITS 8
#define CM_RADIX 256
#define CM_MAX_PASSES 4
#define CM_INVALID_KEY 0xFFFFFFFFu

#define CM_GROUP 32           // tiles per group total in the radix passes
#define CM_DIRECT_GROUPS 64   // up to this many groups (8 M slots) k_scatter sums the group totals itself

#define CM_SEG_TILE 2048      // sorted items per workgroup in the centroid kernel
#define CM_SEG_ITEMS 8
#define CM_MINMAX_BLOCKS 1024 // workgroups (= partial records) of the min/max pass
#define CM_SEG_GROUP 256      // sorted tiles per kept-voxel group total (== CM_BLOCK)
#define CM_SEG_DIRECT_TILES 4096  // up to this many sorted tiles the per-tile counts are summed directly

// Bucket path (cm_kernels_v2.hip): G global 8-bit passes over 16-byte point records on the HIGH key
// bits, then one workgroup-local finish (LDS sort of the low bits + centroids) per bucket-aligned tile.
#define CM2_BLOCK 512         // threads per workgroup in the record passes (8 wave64, 8 points each)
#define CM2_WAVES 8
#define CM2_ITEMS 8
// (the local finish's geometry — 2048-record tiles, room for 4032 (k3_local) / 4096 (k2_local), 512 threads — is fixed where it is launched)
#define CM2_MAX_LOW_BITS 14   // key bits left to the local finish when the global passes allow it

// Quantile passes (cm_kernels_v4.hip): ONE global pass into up to CM4_BINS buckets cut at the quantiles of the previous
// frame's sorted records (about CM4_TARGET records each), one finish workgroup per bucket (room for CM4_CAP records).
#define CM4_BINS 2048         // bins of the wide pass
#define CM4_MAX_BUCKETS 8192  // buckets of a frame: above CM4_BINS two or four neighbouring buckets share a bin of the pass
                              // (cm_quant_sub_shift below; frames of up to 15 M records)
#define CM4_TARGET 1920u
#define CM4_CAP 4032u          // records a finish workgroup of the usual shape holds (k3_local: 512 threads, four per CU)
#define CM4_CAP_BIG 8064u      // ... and of the large shape (1024 threads, one per CU) that takes the few buckets beyond that
#define CM4_MAX_BIG 64u        // at most this many such buckets per frame (more: the frame is handed back); the large launch has this many workgroups
#define CM4_MAX_AVG 2600u     // the host takes the path only while records / buckets stays below this
#define CM4_MAX_TILES 4096u   // ... and the frame has at most this many 4096-slot tiles (k4_colscan's register tile)
// Buckets the NEXT frame uses when this one sorted n records (the finish writes that many splitters; the host sizes the grids).
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t cm_quant_buckets(uint32_t n) {
    if (n == 0) return 0;
    uint32_t b = (n + CM4_TARGET - 1u) / CM4_TARGET;
    // one pass for as long as CM4_BINS buckets of at most CM4_MAX_AVG records hold the frame (cfg2: 2048 x 1953)
    if (b > CM4_BINS && (n + CM4_BINS - 1u) / CM4_BINS <= CM4_MAX_AVG) b = CM4_BINS;
    return b > CM4_MAX_BUCKETS ? CM4_MAX_BUCKETS : b;
}
// More than CM4_BINS buckets, still one global pass ("shared bins"): the pass scatters by bucket >> shift — 2^shift
// neighbouring buckets share a bin — and the finish workgroup of a bucket picks its records out of its bin by their
// index (k3_local<SUB>: the 2^shift workgroups of a bin run on the same XCD and read it through that XCD's L2).
static inline uint32_t cm_quant_sub_shift(uint32_t n_buckets) {
    return n_buckets <= CM4_BINS ? 0u : n_buckets <= 2u * CM4_BINS ? 1u : 2u;
}

// Point layouts the loaders special-case.
#define CM_LAYOUT_XYZI16 0    // x,y,z,intensity @0,4,8,12, step 16, 16-B aligned: one dwordx4 load
#define CM_LAYOUT_PCL32 1     // pcl::PointXYZI image, step 32, intensity @16: dwordx4 + dword
#define CM_LAYOUT_GENERIC 2   // any step/offsets, possibly unaligned: four dword loads

struct CmSensorDev {
    const unsigned char* data;
    uint32_t n;               // points in the cloud
    uint32_t base;            // first padded global index (multiple of CM_TILE)
    uint32_t point_step;
    uint32_t off_x, off_y, off_z, off_i;   // off_i == 0xFFFFFFFF: no intensity
    uint32_t layout;
    float m[12];              // row-major 3x4 [R|t]
    uint32_t slot;            // the caller's sensor number (slots without a cloud are skipped in s[]): what per-sensor
                              // settings — the ground stage's slab tables — are indexed by
    uint32_t _pad;
};

// One entry per 4096-slot tile of the padded index space, written by k_setup with the frame descriptor: where the
// tile's first point lies and how many points of its cloud are left from there, so that the streaming kernels reach
// their points through ONE dependent load instead of descriptor -> sensor -> payload.
struct CmTileDev {
    const unsigned char* data;   // address of the tile's first point
    uint32_t n_left;             // points of the cloud from there to its end (>= 1)
    uint32_t info;               // index into CmFrameDev::s | layout << 8
};

struct CmFrameDev {
    CmSensorDev s[CM_DEV_MAX_SENSORS];
    uint32_t n_sensors;
    uint32_t n_padded;        // size of the padded index space (multiple of CM_TILE)
    uint32_t n_tiles;         // n_padded / CM_TILE
    uint32_t crop_enable;
    float crop_min[3];
    float crop_max[3];
    float inv_leaf[3];        // 1.0f / leaf, computed once on the host in fp32
    uint32_t min_pts;
    uint32_t downsample_all;
    float inv_cell[3];        // outlier stage: 1 / (1.01 * radius) — candidate grid a little wider than the radius
    float outlier_r2;         // float(double(r) * double(r)): FLANN compares squared distances
    uint32_t outlier_min_nb;
    float ext_min[3];         // grid bounds handed in by the host (fused cloud across GPUs: the
    float ext_max[3];         // min/max of the WHOLE merged cloud, all-reduced over the ranks)
    // Bucket path: the grid of the box the frame is sorted in (crop box or predicted box), set up on the
    // host with the arithmetic of compute_grid (cm_common.hpp): cell of the box minimum, cells per axis.
    int32_t box_min_b[3];
    int32_t box_div_b[3];
    uint32_t box_key_bits;
    uint32_t box_predicted;   // 1: the box is a prediction (check every point against it)
    // ... and of the radius grid of the outlier stage over the crop box (cells 1.01 r wide), same set-up
    int32_t cell_min_b[3];
    int32_t cell_div_b[3];
    uint32_t cell_key_bits;
    uint32_t _pad_cell;
};

// Zone-wise ground removal (cm_kernels_ground.hip): per sensor up to 8 x-slabs, each with a z band.
#define CM_DEV_MAX_ZONES 8
#define CM_GROUND_BATCH 32        // RANSAC hypotheses scored per round (PCL's loop usually stops within the first)
#define CM_GROUND_CHUNK 8192      // band points per workgroup of the least-squares sums (fixed: it defines their order)
#define CM_GROUND_SPARE 24        // hypotheses beyond max_iterations, for collinear samples that are skipped
struct CmGroundDev {
    uint32_t n_zones[CM_DEV_MAX_SENSORS];
    float x0[CM_DEV_MAX_SENSORS][CM_DEV_MAX_ZONES], x1[CM_DEV_MAX_SENSORS][CM_DEV_MAX_ZONES];
    float zmax[CM_DEV_MAX_SENSORS][CM_DEV_MAX_ZONES];      // < 0: keep the slab whole
    float zlo[CM_DEV_MAX_SENSORS][CM_DEV_MAX_ZONES];       // float(double(zmax) + 0.01): where the part above the band starts
    float z_keep_max;
    float threshold;
    float probability;
    uint32_t max_iterations;
    uint32_t optimize;
    uint32_t _pad;
    unsigned long long seed;
};
struct CmGroundPlaneDev {                 // == cm_ground_plane
    float plane[4];
    uint32_t band_points, inliers, iterations;
    int32_t found;
};

// Per-frame device state, zeroed before the first kernel of a frame.
struct CmFrameState {
    uint32_t outside;         // bucket path: a point fell outside the predicted box (frame must be redone)
    uint32_t quant_abort;     // quantile passes (cm_kernels_v4.hip): a bucket beyond the finish's capacity — every later kernel leaves
    uint32_t spl_incomplete;  // the finish could not leave every quantile (a voxel reached beyond what a tile holds in LDS): no splitters
    uint32_t quant_big;       // quantile passes: buckets this frame handed to the large finish shape (k4_colscan's list)
    uint32_t _unused[2];
    uint32_t n_valid_k0;      // valid points counted by the min/max pass
    int32_t status;           // cm_status of the frame (0 OK, 1 EMPTY, 2 OVERFLOW)
    int32_t min_b[3], max_b[3], div_b[3];
    float min_p[3], max_p[3];
    uint32_t key_bits;
    uint32_t n_passes;
    uint32_t n_valid;         // points that entered the sort (after pass 0)
    uint32_t n_out;
    uint32_t n_seg_tiles;
    uint32_t err;
};

// Device status codes mirror cm_status.
#define CM_DEV_OK 0
#define CM_DEV_EMPTY 1
#define CM_DEV_OVERFLOW 2
#define CM_DEV_ERR_UNSORTED 2u   // CmFrameState.err: the radix sort's output was not sorted
#define CM_DEV_ERR_LOOKBACK 3u   // ... a workgroup waited too long for its predecessors' counts
#define CM_DEV_ERR_BUCKET 4u     // ... a bucket did not fit the local finish's LDS capacity
#define CM_DEV_ERR_BUCKET_PRE 5u // ... the same in the outlier stage's sort (a radius cell with thousands of points)
#define CM_DEV_ERR_GRID 6u       // ... the kernels behind pass 0 were launched with fewer workgroups than the kept records need
#define CM_DEV_ERR_QUANT 7u      // ... a quantile bucket (cm_kernels_v4.hip) grew beyond the finish's capacity: the splitters are stale
#define CM_DEV_OUTLIER_GRID 3   // the radius grid of the outlier stage does not fit (rows or 32-bit index)
#define CM_DEV_ABORTED 4        // a stage gave up (CmFrameState.err says why): every later kernel of the frame leaves at once —
                                // what the stage left behind (half-sorted keys, unwritten records) must not be indexed with

#define CM_ROW_TABLE_CAP (1u << 22)   // rows (y,z cell pairs) of the outlier stage's candidate grid
