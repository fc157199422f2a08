// cm_kernels.hip — gfx950 kernels of the merge → voxel-grid path.
//
// Pipeline per frame (all on one stream, no host round trip):
//   k_minmax     K0  transform + crop, fp32 min/max of the merged cloud per workgroup (skipped
//                    when the crop box already bounds the grid)               [HBM: 16 B/pt read]
//   k_keys       K1  PCL's overflow guard and grid, transform + crop + voxel key, digit-0 counts
//                    per tile and per group of tiles                          [16 B/pt r, 4 B/pt w]
//   per 8-bit digit: k_hist (digits 1..3: counts per tile and per group), k_scatter (cross-tile
//                    offsets summed in its prologue, stable LDS-tiled scatter); frames of more
//                    than CM_DIRECT_GROUPS groups run k_gscan in between      [LDS-tiled LSD radix]
//   k_seg_count  K3a kept voxels per sorted tile and per group of tiles
//   k_seg_reduce K3b gather + wavefront segmented centroid reduction, threshold, compaction
//                                                                             [16 B/voxel write]
//
// Arithmetic restates what the reference gets from pcl_ros::transformPointCloud
// (pc_preprocessing_main.cpp:322), pcl::PassThrough in getROI (:20-40) and pcl::VoxelGrid (:171-176);
// semantics in SURVEY.md Appendix A. Every fp32 operation that decides occupancy uses the _rn
// intrinsics so it is rounded once, in the reference's order, never contracted into an FMA.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cm_common.hpp"
#include "cm_device.h"
#include "cm_kernels.h"

namespace {


// ------------------------------------------------------------------------------------------------
// k_setup: upload the frame descriptor (passed by value) into HBM.
// ------------------------------------------------------------------------------------------------
__global__ void k_setup(CmFrameDev f, CmFrameDev* __restrict__ dst, CmTileDev* __restrict__ tiles) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = f;
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tiles && tile < f.n_tiles) {
        const uint32_t first = tile * CM_TILE;
        uint32_t k = 0;
        for (uint32_t q = 1; q < f.n_sensors; ++q) k += (first >= f.s[q].base) ? 1u : 0u;
        const CmSensorDev& sd = f.s[k];
        CmTileDev te;
        const uint32_t off = first - sd.base;
        te.data = sd.data + static_cast<size_t>(off) * sd.point_step;
        te.n_left = sd.n > off ? sd.n - off : 0u;
        te.info = k | (sd.layout << 8);
        tiles[tile] = te;
    }
}


// ------------------------------------------------------------------------------------------------
// K0: min/max of the transformed, cropped cloud (pcl::getMinMax3D, A.4 step 2). One record per
// workgroup (min xyz, max xyz, valid count), no atomics; k_keys folds the records.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_minmax(const CmFrameDev* __restrict__ fd,
                                                     float* __restrict__ partials,
                                                     const unsigned char* __restrict__ mask) {
    __shared__ float s_red[CM_WAVES][6];
    __shared__ uint32_t s_cnt[CM_WAVES];
    const uint32_t crop = fd->crop_enable;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inf = __uint_as_float(0x7F800000u);
    float mn0 = inf, mn1 = inf, mn2 = inf, mx0 = -inf, mx1 = -inf, mx2 = -inf;
    uint32_t cnt = 0;
    for (uint32_t tile = blockIdx.x; tile < fd->n_tiles; tile += gridDim.x) {
        const uint32_t s = sensor_of_tile(fd, tile);
        const CmSensorDev& sd = fd->s[s];
        float m[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
        const uint32_t slot0 = tile * CM_TILE + w * (64 * CM_ITEMS) + lane;
        const uint32_t first = slot0 - sd.base;
        Pt p[CM_ITEMS];
        load_tile(sd, first, p);
#pragma unroll
        for (int r = 0; r < CM_ITEMS; ++r) {
            const float x = xf_row(m[0], m[1], m[2], m[3], p[r].x, p[r].y, p[r].z);
            const float y = xf_row(m[4], m[5], m[6], m[7], p[r].x, p[r].y, p[r].z);
            const float z = xf_row(m[8], m[9], m[10], m[11], p[r].x, p[r].y, p[r].z);
            if (point_valid(x, y, z, crop, fd->crop_min, fd->crop_max) && (!mask || mask[slot0 + r * 64])) {
                mn0 = fminf(mn0, x); mx0 = fmaxf(mx0, x);
                mn1 = fminf(mn1, y); mx1 = fmaxf(mx1, y);
                mn2 = fminf(mn2, z); mx2 = fmaxf(mx2, z);
                ++cnt;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, d)); mx0 = fmaxf(mx0, __shfl_xor(mx0, d));
        mn1 = fminf(mn1, __shfl_xor(mn1, d)); mx1 = fmaxf(mx1, __shfl_xor(mx1, d));
        mn2 = fminf(mn2, __shfl_xor(mn2, d)); mx2 = fmaxf(mx2, __shfl_xor(mx2, d));
        cnt += __shfl_xor(cnt, d);
    }
    if (lane == 0) {
        s_red[w][0] = mn0; s_red[w][1] = mn1; s_red[w][2] = mn2;
        s_red[w][3] = mx0; s_red[w][4] = mx1; s_red[w][5] = mx2;
        s_cnt[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 8) {                       // record: min xyz, max xyz, count, pad
        const int k = threadIdx.x;
        float v = 0.f;
        if (k < 6) {
            v = s_red[0][k];
            for (int q = 1; q < CM_WAVES; ++q) v = (k < 3) ? fminf(v, s_red[q][k]) : fmaxf(v, s_red[q][k]);
        } else if (k == 6) {
            v = __uint_as_float(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
        }
        partials[blockIdx.x * 8 + k] = v;
    }
}


// ------------------------------------------------------------------------------------------------
// K1: grid set-up + transform + crop + linear voxel index (A.4 step 5) + digit-0 counts of the
// tile (one coalesced row) and of its group of CM_GROUP tiles. Keys of cropped / non-finite /
// padding slots are CM_INVALID_KEY and never enter the sort. Also clears the group totals the
// later kernels of the frame accumulate into.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_keys(const CmFrameDev* __restrict__ fd,
                                                   CmFrameState* __restrict__ st,
                                                   uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ hist,
                                                   uint32_t* __restrict__ grp_acc,
                                                   uint32_t* __restrict__ grp_clear_a,
                                                   uint32_t* __restrict__ grp_clear_b,
                                                   uint32_t n_group_words, uint32_t n_clear_a_words,
                                                   uint32_t* __restrict__ seg_groups, uint32_t n_seg_groups,
                                                   const float* __restrict__ partials,
                                                   uint32_t n_partials, int from_crop, int use_cell,
                                                   const unsigned char* __restrict__ mask,
                                                   const CmFrameState* __restrict__ st_outlier) {
    __shared__ uint32_t lh[CM_RADIX];
    __shared__ float s_red[CM_WAVES][8];
    const uint32_t tile = blockIdx.x;
    const float* __restrict__ inv = use_cell ? fd->inv_cell : fd->inv_leaf;
    // kept-voxel totals per group of sorted tiles (k_seg_count accumulates into them)
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < n_seg_groups * 32; k += gridDim.x * CM_BLOCK) seg_groups[k] = 0;
    // Clear the group totals of passes 1..3 (grp_clear_b, three arrays) and the pass-0 array the
    // NEXT frame accumulates into (grp_clear_a); this frame's pass-0 array (grp_acc) was cleared
    // by the previous frame.
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < 3 * n_group_words; k += gridDim.x * CM_BLOCK)
        grp_clear_b[k] = 0;
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < n_clear_a_words; k += gridDim.x * CM_BLOCK)
        grp_clear_a[k] = 0;                             // whole array: the next frame may be larger

    Grid g;
    compute_grid(fd, partials, n_partials, from_crop, inv, s_red, g);
    if (use_cell && g.status == CM_DEV_OVERFLOW) g.status = CM_DEV_OUTLIER_GRID;
    if (use_cell && g.status == CM_DEV_OK &&
        static_cast<unsigned long long>(g.div_b[1]) * static_cast<unsigned long long>(g.div_b[2]) > CM_ROW_TABLE_CAP)
        g.status = CM_DEV_OUTLIER_GRID;
    if (st_outlier && (st_outlier->status == CM_DEV_OUTLIER_GRID || st_outlier->status == CM_DEV_ABORTED)) g.status = st_outlier->status;
    if (tile == 0 && threadIdx.x == 0) {
        st->status = g.status;
        st->n_valid_k0 = g.n_valid_k0;
        for (int a = 0; a < 3; ++a) {
            st->min_p[a] = g.min_p[a]; st->max_p[a] = g.max_p[a];
            st->min_b[a] = g.min_b[a]; st->max_b[a] = g.max_b[a]; st->div_b[a] = g.div_b[a];
        }
        st->key_bits = g.key_bits;
        st->n_passes = g.n_passes;
    }
    if (g.status != CM_DEV_OK) return;

    const uint32_t s = sensor_of_tile(fd, tile);
    const CmSensorDev& sd = fd->s[s];
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
    const uint32_t crop = fd->crop_enable;
    const float inv0 = inv[0], inv1 = inv[1], inv2 = inv[2];
    const float fb0 = static_cast<float>(g.min_b[0]), fb1 = static_cast<float>(g.min_b[1]),
                fb2 = static_cast<float>(g.min_b[2]);
    const uint32_t mul1 = static_cast<uint32_t>(g.div_b[0]);
    const uint32_t mul2 = mul1 * static_cast<uint32_t>(g.div_b[1]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t slot0 = tile * CM_TILE + w * (64 * CM_ITEMS) + lane;   // padded global index
    const uint32_t first = slot0 - sd.base;                                // index in the sensor cloud

    Pt p[CM_ITEMS];
    load_tile(sd, first, p);
    lh[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r) {
        uint32_t key = CM_INVALID_KEY;
        const float x = xf_row(m[0], m[1], m[2], m[3], p[r].x, p[r].y, p[r].z);
        const float y = xf_row(m[4], m[5], m[6], m[7], p[r].x, p[r].y, p[r].z);
        const float z = xf_row(m[8], m[9], m[10], m[11], p[r].x, p[r].y, p[r].z);
        if (point_valid(x, y, z, crop, fd->crop_min, fd->crop_max) && (!mask || mask[slot0 + r * 64])) {
            const int c0 = static_cast<int>(__fsub_rn(floorf(__fmul_rn(x, inv0)), fb0));
            const int c1 = static_cast<int>(__fsub_rn(floorf(__fmul_rn(y, inv1)), fb1));
            const int c2 = static_cast<int>(__fsub_rn(floorf(__fmul_rn(z, inv2)), fb2));
            key = static_cast<uint32_t>(c0) + static_cast<uint32_t>(c1) * mul1 +
                  static_cast<uint32_t>(c2) * mul2;
            atomicAdd(&lh[key & (CM_RADIX - 1)], 1u);
        }
        keys[slot0 + r * 64] = key;
    }
    __syncthreads();
    const uint32_t c = lh[threadIdx.x];
    hist[static_cast<size_t>(tile) * CM_RADIX + threadIdx.x] = c;
    if (c) atomicAdd(&grp_acc[static_cast<size_t>(tile / CM_GROUP) * CM_RADIX + threadIdx.x], c);
}

// ------------------------------------------------------------------------------------------------
// LSD radix sort of (key, point index): one 8-bit digit per pass, 4096-item tiles in LDS.
// ------------------------------------------------------------------------------------------------
// Digit counts of every tile (one coalesced 1 KB row) and of every group of CM_GROUP tiles for
// passes >= 1 (the compacted, partially sorted pairs).
__global__ __launch_bounds__(CM_BLOCK) void k_hist(const CmFrameState* __restrict__ st,
                                                   const uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ hist,
                                                   uint32_t* __restrict__ grp, uint32_t pass) {
    __shared__ uint32_t lh[CM_RADIX];
    if (st->status != CM_DEV_OK || pass >= st->n_passes) return;
    const uint32_t n = st->n_valid;
    if (blockIdx.x * CM_TILE >= n) return;             // tiles past the data are never read
    const uint32_t shift = pass * CM_RADIX_BITS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * CM_TILE + w * (64 * CM_ITEMS) + lane;
    lh[threadIdx.x] = 0;
    uint32_t k[CM_ITEMS];
#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * 64;
        k[r] = (i < n) ? keys[i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r)
        if (first + r * 64 < n) atomicAdd(&lh[(k[r] >> shift) & (CM_RADIX - 1)], 1u);
    __syncthreads();
    const uint32_t c = lh[threadIdx.x];
    hist[static_cast<size_t>(blockIdx.x) * CM_RADIX + threadIdx.x] = c;
    if (c) atomicAdd(&grp[static_cast<size_t>(blockIdx.x / CM_GROUP) * CM_RADIX + threadIdx.x], c);
}

// Very large frames only (more than CM_DIRECT_GROUPS groups): group totals -> exclusive prefix
// over the groups, in place, plus the digit totals. One workgroup of 1024 threads: four chunks of
// groups are scanned side by side, thread (c, d) owning digit d of chunk c.
__global__ __launch_bounds__(1024) void k_gscan(const CmFrameState* __restrict__ st,
                                                uint32_t* __restrict__ grp,
                                                uint32_t* __restrict__ totals,
                                                uint32_t pass, uint32_t n_groups) {
    constexpr int GB = 32;            // groups fetched together: a frame of 16 M points (244 groups) costs two round trips per phase
    __shared__ uint32_t chunk_total[4][CM_RADIX];
    if (st->status != CM_DEV_OK || pass >= st->n_passes) return;
    const uint32_t d = threadIdx.x & (CM_RADIX - 1), c = threadIdx.x >> 8;
    const uint32_t per = (n_groups + 3) / 4;
    const uint32_t g0 = c * per, g1 = min(g0 + per, n_groups);
    uint32_t sum = 0;
    for (uint32_t g = g0; g < g1; g += GB) {
        uint32_t v[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {                    // unconditional loads (clamped), all in flight together
            const uint32_t x = grp[static_cast<size_t>(min(g + q, g1 - 1u)) * CM_RADIX + d];
            v[q] = (g + q < g1) ? x : 0u;
        }
#pragma unroll
        for (int q = 0; q < GB; ++q) sum += v[q];
    }
    chunk_total[c][d] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t q = 0; q < c; ++q) run += chunk_total[q][d];
    if (c == 3) totals[d] = run + sum;
    for (uint32_t g = g0; g < g1; g += GB) {
        uint32_t v[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {                    // unconditional loads (clamped), all in flight together
            const uint32_t x = grp[static_cast<size_t>(min(g + q, g1 - 1u)) * CM_RADIX + d];
            v[q] = (g + q < g1) ? x : 0u;
        }
#pragma unroll
        for (int q = 0; q < GB; ++q) {
            if (g + q < g1) grp[static_cast<size_t>(g + q) * CM_RADIX + d] = run;
            run += v[q];
        }
    }
}

// Lanes of the wave that hold the same 8-bit digit (among valid lanes).
__device__ __forceinline__ unsigned long long match_digit(uint32_t digit, bool valid) {
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < CM_RADIX_BITS; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
    }
    return mask;
}

// Device probe: does a returning LDS add hand lanes that hit the same address their old values in
// lane order (and successive instructions of a wave in program order)? Not an architectural
// promise, so the stable ranking built on it (k_scatter<*, true>) is only selected when this
// probe finds no violation on the device at hand; otherwise the ballot-match ranking is used.
__global__ __launch_bounds__(CM_BLOCK) void k_probe_lds_order(uint32_t* __restrict__ violations, uint32_t rounds) {
    __shared__ uint32_t cnt[CM_WAVES][CM_RADIX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t bad = 0;
    uint32_t x = 0x9E3779B9u * (blockIdx.x * CM_BLOCK + threadIdx.x + 1);
    for (uint32_t it = 0; it < rounds; ++it) {
        for (int q = 0; q < CM_WAVES; ++q) cnt[q][threadIdx.x] = 0;
        __syncthreads();
        // four back-to-back DS instructions; digits drawn from 1, 2, 7 or 256 values
        const uint32_t spread = (it & 3u) == 0 ? 1u : (it & 3u) == 1 ? 2u : (it & 3u) == 2 ? 7u : 256u;
        uint32_t dg[4], got[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            dg[r] = (x >> 8) % spread;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) got[r] = atomicAdd(&cnt[w][dg[r]], 1u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // lanes below me with my digit in instruction r + every lane of the earlier instructions
            uint32_t e = __popcll(match_digit(dg[r], true) & lt);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < r)
                    for (int l = 0; l < 64; ++l) e += (static_cast<uint32_t>(__shfl(static_cast<int>(dg[q]), l)) == dg[r]) ? 1u : 0u;
            }
            bad += (got[r] != e) ? 1u : 0u;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(violations, bad);
}

// Stable scatter of one tile by one digit. FIRST: values are the padded global indices and
// CM_INVALID_KEY slots are dropped (this is where the concatenated cloud gets compacted).
template <bool FIRST, bool LDS_RANK>
__global__ __launch_bounds__(CM_BLOCK, 4) void k_scatter(CmFrameState* __restrict__ st,
                                                      const uint32_t* __restrict__ keys_in,
                                                      const uint32_t* __restrict__ vals_in,
                                                      uint32_t* __restrict__ keys_out,
                                                      uint32_t* __restrict__ vals_out,
                                                      const uint32_t* __restrict__ hist,
                                                      const uint32_t* __restrict__ grp,
                                                      const uint32_t* __restrict__ totals,
                                                      uint32_t pass, uint32_t n_groups, uint32_t n_padded,
                                                      uint32_t* __restrict__ tile_kept) {
    __shared__ uint32_t whist[CM_WAVES][CM_RADIX];
    __shared__ uint32_t gofs[CM_RADIX];
    __shared__ uint32_t skey[CM_TILE];
    __shared__ uint32_t sval[CM_TILE];
    __shared__ uint32_t lds[CM_WAVES];
    __shared__ uint32_t s_tile_valid;
    if (st->status != CM_DEV_OK || pass >= st->n_passes) return;
    // Workgroups are dealt round-robin over the 8 XCDs (speed only, never correctness): give each
    // XCD a contiguous range of tiles, so the digit runs of neighbouring tiles — adjacent in the
    // output — meet in one L2 instead of leaving as partial lines from eight.
    uint32_t tile = blockIdx.x;
    {
        const uint32_t per = gridDim.x / 8;
        if (blockIdx.x < per * 8) tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    // After pass 0 only the valid items are left, and pass 0 recorded how many: a crop box may have dropped most
    // of the frame, and the tiles past the end have nothing to read.
    if (!FIRST && tile * CM_TILE >= st->n_valid) return;
    const uint32_t shift = pass * CM_RADIX_BITS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t first = tile * CM_TILE + w * (64 * CM_ITEMS) + lane;

    // All loads of the tile go out first (keys, and the values they carry); the cross-tile sums
    // below run while they are in flight. Slots past n are never counted (n is known only after
    // the sums, so bound the loads by the launch-time upper bound).
    const uint32_t n_max = n_padded;
    uint32_t key[CM_ITEMS], val[CM_ITEMS];
#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * 64;
        key[r] = (i < n_max) ? keys_in[i] : CM_INVALID_KEY;
        val[r] = FIRST ? i : ((i < n_max) ? vals_in[i] : 0u);
    }

    // Items of digit d written before this tile's (thread d): every smaller digit of the whole
    // frame (gbase) + digit d in earlier groups + digit d in earlier tiles of this group.
    const uint32_t grp_id = tile / CM_GROUP;
    uint32_t before = 0, my_total = 0;
    if (totals) {                                       // large frame: k_gscan left prefixes
        my_total = totals[threadIdx.x];
        before = grp[static_cast<size_t>(grp_id) * CM_RADIX + threadIdx.x];
    } else {
        for (uint32_t g = 0; g < n_groups; g += 16) {
            uint32_t v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = (g + q < n_groups) ? grp[static_cast<size_t>(g + q) * CM_RADIX + threadIdx.x] : 0u;
#pragma unroll
            for (int q = 0; q < 16; ++q) { my_total += v[q]; before += (g + q < grp_id) ? v[q] : 0u; }
        }
    }
    {
        const uint32_t t0 = grp_id * CM_GROUP;
        for (uint32_t t = t0; t < tile; t += 16) {
            uint32_t v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = (t + q < tile) ? hist[static_cast<size_t>(t + q) * CM_RADIX + threadIdx.x] : 0u;
#pragma unroll
            for (int q = 0; q < 16; ++q) before += v[q];
        }
    }
    uint32_t gtot;
    const uint32_t gbase = block_excl_scan_u32(my_total, lds, &gtot);
    if (FIRST && tile == 0 && threadIdx.x == 0) st->n_valid = gtot;
    const uint32_t n = FIRST ? n_padded : gtot;        // every pass sorts the same valid items
    if (tile * CM_TILE >= n) return;                   // uniform: empty tile (after the scan's barriers)

#pragma unroll
    for (int q = 0; q < CM_WAVES; ++q) whist[q][threadIdx.x] = 0;
    __syncthreads();

    // Rank inside the wave, rounds in order, lanes in order: stable.
    uint32_t rank[CM_ITEMS];
    if (LDS_RANK) {
        // One returning LDS add per item. Only used after k_probe_lds_order has shown, on this
        // device, that same-address lanes of one DS instruction are served in lane order.
#pragma unroll
        for (int r = 0; r < CM_ITEMS; ++r) {
            const bool valid = FIRST ? (key[r] != CM_INVALID_KEY) : (first + r * 64 < n);
            const uint32_t digit = (key[r] >> shift) & (CM_RADIX - 1);
            rank[r] = valid ? atomicAdd(&whist[w][digit], 1u) : 0u;
        }
    } else {
        volatile uint32_t* wh = whist[w];
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < CM_ITEMS; ++r) {
            const bool valid = FIRST ? (key[r] != CM_INVALID_KEY) : (first + r * 64 < n);
            const uint32_t digit = (key[r] >> shift) & (CM_RADIX - 1);
            const unsigned long long peers = match_digit(digit, valid);
            const uint32_t below = __popcll(peers & lt);
            const uint32_t base = wh[digit];
            __builtin_amdgcn_wave_barrier();
            if (valid && below == 0) wh[digit] = base + __popcll(peers);
            __builtin_amdgcn_wave_barrier();
            rank[r] = base + below;
        }
    }
    __syncthreads();

    // Digit d: tile count, position of digit d in the tile, per-wave bases.
    {
        const uint32_t d = threadIdx.x;
        const uint32_t c0 = whist[0][d], c1 = whist[1][d], c2 = whist[2][d], c3 = whist[3][d];
        uint32_t tile_valid;
        const uint32_t dbase = block_excl_scan_u32(c0 + c1 + c2 + c3, lds, &tile_valid);
        whist[0][d] = dbase;
        whist[1][d] = dbase + c0;
        whist[2][d] = dbase + c0 + c1;
        whist[3][d] = dbase + c0 + c1 + c2;
        gofs[d] = gbase + before - dbase;
        if (d == 0) s_tile_valid = tile_valid;
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * 64;
        const bool valid = FIRST ? (key[r] != CM_INVALID_KEY) : (i < n);
        if (valid) {
            const uint32_t digit = (key[r] >> shift) & (CM_RADIX - 1);
            const uint32_t pos = whist[w][digit] + rank[r];
            skey[pos] = key[r];
            sval[pos] = val[r];
        }
    }
    __syncthreads();

    const uint32_t tile_valid = s_tile_valid;
    if (FIRST && tile_kept && threadIdx.x == 0) tile_kept[tile] = tile_valid;   // cm_get_frame_stats: points per sensor that entered the grid
#pragma unroll
    for (int j = 0; j < CM_ITEMS; ++j) {
        const uint32_t t = j * CM_BLOCK + threadIdx.x;
        if (t < tile_valid) {
            const uint32_t k = skey[t];
            const uint32_t p = gofs[(k >> shift) & (CM_RADIX - 1)] + t;
            keys_out[p] = k;
            vals_out[p] = sval[t];
        }
    }
}

__device__ __forceinline__ const uint32_t* pick(const CmFrameState* st, const uint32_t* a, const uint32_t* b) {
    return (st->n_passes & 1u) ? b : a;      // pass p reads A when p is even and writes the other
}

// ------------------------------------------------------------------------------------------------
// K3a: kept voxels per tile of the sorted keys (+ totals per group of CM_SEG_GROUP tiles), and a
// check that the keys really are sorted (st->err = 2 otherwise). A run is
// owned by the tile holding its head; it is kept iff it reaches min_points_per_voxel (A.4 step 7),
// i.e. keys[head + min_pts - 1] == key.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_seg_count(CmFrameState* __restrict__ st,
                                                        const uint32_t* __restrict__ keys_a,
                                                        const uint32_t* __restrict__ keys_b,
                                                        uint32_t* __restrict__ counts,
                                                        uint32_t* __restrict__ group_counts,
                                                        uint32_t min_pts) {
    __shared__ uint32_t lds[CM_WAVES];
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t base = blockIdx.x * CM_SEG_TILE;
    if (base >= n) return;
    const uint32_t* __restrict__ keys = pick(st, keys_a, keys_b);
    uint32_t cnt = 0;
    bool unsorted = false;
#pragma unroll
    for (int j = 0; j < CM_SEG_ITEMS; ++j) {
        const uint32_t i = base + j * CM_BLOCK + threadIdx.x;
        if (i < n) {
            const uint32_t k = keys[i];
            const uint32_t kp = (i == 0) ? 0u : keys[i - 1];
            unsorted = unsorted || (i > 0 && kp > k);   // every adjacent pair is seen here: free check of the sort
            const bool head = (i == 0) || (kp != k);
            bool keep = head;
            if (head && min_pts > 1) {
                const uint32_t e = i + min_pts - 1;
                keep = (e >= i) && (e < n) && (keys[e] == k);
            }
            cnt += keep ? 1u : 0u;
        }
    }
    if (unsorted) st->err = 2u;
    const uint32_t tot = block_sum_u32(cnt, lds);
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = tot;
        // Large frames only. One 128-byte line per group: same-line atomic requests serialise.
        if (group_counts && tot) atomicAdd(&group_counts[(blockIdx.x / CM_SEG_GROUP) * 32], tot);
    }
}

// ------------------------------------------------------------------------------------------------
// K3: kept voxels per sorted tile (published), gather, segmented centroid reduction, threshold,
// compaction.
// Thread t owns 8 consecutive sorted items; runs closed inside a thread are summed sequentially in
// sorted (= stable point) order, runs crossing threads by a wave64 segmented suffix scan (DPP
// shuffles), runs crossing waves through LDS, runs crossing the tile by a cooperative extension
// loop of the owning workgroup. No atomics: sums are deterministic.
// ------------------------------------------------------------------------------------------------

struct SensorLds {
    const unsigned char* data;
    uint32_t base, step, ox, oy, oz, oi, layout, _pad;
    float m[12];
};

// What the centroid kernel sums. SEG_CENTROIDS: points -> voxel means (the path). SEG_PARTIAL:
// points -> per-voxel sums and counts, no threshold (one GPU's share of a fused cloud).
// SEG_TABLES: 32-byte partial entries of several GPUs -> merged sums and counts.
enum { SEG_CENTROIDS = 0, SEG_PARTIAL = 1, SEG_TABLES = 2 };

template <int MODE>
__device__ __forceinline__ Acc gather_item(const SensorLds* __restrict__ tab, uint32_t n_sensors, uint32_t gidx,
                                           bool all_fields) {
    uint32_t s = 0;
    for (uint32_t q = 1; q < n_sensors; ++q) s += (gidx >= tab[q].base) ? 1u : 0u;
    const SensorLds& sd = tab[s];
    Acc o;
    if (MODE == SEG_TABLES) {
        const float4* e = reinterpret_cast<const float4*>(sd.data + static_cast<size_t>(gidx - sd.base) * 32);
        const float4 lo = e[0], hi = e[1];
        o.c = __float_as_uint(lo.y);
        o.x = lo.z; o.y = lo.w; o.z = hi.x; o.i = hi.y;
    } else {
        const Pt p = load_point(sd.data, sd.layout, sd.step, sd.ox, sd.oy, sd.oz, sd.oi, gidx - sd.base);
        o.x = xf_row(sd.m[0], sd.m[1], sd.m[2], sd.m[3], p.x, p.y, p.z);
        o.y = xf_row(sd.m[4], sd.m[5], sd.m[6], sd.m[7], p.x, p.y, p.z);
        o.z = xf_row(sd.m[8], sd.m[9], sd.m[10], sd.m[11], p.x, p.y, p.z);
        o.i = all_fields ? p.i : 0.f;
        o.c = 1u;
    }
    return o;
}


template <int MODE>
__global__ __launch_bounds__(CM_BLOCK) void k_seg_reduce(const CmFrameDev* __restrict__ fd,
                                                         CmFrameState* __restrict__ st,
                                                         CmFrameState* __restrict__ st_next,
                                                         uint32_t* __restrict__ host_state,
                                                         const uint32_t* __restrict__ keys_a,
                                                         const uint32_t* __restrict__ vals_a,
                                                         const uint32_t* __restrict__ keys_b,
                                                         const uint32_t* __restrict__ vals_b,
                                                         const uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ group_counts,
                                                         float4* __restrict__ out,
                                                         uint32_t* __restrict__ out_key,
                                                         uint32_t* __restrict__ out_cnt) {
    __shared__ SensorLds tab[CM_DEV_MAX_SENSORS];
    __shared__ float s_acc[CM_WAVES + 1][4];      // resolved suffix sums at the first lane of each wave
    __shared__ uint32_t s_accc[CM_WAVES + 1];
    __shared__ uint32_t s_flag[CM_WAVES];
    __shared__ float s_ext[CM_WAVES][4];
    __shared__ uint32_t s_extc[CM_WAVES];
    __shared__ uint32_t lds[CM_WAVES];
    const uint32_t n_sensors = fd->n_sensors;
    if (threadIdx.x < n_sensors) {
        const CmSensorDev& g = fd->s[threadIdx.x];
        SensorLds& t = tab[threadIdx.x];
        t.data = g.data; t.base = g.base; t.step = g.point_step;
        t.ox = g.off_x; t.oy = g.off_y; t.oz = g.off_z; t.oi = g.off_i; t.layout = g.layout;
        for (int k = 0; k < 12; ++k) t.m[k] = g.m[k];
    }
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    if (tile == 0 && st_next && threadIdx.x < sizeof(CmFrameState) / 4)
        reinterpret_cast<uint32_t*>(st_next)[threadIdx.x] = 0;          // next frame starts from zero
    if (st->status != CM_DEV_OK) {
        if (tile == 0) report_state(host_state, st, st->status, 0u);
        return;
    }
    const uint32_t n = st->n_valid;
    if (n == 0) {
        if (tile == 0) report_state(host_state, st, CM_DEV_EMPTY, 0u);
        return;
    }
    const uint32_t base = tile * CM_SEG_TILE;
    if (base >= n) return;
    const uint32_t n_tiles = (n + CM_SEG_TILE - 1) / CM_SEG_TILE;
    const uint32_t* __restrict__ keys = pick(st, keys_a, keys_b);
    const uint32_t* __restrict__ vals = pick(st, vals_a, vals_b);
    const uint32_t min_pts = (MODE == SEG_CENTROIDS && fd->min_pts > 1) ? fd->min_pts : 1u;
    const bool all_fields = fd->downsample_all != 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

    const uint32_t tile_n = min(static_cast<uint32_t>(CM_SEG_TILE), n - base);
    const uint32_t i0 = base + threadIdx.x * CM_SEG_ITEMS;

    // Every load that does not depend on another goes out now: the keys of the chunk, the key
    // before and after it, the point indices the keys carry, and the first keys past the tile
    // (for the extension of its last run).
    uint32_t k[CM_SEG_ITEMS], v[CM_SEG_ITEMS];
    if (i0 + CM_SEG_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(keys + i0);
        const uint4 b = *reinterpret_cast<const uint4*>(keys + i0 + 4);
        const uint4 c = *reinterpret_cast<const uint4*>(vals + i0);
        const uint4 d = *reinterpret_cast<const uint4*>(vals + i0 + 4);
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
        v[0] = c.x; v[1] = c.y; v[2] = c.z; v[3] = c.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
    } else {
#pragma unroll
        for (int j = 0; j < CM_SEG_ITEMS; ++j) {
            k[j] = (i0 + j < n) ? keys[i0 + j] : 0u;
            v[j] = (i0 + j < n) ? vals[i0 + j] : 0u;
        }
    }
    const uint32_t kprev = (i0 > 0 && i0 < n) ? keys[i0 - 1] : 0u;
    const bool has_next = i0 + CM_SEG_ITEMS < n;
    const uint32_t knext = has_next ? keys[i0 + CM_SEG_ITEMS] : 0u;
    const uint32_t tile_end = base + tile_n;
    const uint32_t klast = keys[tile_end - 1];
    const uint32_t jext0 = tile_end + threadIdx.x;
    const uint32_t kext0 = (jext0 < n) ? keys[jext0] : 0u;
    const uint32_t vext0 = (jext0 < n) ? vals[jext0] : 0u;

    // head flags; `need`: items whose run can reach min_points_per_voxel (all, or runs of >= 2).
    uint32_t heads = 0, live = 0, need = 0;
#pragma unroll
    for (int j = 0; j < CM_SEG_ITEMS; ++j) {
        const uint32_t i = i0 + j;
        if (i < n) {
            live |= 1u << j;
            const uint32_t pk = (j == 0) ? kprev : k[(j + CM_SEG_ITEMS - 1) % CM_SEG_ITEMS];
            const bool head = (i == 0) || (pk != k[j]);
            const bool same_next = (j == CM_SEG_ITEMS - 1) ? (has_next && knext == k[j])
                                                           : (i + 1 < n && k[(j + 1) % CM_SEG_ITEMS] == k[j]);
            if (head) heads |= 1u << j;
            if (min_pts <= 1 || !head || same_next) need |= 1u << j;
        }
    }
    // Output offset of this tile: kept voxels of the earlier groups + of the earlier tiles of its
    // own group (k_seg_count); loads issued here, summed after the gather.
    uint32_t before = 0;
    if (group_counts) {
        const uint32_t g = tile / CM_SEG_GROUP;
        for (uint32_t q = threadIdx.x; q < g; q += CM_BLOCK) before += group_counts[q * 32];
        const uint32_t t = g * CM_SEG_GROUP + threadIdx.x;      // CM_SEG_GROUP == CM_BLOCK
        if (t < tile) before += counts[t];
    } else {
        for (uint32_t t = threadIdx.x; t < tile; t += CM_BLOCK) before += counts[t];
    }

    // gather + transform (only points whose run can survive the threshold)
    Acc it[CM_SEG_ITEMS];
#pragma unroll
    for (int j = 0; j < CM_SEG_ITEMS; ++j) {
        if (need >> j & 1u) it[j] = gather_item<MODE>(tab, n_sensors, v[j], all_fields);
        else { it[j].x = it[j].y = it[j].z = it[j].i = 0.f; it[j].c = 1u; }
    }

    // Extension: items after the tile that continue its last run (owned by this workgroup).
    Acc ext = {0.f, 0.f, 0.f, 0.f, 0u};
    {
        bool any = false;
        for (uint32_t off = 0;; off += CM_BLOCK) {
            const uint32_t j = tile_end + off + threadIdx.x;
            const uint32_t kj = (off == 0) ? kext0 : ((j < n) ? keys[j] : 0u);
            const bool ok = (j < n) && (kj == klast);
            if (ok) {
                const Acc q = gather_item<MODE>(tab, n_sensors, (off == 0) ? vext0 : vals[j], all_fields);
                if (ext.c == 0) ext = q; else acc_add(ext, q);
            }
            any = any || ok;
            if (!__syncthreads_or(ok && threadIdx.x == CM_BLOCK - 1)) break;
        }
        if (__syncthreads_or(any)) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                Acc o;
                o.x = __shfl_xor(ext.x, d); o.y = __shfl_xor(ext.y, d); o.z = __shfl_xor(ext.z, d);
                o.i = __shfl_xor(ext.i, d); o.c = __shfl_xor(ext.c, d);
                acc_add(ext, o);
            }
            if (lane == 0) {
                s_ext[w][0] = ext.x; s_ext[w][1] = ext.y; s_ext[w][2] = ext.z; s_ext[w][3] = ext.i;
                s_extc[w] = ext.c;
            }
            __syncthreads();
            ext.x = ext.y = ext.z = ext.i = 0.f; ext.c = 0;
            for (int q = 0; q < CM_WAVES; ++q) {
                Acc o = {s_ext[q][0], s_ext[q][1], s_ext[q][2], s_ext[q][3], s_extc[q]};
                acc_add(ext, o);
            }
        }
    }

    // Thread-local pass in sorted order. `pre` = items before the first head (they belong to a
    // run owned further left); a run that ends inside the chunk is finished at the item that
    // closes it (static register index); `run` = the last, still open run.
    Acc pre = {0.f, 0.f, 0.f, 0.f, 0u};
    Acc run = {0.f, 0.f, 0.f, 0.f, 0u};
    Acc fin[CM_SEG_ITEMS];
    uint32_t fkey[CM_SEG_ITEMS];
    uint32_t fmask = 0, run_key = 0;
    bool open = false;
#pragma unroll
    for (int j = 0; j < CM_SEG_ITEMS; ++j) {
        fin[j].x = fin[j].y = fin[j].z = fin[j].i = 0.f; fin[j].c = 0; fkey[j] = 0;
        if (live >> j & 1u) {
            if (heads >> j & 1u) {
                if (open) { fin[j] = run; fkey[j] = run_key; fmask |= 1u << j; }
                run = it[j]; run_key = k[j]; open = true;
            } else if (open) {
                acc_add(run, it[j]);
            } else if (pre.c == 0) {
                pre = it[j];
            } else {
                acc_add(pre, it[j]);
            }
        }
    }
    const bool has_head = open;

    // Wave64 segmented suffix scan of `pre`: S[t] = pre[t] + (has_head[t] ? 0 : S[t+1]).
    Acc S = pre;
    uint32_t flag = has_head ? 1u : 0u;
    // Common case: every lane of the wave has a head in its chunk, so nothing propagates further
    // than one lane (S[t] = pre[t]) and the scan is skipped (wave-uniform branch).
    if (__ballot(has_head) != ~0ull) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const Acc o = acc_shfl_down(S, d);
            const uint32_t of = __shfl_down(flag, d);
            if (lane + d < 64 && !flag) {
                if (S.c == 0) S = o; else if (o.c) acc_add(S, o);
                flag |= of;
            }
        }
    }
    if (lane == 0) {
        s_acc[w][0] = S.x; s_acc[w][1] = S.y; s_acc[w][2] = S.z; s_acc[w][3] = S.i;
        s_accc[w] = S.c; s_flag[w] = flag;
    }
    if (threadIdx.x == 0) {
        s_acc[CM_WAVES][0] = ext.x; s_acc[CM_WAVES][1] = ext.y; s_acc[CM_WAVES][2] = ext.z;
        s_acc[CM_WAVES][3] = ext.i; s_accc[CM_WAVES] = ext.c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                       // resolve the wave chain right to left
        for (int q = CM_WAVES - 1; q >= 0; --q) {
            if (!s_flag[q] && s_accc[q + 1]) {
                if (s_accc[q] == 0) {
                    for (int e = 0; e < 4; ++e) s_acc[q][e] = s_acc[q + 1][e];
                } else {
                    for (int e = 0; e < 4; ++e) s_acc[q][e] = __fadd_rn(s_acc[q][e], s_acc[q + 1][e]);
                }
                s_accc[q] += s_accc[q + 1];
            }
        }
    }
    __syncthreads();
    if (!flag) {                                   // no head from this lane to the end of the wave
        const Acc o = {s_acc[w + 1][0], s_acc[w + 1][1], s_acc[w + 1][2], s_acc[w + 1][3], s_accc[w + 1]};
        if (S.c == 0) S = o; else if (o.c) acc_add(S, o);
    }
    // carry = S of the next thread (exclusive); the last lane takes the next wave's resolved value.
    Acc carry = acc_shfl_down(S, 1);
    if (lane == 63) {
        carry.x = s_acc[w + 1][0]; carry.y = s_acc[w + 1][1]; carry.z = s_acc[w + 1][2];
        carry.i = s_acc[w + 1][3]; carry.c = s_accc[w + 1];
    }
    if (has_head && carry.c) acc_add(run, carry);

    // keep flags and output slots (runs in sorted order: closed ones first, the open one last)
    uint32_t nkeep = (has_head && run.c >= min_pts) ? 1u : 0u;
#pragma unroll
    for (int j = 0; j < CM_SEG_ITEMS; ++j)
        if ((fmask >> j & 1u) && fin[j].c >= min_pts) ++nkeep;
    const uint32_t tile_off = block_sum_u32(before, lds);
    if (tile == n_tiles - 1) report_state(host_state, st, CM_DEV_OK, tile_off + counts[tile]);
    uint32_t tot;
    uint32_t slot = tile_off + block_excl_scan_u32(nkeep, lds, &tot);
#pragma unroll
    for (int j = 0; j <= CM_SEG_ITEMS; ++j) {
        const bool is_last = (j == CM_SEG_ITEMS);
        const bool emit = is_last ? (has_head && run.c >= min_pts)
                                  : ((fmask >> (j & 7) & 1u) && fin[j & 7].c >= min_pts);
        if (emit) {
            const Acc a = is_last ? run : fin[j & 7];
            const uint32_t ak = is_last ? run_key : fkey[j & 7];
            if (MODE == SEG_CENTROIDS) {
                const float c = static_cast<float>(a.c);
                float4 o;
                o.x = __fdiv_rn(a.x, c); o.y = __fdiv_rn(a.y, c);
                o.z = __fdiv_rn(a.z, c); o.w = __fdiv_rn(a.i, c);
                out[slot] = o;
                if (out_key) { out_key[slot] = ak; out_cnt[slot] = a.c; }
            } else {                                   // cm_partial_entry: key, count, sx, sy | sz, si, 0, 0
                out[2 * static_cast<size_t>(slot)] = make_float4(__uint_as_float(ak), __uint_as_float(a.c), a.x, a.y);
                out[2 * static_cast<size_t>(slot) + 1] = make_float4(a.z, a.i, 0.f, 0.f);
            }
            ++slot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused cloud across GPUs (nothing like it exists in the reference). Every GPU turns
// its share of the sensors into a table of per-voxel sums (SEG_PARTIAL), the host all-gathers the
// tables over RCCL, and the merge below re-sorts the concatenated entries by voxel index (stable:
// equal voxels stay in rank order, so sums are deterministic), adds them (SEG_TABLES), and only
// then applies min_points_per_voxel and divides — a voxel may hold one point on each of two GPUs.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_table_keys(const CmFrameDev* __restrict__ fd,
                                                         CmFrameState* __restrict__ st,
                                                         uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ hist,
                                                         uint32_t* __restrict__ grp_acc,
                                                         uint32_t* __restrict__ grp_clear_a,
                                                         uint32_t* __restrict__ grp_clear_b,
                                                         uint32_t n_group_words, uint32_t n_clear_a_words,
                                                         uint32_t* __restrict__ seg_groups, uint32_t n_seg_groups,
                                                         uint32_t key_bits) {
    __shared__ uint32_t lh[CM_RADIX];
    const uint32_t tile = blockIdx.x;
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < n_seg_groups * 32; k += gridDim.x * CM_BLOCK) seg_groups[k] = 0;
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < 3 * n_group_words; k += gridDim.x * CM_BLOCK) grp_clear_b[k] = 0;
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < n_clear_a_words; k += gridDim.x * CM_BLOCK) grp_clear_a[k] = 0;
    if (tile == 0 && threadIdx.x == 0) {
        st->status = CM_DEV_OK;
        st->key_bits = key_bits;
        st->n_passes = (key_bits + CM_RADIX_BITS - 1) / CM_RADIX_BITS;
    }
    const uint32_t s = sensor_of_tile(fd, tile);
    const CmSensorDev& sd = fd->s[s];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t slot0 = tile * CM_TILE + w * (64 * CM_ITEMS) + lane;
    const uint32_t first = slot0 - sd.base;
    lh[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * 64;
        uint32_t key = CM_INVALID_KEY;
        if (i < sd.n) {
            key = *reinterpret_cast<const uint32_t*>(sd.data + static_cast<size_t>(i) * 32);
            atomicAdd(&lh[key & (CM_RADIX - 1)], 1u);
        }
        keys[slot0 + r * 64] = key;
    }
    __syncthreads();
    const uint32_t c = lh[threadIdx.x];
    hist[static_cast<size_t>(tile) * CM_RADIX + threadIdx.x] = c;
    if (c) atomicAdd(&grp_acc[static_cast<size_t>(tile / CM_GROUP) * CM_RADIX + threadIdx.x], c);
}

// Threshold + divide + compaction of merged entries (two launches around k_scan_counts).
__global__ __launch_bounds__(CM_BLOCK) void k_table_finish(const float4* __restrict__ entries, uint32_t n,
                                                           uint32_t min_pts, uint32_t* __restrict__ tile_counts,
                                                           float4* __restrict__ out, uint32_t* __restrict__ out_key,
                                                           uint32_t* __restrict__ out_cnt, int write) {
    __shared__ uint32_t lds[CM_WAVES];
    const uint32_t need = min_pts > 1 ? min_pts : 1u;
    uint32_t slot = write ? tile_counts[blockIdx.x] : 0u;
    uint32_t total = 0;
    for (int r = 0; r < CM_ITEMS; ++r) {            // rounds in order, threads in order: stable
        const uint32_t i = blockIdx.x * CM_TILE + r * CM_BLOCK + threadIdx.x;
        bool keep = false;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (i < n) {
            lo = entries[2 * static_cast<size_t>(i)];
            hi = entries[2 * static_cast<size_t>(i) + 1];
            keep = __float_as_uint(lo.y) >= need;
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(keep ? 1u : 0u, lds, &tot);
        if (write && keep) {
            const uint32_t cnt = __float_as_uint(lo.y);
            const float c = static_cast<float>(cnt);
            out[slot + ex] = make_float4(__fdiv_rn(lo.z, c), __fdiv_rn(lo.w, c), __fdiv_rn(hi.x, c), __fdiv_rn(hi.y, c));
            if (out_key) { out_key[slot + ex] = __float_as_uint(lo.x); out_cnt[slot + ex] = cnt; }
        }
        slot += tot;
        total += tot;
    }
    if (!write && threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

// ------------------------------------------------------------------------------------------------
// Radius outlier removal on the fused cloud (reference remove_outliers,
// my_cloud_fusion/src/CloudFusionNode.h:74-85, applied before voxelgrid at cloud_fusion_node.cpp:72;
// live node outlierRemoval pc_preprocessing_main.cpp:184-192). pcl::RadiusOutlierRemoval keeps a
// point iff k > min_neighbors, k = points (itself included) with fp32 squared distance
// ((dx*dx + dy*dy) + dz*dz) < float(r*r).
// Candidates come from a grid a little wider than the radius: the fused cloud is sorted by that
// grid's linear cell index with the same radix sort, the points are gathered once into sorted
// order (so the cells of a row are contiguous), a (y,z)-row table gives each row's range, and one
// thread per point walks the 9 neighbouring rows x 3 cells, stopping at min_neighbors + 1 hits.
// The result is a keep-mask over the padded point indices that k_minmax / k_keys honour.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_gather_sorted(const CmFrameDev* __restrict__ fd,
                                                            const CmFrameState* __restrict__ st,
                                                            const uint32_t* __restrict__ vals_a,
                                                            const uint32_t* __restrict__ vals_b,
                                                            float4* __restrict__ sorted_pts) {
    __shared__ SensorLds tab[CM_DEV_MAX_SENSORS];
    const uint32_t n_sensors = fd->n_sensors;
    if (threadIdx.x < n_sensors) {
        const CmSensorDev& g = fd->s[threadIdx.x];
        SensorLds& t = tab[threadIdx.x];
        t.data = g.data; t.base = g.base; t.step = g.point_step;
        t.ox = g.off_x; t.oy = g.off_y; t.oz = g.off_z; t.oi = g.off_i; t.layout = g.layout;
        for (int k = 0; k < 12; ++k) t.m[k] = g.m[k];
    }
    __syncthreads();
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t* __restrict__ vals = pick(st, vals_a, vals_b);
    for (uint32_t p = blockIdx.x * CM_BLOCK + threadIdx.x; p < n; p += gridDim.x * CM_BLOCK) {
        const uint32_t idx = vals[p];
        const Acc a = gather_item<SEG_PARTIAL>(tab, n_sensors, idx, false);
        sorted_pts[p] = make_float4(a.x, a.y, a.z, __uint_as_float(idx));
    }
}

__global__ __launch_bounds__(CM_BLOCK) void k_row_clear(const CmFrameState* __restrict__ st, uint2* __restrict__ rows) {
    if (st->status != CM_DEV_OK) return;
    const uint32_t n_rows = static_cast<uint32_t>(st->div_b[1]) * static_cast<uint32_t>(st->div_b[2]);
    for (uint32_t r = blockIdx.x * CM_BLOCK + threadIdx.x; r < n_rows; r += gridDim.x * CM_BLOCK) rows[r] = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(CM_BLOCK) void k_row_table(const CmFrameState* __restrict__ st,
                                                        const uint32_t* __restrict__ keys_a,
                                                        const uint32_t* __restrict__ keys_b,
                                                        uint2* __restrict__ rows) {
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t dx = static_cast<uint32_t>(st->div_b[0]);
    const uint32_t* __restrict__ keys = pick(st, keys_a, keys_b);
    for (uint32_t p = blockIdx.x * CM_BLOCK + threadIdx.x; p < n; p += gridDim.x * CM_BLOCK) {
        const uint32_t row = keys[p] / dx;
        const uint32_t prev = p ? keys[p - 1] / dx : 0xFFFFFFFFu;
        if (row != prev) {
            rows[row].x = p;
            if (p) rows[prev].y = p;
        }
        if (p == n - 1) rows[row].y = n;
    }
}

// Neighbour search, first launch: every point looks along its own row only — its neighbours in the sorted
// order — inside an LDS window (1024 points of the workgroup + 256 on either side), so the scan costs no
// dependent global loads. Points that reach min_neighbors are marked kept; the others go, with the count they
// have, onto a list (per workgroup in LDS, one global add per workgroup, then the copy) for k_neighbors. A scan
// that runs into the edge of the window before its cell range ends is flagged: k_neighbors redoes that row scan
// through global memory. pend_a/pend_b: the two key buffers of the sort (the list goes into the one the sort did
// not end in); pend_c: a vals buffer (free once the points are gathered).
#define CM_NB_TILE 1024
#define CM_NB_HALO 256
__global__ __launch_bounds__(CM_BLOCK) void k_neighbors_row(const CmFrameDev* __restrict__ fd,
                                                            const CmFrameState* __restrict__ st,
                                                            const uint32_t* __restrict__ keys_a,
                                                            const uint32_t* __restrict__ keys_b,
                                                            const float4* __restrict__ sorted_pts,
                                                            unsigned char* __restrict__ mask,
                                                            const unsigned char* __restrict__ cls,
                                                            uint32_t* __restrict__ pend_a, uint32_t* __restrict__ pend_b,
                                                            uint32_t* __restrict__ pend_c, uint32_t* __restrict__ pend_n) {
    __shared__ uint32_t wk[CM_NB_TILE + 2 * CM_NB_HALO];
    __shared__ float4 wp[CM_NB_TILE + 2 * CM_NB_HALO];
    __shared__ uint32_t s_pp[CM_NB_TILE], s_pc[CM_NB_TILE];
    __shared__ uint32_t s_np, s_base;
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t base = blockIdx.x * CM_NB_TILE;
    if (base >= n) return;
    uint32_t* __restrict__ pend_p = (st->n_passes & 1u) ? pend_a : pend_b;
    const uint32_t* __restrict__ keys = pick(st, keys_a, keys_b);
    const uint32_t dx = static_cast<uint32_t>(st->div_b[0]);
    const float r2 = fd->outlier_r2;
    const uint32_t need = fd->outlier_min_nb;
    const uint32_t w0 = base >= CM_NB_HALO ? base - CM_NB_HALO : 0u;
    const uint32_t w1 = min(n, base + CM_NB_TILE + CM_NB_HALO);
    const uint32_t wn = w1 - w0;
    if (threadIdx.x == 0) s_np = 0;
    for (uint32_t q = threadIdx.x; q < wn; q += CM_BLOCK) { wk[q] = keys[w0 + q]; wp[q] = sorted_pts[w0 + q]; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CM_NB_TILE / CM_BLOCK; ++u) {
        const uint32_t p = base + u * CM_BLOCK + threadIdx.x;
        if (p >= n) continue;
        const uint32_t lp = p - w0;
        const uint32_t key = wk[lp];
        const float4 me = wp[lp];
        const uint32_t jk = key / dx, i = key - jk * dx;
        const uint32_t lo_key = jk * dx + (i ? i - 1 : 0u), hi_key = jk * dx + ((i + 1 < dx) ? i + 1 : dx - 1);
        const uint32_t my_cls = cls ? cls[__float_as_uint(me.w)] : 0u;
        auto test = [&](const float4& pt) {
            const float ex = __fsub_rn(me.x, pt.x), ey = __fsub_rn(me.y, pt.y), ez = __fsub_rn(me.z, pt.z);
            const bool near = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)) < r2;
            return near && (!cls || cls[__float_as_uint(pt.w)] == my_cls);
        };
        uint32_t cnt = 1;                                      // the point itself (distance 0)
        bool incomplete = false;
        for (uint32_t q = lp + 1; cnt <= need; ++q) {
            if (q >= wn) { incomplete = w1 < n; break; }
            if (wk[q] > hi_key) break;
            if (test(wp[q])) ++cnt;
        }
        for (uint32_t q = lp; cnt <= need;) {
            if (q == 0) { incomplete = incomplete || w0 > 0; break; }
            --q;
            if (wk[q] < lo_key) break;
            if (test(wp[q])) ++cnt;
        }
        if (cnt > need) mask[__float_as_uint(me.w)] = 1;
        else {
            const uint32_t at = atomicAdd(&s_np, 1u);
            s_pp[at] = p;
            s_pc[at] = incomplete ? 0x80000000u : cnt;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_np ? atomicAdd(pend_n, s_np) : 0u;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < s_np; q += CM_BLOCK) {
        pend_p[s_base + q] = s_pp[q];
        pend_c[s_base + q] = s_pc[q];
    }
}

// Second launch: the points on the list — the eight rows around each (and its own row again where the first
// launch could not finish it): full waves of hard cases instead of a few slow lanes holding up every wave.
__global__ __launch_bounds__(CM_BLOCK) void k_neighbors(const CmFrameDev* __restrict__ fd,
                                                        const CmFrameState* __restrict__ st,
                                                        const uint32_t* __restrict__ keys_a,
                                                        const uint32_t* __restrict__ keys_b,
                                                        const float4* __restrict__ sorted_pts,
                                                        const uint2* __restrict__ rows,
                                                        unsigned char* __restrict__ mask,
                                                        const unsigned char* __restrict__ cls,
                                                        uint32_t* __restrict__ pend_a,
                                                        uint32_t* __restrict__ pend_b, uint32_t* __restrict__ pend_c,
                                                        uint32_t* __restrict__ pend_n) {
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    uint32_t* __restrict__ pend_p = (st->n_passes & 1u) ? pend_a : pend_b;      // the keys array the sort did not end in
    const uint32_t n_items = *pend_n;
    const uint32_t dx = static_cast<uint32_t>(st->div_b[0]), dy = static_cast<uint32_t>(st->div_b[1]),
                   dz = static_cast<uint32_t>(st->div_b[2]);
    const uint32_t* __restrict__ keys = pick(st, keys_a, keys_b);
    const float r2 = fd->outlier_r2;
    const uint32_t need = fd->outlier_min_nb;            // keep iff k > need, k counts the point itself
    const float cell = 1.0f / fd->inv_cell[0];
    const float fb0 = static_cast<float>(st->min_b[0]), fb1 = static_cast<float>(st->min_b[1]),
                fb2 = static_cast<float>(st->min_b[2]);
    for (uint32_t t = blockIdx.x * CM_BLOCK + threadIdx.x; t < n_items; t += gridDim.x * CM_BLOCK) {
        const uint32_t p = pend_p[t];
        const uint32_t pc = pend_c[t];
        const bool redo_row = (pc >> 31) != 0;                    // the first launch ran out of its LDS window
        const uint32_t key = keys[p];
        const float4 me = sorted_pts[p];
        const uint32_t jk = key / dx, i = key - jk * dx, k = jk / dy, j = jk - k * dy;
        uint32_t cnt = redo_row ? 1u : pc;                        // 1: the point itself (distance 0)
        // cls: neighbours count only inside the point's own class (the ground stage filters every slab's
        // band on its own, like the reference's per-slab outlierRemoval call)
        const uint32_t my_cls = cls ? cls[__float_as_uint(me.w)] : 0u;
        auto test = [&](const float4& pt) {
            const float ex = __fsub_rn(me.x, pt.x), ey = __fsub_rn(me.y, pt.y), ez = __fsub_rn(me.z, pt.z);
            const bool near = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)) < r2;
            return near && (!cls || cls[__float_as_uint(pt.w)] == my_cls);
        };
        // Own row first, outward from the point's own sorted position: the points of its own cell
        // are its immediate neighbours in the sorted order (no search, early exit for most points).
        if (redo_row) {
            const uint32_t lo_key = jk * dx + (i ? i - 1 : 0u), hi_key = jk * dx + ((i + 1 < dx) ? i + 1 : dx - 1);
            for (uint32_t q = p + 1; q < n && cnt <= need; ++q) {
                if (keys[q] > hi_key) break;
                if (test(sorted_pts[q])) ++cnt;
            }
            for (uint32_t q = p; q > 0 && cnt <= need;) {
                --q;
                if (keys[q] < lo_key) break;
                if (test(sorted_pts[q])) ++cnt;
            }
        }
        if (cnt <= need) {
            // Distance from the point to the faces of its own cell, shrunk by 1 % of a cell so fp32
            // rounding of the cell coordinates can only make the pruning below more cautious.
            const float ux = __fsub_rn(__fmul_rn(me.x, fd->inv_cell[0]), __fadd_rn(fb0, static_cast<float>(i)));
            const float uy = __fsub_rn(__fmul_rn(me.y, fd->inv_cell[1]), __fadd_rn(fb1, static_cast<float>(j)));
            const float uz = __fsub_rn(__fmul_rn(me.z, fd->inv_cell[2]), __fadd_rn(fb2, static_cast<float>(k)));
            auto gap = [&](float u, int d) {                      // metres to the neighbouring cell in direction d
                const float g = (d < 0) ? u : (d > 0) ? (1.0f - u) : 0.0f;
                return fmaxf(g - 0.01f, 0.0f) * cell;
            };
            // The eight rows around it: their lookups and binary searches advance together, so the
            // dependent-load chain is one search deep instead of eight. Rows and end cells that lie
            // farther than the radius are skipped.
            uint32_t a[8], b[8], end[8], hi[8];
            uint32_t lo[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int t = o < 4 ? o : o + 1;
                const int dj = (t % 3) - 1, dk = (t / 3) - 1;
                const int jj = static_cast<int>(j) + dj, kk = static_cast<int>(k) + dk;
                const float gy = gap(uy, dj), gz = gap(uz, dk);
                const float g2 = gy * gy + gz * gz;
                const bool ok = kk >= 0 && kk < static_cast<int>(dz) && jj >= 0 && jj < static_cast<int>(dy) && g2 < r2;
                const uint32_t row = ok ? static_cast<uint32_t>(jj) + static_cast<uint32_t>(kk) * dy : 0u;
                const uint2 range = ok ? rows[row] : make_uint2(0u, 0u);
                const float gl = gap(ux, -1), gh = gap(ux, 1);
                const uint32_t il = (i && gl * gl + g2 < r2) ? i - 1 : i;
                const uint32_t ih = (i + 1 < dx && gh * gh + g2 < r2) ? i + 1 : i;
                lo[o] = row * dx + il; hi[o] = row * dx + ih;
                a[o] = range.x; b[o] = range.y; end[o] = range.y;
            }
            for (int step = 0; step < 32; ++step) {
                bool any = false;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    if (a[o] < b[o]) {
                        const uint32_t mid = (a[o] + b[o]) >> 1;
                        if (keys[mid] < lo[o]) a[o] = mid + 1; else b[o] = mid;
                        any = true;
                    }
                }
                if (!any) break;
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                uint32_t q = a[o];
                const uint32_t e = end[o], hk = hi[o];
                while (q + 4 <= e && cnt <= need) {                // four candidates per step: loads overlap
                    const uint32_t k0 = keys[q], k1 = keys[q + 1], k2 = keys[q + 2], k3 = keys[q + 3];
                    const float4 p0 = sorted_pts[q], p1 = sorted_pts[q + 1], p2 = sorted_pts[q + 2], p3 = sorted_pts[q + 3];
                    if (k0 > hk) { q = e; break; }
                    if (test(p0)) ++cnt;
                    if (k1 > hk) { q = e; break; }
                    if (test(p1)) ++cnt;
                    if (k2 > hk) { q = e; break; }
                    if (test(p2)) ++cnt;
                    if (k3 > hk) { q = e; break; }
                    if (test(p3)) ++cnt;
                    q += 4;
                }
                for (; q < e && cnt <= need; ++q) {
                    if (keys[q] > hk) break;
                    if (test(sorted_pts[q])) ++cnt;
                }
            }
        }
        if (cnt > need) mask[__float_as_uint(me.w)] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Merged-cloud materialisation (the reference's fused cloud, :137-142): stable compaction of the
// valid transformed points into 16-byte records. Used for CM_GRID_OVERFLOW (output = input) and
// by parity tests; not on the timed path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_merged_count(const CmFrameDev* __restrict__ fd,
                                                           uint32_t* __restrict__ tile_counts,
                                                           const unsigned char* __restrict__ mask) {
    __shared__ uint32_t lds[CM_WAVES];
    const uint32_t tile = blockIdx.x;
    const uint32_t s = sensor_of_tile(fd, tile);
    const CmSensorDev& sd = fd->s[s];
    const uint32_t first = tile * CM_TILE - sd.base;
    uint32_t cnt = 0;
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * CM_BLOCK + threadIdx.x;
        if (i < sd.n) {
            const Pt p = load_point(sd.data, sd.layout, sd.point_step, sd.off_x, sd.off_y, sd.off_z, sd.off_i, i);
            const float x = xf_row(sd.m[0], sd.m[1], sd.m[2], sd.m[3], p.x, p.y, p.z);
            const float y = xf_row(sd.m[4], sd.m[5], sd.m[6], sd.m[7], p.x, p.y, p.z);
            const float z = xf_row(sd.m[8], sd.m[9], sd.m[10], sd.m[11], p.x, p.y, p.z);
            cnt += (point_valid(x, y, z, fd->crop_enable, fd->crop_min, fd->crop_max) &&
                    (!mask || mask[tile * CM_TILE + r * CM_BLOCK + threadIdx.x])) ? 1u : 0u;
        }
    }
    const uint32_t tot = block_sum_u32(cnt, lds);
    if (threadIdx.x == 0) tile_counts[tile] = tot;
}

__global__ __launch_bounds__(CM_BLOCK) void k_scan_counts(uint32_t* __restrict__ counts, uint32_t n,
                                                          uint32_t* __restrict__ total) {
    __shared__ uint32_t lds[CM_WAVES];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += CM_BLOCK) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = (t < n) ? counts[t] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(v, lds, &tot);
        if (t < n) counts[t] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(CM_BLOCK) void k_merged_write(const CmFrameDev* __restrict__ fd,
                                                           const uint32_t* __restrict__ tile_offs,
                                                           float4* __restrict__ out,
                                                           const unsigned char* __restrict__ mask) {
    __shared__ uint32_t lds[CM_WAVES];
    const uint32_t tile = blockIdx.x;
    const uint32_t s = sensor_of_tile(fd, tile);
    const CmSensorDev& sd = fd->s[s];
    const uint32_t first = tile * CM_TILE - sd.base;
    uint32_t slot = tile_offs[tile];
    for (int r = 0; r < CM_ITEMS; ++r) {            // rounds in order, threads in order: stable
        const uint32_t i = first + r * CM_BLOCK + threadIdx.x;
        bool ok = false;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < sd.n) {
            const Pt p = load_point(sd.data, sd.layout, sd.point_step, sd.off_x, sd.off_y, sd.off_z, sd.off_i, i);
            o.x = xf_row(sd.m[0], sd.m[1], sd.m[2], sd.m[3], p.x, p.y, p.z);
            o.y = xf_row(sd.m[4], sd.m[5], sd.m[6], sd.m[7], p.x, p.y, p.z);
            o.z = xf_row(sd.m[8], sd.m[9], sd.m[10], sd.m[11], p.x, p.y, p.z);
            o.w = p.i;
            ok = point_valid(o.x, o.y, o.z, fd->crop_enable, fd->crop_min, fd->crop_max) &&
                 (!mask || mask[tile * CM_TILE + r * CM_BLOCK + threadIdx.x]);
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(ok ? 1u : 0u, lds, &tot);
        if (ok) out[slot + ex] = o;
        slot += tot;
    }
}

// The image pcl::toROSMsg puts on the wire for pcl::PointXYZI: x, y, z, 1.0f | intensity, 0, 0, 0.
__global__ __launch_bounds__(256) void k_to_pcl32(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float4 v = in[i];
        out[2 * static_cast<size_t>(i)] = make_float4(v.x, v.y, v.z, 1.0f);
        out[2 * static_cast<size_t>(i) + 1] = make_float4(v.w, 0.f, 0.f, 0.f);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from cm_api.cpp)
// ------------------------------------------------------------------------------------------------
#define CM_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)

void cmk_setup(hipStream_t s, const CmFrameDev& f, CmFrameDev* d_frame, CmTileDev* d_tiles) {
    CM_LAUNCH(k_setup, d_tiles ? (f.n_tiles + 255) / 256 + (f.n_tiles == 0) : 1, d_tiles ? 256 : 64, s, f, d_frame, d_tiles);
}
void cmk_minmax(hipStream_t s, const CmFrameDev* fd, float* partials, uint32_t n_blocks, const unsigned char* mask) {
    CM_LAUNCH(k_minmax, n_blocks, CM_BLOCK, s, fd, partials, mask);
}
void cmk_keys(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* keys, uint32_t* hist,
              uint32_t* grp_acc, uint32_t* grp_clear_a, uint32_t* grp_clear_b, uint32_t n_group_words,
              uint32_t n_clear_a_words, uint32_t* seg_groups, uint32_t n_seg_groups, const float* partials,
              uint32_t n_partials, int from_crop, int use_cell, const unsigned char* mask,
              const CmFrameState* st_outlier, uint32_t n_tiles) {
    CM_LAUNCH(k_keys, n_tiles, CM_BLOCK, s, fd, st, keys, hist, grp_acc, grp_clear_a, grp_clear_b,
              n_group_words, n_clear_a_words, seg_groups, n_seg_groups, partials, n_partials, from_crop,
              use_cell, mask, st_outlier);
}
void cmk_outlier_mask(hipStream_t s, const CmFrameDev* fd, const CmFrameState* st, const uint32_t* keys_a,
                      const uint32_t* vals_a, const uint32_t* keys_b, const uint32_t* vals_b, void* sorted_pts,
                      void* rows, unsigned char* mask, uint32_t n_padded, const unsigned char* cls, uint32_t* pend_n,
                      bool already_gathered) {
    const uint32_t blocks = (n_padded + CM_BLOCK * 4 - 1) / (CM_BLOCK * 4);
    if (!already_gathered)
        CM_LAUNCH(k_gather_sorted, blocks, CM_BLOCK, s, fd, st, vals_a, vals_b, reinterpret_cast<float4*>(sorted_pts));
    CM_LAUNCH(k_row_clear, 1024, CM_BLOCK, s, st, reinterpret_cast<uint2*>(rows));
    CM_LAUNCH(k_row_table, blocks, CM_BLOCK, s, st, keys_a, keys_b, reinterpret_cast<uint2*>(rows));
    // pending list: point numbers in the keys buffer the sort did not end in, their counts in a vals buffer
    (void)hipMemsetAsync(pend_n, 0, 4, s);
    uint32_t* pa = const_cast<uint32_t*>(keys_a);
    uint32_t* pb = const_cast<uint32_t*>(keys_b);
    uint32_t* pc = const_cast<uint32_t*>(vals_a);
    const float4* sp = reinterpret_cast<const float4*>(sorted_pts);
    CM_LAUNCH(k_neighbors_row, (n_padded + CM_NB_TILE - 1) / CM_NB_TILE, CM_BLOCK, s, fd, st, keys_a, keys_b, sp, mask, cls,
              pa, pb, pc, pend_n);
    CM_LAUNCH(k_neighbors, (n_padded + CM_BLOCK - 1) / CM_BLOCK, CM_BLOCK, s, fd, st, keys_a, keys_b, sp,
              reinterpret_cast<const uint2*>(rows), mask, cls, pa, pb, pc, pend_n);
}
void cmk_hist(hipStream_t s, const CmFrameState* st, const uint32_t* keys, uint32_t* hist, uint32_t* grp,
              uint32_t pass, uint32_t n_tiles) {
    CM_LAUNCH(k_hist, n_tiles, CM_BLOCK, s, st, keys, hist, grp, pass);
}
void cmk_gscan(hipStream_t s, const CmFrameState* st, uint32_t* grp, uint32_t* totals, uint32_t pass,
               uint32_t n_groups) {
    CM_LAUNCH(k_gscan, 1, 1024, s, st, grp, totals, pass, n_groups);
}
void cmk_scatter(hipStream_t s, CmFrameState* st, const uint32_t* keys_in, const uint32_t* vals_in,
                 uint32_t* keys_out, uint32_t* vals_out, const uint32_t* hist, const uint32_t* grp,
                 const uint32_t* totals, uint32_t pass, uint32_t n_tiles, uint32_t n_groups,
                 uint32_t n_padded, bool lds_rank, uint32_t* tile_kept) {
    if (pass == 0) {
        if (lds_rank)
            CM_LAUNCH((k_scatter<true, true>), n_tiles, CM_BLOCK, s, st, keys_in, vals_in, keys_out, vals_out,
                      hist, grp, totals, pass, n_groups, n_padded, tile_kept);
        else
            CM_LAUNCH((k_scatter<true, false>), n_tiles, CM_BLOCK, s, st, keys_in, vals_in, keys_out, vals_out,
                      hist, grp, totals, pass, n_groups, n_padded, tile_kept);
    } else {
        if (lds_rank)
            CM_LAUNCH((k_scatter<false, true>), n_tiles, CM_BLOCK, s, st, keys_in, vals_in, keys_out, vals_out,
                      hist, grp, totals, pass, n_groups, n_padded, tile_kept);
        else
            CM_LAUNCH((k_scatter<false, false>), n_tiles, CM_BLOCK, s, st, keys_in, vals_in, keys_out, vals_out,
                      hist, grp, totals, pass, n_groups, n_padded, tile_kept);
    }
}
void cmk_probe_lds_order(hipStream_t s, uint32_t* violations, uint32_t rounds) {
    CM_LAUNCH(k_probe_lds_order, 512, CM_BLOCK, s, violations, rounds);
}
void cmk_seg_count(hipStream_t s, CmFrameState* st, const uint32_t* keys_a, const uint32_t* keys_b,
                   uint32_t* counts, uint32_t* group_counts, uint32_t min_pts, uint32_t n_seg_tiles) {
    CM_LAUNCH(k_seg_count, n_seg_tiles, CM_BLOCK, s, st, keys_a, keys_b, counts, group_counts, min_pts);
}
void cmk_seg_reduce(hipStream_t s, int mode, const CmFrameDev* fd, CmFrameState* st, CmFrameState* st_next,
                    uint32_t* host_state, const uint32_t* keys_a, const uint32_t* vals_a,
                    const uint32_t* keys_b, const uint32_t* vals_b, const uint32_t* counts,
                    const uint32_t* group_counts, void* out, uint32_t* out_key, uint32_t* out_cnt,
                    uint32_t n_seg_tiles) {
    float4* o = reinterpret_cast<float4*>(out);
    if (mode == SEG_CENTROIDS)
        CM_LAUNCH(k_seg_reduce<SEG_CENTROIDS>, n_seg_tiles, CM_BLOCK, s, fd, st, st_next, host_state, keys_a, vals_a,
                  keys_b, vals_b, counts, group_counts, o, out_key, out_cnt);
    else if (mode == SEG_PARTIAL)
        CM_LAUNCH(k_seg_reduce<SEG_PARTIAL>, n_seg_tiles, CM_BLOCK, s, fd, st, st_next, host_state, keys_a, vals_a,
                  keys_b, vals_b, counts, group_counts, o, out_key, out_cnt);
    else
        CM_LAUNCH(k_seg_reduce<SEG_TABLES>, n_seg_tiles, CM_BLOCK, s, fd, st, st_next, host_state, keys_a, vals_a,
                  keys_b, vals_b, counts, group_counts, o, out_key, out_cnt);
}
void cmk_table_keys(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* keys, uint32_t* hist,
                    uint32_t* grp_acc, uint32_t* grp_clear_a, uint32_t* grp_clear_b, uint32_t n_group_words,
                    uint32_t n_clear_a_words, uint32_t* seg_groups, uint32_t n_seg_groups, uint32_t key_bits,
                    uint32_t n_tiles) {
    CM_LAUNCH(k_table_keys, n_tiles, CM_BLOCK, s, fd, st, keys, hist, grp_acc, grp_clear_a, grp_clear_b,
              n_group_words, n_clear_a_words, seg_groups, n_seg_groups, key_bits);
}
void cmk_table_finish(hipStream_t s, const void* entries, uint32_t n, uint32_t min_pts, uint32_t* tile_counts,
                      uint32_t* total, void* out, uint32_t* out_key, uint32_t* out_cnt) {
    const uint32_t nt = (n + CM_TILE - 1) / CM_TILE;
    if (nt == 0) return;
    const float4* e = reinterpret_cast<const float4*>(entries);
    CM_LAUNCH(k_table_finish, nt, CM_BLOCK, s, e, n, min_pts, tile_counts, reinterpret_cast<float4*>(out), out_key, out_cnt, 0);
    CM_LAUNCH(k_scan_counts, 1, CM_BLOCK, s, tile_counts, nt, total);
    CM_LAUNCH(k_table_finish, nt, CM_BLOCK, s, e, n, min_pts, tile_counts, reinterpret_cast<float4*>(out), out_key, out_cnt, 1);
}
void cmk_merged(hipStream_t s, const CmFrameDev* fd, uint32_t* tile_counts, uint32_t* total, void* out,
                uint32_t n_tiles, const unsigned char* mask) {
    CM_LAUNCH(k_merged_count, n_tiles, CM_BLOCK, s, fd, tile_counts, mask);
    CM_LAUNCH(k_scan_counts, 1, CM_BLOCK, s, tile_counts, n_tiles, total);
    CM_LAUNCH(k_merged_write, n_tiles, CM_BLOCK, s, fd, tile_counts, reinterpret_cast<float4*>(out), mask);
}
void cmk_to_pcl32(hipStream_t s, const void* in, void* out, uint32_t n) {
    if (n) CM_LAUNCH(k_to_pcl32, (n + 255) / 256, 256, s, reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), n);
}
