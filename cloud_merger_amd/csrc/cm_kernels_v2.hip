// cm_kernels_v2.hip — the "bucket" path of merge -> voxel grid for gfx950: same results as
// cm_kernels.hip, about half its HBM traffic.
//
// cm_kernels.hip sorts (voxel key, point index) pairs by the whole key (4 passes for a 30-bit grid)
// and then gathers every point again through its index (a 128-byte line per 16-byte point). Here the
// 16-byte transformed point itself is what moves, and only as far as it has to:
//   k2_hist0     transform + crop + key of every raw point, counts of the first HIGH digit per tile;
//                min/max records of the cloud (for the next frame's box)        [16 B/pt read]
//   k2_scatter<1> raw points -> records (x,y,z,intensity in the target frame), stable scatter by that
//                digit, next digit of every record as one byte                  [16 B r, 17 B w]
//   k2_hist      counts of the next digit per tile, from the bytes              [1 B/pt read]
//   k2_scatter<0> records -> records by the next digit                          [16 B r, 16 B w]
//   k2_local     the records are now grouped by the high key bits ("buckets"). One workgroup takes the
//                buckets that start in its 4096-record tile, sorts them by the remaining low bits
//                inside LDS, reduces the runs to centroids (PCL's order: ascending voxel index,
//                points of a voxel in stable order) and writes them at the offset it learns from
//                its predecessors' published counts (decoupled look-back)       [16 B r, 16 B/voxel w]
// G = 1..3 global passes depending on the key width; the local finish handles what is left.
//
// The voxel keys need the grid before the first point is read, so this path runs only when the host
// knows a box that contains the cloud: the crop box, or the previous frame's bounds plus a margin
// (cm_api.cpp). Keys are linear indices in that box: the same order as PCL's (x fastest), so kept
// voxels, their order and their point sums are those of pcl::VoxelGrid; a point
// outside a predicted box raises CmFrameState.outside and the host redoes the frame with
// cm_kernels.hip. A bucket that does not fit LDS raises CM_DEV_ERR_BUCKET, same remedy.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cm_common.hpp"
#include "cm_device.h"
#include "cm_kernels.h"

namespace {

template <int WAVES>
__device__ __forceinline__ uint32_t block_excl_scan_w(uint32_t v, uint32_t* lds, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) {
        const uint32_t c = lds[k];
        if (k < w) woff += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

template <int WAVES>
__device__ __forceinline__ uint32_t block_sum_w(uint32_t v, uint32_t* lds) {
    v = wave_sum_u32(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) tot += lds[k];
    __syncthreads();
    return tot;
}

// x / c for a point count c (an integer below 2^16) given rc = RN(1/c): quotient estimate, exact
// residual by FMA, one correction — the correctly rounded x / c that pcl::VoxelGrid's float division
// gives (A.4 step 6) whenever x / c is a normal number; inf / NaN sums pass through unchanged.
__device__ __forceinline__ float div_by_count(float x, float c, float rc) {
    const float q = __fmul_rn(x, rc);
    const float r = __fmaf_rn(-q, c, x);
    const float q2 = __fmaf_rn(r, rc, q);
    return finite_f32(q) ? q2 : q;
}

__device__ __forceinline__ uint32_t sensor_of_slot(const CmFrameDev* __restrict__ fd, uint32_t first) {
    uint32_t s = 0;
    for (uint32_t q = 1; q < fd->n_sensors; ++q) s += (first >= fd->s[q].base) ? 1u : 0u;
    return s;
}

// ------------------------------------------------------------------------------------------------
// k2_hist0: per 4096-slot tile of the padded index space, counts of the first (lowest of the HIGH)
// digit among the valid points; grid set-up recorded by workgroup 0; min/max/count record per tile.
// Also clears what the later kernels of this frame (and the first kernel of the next) accumulate into.
// ------------------------------------------------------------------------------------------------
template <bool PACK>
__global__ __launch_bounds__(CM2_BLOCK, PACK ? 8 : 1) void k2_hist0(const CmFrameDev fv,
                                                      CmFrameDev* __restrict__ fd_dst,
                                                      CmTileDev* __restrict__ tiles_dst, int do_setup,
                                                      CmFrameState* __restrict__ st,
                                                      uint32_t* __restrict__ hist,
                                                      uint32_t* __restrict__ grp_acc,
                                                      uint32_t* __restrict__ grp_clear_a,
                                                      uint32_t* __restrict__ grp_clear_b,
                                                      uint32_t n_group_words, uint32_t n_clear_a_words,
                                                      unsigned long long* __restrict__ tile_state,
                                                      uint32_t n_tile_state,
                                                      float* __restrict__ records,
                                                      int grid_mode, int check_box, uint32_t shift0, uint32_t n_global_passes,
                                                      const unsigned char* __restrict__ mask,
                                                      const CmFrameState* __restrict__ st_outlier, int use_cell,
                                                      float4* __restrict__ compact_out, uint32_t* __restrict__ wave_cnt) {
    // The frame descriptor arrives by value (kernel arguments). do_setup (it changed since the context's last frame:
    // new clouds, new poses): this kernel also leaves it in HBM for the kernels behind it, with the tile table — what
    // a launch of k_setup in front of every frame of a moving stream used to do (6 us on a frame's critical path).
    // compact_out (frames whose crop box drops most points): the surviving records are written here as well, wave w of
    // tile t packing its own in slot order at [t * 4096 + w * 512, ...) and leaving their number in wave_cnt[t * 8 + w];
    // the first scatter then reads these few records instead of every raw point a second time.
    __shared__ uint32_t lh[CM_RADIX];
    __shared__ float s_mm[CM2_WAVES][6];
    __shared__ uint32_t s_cnt[CM2_WAVES];
    __shared__ uint32_t s_out;
    const uint32_t tile = blockIdx.x;
    const CmFrameDev* __restrict__ fd = &fv;
    CmTileDev te;                                         // where this tile's points lie (k_setup's arithmetic)
    {
        const uint32_t first = tile * CM_TILE;
        uint32_t k = 0;
        for (uint32_t q = 1; q < fv.n_sensors; ++q) k += (first >= fv.s[q].base) ? 1u : 0u;
        const CmSensorDev& sd0 = fv.s[k];
        const uint32_t off = first - sd0.base;
        te.data = sd0.data + static_cast<size_t>(off) * sd0.point_step;
        te.n_left = sd0.n > off ? sd0.n - off : 0u;
        te.info = k | (sd0.layout << 8);
    }
    if (do_setup) {
        static_assert(sizeof(CmFrameDev) % 4 == 0 && sizeof(CmFrameDev) / 4 <= CM2_BLOCK, "one word of the descriptor per thread");
        if (threadIdx.x == 0) tiles_dst[tile] = te;
        if (tile == 0 && threadIdx.x < sizeof(CmFrameDev) / 4)
            reinterpret_cast<uint32_t*>(fd_dst)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&fv)[threadIdx.x];
    }
    for (uint32_t k = tile * CM2_BLOCK + threadIdx.x; k < 3 * n_group_words; k += gridDim.x * CM2_BLOCK) grp_clear_b[k] = 0;
    for (uint32_t k = tile * CM2_BLOCK + threadIdx.x; k < n_clear_a_words; k += gridDim.x * CM2_BLOCK) grp_clear_a[k] = 0;
    for (uint32_t k = tile * CM2_BLOCK + threadIdx.x; k < n_tile_state; k += gridDim.x * CM2_BLOCK) tile_state[k] = 0ull;

    if (tile == 0 && threadIdx.x == 0) {                 // the box and its grid, as the host set them up
        const int32_t so = st_outlier ? st_outlier->status : CM_DEV_OK;      // a stage before this one failed: so does the frame
        st->status = (so == CM_DEV_OUTLIER_GRID || so == CM_DEV_ABORTED) ? so : CM_DEV_OK;
        for (int a = 0; a < 3; ++a) {
            st->min_p[a] = grid_mode == 2 ? fd->ext_min[a] : fd->crop_min[a];
            st->max_p[a] = grid_mode == 2 ? fd->ext_max[a] : fd->crop_max[a];
            const int32_t mb = use_cell ? fd->cell_min_b[a] : fd->box_min_b[a], db = use_cell ? fd->cell_div_b[a] : fd->box_div_b[a];
            st->min_b[a] = mb; st->max_b[a] = mb + db - 1;
            st->div_b[a] = db;
        }
        st->key_bits = use_cell ? fd->cell_key_bits : fd->box_key_bits;
        st->n_passes = n_global_passes;
    }
    const BoxGrid b = box_grid_of(fd, use_cell);
    const bool predicted = check_box != 0;        // the box is a prediction: verify every point, record the true bounds

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t slot0 = tile * CM_TILE + w * (64 * CM2_ITEMS) + lane;
    const CmSensorDev& sd = fd->s[te.info & 0xFFu];
    Pt p[CM2_ITEMS];
    load_tile_te<CM2_ITEMS>(te, sd, w * (64 * CM2_ITEMS) + lane, p);
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
    const uint32_t crop = fd->crop_enable;
    float cmn0 = 0.f, cmn1 = 0.f, cmn2 = 0.f, cmx0 = 0.f, cmx1 = 0.f, cmx2 = 0.f;
    if (crop) {
        cmn0 = fd->crop_min[0]; cmn1 = fd->crop_min[1]; cmn2 = fd->crop_min[2];
        cmx0 = fd->crop_max[0]; cmx1 = fd->crop_max[1]; cmx2 = fd->crop_max[2];
    }
    if (threadIdx.x < CM_RADIX) lh[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_out = 0;
    uint32_t mk[CM2_ITEMS];                               // keep-mask bytes of the pre-stages (outlier / ground removal)
    if (mask) {
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) mk[r] = mask[slot0 + r * 64];
    }
    __syncthreads();
    const float inf = __uint_as_float(0x7F800000u);
    float mn0 = inf, mn1 = inf, mn2 = inf, mx0 = -inf, mx1 = -inf, mx2 = -inf;
    uint32_t cnt = 0;
    bool any_out = false;
    const bool all_fields = fd->downsample_all != 0;
    uint32_t wrun = 0;                                    // records this wave has packed so far (wave-uniform)
    // Straight-line code per point: every test is formed as a flag (no short-circuit evaluation — the compiler turns
    // that into a chain of exec-mask branches with the running min/max re-materialised at every join), and only the
    // LDS add sits under a mask.
    float tx[CM2_ITEMS], ty[CM2_ITEMS], tz[CM2_ITEMS];
    uint32_t okm = 0;
#pragma unroll
    for (int r = 0; r < CM2_ITEMS; ++r) {
        const float x = xf_row(m[0], m[1], m[2], m[3], p[r].x, p[r].y, p[r].z);
        const float y = xf_row(m[4], m[5], m[6], m[7], p[r].x, p[r].y, p[r].z);
        const float z = xf_row(m[8], m[9], m[10], m[11], p[r].x, p[r].y, p[r].z);
        tx[r] = x; ty[r] = y; tz[r] = z;
        bool ok = finite_f32(x) & finite_f32(y) & finite_f32(z);
        if (crop) ok = ok & !((x < cmn0) | (x > cmx0) | (y < cmn1) | (y > cmx1) | (z < cmn2) | (z > cmx2));
        if (mask) ok = ok & (mk[r] != 0u);
        bool in;
        const uint32_t key = key_of(b, x, y, z, &in);
        if (predicted) any_out = any_out | (ok & !in);     // a crop box holds every valid point by construction
        else in = true;
        const bool keep = ok & in;
        okm |= ok ? (1u << r) : 0u;
        if (keep) atomicAdd(&lh[(key >> shift0) & (CM_RADIX - 1)], 1u);
        if (PACK) {
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const uint32_t at = wrun + static_cast<uint32_t>(__popcll(bal & ((1ull << lane) - 1ull)));
                compact_out[static_cast<size_t>(tile) * CM_TILE + w * (64 * CM2_ITEMS) + at] =
                    make_float4(x, y, z, use_cell ? __uint_as_float(slot0 + r * 64) : (all_fields ? p[r].i : 0.f));
            }
            wrun += static_cast<uint32_t>(__popcll(bal));
        }
    }
    if (predicted) {
        // The exact bounds of the valid points (pcl::getMinMax3D). Nearly every wave holds valid points only: then the
        // eight values of a lane fold with three-operand min / max, no masking.
        cnt = static_cast<uint32_t>(__builtin_popcount(okm));
        if (__ballot(okm != (1u << CM2_ITEMS) - 1u) == 0ull) {
#pragma unroll
            for (int r = 0; r < CM2_ITEMS; r += 2) {
                mn0 = fminf(fminf(mn0, tx[r]), tx[r + 1]); mx0 = fmaxf(fmaxf(mx0, tx[r]), tx[r + 1]);
                mn1 = fminf(fminf(mn1, ty[r]), ty[r + 1]); mx1 = fmaxf(fmaxf(mx1, ty[r]), ty[r + 1]);
                mn2 = fminf(fminf(mn2, tz[r]), tz[r + 1]); mx2 = fmaxf(fmaxf(mx2, tz[r]), tz[r + 1]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < CM2_ITEMS; ++r) {
                const bool ok = (okm >> r) & 1u;
                mn0 = fminf(mn0, ok ? tx[r] : inf); mx0 = fmaxf(mx0, ok ? tx[r] : -inf);
                mn1 = fminf(mn1, ok ? ty[r] : inf); mx1 = fmaxf(mx1, ok ? ty[r] : -inf);
                mn2 = fminf(mn2, ok ? tz[r] : inf); mx2 = fmaxf(mx2, ok ? tz[r] : -inf);
            }
        }
    }
    if (PACK && lane == 0) wave_cnt[tile * CM2_WAVES + w] = wrun;
    if (predicted) {
        if (any_out) s_out = 1u;
        mn0 = wave_min_f32_l63(mn0); mn1 = wave_min_f32_l63(mn1); mn2 = wave_min_f32_l63(mn2);
        mx0 = wave_max_f32_l63(mx0); mx1 = wave_max_f32_l63(mx1); mx2 = wave_max_f32_l63(mx2);
        cnt = wave_sum_u32(cnt);
        if (lane == 63) {
            s_mm[w][0] = mn0; s_mm[w][1] = mn1; s_mm[w][2] = mn2;
            s_mm[w][3] = mx0; s_mm[w][4] = mx1; s_mm[w][5] = mx2;
            s_cnt[w] = cnt;
        }
    }
    __syncthreads();
    if (threadIdx.x < CM_RADIX) {
        const uint32_t c = lh[threadIdx.x];
        hist[static_cast<size_t>(tile) * CM_RADIX + threadIdx.x] = c;
        if (c) atomicAdd(&grp_acc[static_cast<size_t>(tile / CM_GROUP) * CM_RADIX + threadIdx.x], c);
    } else if (predicted && threadIdx.x < CM_RADIX + 8) {         // record: min xyz, max xyz, count, pad
        const int k = threadIdx.x - CM_RADIX;
        float v = 0.f;
        if (k < 6) {
            v = s_mm[0][k];
            for (int q = 1; q < CM2_WAVES; ++q) v = (k < 3) ? fminf(v, s_mm[q][k]) : fmaxf(v, s_mm[q][k]);
        } else if (k == 6) {
            uint32_t c = 0;
            for (int q = 0; q < CM2_WAVES; ++q) c += s_cnt[q];
            v = __uint_as_float(c);
        }
        records[static_cast<size_t>(tile) * 8 + k] = v;
    }
    if (threadIdx.x == 0 && s_out) st->outside = 1u;
}

// Counts of the next digit per tile of the records, from the digit bytes the last scatter left.
__global__ __launch_bounds__(CM2_BLOCK) void k2_hist(const CmFrameState* __restrict__ st,
                                                     const unsigned char* __restrict__ dig,
                                                     uint32_t* __restrict__ hist,
                                                     uint32_t* __restrict__ grp) {
    __shared__ uint32_t lh[CM_RADIX];
    if (st->status != CM_DEV_OK || st->outside) return;
    const uint32_t n = st->n_valid;
    const uint32_t base = blockIdx.x * CM_TILE;
    if (base >= n) return;
    if (threadIdx.x < CM_RADIX) lh[threadIdx.x] = 0;
    const uint32_t i0 = base + threadIdx.x * 8;
    uint2 v = make_uint2(0u, 0u);
    if (i0 < n) v = *reinterpret_cast<const uint2*>(dig + i0);     // the buffer is padded to whole tiles
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t d = ((q < 4 ? v.x : v.y) >> ((q & 3) * 8)) & 0xFFu;
        if (i0 + q < n) atomicAdd(&lh[d], 1u);
    }
    __syncthreads();
    if (threadIdx.x < CM_RADIX) {
        const uint32_t c = lh[threadIdx.x];
        hist[static_cast<size_t>(blockIdx.x) * CM_RADIX + threadIdx.x] = c;
        if (c) atomicAdd(&grp[static_cast<size_t>(blockIdx.x / CM_GROUP) * CM_RADIX + threadIdx.x], c);
    }
}

// Phase timing of the local finish (scripts/phase_times.py; build with CM_PHASE_TIMING=1): thread 0 of every
// workgroup stores the 100 MHz ticks between phase boundaries. Compiled out of the product build.
#ifdef CM_PHASE_TIMING
__device__ unsigned long long g_phase[4096 * 16];
#define PH_START() long long t0_ = wall_clock64()
#define PH(k) do { if (threadIdx.x == 0) { const long long t1_ = wall_clock64(); g_phase[(blockIdx.x & 4095) * 16 + (k)] = (unsigned long long)(t1_ - t0_); t0_ = t1_; } } while (0)
#else
#define PH_START() do {} while (0)
#define PH(k) do {} while (0)
#endif
// The exact bounds of the frame's valid points from k2_hist0's per-tile records (min xyz, max xyz, count): one workgroup,
// s_f = CM2_WAVES x 8 floats of LDS; writes st->min_p / max_p / n_valid_k0.
__device__ __forceinline__ void fold_bounds(float* s_f, CmFrameState* __restrict__ st, const float* __restrict__ records,
                                            uint32_t n_records) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inf = __uint_as_float(0x7F800000u);
    float v[6] = {inf, inf, inf, -inf, -inf, -inf};
    uint32_t cnt = 0;
    for (uint32_t r = threadIdx.x; r < n_records; r += CM2_BLOCK) {
        const float4 lo = *reinterpret_cast<const float4*>(records + static_cast<size_t>(r) * 8);
        const float4 hi = *reinterpret_cast<const float4*>(records + static_cast<size_t>(r) * 8 + 4);
        v[0] = fminf(v[0], lo.x); v[1] = fminf(v[1], lo.y); v[2] = fminf(v[2], lo.z);
        v[3] = fmaxf(v[3], lo.w); v[4] = fmaxf(v[4], hi.x); v[5] = fmaxf(v[5], hi.y);
        cnt += __float_as_uint(hi.z);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = fminf(v[k], __shfl_xor(v[k], d));
#pragma unroll
        for (int k = 3; k < 6; ++k) v[k] = fmaxf(v[k], __shfl_xor(v[k], d));
        cnt += __shfl_xor(cnt, d);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_f[w * 8 + k] = v[k];
        s_f[w * 8 + 6] = __uint_as_float(cnt);
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int k = threadIdx.x;
        if (k < 6) {
            float r = s_f[k];
            for (int q = 1; q < CM2_WAVES; ++q) r = (k < 3) ? fminf(r, s_f[q * 8 + k]) : fmaxf(r, s_f[q * 8 + k]);
            if (k < 3) st->min_p[k] = r; else st->max_p[k - 3] = r;
        } else {
            uint32_t c = 0;
            for (int q = 0; q < CM2_WAVES; ++q) c += __float_as_uint(s_f[q * 8 + 6]);
            st->n_valid_k0 = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k2_scatter: stable scatter of one 4096-record tile by one 8-bit digit; the record itself moves.
// FIRST: reads the raw sensor points (transform + crop once more, dropping invalid slots: this is
// where the clouds get concatenated) and folds the min/max records (workgroup of tile 0).
// Ranking: returning LDS adds per wave (cm_kernels.hip k_scatter<*, true>; the path is only
// selected after the device probe passed).
// ------------------------------------------------------------------------------------------------
template <bool FIRST, bool BALLOT>
__global__ __launch_bounds__(CM2_BLOCK, 6) void k2_scatter(const CmFrameDev* __restrict__ fd,
                                                           const CmTileDev* __restrict__ tiles,
                                                           CmFrameState* __restrict__ st,
                                                           const float4* __restrict__ rec_in,
                                                           float4* __restrict__ rec_out,
                                                           unsigned char* __restrict__ dig_out,
                                                           const uint32_t* __restrict__ hist,
                                                           const uint32_t* __restrict__ grp,
                                                           const uint32_t* __restrict__ totals,
                                                           uint32_t shift, uint32_t next_shift,
                                                           uint32_t n_groups, uint32_t n_padded,
                                                           const float* __restrict__ records,
                                                           uint32_t n_records, int fold,
                                                           const unsigned char* __restrict__ mask, int use_cell,
                                                           const float4* __restrict__ compact_in,
                                                           const uint32_t* __restrict__ wave_cnt, int debug_swap,
                                                           uint32_t* __restrict__ tile_kept) {
    // debug_swap (the CM_TEST_HOOKS build only, with CM_DEBUG_MISRANK=1): the first and the last record of tile 0's sorted tile change places on
    // their way out — what a mis-ranked pass would look like to the finish, which must notice (CM_DEV_ERR_UNSORTED).
    // 37 KB of LDS and at most 64 VGPRs: four workgroups per CU, so that the 977 tiles of a 4 M-point frame are all
    // resident at once (with three per CU the last 209 tiles ran as a second, nearly empty generation)
    __shared__ float4 srec[CM_TILE / 2];                // staging in two halves
    __shared__ uint32_t whist[CM2_WAVES][CM_RADIX / 2]; // digit counts per wave, two 16-bit counters per word (a wave ranks 512 records)
    uint16_t* sdig = reinterpret_cast<uint16_t*>(&whist[0][0]);   // once the ranks are known: this pass's digit | the next one's << 8 of every staged record
    static_assert(sizeof(uint32_t) * CM2_WAVES * (CM_RADIX / 2) >= sizeof(uint16_t) * (CM_TILE / 2), "digit pairs fit the counters");
    __shared__ uint32_t gofs[CM_RADIX];
    __shared__ uint16_t s_dbase[CM_RADIX];
    __shared__ uint32_t lds[CM2_WAVES];
    __shared__ uint32_t s_tile_valid;
    if (st->status != CM_DEV_OK) return;
    if (st->outside) {
        // A point left the predicted box: the frame is handed back — with the cloud's exact bounds, which k2_hist0's records
        // hold all the same, so that the host can redo it in a box that fits (cm_api.cpp wait_frame) instead of measuring again.
        if (FIRST && fold && blockIdx.x == 0) fold_bounds(reinterpret_cast<float*>(whist), st, records, n_records);
        return;
    }
    PH_START();
    uint32_t tile = blockIdx.x;
    {
        const uint32_t per = gridDim.x / 8;              // contiguous tile range per XCD (see k_scatter)
        if (blockIdx.x < per * 8) tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    if (!FIRST && tile * CM_TILE >= st->n_valid) return;    // (the first pass recorded how many records are left: see k_scatter)
    const BoxGrid b = box_grid_of(fd, use_cell);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t first = tile * CM_TILE + w * (64 * CM2_ITEMS) + lane;

    // Records of digit d written before this tile's: the totals of the groups before its group (threads 0-255, one
    // digit each) and the counts of the tiles before it in its group (threads 256-511), every load of a thread in
    // flight at once and asked for before the tile's records, so that the latencies overlap (summed one batch after
    // the other behind the finished record loads they were 39 % of a tile's time; asking for the records first and
    // for these behind them was slower still: 94 spilled registers in the first pass).
    const uint32_t grp_id = tile / CM_GROUP;
    uint32_t before = 0, my_total = 0;
    {
        const uint32_t d = threadIdx.x & (CM_RADIX - 1);
        if (threadIdx.x < CM_RADIX) {
            if (totals) {
                my_total = totals[d];
                before = grp[static_cast<size_t>(grp_id) * CM_RADIX + d];
            } else {
                for (uint32_t g = 0; g < n_groups; g += 16) {
                    uint32_t v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = (g + q < n_groups) ? grp[static_cast<size_t>(g + q) * CM_RADIX + d] : 0u;
#pragma unroll
                    for (int q = 0; q < 16; ++q) { my_total += v[q]; before += (g + q < grp_id) ? v[q] : 0u; }
                }
            }
        } else {
            for (uint32_t t = grp_id * CM_GROUP; t < tile; t += 16) {
                uint32_t v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = (t + q < tile) ? hist[static_cast<size_t>(t + q) * CM_RADIX + d] : 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) before += v[q];
            }
            gofs[d] = before;                                  // (picked up by thread d behind the barriers of the scans below)
        }
    }

    float4 rec[CM2_ITEMS];
    uint32_t lp[CM2_ITEMS];                               // digit | next pass's digit << 8, then | rank << 16, then position in the sorted tile | digits << 16
    uint32_t vmask = 0;
    Pt p[CM2_ITEMS];                                      // (first pass: the raw points)
    uint32_t sidx = 0;
    if (FIRST && compact_in) {
        // k2_hist0 left this tile's surviving records packed per wave, in slot order: same (wave, round, lane) layout
        // as the raw read below, so the ranking stays stable; nothing to transform or to test again.
        uint32_t tile_cnt = 0;
#pragma unroll
        for (int q = 0; q < CM2_WAVES; ++q) tile_cnt += wave_cnt[tile * CM2_WAVES + q];
        if (tile_cnt == 0 && tile != 0) {                  // nothing of this tile survived (tile 0 also records the frame's totals)
            if (tile_kept && threadIdx.x == 0) tile_kept[tile] = 0;
            return;
        }
        const uint32_t cw = wave_cnt[tile * CM2_WAVES + w];
        const float4* __restrict__ src = compact_in + static_cast<size_t>(tile) * CM_TILE + w * (64 * CM2_ITEMS);
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) {
            const uint32_t i = r * 64 + lane;
            rec[r] = src[i < cw ? i : 0u];                 // (unconditional load; slot 0 of the wave's range is always mapped)
            const uint32_t k = key_of(b, rec[r]);
            lp[r] = ((k >> shift) & (CM_RADIX - 1)) | (next_shift < 32u ? ((k >> next_shift) & 0xFFu) << 8 : 0u);
            if (i < cw) vmask |= 1u << r;
        }
    } else if (FIRST) {
        const CmTileDev te = tiles[tile];
        // (the same in every lane; said so, the sensor's matrix and the crop box below are scalar loads, not twenty VGPRs)
        sidx = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(te.info & 0xFFu)));
        // (non-temporal: this is the last anybody reads of the raw clouds — with a new cloud every frame they would only push
        // the record buffers out of the Infinity Cache: moving stream, one frame alone 0.144 -> 0.139 ms; nothing in flight)
        load_tile_te<CM2_ITEMS, true>(te, fd->s[sidx], w * (64 * CM2_ITEMS) + lane, p);
    } else {
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) {
            const uint32_t i = first + r * 64;
            rec[r] = (i < n_padded) ? rec_in[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // The last pass needs no index at all: its digit is the byte the pass before left beside every record (dig_out is
        // only written by passes that have a successor, so nobody writes the array while this pass reads it).
        if (next_shift >= 32u) {
#pragma unroll
            for (int r = 0; r < CM2_ITEMS; ++r) {
                const uint32_t i = first + r * 64;
                lp[r] = (i < n_padded) ? dig_out[i] : 0u;
            }
        }
    }

    if (FIRST && !compact_in) {
        const CmSensorDev& sd = fd->s[sidx];
        float m[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
        const uint32_t crop = fd->crop_enable;
        float cmn0 = 0.f, cmn1 = 0.f, cmn2 = 0.f, cmx0 = 0.f, cmx1 = 0.f, cmx2 = 0.f;
        if (crop) {
            cmn0 = fd->crop_min[0]; cmn1 = fd->crop_min[1]; cmn2 = fd->crop_min[2];
            cmx0 = fd->crop_max[0]; cmx1 = fd->crop_max[1]; cmx2 = fd->crop_max[2];
        }
        const bool all_fields = fd->downsample_all != 0;
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) {
            rec[r].x = xf_row(m[0], m[1], m[2], m[3], p[r].x, p[r].y, p[r].z);
            rec[r].y = xf_row(m[4], m[5], m[6], m[7], p[r].x, p[r].y, p[r].z);
            rec[r].z = xf_row(m[8], m[9], m[10], m[11], p[r].x, p[r].y, p[r].z);
            rec[r].w = use_cell ? __uint_as_float(first + r * 64) : (all_fields ? p[r].i : 0.f);   // outlier stage: the point's padded index rides along
            bool ok = finite_f32(rec[r].x) & finite_f32(rec[r].y) & finite_f32(rec[r].z);
            if (crop) ok = ok & !((rec[r].x < cmn0) | (rec[r].x > cmx0) | (rec[r].y < cmn1) | (rec[r].y > cmx1) |
                                  (rec[r].z < cmn2) | (rec[r].z > cmx2));
            if (mask) ok = ok & (mask[first + r * 64] != 0);        // (frames behind a pre-stage only)
            bool in;
            const uint32_t k = key_of(b, rec[r].x, rec[r].y, rec[r].z, &in);
            ok = ok & in;
            lp[r] = ok ? ((k >> shift) & (CM_RADIX - 1)) | (next_shift < 32u ? ((k >> next_shift) & 0xFFu) << 8 : 0u) : 0u;
            vmask |= ok ? (1u << r) : 0u;
        }
    }

    PH((FIRST ? 0 : 8) + 0);
    uint32_t gtot;
    const uint32_t gbase = block_excl_scan_w<CM2_WAVES>(my_total, lds, &gtot);
    if (FIRST && tile == 0 && threadIdx.x == 0) st->n_valid = gtot;
    const uint32_t n = FIRST ? n_padded : gtot;
    if (tile * CM_TILE >= n) return;                     // uniform (n is the same in every workgroup)
    PH((FIRST ? 0 : 8) + 1);

    if (!FIRST) {
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) {
            if (next_shift < 32u) {                              // (uniform; the last pass has its digits already)
                const uint32_t k = key_of(b, rec[r]);
                lp[r] = ((k >> shift) & (CM_RADIX - 1)) | (((k >> next_shift) & 0xFFu) << 8);
            }
            if (first + r * 64 < n) vmask |= 1u << r;
        }
    }
    for (uint32_t q = threadIdx.x; q < CM2_WAVES * CM_RADIX / 2; q += CM2_BLOCK) (&whist[0][0])[q] = 0;
    __syncthreads();
    if (BALLOT) {                                          // (ranking by ballots: cm_common.hpp wave_rank_ballot)
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) {
            lp[r] |= wave_rank_ballot(whist[w], lp[r] & 0xFFu, 8u, (vmask >> r) & 1u, lane) << 16;
            asm volatile("" : "+v"(lp[r]));
        }
    } else {
    // The returning adds go out one behind the other (a slot without a record adds nothing), the ranks are taken once all
    // are back: under a branch each the wave waited for every single one.
#pragma unroll
    for (int r0 = 0; r0 < CM2_ITEMS; r0 += 4) {               // (four at a time: eight returns in flight cost registers the records need)
        uint32_t got[4];
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) {
            // (a slot without a record adds nothing — to a word of its own: the LDS serialises same-address adds of a
            // wave, and in a frame that a crop box empties nearly every lane would meet on one counter)
            const uint32_t sh = (lp[r] & 1u) * 16u;
            const bool has = vmask >> r & 1u;
            got[r - r0] = atomicAdd(&whist[w][has ? (lp[r] & 0xFFu) >> 1 : static_cast<uint32_t>(lane)], (has ? 1u : 0u) << sh);
        }
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) {
            lp[r] |= ((got[r - r0] >> ((lp[r] & 1u) * 16u)) & 0xFFFFu) << 16;
            asm volatile("" : "+v"(lp[r]));                 // (formed here, not where it is next used: the raw returns would have to stay alive)
        }
    }
    }
    __syncthreads();
    PH((FIRST ? 0 : 8) + 2);
    {
        // thread t < 128: digits 2t and 2t+1 — totals over the waves, exclusive prefix over the digits, then every wave's
        // counter becomes the first sorted position of its records of that digit
        uint32_t cw[CM2_WAVES], t0 = 0, t1 = 0;
        if (threadIdx.x < CM_RADIX / 2) {
#pragma unroll
            for (int q = 0; q < CM2_WAVES; ++q) { cw[q] = whist[q][threadIdx.x]; t0 += cw[q] & 0xFFFFu; t1 += cw[q] >> 16; }
        }
        uint32_t tile_valid;
        const uint32_t db = block_excl_scan_w<CM2_WAVES>(t0 + t1, lds, &tile_valid);
        if (threadIdx.x < CM_RADIX / 2) {
            uint32_t r0 = db, r1 = db + t0;
            s_dbase[2 * threadIdx.x] = static_cast<uint16_t>(r0);
            s_dbase[2 * threadIdx.x + 1] = static_cast<uint16_t>(r1);
#pragma unroll
            for (int q = 0; q < CM2_WAVES; ++q) {
                whist[q][threadIdx.x] = r0 | (r1 << 16);
                r0 += cw[q] & 0xFFFFu; r1 += cw[q] >> 16;
            }
            if (threadIdx.x == 0) s_tile_valid = tile_valid;
        }
    }
    __syncthreads();
    PH((FIRST ? 0 : 8) + 3);
    if (threadIdx.x < CM_RADIX)                                  // (read again behind the barrier of the first staging round)
        gofs[threadIdx.x] = gbase + before + gofs[threadIdx.x] - s_dbase[threadIdx.x];   // (+ the part threads 256-511 summed)
    const uint32_t tile_valid = s_tile_valid;
    if (FIRST && tile_kept && threadIdx.x == 0) tile_kept[tile] = tile_valid;   // cm_get_frame_stats: points per sensor that entered the grid
#pragma unroll
    for (int r = 0; r < CM2_ITEMS; ++r) {
        const uint32_t digit = lp[r] & 0xFFu;
        const uint32_t wv = whist[w][digit >> 1];               // (read whether the slot holds a record or not: no branch)
        const uint32_t at = ((wv >> ((digit & 1u) * 16u)) & 0xFFFFu) + (lp[r] >> 16);
        lp[r] = ((vmask >> r & 1u) ? at : 0xFFFFu) | (lp[r] << 16);
        asm volatile("" : "+v"(lp[r]));
    }
    // Two rounds through the staging buffer: sorted positions [0, 2048), then [2048, 4096). Beside every record goes its
    // pair of digits (where the counters were: they are read out by now), so that whoever carries the record to HBM
    // neither forms its index a second time nor needs the grid.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t lo = h * (CM_TILE / 2);
        if (h == 1) {
            PH((FIRST ? 0 : 8) + 4);
            if (tile_valid <= lo) break;                   // uniform
        }
        __syncthreads();                                   // (round 0: the last reads of the counters; round 1: of the staged records)
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r)
            if ((lp[r] & 0xFFFFu) - lo < CM_TILE / 2) {
                srec[(lp[r] & 0xFFFFu) - lo] = rec[r];
                sdig[(lp[r] & 0xFFFFu) - lo] = static_cast<uint16_t>(lp[r] >> 16);
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CM2_ITEMS / 2; ++j) {
            const uint32_t t = lo + j * CM2_BLOCK + threadIdx.x;
            if (t < tile_valid) {
                const float4 r4 = srec[t - lo];
                const uint32_t dd = sdig[t - lo];
                uint32_t pos = gofs[dd & 0xFFu] + t;
#ifdef CM_TEST_HOOKS
                if (debug_swap && tile == 0 && h == 0 && tile_valid > 1) {
                    const uint32_t last = min(tile_valid, static_cast<uint32_t>(CM_TILE / 2)) - 1u;   // (both in the first staging round)
                    if (t == 0 || t == last) {
                        const uint32_t to = t == 0 ? last : 0u;
                        pos = gofs[sdig[to] & 0xFFu] + to;
                    }
                }
#endif
                rec_out[pos] = r4;
                if (next_shift < 32u) dig_out[pos] = static_cast<unsigned char>(dd >> 8);
            }
        }
    }

    PH((FIRST ? 0 : 8) + 5);
    // The exact bounds of the cloud (pcl::getMinMax3D) for the result and for the next frame's box.
    if (FIRST && fold && tile == 0) {
        __syncthreads();
        fold_bounds(reinterpret_cast<float*>(whist), st, records, n_records);
    }
}

// ------------------------------------------------------------------------------------------------
// k2_scatter_sparse: the first scatter of a frame whose crop box drops most of its points, from the records k2_hist0<PACK>
// left packed per wave. k2_scatter<true> gives every 4096-slot tile a workgroup of its own — offsets prologue, ranking,
// two staging rounds — for what may be a hundred survivors (cfg3: 3906 workgroups, 38 us for 10 MB of records). Here a
// workgroup takes EIGHT consecutive tiles: one prologue for all of them (the offsets of tile T + k are those of tile T
// plus the counts of the tiles between), then wave k carries tile T + k's survivors — chunk by chunk in (wave, slot)
// order, ranked by returning LDS adds on counters of its own, written straight from registers (few records: no
// staging) — without another workgroup barrier. Same positions as k2_scatter<true>, record for record.
// ------------------------------------------------------------------------------------------------
template <bool BALLOT>
__global__ __launch_bounds__(CM2_BLOCK) void k2_scatter_sparse(const CmFrameDev* __restrict__ fd, CmFrameState* __restrict__ st,
                                                               const float4* __restrict__ compact_in,
                                                               const uint32_t* __restrict__ wave_cnt,
                                                               float4* __restrict__ rec_out, unsigned char* __restrict__ dig_out,
                                                               const uint32_t* __restrict__ hist, const uint32_t* __restrict__ grp,
                                                               const uint32_t* __restrict__ totals, uint32_t shift,
                                                               uint32_t next_shift, uint32_t n_groups, uint32_t n_tiles,
                                                               int use_cell, uint32_t* __restrict__ tile_kept) {
    static_assert(CM2_WAVES == 8 && CM_GROUP % CM2_WAVES == 0, "eight tiles of one group per workgroup");
    __shared__ uint32_t offs[CM2_WAVES][CM_RADIX];        // first position of digit d among tile T + k's records
    __shared__ uint32_t wcnt[CM2_WAVES][CM_RADIX / 2];    // a wave's digit counters, two 16-bit counters per word
    __shared__ uint32_t part[CM_RADIX];
    __shared__ uint32_t lds[CM2_WAVES];
    if (st->status != CM_DEV_OK || st->outside) return;
    const uint32_t T = blockIdx.x * CM2_WAVES;
    const BoxGrid b = box_grid_of(fd, use_cell);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t grp_id = T / CM_GROUP;
    const uint32_t d = threadIdx.x & (CM_RADIX - 1);
    uint32_t before = 0, my_total = 0;
    if (threadIdx.x < CM_RADIX) {                          // records of digit d in the groups before this one (see k2_scatter)
        if (totals) {
            my_total = totals[d];
            before = grp[static_cast<size_t>(grp_id) * CM_RADIX + d];
        } else {
            for (uint32_t g = 0; g < n_groups; g += 16) {
                uint32_t v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = (g + q < n_groups) ? grp[static_cast<size_t>(g + q) * CM_RADIX + d] : 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) { my_total += v[q]; before += (g + q < grp_id) ? v[q] : 0u; }
            }
        }
    } else {                                               // ... and in the tiles of this group before tile T
        for (uint32_t t = grp_id * CM_GROUP; t < T; t += 16) {
            uint32_t v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = (t + q < T) ? hist[static_cast<size_t>(t + q) * CM_RADIX + d] : 0u;
#pragma unroll
            for (int q = 0; q < 16; ++q) before += v[q];
        }
        part[d] = before;
    }
    (&wcnt[0][0])[threadIdx.x] = 0;                        // (CM2_WAVES * CM_RADIX / 2 == 2 * CM2_BLOCK words)
    (&wcnt[0][0])[CM2_BLOCK + threadIdx.x] = 0;
    static_assert(CM2_WAVES * (CM_RADIX / 2) == 2 * CM2_BLOCK, "counter words per thread");
    uint32_t gtot;
    const uint32_t gbase = block_excl_scan_w<CM2_WAVES>(my_total, lds, &gtot);     // (its barriers publish part[])
    if (blockIdx.x == 0 && threadIdx.x == 0) st->n_valid = gtot;
    if (threadIdx.x < CM_RADIX) {
        uint32_t hv[CM2_WAVES];
#pragma unroll
        for (int k = 0; k < CM2_WAVES; ++k) hv[k] = (T + k < n_tiles) ? hist[static_cast<size_t>(T + k) * CM_RADIX + d] : 0u;
        uint32_t run = gbase + before + part[d];
#pragma unroll
        for (int k = 0; k < CM2_WAVES; ++k) { offs[k][d] = run; run += hv[k]; }
    }
    __syncthreads();
    const uint32_t tile = T + w;
    if (tile >= n_tiles) return;
    uint32_t cw[CM2_WAVES], total = 0, most = 0;
#pragma unroll
    for (int q = 0; q < CM2_WAVES; ++q) {
        cw[q] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(wave_cnt[tile * CM2_WAVES + q])));
        total += cw[q];
        most = max(most, cw[q]);
    }
    const float4* __restrict__ src = compact_in + static_cast<size_t>(tile) * CM_TILE;
    // one record of one chunk: ranked among the tile's records of its digit by a returning LDS add (the chunks the eight
    // waves of k2_hist0 packed are taken in their order, a chunk's records 64 at a time: the order k2_scatter ranks in)
    auto place = [&](const float4& r4, bool has) {
        const uint32_t k = key_of(b, r4);
        const uint32_t dg = (k >> shift) & (CM_RADIX - 1), sh = (dg & 1u) * 16u;
        // (a lane without a record adds nothing, to a word of its own: see k2_scatter)
        const uint32_t rank = BALLOT ? wave_rank_ballot(wcnt[w], dg, 8u, has, lane)
                                     : (atomicAdd(&wcnt[w][has ? dg >> 1 : static_cast<uint32_t>(lane)], (has ? 1u : 0u) << sh) >> sh) & 0xFFFFu;
        if (has) {
            const uint32_t pos = offs[w][dg] + rank;
            rec_out[pos] = r4;
            if (next_shift < 32u) dig_out[pos] = static_cast<unsigned char>((k >> next_shift) & 0xFFu);
        }
    };
    if (most <= 64u) {                                     // the usual case: every chunk is one load — all eight in flight at once
        float4 r4[CM2_WAVES];
#pragma unroll
        for (int q = 0; q < CM2_WAVES; ++q)
            r4[q] = src[q * (64 * CM2_ITEMS) + (static_cast<uint32_t>(lane) < cw[q] ? lane : 0)];
#pragma unroll
        for (int q = 0; q < CM2_WAVES; ++q) place(r4[q], static_cast<uint32_t>(lane) < cw[q]);
    } else {
        for (int q = 0; q < CM2_WAVES; ++q)
            for (uint32_t i0 = 0; i0 < cw[q]; i0 += 64) {
                const uint32_t i = i0 + lane;
                place(src[q * (64 * CM2_ITEMS) + (i < cw[q] ? i : 0u)], i < cw[q]);
            }
    }
    if (tile_kept && lane == 0) tile_kept[tile] = total;
}

// ------------------------------------------------------------------------------------------------
// k2_local: the finish. `rec` is grouped by H = key >> low_bits (ascending). Workgroup t owns the
// buckets (runs of equal H) that START inside records [t*4096, (t+1)*4096): it skips the head of
// its tile that continues the previous workgroup's last bucket and reads past its end until its
// own last bucket closes. A voxel's points share H, so no voxel is ever split between workgroups.
// ------------------------------------------------------------------------------------------------
#define CM2_FLAG_AGG (1ull << 32)
#define CM2_FLAG_PREFIX (2ull << 32)

template <int LT, int LCAP, int LBLOCK, bool WRITEBACK>
__global__ __launch_bounds__(LBLOCK, LBLOCK <= 512 ? 6 : 4) void k2_local(const CmFrameDev* __restrict__ fd,
                                                       CmFrameState* __restrict__ st,
                                                       CmFrameState* __restrict__ st_next,
                                                       uint32_t* __restrict__ host_state,
                                                       const float4* __restrict__ rec,
                                                       unsigned long long* __restrict__ tile_state,
                                                       uint32_t* __restrict__ ticket,
                                                       float4* __restrict__ out,
                                                       uint32_t* __restrict__ out_key,
                                                       uint32_t* __restrict__ out_cnt,
                                                       float4* __restrict__ partial_out,
                                                       uint32_t low_bits,
                                                       uint32_t* __restrict__ keys_sorted, float4* __restrict__ recs_sorted,
                                                       int use_cell) {
    // WRITEBACK (outlier stage): no voxels — the tile's records and keys go back to HBM in key order; since the
    // tiles' owned ranges partition the array in order, that leaves the whole array sorted.
    constexpr int LWAVES = LBLOCK / 64, LITEMS = (LCAP + LBLOCK - 1) / LBLOCK, EXT0 = LBLOCK < 256 ? LBLOCK : 256;
    constexpr int BINS = 1024, HWORDS = BINS / 2;          // two 16-bit counters per LDS word
    static_assert(LT == 4 * LBLOCK && LCAP <= 0xFFFF && HWORDS <= LBLOCK && LWAVES * HWORDS * 2 >= LCAP, "tile geometry");
    __shared__ uint32_t sk[LCAP];                      // key of every slot
    __shared__ uint16_t si[LCAP];                      // slots in sorted order
    __shared__ uint32_t whist[LWAVES][HWORDS];         // digit counts per wave; after the sort: head positions
    __shared__ uint16_t dbase[BINS];                   // first sorted position of every digit
    __shared__ uint32_t lds[LWAVES];
    __shared__ uint32_t s_a, s_keyprev, s_off;
    uint16_t* hpos = reinterpret_cast<uint16_t*>(&whist[0][0]);

    PH_START();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (blockIdx.x == 0 && st_next && threadIdx.x < sizeof(CmFrameState) / 4)
        reinterpret_cast<uint32_t*>(st_next)[threadIdx.x] = 0;
    if (st->status != CM_DEV_OK || st->outside) {
        if (blockIdx.x == 0) report_state(host_state, st, st->status, 0u, true);
        return;
    }
    const uint32_t n = st->n_valid;
    if (n == 0) {
        if (blockIdx.x == 0) report_state(host_state, st, CM_DEV_EMPTY, 0u, true);
        return;
    }
    const uint32_t n_lt = (n + LT - 1) / LT;
    // The grid is sized for the padded frame; only n_lt workgroups are needed (a crop may leave far fewer
    // records than slots): the others leave without drawing a ticket.
    if (blockIdx.x >= n_lt) return;
    // Which tile this workgroup works on is decided when it starts running, by a ticket: it will wait
    // (look-back below) only for tiles with smaller tickets, i.e. for workgroups that are already running —
    // whatever order and placement the hardware dispatches workgroups in (several XCDs, other streams, other
    // processes on the same GPU). The n_lt workgroups that stay draw the tickets 0 .. n_lt-1 between them.
    if (threadIdx.x == 0) s_a = WRITEBACK ? blockIdx.x : atomicAdd(ticket, 1u);     // nobody waits for anybody when writing back
    const BoxGrid b0 = box_grid_of(fd, use_cell);
    __syncthreads();
    const uint32_t tile = s_a;
    __syncthreads();
    const BoxGrid b = b0;
    PH(0);
    const uint32_t L = low_bits;
    // partial_out: this GPU's share of a fused cloud — per-voxel sums and counts, no threshold, no division (§6)
    const uint32_t min_pts = (!partial_out && fd->min_pts > 1) ? fd->min_pts : 1u;

    // ---- load: the nominal tile, the key before it, and the first records after it
    const uint32_t base = tile * LT;
    const uint32_t nom = min(static_cast<uint32_t>(LT), n - base);
    {
        float4 r4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t q = r * LBLOCK + threadIdx.x;
            r4[r] = (q < nom) ? rec[base + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const uint32_t j0 = base + LT + threadIdx.x;
        const bool has_e = threadIdx.x < EXT0 && nom == LT && j0 < n;
        const float4 e4 = has_e ? rec[j0] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x == 0 && base > 0) pv = rec[base - 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t q = r * LBLOCK + threadIdx.x;
            if (q < nom) sk[q] = key_of(b, r4[r]);
        }
        if (has_e) sk[LT + threadIdx.x] = key_of(b, e4);
        if (threadIdx.x == 0) { s_keyprev = base > 0 ? key_of(b, pv) : 0u; s_a = 0xFFFFFFFFu; }
    }
    __syncthreads();
    PH(1);

    // ---- a: first bucket start in the nominal tile; the bucket number must not decrease anywhere (free check of the
    // global passes' ranking, as k3_local does: CM_DEV_ERR_UNSORTED hands the frame back)
    bool bad_order = false;
    {
        uint32_t best = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 3; r >= 0; --r) {
            const uint32_t q = r * LBLOCK + threadIdx.x;
            if (q < nom) {
                const uint32_t kp = (q == 0) ? s_keyprev : sk[q - 1];
                if ((base + q == 0) || ((sk[q] >> L) != (kp >> L))) best = q;
                bad_order = bad_order || (base + q > 0 && (sk[q] >> L) < (kp >> L));
            }
        }
        if (best != 0xFFFFFFFFu) atomicMin(&s_a, best);
    }
    if (bad_order) {
        host_state[offsetof(CmFrameState, err) / 4] = CM_DEV_ERR_UNSORTED;
        if (WRITEBACK) st->status = CM_DEV_ABORTED;          // (nobody may index with what this stage leaves behind)
    }
    const uint32_t h_last = sk[nom - 1] >> L;
    // ---- tail of the last bucket past the nominal end (a prefix of what follows: H is ascending)
    const bool m0 = threadIdx.x < EXT0 && nom == LT && (base + LT + threadIdx.x) < n &&
                    (sk[LT + threadIdx.x] >> L) == h_last;
    uint32_t ext = __syncthreads_count(m0);               // also orders the atomicMin above
    const uint32_t a = s_a;
    bool too_big = false;
    if (ext == EXT0 && a != 0xFFFFFFFFu) {
        for (uint32_t off = EXT0;; off += LBLOCK) {
            const uint32_t j = base + LT + off + threadIdx.x;
            const uint32_t pos = LT + off + threadIdx.x;
            bool mm = false;
            if (j < n) {
                const float4 r4 = rec[j];
                const uint32_t k = key_of(b, r4);
                mm = (k >> L) == h_last;
                if (mm && pos < LCAP) sk[pos] = k;
            }
            const uint32_t c = __syncthreads_count(mm);
            ext += c;
            if (LT + ext > LCAP) { too_big = true; break; }
            if (c < LBLOCK) break;
        }
    }
    const uint32_t m = (a == 0xFFFFFFFFu || too_big) ? 0u : nom + ext - a;
    if (too_big && threadIdx.x == 0) {
        host_state[offsetof(CmFrameState, err) / 4] = WRITEBACK ? CM_DEV_ERR_BUCKET_PRE : CM_DEV_ERR_BUCKET;
        // This tile's part of the sorted arrays stays unwritten: the row table and the neighbour search behind this
        // kernel would index memory with whatever the buffers held before. They all leave on a status other than OK.
        if (WRITEBACK) st->status = CM_DEV_ABORTED;
    }

    PH(2);
    // ---- sort the owned slots [a, a+m) by key: LSD over the bits in which the keys of this tile can
    // differ, up to 10 per pass, stable. Only the slot numbers move (si); a pass reads its digit
    // through the slot. Ranking: returning LDS adds on per-wave counters (lane order, see k_scatter).
    if (m) {
        const uint32_t kbase = (sk[a] >> L) << L;
        const unsigned long long span = static_cast<unsigned long long>(h_last - (sk[a] >> L) + 1u) << L;
        const uint32_t nb = span > 1ull ? 64u - static_cast<uint32_t>(__builtin_clzll(span - 1ull)) : 0u;
        const uint32_t npass = nb ? (nb + 9u) / 10u : 1u;
        const uint32_t width = nb ? (nb + npass - 1u) / npass : 0u;
        const uint32_t dmask = (1u << width) - 1u;
        // (per pass: only the counter words its digit can reach are cleared and scanned)

        // Wave w ranks the contiguous chunk [w * 64 * rounds, (w + 1) * 64 * rounds) of the owned records: all
        // waves equally busy whatever m is (LITEMS only bounds the capacity).
        const uint32_t rounds = (m + LBLOCK - 1) / LBLOCK;
        for (uint32_t p = 0; p < npass; ++p) {
            const uint32_t wp = nb > p * width ? min(width, nb - p * width) : 0u;     // bits this pass really sorts
            const uint32_t words = wp ? ((1u << wp) + 1u) / 2u : 1u;
            uint32_t dg[LITEMS], rk[LITEMS];
            uint16_t ei[LITEMS];
#pragma unroll
            for (int r = 0; r < LITEMS; ++r) {
                const uint32_t e = w * (64 * rounds) + r * 64 + lane;
                ei[r] = 0; dg[r] = 0;
                if (r < rounds && e < m) {
                    ei[r] = (p == 0) ? static_cast<uint16_t>(a + e) : si[e];
                    dg[r] = ((sk[ei[r]] - kbase) >> (p * width)) & dmask;
                }
            }
#pragma unroll
            for (int q = 0; q < LWAVES * HWORDS / LBLOCK; ++q) {
                const uint32_t flat = q * LBLOCK + threadIdx.x;       // row = flat / HWORDS, column = flat % HWORDS
                if ((flat & (HWORDS - 1)) < words) (&whist[0][0])[flat] = 0;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < LITEMS; ++r) {
                const uint32_t e = w * (64 * rounds) + r * 64 + lane;
                const uint32_t sh = (dg[r] & 1u) * 16u;
                rk[r] = 0;
                if (r < rounds && e < m) rk[r] = (atomicAdd(&whist[w][dg[r] >> 1], 1u << sh) >> sh) & 0xFFFFu;
            }
            __syncthreads();
            // thread t < HWORDS: digits 2t and 2t+1 — exclusive prefix over the waves, then over the digits
            uint32_t t0 = 0, t1 = 0;
            if (threadIdx.x < words) {
#pragma unroll
                for (int q = 0; q < LWAVES; ++q) {
                    const uint32_t c = whist[q][threadIdx.x];
                    whist[q][threadIdx.x] = t0 | (t1 << 16);
                    t0 += c & 0xFFFFu; t1 += c >> 16;
                }
            }
            uint32_t all;
            const uint32_t db = block_excl_scan_w<LWAVES>(t0 + t1, lds, &all);
            if (threadIdx.x < words) {
                dbase[2 * threadIdx.x] = static_cast<uint16_t>(db);
                dbase[2 * threadIdx.x + 1] = static_cast<uint16_t>(db + t0);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < LITEMS; ++r) {
                const uint32_t e = w * (64 * rounds) + r * 64 + lane;
                if (r < rounds && e < m) {
                    const uint32_t pos = dbase[dg[r]] + ((whist[w][dg[r] >> 1] >> ((dg[r] & 1u) * 16u)) & 0xFFFFu) + rk[r];
                    si[pos] = ei[r];
                }
            }
            __syncthreads();
        }
    }

    PH(3);
    if (WRITEBACK) {
        const uint32_t rounds_w = (m + LBLOCK - 1) / LBLOCK;
        for (uint32_t r = 0; r < rounds_w; ++r) {
            const uint32_t e = r * LBLOCK + threadIdx.x;
            if (e < m) {
                const uint32_t slot = si[e];
                keys_sorted[base + a + e] = sk[slot];
                recs_sorted[base + a + e] = rec[base + slot];
            }
        }
        return;
    }
    // ---- voxels of the tile: a head is a sorted item whose key differs from the one before it.
    // hpos[v] = sorted position of voxel v's first point (the counters' LDS is free now).
    uint32_t heads = 0, nh = 0;
    const uint32_t per = (m + LBLOCK - 1) / LBLOCK;        // sorted items per thread (<= LITEMS)
    {
        const uint32_t i0 = threadIdx.x * per;
        uint32_t kp = (i0 > 0 && i0 < m) ? sk[si[i0 - 1]] : 0u;
#pragma unroll
        for (int j = 0; j < LITEMS; ++j) {
            if (static_cast<uint32_t>(j) < per && i0 + j < m) {
                const uint32_t k = sk[si[i0 + j]];
                if (i0 + j == 0 || k != kp) { heads |= 1u << j; ++nh; }
                kp = k;
            }
        }
    }
    uint32_t n_vox;
    {
        uint32_t v = block_excl_scan_w<LWAVES>(nh, lds, &n_vox);     // its barriers also retire the last reads of whist
        const uint32_t i0 = threadIdx.x * per;
#pragma unroll
        for (int j = 0; j < LITEMS; ++j)
            if (heads >> j & 1u) hpos[v++] = static_cast<uint16_t>(i0 + j);
    }
    __syncthreads();

    PH(4);
    // ---- kept voxels (A.4 step 7: at least min_pts points), published at once for the look-back
    // Thread t takes voxels [t * pv, (t + 1) * pv): a wave covers 64 * pv consecutive voxels (similar run
    // lengths), and the kept ones get consecutive output slots from one scan.
    uint32_t vstart[LITEMS], vlen[LITEMS];
    uint32_t nkeep = 0;
    const uint32_t pv = (n_vox + LBLOCK - 1) / LBLOCK;
    {
        const uint32_t v0 = threadIdx.x * pv;
        uint32_t hp = (v0 < n_vox) ? hpos[v0] : m;
#pragma unroll
        for (int j = 0; j < LITEMS; ++j) {
            const uint32_t v = v0 + j;
            vstart[j] = hp; vlen[j] = 0;
            if (static_cast<uint32_t>(j) < pv && v < n_vox) {
                const uint32_t nx = (v + 1 < n_vox) ? hpos[v + 1] : m;
                vlen[j] = (nx - hp >= min_pts) ? nx - hp : 0u;
                nkeep += vlen[j] ? 1u : 0u;
                hp = nx;
            }
        }
    }
    uint32_t c_t;
    const uint32_t my_slot = block_excl_scan_w<LWAVES>(nkeep, lds, &c_t);
    if (threadIdx.x == (LWAVES - 1) * 64) {              // the lane that later publishes the prefix: same-address stores stay in order
        const unsigned long long v = (tile == 0 ? CM2_FLAG_PREFIX : CM2_FLAG_AGG) | c_t;
        __hip_atomic_store(&tile_state[tile], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PH(5);
    // ---- output offset: kept voxels of every earlier tile (decoupled look-back). The last wave does
    // it while the others add up their voxels: it has the fewest voxels (often none), and what it waits for —
    // the counts of the tiles before this one — is being published in the meantime.
#ifdef CM_PHASE_TIMING
    const long long tw0_ = wall_clock64();
#endif
    if (w == LWAVES - 1) {
        uint32_t excl = 0;
        bool timed_out = false, done = false;
        // Eight wave-wide reads (512 predecessors: a whole generation of resident tiles) go out together, so
        // the walk back to the nearest published prefix usually costs one memory latency, not eight.
        constexpr int LB = 8;
        for (int j0 = static_cast<int>(tile) - 1; j0 >= 0 && !done; j0 -= 64 * LB) {
            unsigned long long v[LB];
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int jj = j0 - 64 * k - lane;
                v[k] = jj >= 0 ? __hip_atomic_load(&tile_state[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : CM2_FLAG_PREFIX;              // before tile 0: an empty prefix
            }
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                if (!done) {                                   // wave-uniform
                    const int jj = j0 - 64 * k - lane;
                    uint32_t spins = 0;
                    while (!(v[k] >> 32) && spins < (1u << 18)) {
                        __builtin_amdgcn_s_sleep(2);
                        v[k] = __hip_atomic_load(&tile_state[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ++spins;
                    }
                    if (!(v[k] >> 32)) { timed_out = true; v[k] = CM2_FLAG_PREFIX; }
                    const unsigned long long pm = __ballot((v[k] & CM2_FLAG_PREFIX) != 0);
                    const int first_p = pm ? __builtin_ctzll(pm) : 64;     // nearest predecessor with a full prefix
                    excl += wave_sum_u32((lane <= first_p) ? static_cast<uint32_t>(v[k]) : 0u);
                    done = pm != 0;
                }
            }
        }
        if (lane == 0) {
            if (tile > 0)
                __hip_atomic_store(&tile_state[tile], CM2_FLAG_PREFIX | static_cast<unsigned long long>(excl + c_t),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_off = excl;
        }
        if (__ballot(timed_out) && lane == 0) host_state[offsetof(CmFrameState, err) / 4] = CM_DEV_ERR_LOOKBACK;
#ifdef CM_PHASE_TIMING
        if (lane == 0) g_phase[(blockIdx.x & 4095) * 16 + 9] = (unsigned long long)(wall_clock64() - tw0_);
#endif
    }
    // ---- centroid sums: one lane per kept voxel adds its points in sorted (= stable) order, the
    // order pcl::VoxelGrid itself adds them in (A.4 step 6).
    Acc acc[LITEMS];
#pragma unroll
    for (int j = 0; j < LITEMS; ++j) {
        acc[j].x = acc[j].y = acc[j].z = acc[j].i = 0.f; acc[j].c = 0;
        if (vlen[j]) {
            float4 s4 = rec[base + si[vstart[j]]];
            for (uint32_t e = 1; e < vlen[j]; ++e) {
                const float4 r4 = rec[base + si[vstart[j] + e]];
                s4.x = __fadd_rn(s4.x, r4.x); s4.y = __fadd_rn(s4.y, r4.y);
                s4.z = __fadd_rn(s4.z, r4.z); s4.w = __fadd_rn(s4.w, r4.w);
            }
            acc[j].x = s4.x; acc[j].y = s4.y; acc[j].z = s4.z; acc[j].i = s4.w; acc[j].c = vlen[j];
        }
    }

#ifdef CM_PHASE_TIMING
    if (lane == 0 && w >= 1 && w <= 6) g_phase[(blockIdx.x & 4095) * 16 + 9 + w] = (unsigned long long)(wall_clock64() - tw0_);   // 10..15: waves 1..6
#endif
    PH(6);
    __syncthreads();
    PH(7);
    const uint32_t tile_off = s_off;
    if (tile == n_lt - 1) report_state(host_state, st, CM_DEV_OK, tile_off + c_t, true);

    uint32_t slot = tile_off + my_slot;
#pragma unroll
    for (int j = 0; j < LITEMS; ++j) {
        if (vlen[j] && partial_out) {                      // cm_partial_entry: key, count, sx, sy | sz, si, 0, 0
            partial_out[2 * static_cast<size_t>(slot)] = make_float4(__uint_as_float(sk[si[vstart[j]]]), __uint_as_float(acc[j].c), acc[j].x, acc[j].y);
            partial_out[2 * static_cast<size_t>(slot) + 1] = make_float4(acc[j].z, acc[j].i, 0.f, 0.f);
            ++slot;
        } else if (vlen[j]) {
            const float c = static_cast<float>(acc[j].c);
            const float rc = __frcp_rn(c);                  // RN(1/c), shared by the four quotients
            float4 o;
            o.x = div_by_count(acc[j].x, c, rc); o.y = div_by_count(acc[j].y, c, rc);
            o.z = div_by_count(acc[j].z, c, rc); o.w = div_by_count(acc[j].i, c, rc);
            out[slot] = o;
            if (out_key) { out_key[slot] = sk[si[vstart[j]]]; out_cnt[slot] = acc[j].c; }
            ++slot;
        }
    }
    PH(8);
}

}  // namespace
#ifdef CM_PHASE_TIMING
extern "C" __attribute__((visibility("default"))) void cm_debug_phases(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16 * 4096);
    if (reset) { void* p_; (void)hipGetSymbolAddress(&p_, HIP_SYMBOL(g_phase)); (void)hipMemset(p_, 0, sizeof(unsigned long long) * 16 * 4096); }
}
#endif

void cmk2_hist0(hipStream_t s, const CmFrameDev& f, CmFrameDev* fd, CmTileDev* tiles, bool do_setup, CmFrameState* st, uint32_t* hist, uint32_t* grp_acc,
                uint32_t* grp_clear_a, uint32_t* grp_clear_b, uint32_t n_group_words, uint32_t n_clear_a_words,
                unsigned long long* tile_state, uint32_t n_tile_state, float* records, int grid_mode, int check_box,
                uint32_t shift0, uint32_t n_global_passes, uint32_t n_tiles, const unsigned char* mask,
                const CmFrameState* st_outlier, int use_cell, void* compact_out, uint32_t* wave_cnt) {
    if (compact_out)
        hipLaunchKernelGGL(k2_hist0<true>, dim3(n_tiles), dim3(CM2_BLOCK), 0, s, f, fd, tiles, do_setup ? 1 : 0, st, hist, grp_acc, grp_clear_a, grp_clear_b,
                           n_group_words, n_clear_a_words, tile_state, n_tile_state, records, grid_mode, check_box, shift0,
                           n_global_passes, mask, st_outlier, use_cell, reinterpret_cast<float4*>(compact_out), wave_cnt);
    else
        hipLaunchKernelGGL(k2_hist0<false>, dim3(n_tiles), dim3(CM2_BLOCK), 0, s, f, fd, tiles, do_setup ? 1 : 0, st, hist, grp_acc, grp_clear_a, grp_clear_b,
                           n_group_words, n_clear_a_words, tile_state, n_tile_state, records, grid_mode, check_box, shift0,
                           n_global_passes, mask, st_outlier, use_cell, nullptr, nullptr);
}
void cmk2_hist(hipStream_t s, const CmFrameState* st, const unsigned char* dig, uint32_t* hist, uint32_t* grp,
               uint32_t n_tiles) {
    hipLaunchKernelGGL(k2_hist, dim3(n_tiles), dim3(CM2_BLOCK), 0, s, st, dig, hist, grp);
}
void cmk2_scatter(hipStream_t s, bool first, const CmFrameDev* fd, const CmTileDev* tiles, CmFrameState* st, const void* rec_in, void* rec_out,
                  unsigned char* dig_out, const uint32_t* hist, const uint32_t* grp, const uint32_t* totals,
                  uint32_t shift, uint32_t next_shift, uint32_t n_tiles, uint32_t n_groups, uint32_t n_padded,
                  const float* records, uint32_t n_records, int fold, const unsigned char* mask, int use_cell,
                  const void* compact_in, const uint32_t* wave_cnt, int debug_swap, uint32_t* tile_kept, bool sparse, bool ballot) {
    const float4* in = reinterpret_cast<const float4*>(rec_in);
    float4* o = reinterpret_cast<float4*>(rec_out);
    if (first && compact_in && sparse) {
#define CM2_SPARSE(BAL) hipLaunchKernelGGL(k2_scatter_sparse<BAL>, dim3((n_tiles + CM2_WAVES - 1) / CM2_WAVES), dim3(CM2_BLOCK), 0, s, fd, st, \
                                           reinterpret_cast<const float4*>(compact_in), wave_cnt, o, dig_out, hist, grp, totals, shift,       \
                                           next_shift, n_groups, n_tiles, use_cell, tile_kept)
        if (ballot) CM2_SPARSE(true); else CM2_SPARSE(false);
#undef CM2_SPARSE
        return;
    }
#define CM2_SCATTER(FIRST, BAL) hipLaunchKernelGGL((k2_scatter<FIRST, BAL>), dim3(n_tiles), dim3(CM2_BLOCK), 0, s, fd, tiles, st, in, o, dig_out, hist, grp, \
                                                   totals, shift, next_shift, n_groups, n_padded, records, n_records, fold, mask, use_cell,          \
                                                   reinterpret_cast<const float4*>(compact_in), wave_cnt, debug_swap, tile_kept)
    if (first) { if (ballot) CM2_SCATTER(true, true); else CM2_SCATTER(true, false); }
    else { if (ballot) CM2_SCATTER(false, true); else CM2_SCATTER(false, false); }
#undef CM2_SCATTER
}
void cmk2_local(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, CmFrameState* st_next, uint32_t* host_state,
                const void* rec, unsigned long long* tile_state, uint32_t* ticket, void* out, uint32_t* out_key,
                uint32_t* out_cnt, void* partial_out, uint32_t low_bits, uint32_t n_padded) {
    // 2048-record tiles, room for 4096 (bucket tails up to 2048 records), 512 threads: 42 KB of LDS, three per CU
    hipLaunchKernelGGL((k2_local<2048, 4096, 512, false>), dim3(n_padded / 2048), dim3(512), 0, s, fd, st, st_next, host_state,
                       reinterpret_cast<const float4*>(rec), tile_state, ticket, reinterpret_cast<float4*>(out), out_key,
                       out_cnt, reinterpret_cast<float4*>(partial_out), low_bits, nullptr, nullptr, 0);
}
void cmk2_local_sort(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t* host_state, const void* rec,
                     uint32_t* keys_sorted, void* recs_sorted, uint32_t low_bits, uint32_t n_padded) {
    hipLaunchKernelGGL((k2_local<2048, 4096, 512, true>), dim3(n_padded / 2048), dim3(512), 0, s, fd, st, nullptr, host_state,
                       reinterpret_cast<const float4*>(rec), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, low_bits,
                       keys_sorted, reinterpret_cast<float4*>(recs_sorted), 1);
}
