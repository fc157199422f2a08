// cm_kernels_v4.hip — ONE global pass instead of two or three: the "quantile" variant of the bucket path (gfx950).
//
// The bucket path (cm_kernels_v2.hip) groups the point records by the HIGH bits of their voxel index: buckets are cells of
// a fixed grid, and the cloud decides how full each one is — at cfg2 the finish's LDS (4032 records) is only safe once 16
// index bits are sorted, i.e. after TWO 8-bit passes (2 x 128 MB of fabric traffic). Here the buckets are ranges of the
// voxel index cut at the QUANTILES of the previous frame's sorted records (k3_local leaves them, cm_kernels_v3.hip):
// every bucket holds about 1950 records whatever the density, so n / 1950 <= 2048 buckets are enough — one pass:
//
//   k4_hist     transform + crop + index of every raw point, bucket by binary search over the splitters (LDS), counts
//               per (4096-slot tile, bucket) as one row of 16-bit words; min/max records, frame set-up  [16 B/pt read]
//   k4_colscan  the rows become "records of bucket b in the tiles before this one" (column prefix, in place); totals
//               per bucket; a bucket that would not fit the finish aborts the frame                      [4 MB r + w]
//   k4_scatter  raw points again -> records, ranked per wave by returning LDS adds, staged through LDS in sorted order
//               and written as runs: stable, every record to its final bucket                   [16 B r, 16 B w per pt]
//   k3_local<QUANT> one workgroup per bucket: LDS sort by the index, centroids (cm_kernels_v3.hip), next frame's splitters
//
// Whether 32-byte runs (two records per tile and bucket) reach the fabric as whole lines was the open question:
// scripts/micro/wide_scatter.hip — staged in sorted order they do (2048 bins: 71.5 MB written for 65.5 MB of records,
// 34 us; written straight from registers 117 MB, 43 us; the two 8-bit passes this replaces: 2 x 66 MB, 55 us + k2_hist).
// Same records in the same (stable) order inside every voxel as the bucket path, hence the same sums bit for bit.
// The splitters are a prediction like the predicted box: verified on the device (k4_colscan: no bucket beyond the
// finish's capacity; k3_local: every index inside its bucket's range), handed back otherwise (CM_DEV_ERR_QUANT: the frame
// is redone with the fixed-grid passes, which also leave fresh splitters).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cm_common.hpp"
#include "cm_device.h"
#include "cm_kernels.h"

namespace {

template <int WAVES>
__device__ __forceinline__ uint32_t block_excl_scan4(uint32_t v, uint32_t* lds, uint32_t* total) {
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

// Bucket of an index = the number of splitters S[1 .. CM4_BINS - 1] that are <= key (S[0] = 0 is below every index; S is
// ascending, 0xFFFFFFFF beyond the frame's buckets; an index is below 2^31: the host only takes this path for key_bits < 32).
// The splitters sit in LDS as a perfect binary tree in breadth-first order (node i: children 2i + 1, 2i + 2; node of level l,
// position p = S[(2p + 1) << (10 - l)]): the nodes a wave's lanes read in one step are NEIGHBOURS in memory, so they spread
// over the banks — the same search over the sorted array reads addresses that are multiples of the step, i.e. ONE bank for
// every step of 32 or more and a few for the smaller ones (k4_hist 40 us against 19 us for k2_hist0 when it was written
// that way). Eight searches side by side.
static_assert(CM4_BINS == 2048 && CM4_MAX_BUCKETS == 8192, "eleven to thirteen levels");
// (nodes numbered from 1, heap fashion: children of node b are 2b and 2b + 1, so a step is b = 2b + (tree[b] <= key) — one
// add-with-carry behind the compare; word 0 of the tree is unused. After LEVELS steps b - 2^LEVELS is the bucket.)
// LEVELS = 11: up to 2048 buckets. 12, 13: up to 8192 buckets for frames of up to 15 M records — 2 or 4 neighbouring buckets
// then share one of the pass's 2048 bins (cm_device.h cm_quant_sub_shift): the pass scatters by bucket >> shift and leaves
// the low bits as a byte beside every record; the finish workgroup of a bucket picks its records out of its bin by them.
template <int LEVELS>
__device__ __forceinline__ void load_splitter_tree(uint32_t* __restrict__ tree, const uint32_t* __restrict__ spl_g) {
#pragma unroll
    for (int q = 0; q < (1 << LEVELS) / CM2_BLOCK; ++q) {
        const uint32_t e = q * CM2_BLOCK + threadIdx.x;            // node e (1 .. 2^LEVELS - 1)
        const uint32_t l = 31u - static_cast<uint32_t>(__builtin_clz(e | 1u));
        const uint32_t p = e - (1u << l);
        tree[e] = e ? spl_g[((2u * p + 1u) << (LEVELS - 1 - l))] : 0u;
    }
}
template <int LEVELS, int N>
__device__ __forceinline__ void buckets_of(const uint32_t* __restrict__ tree, const uint32_t (&key)[N], uint32_t (&bk)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) bk[r] = 1;
#pragma unroll
    for (int l = 0; l < LEVELS; ++l) {
#pragma unroll
        for (int r = 0; r < N; ++r) bk[r] = bk[r] + bk[r] + ((tree[bk[r]] <= key[r]) ? 1u : 0u);
    }
#pragma unroll
    for (int r = 0; r < N; ++r) bk[r] -= 1u << LEVELS;
}

// Where counter word `word` (two 16-bit bucket counters) of tile `tile` lives: column blocks of eight words, the tiles of a
// block one after the other — [word / 8][tile][word % 8]. k4_hist writes and k4_scatter reads 32-byte pieces of it;
// k4_colscan, whose whole work is this table, reads and writes one contiguous block per workgroup (with rows of 1024 words
// per tile it fetched every 128-byte line four times, from four workgroups: 15 MB for 4).
__device__ __forceinline__ size_t cnt_at(uint32_t tile, uint32_t word, uint32_t n_tiles) {
    return (static_cast<size_t>(word >> 3) * n_tiles + tile) * 8u + (word & 7u);
}

// The exact bounds of the frame's valid points from the per-tile records (see fold_bounds, cm_kernels_v2.hip).
__device__ __forceinline__ void fold_bounds4(float* s_f, CmFrameState* __restrict__ st, const float* __restrict__ records,
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
// k4_hist: what k2_hist0 does for the fixed-grid passes (frame set-up, clears, box check, min/max records), with the
// counts taken per quantile bucket: row `tile` of cnt = CM4_BINS 16-bit counters (a tile holds 4096 points: no overflow).
// ------------------------------------------------------------------------------------------------
template <int LEVELS>
__global__ __launch_bounds__(CM2_BLOCK) void k4_hist(const CmFrameDev fv, CmFrameDev* __restrict__ fd_dst,
                                                     CmTileDev* __restrict__ tiles_dst, int do_setup,
                                                     CmFrameState* __restrict__ st, const uint32_t* __restrict__ spl_g,
                                                     uint32_t* __restrict__ cnt, uint16_t* __restrict__ bid,
                                                     unsigned long long* __restrict__ tile_state, uint32_t n_tile_state,
                                                     float* __restrict__ records, int grid_mode, int check_box,
                                                     uint32_t* __restrict__ big_list, uint32_t bin_shift) {
    __shared__ uint32_t spl[1 << LEVELS];
    __shared__ uint32_t lh[CM4_BINS / 2];
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
    for (uint32_t k = tile * CM2_BLOCK + threadIdx.x; k < n_tile_state; k += gridDim.x * CM2_BLOCK) tile_state[k] = 0ull;
    if (tile == 0 && threadIdx.x == 0) {                 // the box and its grid, as the host set them up
        st->status = CM_DEV_OK;
        for (int a = 0; a < 3; ++a) {
            st->min_p[a] = grid_mode == 2 ? fd->ext_min[a] : fd->crop_min[a];
            st->max_p[a] = grid_mode == 2 ? fd->ext_max[a] : fd->crop_max[a];
            const int32_t mb = fd->box_min_b[a], db = fd->box_div_b[a];
            st->min_b[a] = mb; st->max_b[a] = mb + db - 1;
            st->div_b[a] = db;
        }
        st->key_bits = fd->box_key_bits;
        st->n_passes = 1u;
        if (big_list) big_list[0] = 0u;                    // (k4_colscan's list of buckets for the large finish shape)
    }
    const BoxGrid b = box_grid_of(fd);
    const bool predicted = check_box != 0;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const CmSensorDev& sd = fd->s[te.info & 0xFFu];
    Pt p[CM2_ITEMS];
    load_tile_te<CM2_ITEMS>(te, sd, w * (64 * CM2_ITEMS) + lane, p);
    load_splitter_tree<LEVELS>(spl, spl_g);
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
    const uint32_t crop = fd->crop_enable;
    float cmn0 = 0.f, cmn1 = 0.f, cmn2 = 0.f, cmx0 = 0.f, cmx1 = 0.f, cmx2 = 0.f;
    if (crop) {
        cmn0 = fd->crop_min[0]; cmn1 = fd->crop_min[1]; cmn2 = fd->crop_min[2];
        cmx0 = fd->crop_max[0]; cmx1 = fd->crop_max[1]; cmx2 = fd->crop_max[2];
    }
#pragma unroll
    for (int q = 0; q < CM4_BINS / 2 / CM2_BLOCK; ++q) lh[q * CM2_BLOCK + threadIdx.x] = 0;
    if (threadIdx.x == 0) s_out = 0;
    __syncthreads();
    const float inf = __uint_as_float(0x7F800000u);
    float mn0 = inf, mn1 = inf, mn2 = inf, mx0 = -inf, mx1 = -inf, mx2 = -inf;
    uint32_t cnt_ok = 0;
    bool any_out = false;
    float tx[CM2_ITEMS], ty[CM2_ITEMS], tz[CM2_ITEMS];
    uint32_t key[CM2_ITEMS], bk[CM2_ITEMS];
    uint32_t okm = 0, keepm = 0;
#pragma unroll
    for (int r = 0; r < CM2_ITEMS; ++r) {
        const float x = xf_row(m[0], m[1], m[2], m[3], p[r].x, p[r].y, p[r].z);
        const float y = xf_row(m[4], m[5], m[6], m[7], p[r].x, p[r].y, p[r].z);
        const float z = xf_row(m[8], m[9], m[10], m[11], p[r].x, p[r].y, p[r].z);
        tx[r] = x; ty[r] = y; tz[r] = z;
        bool ok = finite_f32(x) & finite_f32(y) & finite_f32(z);
        if (crop) ok = ok & !((x < cmn0) | (x > cmx0) | (y < cmn1) | (y > cmx1) | (z < cmn2) | (z > cmx2));
        bool in;
        key[r] = key_of(b, x, y, z, &in);
        if (predicted) any_out = any_out | (ok & !in);     // a crop box holds every valid point by construction
        else in = true;
        okm |= ok ? (1u << r) : 0u;
        keepm |= (ok & in) ? (1u << r) : 0u;
    }
    buckets_of<LEVELS, CM2_ITEMS>(spl, key, bk);
    const uint32_t slot0 = tile * CM_TILE + w * (64 * CM2_ITEMS) + lane;
#pragma unroll
    for (int r = 0; r < CM2_ITEMS; ++r) {
        // (a slot without a record adds nothing, to a word of its own: same-address LDS adds of a wave serialise;
        // the counters are per bin — what this pass scatters by: the bucket number without its low bin_shift bits
        // (shared bins above CM4_BINS buckets, cm_device.h; bin_shift == 0: a bin is a bucket))
        const bool keep = (keepm >> r) & 1u;
        const uint32_t lo = bk[r] >> bin_shift;
        atomicAdd(&lh[keep ? lo >> 1 : static_cast<uint32_t>(lane)], (keep ? 1u : 0u) << ((lo & 1u) * 16u));
        // the bucket of every slot (0xFFFF: no record), so that k4_scatter neither tests nor searches a second time
        bid[slot0 + r * 64] = static_cast<uint16_t>(keep ? bk[r] : 0xFFFFu);
    }
    if (predicted) {
        cnt_ok = static_cast<uint32_t>(__builtin_popcount(okm));
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
        if (any_out) s_out = 1u;
        mn0 = wave_min_f32_l63(mn0); mn1 = wave_min_f32_l63(mn1); mn2 = wave_min_f32_l63(mn2);
        mx0 = wave_max_f32_l63(mx0); mx1 = wave_max_f32_l63(mx1); mx2 = wave_max_f32_l63(mx2);
        cnt_ok = wave_sum_u32(cnt_ok);
        if (lane == 63) {
            s_mm[w][0] = mn0; s_mm[w][1] = mn1; s_mm[w][2] = mn2;
            s_mm[w][3] = mx0; s_mm[w][4] = mx1; s_mm[w][5] = mx2;
            s_cnt[w] = cnt_ok;
        }
    }
    __syncthreads();
    // The counts go out in column blocks of eight words (cnt_at): k4_colscan then streams one contiguous block per workgroup.
#pragma unroll
    for (int q = 0; q < CM4_BINS / 2 / CM2_BLOCK; ++q)
        cnt[cnt_at(tile, q * CM2_BLOCK + threadIdx.x, gridDim.x)] = lh[q * CM2_BLOCK + threadIdx.x];
    if (predicted && threadIdx.x < 8) {                    // record: min xyz, max xyz, count, pad
        const int k = threadIdx.x;
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

// ------------------------------------------------------------------------------------------------
// k4_colscan: cnt[tile][bucket] (16-bit, two per word) -> records of that bucket in the tiles before `tile` (in place);
// totals[bucket]. Workgroup = 16 words (32 buckets) x 64 chunks of consecutive tiles: every thread holds its chunk's words
// in registers (one batch of loads), the chunks' sums meet in LDS. The 16-bit halves are summed apart for the totals (a
// bucket beyond the finish's capacity aborts the frame — its packed prefixes may then have carried into their neighbours,
// and nothing reads them).
// ------------------------------------------------------------------------------------------------
// TPC: tiles per chunk at most — 12: 128 chunks -> 1536 tiles (6.3 M slots); 32: 4096 tiles (16.8 M slots)
static_assert(32 * 128 >= CM4_MAX_TILES, "k4_colscan's register tile covers the frames the host sends here");
template <int CM4_SCAN_TPC>
__global__ __launch_bounds__(1024) void k4_colscan(CmFrameState* __restrict__ st, uint32_t* __restrict__ host_state,
                                                   uint32_t* __restrict__ cnt, uint32_t* __restrict__ totals,
                                                   uint32_t n_tiles, uint32_t cap, uint32_t cap_big, uint32_t* __restrict__ big_list) {
    // cap: what a finish workgroup of the usual shape holds; a bucket beyond it (up to cap_big) goes on big_list — word 0 the
    // count (zeroed by k4_hist), then the bucket numbers — for the large shape's launch; beyond cap_big, or more than
    // CM4_MAX_BIG of them, the frame is handed back. (Shared bins: cap == cap_big == what 2^shift finish workgroups hold, no
    // list: a bin beyond that holds a bucket beyond one workgroup; k3_local<SUB> checks the buckets themselves.)
    __shared__ uint32_t s_lo[16][8], s_hi[16][8];
    if (st->status != CM_DEV_OK || st->outside) return;
    const uint32_t j = threadIdx.x & 7u, c = threadIdx.x >> 3;
    const uint32_t tpc = (n_tiles + 127u) / 128u;
    const uint32_t word = blockIdx.x * 8u + j;
    uint32_t v[CM4_SCAN_TPC];
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < CM4_SCAN_TPC; ++k) {
        const uint32_t t = c * tpc + k;
        v[k] = (static_cast<uint32_t>(k) < tpc && t < n_tiles) ? cnt[cnt_at(t, word, n_tiles)] : 0u;
        lo += v[k] & 0xFFFFu; hi += v[k] >> 16;
    }
    // prefix over the chunks: eight of them sit in one wave (lanes 8 apart), the sixteen waves meet in LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t ilo = lo, ihi = hi;
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
        const uint32_t a = __shfl_up(ilo, d), b2 = __shfl_up(ihi, d);
        if (lane >= d) { ilo += a; ihi += b2; }
    }
    if (lane >= 56) { s_lo[w][j] = ilo; s_hi[w][j] = ihi; }
    __syncthreads();
    uint32_t plo = ilo - lo, phi = ihi - hi;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t a = s_lo[q][j], b2 = s_hi[q][j];
        if (q < w) { plo += a; phi += b2; }
    }
    uint32_t run = plo | (phi << 16);
#pragma unroll
    for (int k = 0; k < CM4_SCAN_TPC; ++k) {
        const uint32_t t = c * tpc + k;
        if (static_cast<uint32_t>(k) < tpc && t < n_tiles) cnt[cnt_at(t, word, n_tiles)] = run;
        run += v[k];
    }
    if (c == 127u) {
        const uint32_t tlo = plo + lo, thi = phi + hi;
        totals[2 * word] = tlo; totals[2 * word + 1] = thi;
        bool bad = tlo > cap_big || thi > cap_big;
        if (big_list) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t tv = h ? thi : tlo;
                if (tv > cap && tv <= cap_big) {
                    const uint32_t at = atomicAdd(&big_list[0], 1u);
                    if (at < CM4_MAX_BIG) big_list[1 + at] = 2 * word + h; else bad = true;
                }
            }
        }
        if (bad) {                                         // a bucket no finish workgroup can hold: the frame goes back
            st->quant_abort = 1u;
            host_state[offsetof(CmFrameState, err) / 4] = CM_DEV_ERR_QUANT;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k4_scatter: every valid point -> its record (x, y, z, intensity in the target frame) at
//   bucket base + records of the bucket in earlier tiles + rank among this tile's records of the bucket   (stable).
// Ranks: returning LDS adds on per-wave counters (16-bit pairs), waves in order. The records leave through an LDS
// staging buffer in sorted order, 2048 positions per round: a wave's store instruction then covers neighbouring
// buckets in ascending address order with same-bucket records side by side, which is what lets the XCD's L2 put the
// 32-byte runs of neighbouring tiles together (file header). Tiles are dealt to the XCDs in contiguous ranges.
// ------------------------------------------------------------------------------------------------
template <bool SUB, bool BALLOT>
__global__ __launch_bounds__(CM2_BLOCK, 6) void k4_scatter(const CmFrameDev* __restrict__ fd, const CmTileDev* __restrict__ tiles,
                                                           CmFrameState* __restrict__ st, const uint16_t* __restrict__ bid,
                                                           const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ totals,
                                                           uint32_t* __restrict__ bofs, uint32_t n_buckets,
                                                           float4* __restrict__ rec_out, const float* __restrict__ records,
                                                           uint32_t n_records, int fold, uint32_t* __restrict__ tile_kept,
                                                           unsigned char* __restrict__ dig_out, const uint32_t* __restrict__ big_list,
                                                           uint32_t sub_shift) {
    // SUB (shared bins, cm_device.h): the bucket numbers have up to thirteen bits; this pass scatters by all but the low
    // sub_shift and leaves THOSE as a byte beside every record (dig_out): which of its bin's buckets a record belongs to, for
    // the finish (k3_local<SUB>).
    auto bin_of = [&](uint32_t id) { return SUB ? id >> sub_shift : id; };
    constexpr int HW = CM4_BINS / 2;                      // counter words per wave
    constexpr int STG = 2048;                             // staged records per round
    __shared__ uint32_t buf[STG * 4 + STG / 2];           // per-wave counters (8 x HW) | staging: records + buckets
    __shared__ uint32_t gofs[CM4_BINS];
    __shared__ uint32_t lds[CM2_WAVES];
    static_assert(CM2_WAVES * HW <= STG * 4 + STG / 2 && CM4_BINS <= STG * 4, "overlays fit");
    static_assert(HW == 2 * CM2_BLOCK, "two counter words per thread");
    uint32_t (*wcnt)[HW] = reinterpret_cast<uint32_t (*)[HW]>(buf);
    float4* srec = reinterpret_cast<float4*>(buf);
    uint16_t* sbk = reinterpret_cast<uint16_t*>(buf + STG * 4);
    if (st->status != CM_DEV_OK) return;
    if (st->outside) {                                     // handed back — with the cloud's exact bounds (see k2_scatter)
        if (fold && blockIdx.x == 0) fold_bounds4(reinterpret_cast<float*>(buf), st, records, n_records);
        return;
    }
    if (st->quant_abort) return;
    uint32_t tile = blockIdx.x;
    {
        const uint32_t per = gridDim.x / 8;              // contiguous tile range per XCD
        if (blockIdx.x < per * 8) tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

    const CmTileDev te = tiles[tile];
    const uint32_t sidx = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(te.info & 0xFFu)));   // (uniform: scalar loads of the matrix)
    Pt p[CM2_ITEMS];
    load_tile_te<CM2_ITEMS, true>(te, fd->s[sidx], w * (64 * CM2_ITEMS) + lane, p);     // (non-temporal: the last reader of the raw clouds)
    uint32_t bk[CM2_ITEMS];                               // k4_hist left every slot's bucket (0xFFFF: the slot holds no record)
    {
        const uint32_t slot0 = tile * CM_TILE + w * (64 * CM2_ITEMS) + lane;
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) bk[r] = bid[slot0 + r * 64];
    }
    for (uint32_t q = threadIdx.x; q < CM2_WAVES * HW; q += CM2_BLOCK) buf[q] = 0;

    float4 rec[CM2_ITEMS];
    uint32_t vmask = 0;
    {
        const CmSensorDev& sd = fd->s[sidx];
        float m[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
        const bool all_fields = fd->downsample_all != 0;
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r) {
            rec[r].x = xf_row(m[0], m[1], m[2], m[3], p[r].x, p[r].y, p[r].z);
            rec[r].y = xf_row(m[4], m[5], m[6], m[7], p[r].x, p[r].y, p[r].z);
            rec[r].z = xf_row(m[8], m[9], m[10], m[11], p[r].x, p[r].y, p[r].z);
            rec[r].w = all_fields ? p[r].i : 0.f;
            vmask |= (bk[r] != 0xFFFFu) ? (1u << r) : 0u;
            bk[r] &= SUB ? (CM4_MAX_BUCKETS - 1) : (CM4_BINS - 1);
        }
    }
    __syncthreads();
    // this tile's row of the column prefix and the bucket totals: asked for now, used behind the ranking
    const uint2 trow = *reinterpret_cast<const uint2*>(cnt + cnt_at(tile, 2 * threadIdx.x, gridDim.x));
    const uint4 tot4 = *reinterpret_cast<const uint4*>(totals + 4 * threadIdx.x);
    // (from here on bk[r] = bucket | rank among the wave's records of the bucket << 16)
    if (BALLOT) {                                          // (ranking by ballots: cm_common.hpp wave_rank_ballot)
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r)
            bk[r] |= wave_rank_ballot(wcnt[w], bin_of(bk[r]), 11u, (vmask >> r) & 1u, lane) << 16;
    } else {
#pragma unroll
    for (int r0 = 0; r0 < CM2_ITEMS; r0 += 4) {
        uint32_t got[4];
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) {
            const bool has = (vmask >> r) & 1u;
            const uint32_t lo = bin_of(bk[r]);
            got[r - r0] = atomicAdd(&wcnt[w][has ? lo >> 1 : static_cast<uint32_t>(lane)], (has ? 1u : 0u) << ((lo & 1u) * 16u));
        }
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) {
            bk[r] |= ((got[r - r0] >> ((bin_of(bk[r]) & 1u) * 16u)) & 0xFFFFu) << 16;
            asm volatile("" : "+v"(bk[r]));                 // (formed here: the raw returns need not stay alive)
        }
    }
    }
    __syncthreads();
    // thread t: counter words 2t, 2t+1 = buckets 4t .. 4t+3. Per bucket: prefix over the waves; first sorted position of
    // the bucket in this tile (exclusive scan over the buckets); first record of the bucket in the frame (the same over
    // the totals); gofs[b] = where sorted position 0 of this tile would go if it belonged to bucket b.
    uint32_t c4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < CM2_WAVES; ++q) {
        const uint2 cwq = *reinterpret_cast<const uint2*>(&wcnt[q][2 * threadIdx.x]);
        c4[0] += cwq.x & 0xFFFFu; c4[1] += cwq.x >> 16; c4[2] += cwq.y & 0xFFFFu; c4[3] += cwq.y >> 16;
    }
    uint32_t tile_valid, n_total;
    const uint32_t db = block_excl_scan4<CM2_WAVES>(c4[0] + c4[1] + c4[2] + c4[3], lds, &tile_valid);
    const uint32_t t4[4] = {tot4.x, tot4.y, tot4.z, tot4.w};
    const uint32_t gb = block_excl_scan4<CM2_WAVES>(t4[0] + t4[1] + t4[2] + t4[3], lds, &n_total);
    {
        const uint32_t pre[4] = {trow.x & 0xFFFFu, trow.x >> 16, trow.y & 0xFFFFu, trow.y >> 16};
        uint32_t d = db, g = gb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gofs[4 * threadIdx.x + k] = g + pre[k] - d;
            if (tile == 0) bofs[4 * threadIdx.x + k] = g;  // first record of every bucket: what the finish tiles by
            d += c4[k]; g += t4[k];
        }
        uint32_t r0 = db, r1 = db + c4[0], r2 = db + c4[0] + c4[1], r3 = db + c4[0] + c4[1] + c4[2];
#pragma unroll
        for (int q = 0; q < CM2_WAVES; ++q) {                 // (read again: sixteen counts per thread held over the scans spill)
            const uint2 cwq = *reinterpret_cast<const uint2*>(&wcnt[q][2 * threadIdx.x]);
            *reinterpret_cast<uint2*>(&wcnt[q][2 * threadIdx.x]) = make_uint2(r0 | (r1 << 16), r2 | (r3 << 16));
            r0 += cwq.x & 0xFFFFu; r1 += cwq.x >> 16; r2 += cwq.y & 0xFFFFu; r3 += cwq.y >> 16;
        }
    }
    if (tile == 0 && threadIdx.x == 0) { st->n_valid = n_total; bofs[CM4_BINS] = n_total; st->quant_big = big_list ? big_list[0] : 0u; }
    if (tile_kept && threadIdx.x == 0) tile_kept[tile] = tile_valid;
    __syncthreads();
    // (now bk[r] = bucket | sorted position in the tile << 16; a slot without a record: position 0xFFFF, beyond every round)
#pragma unroll
    for (int r = 0; r < CM2_ITEMS; ++r) {
        const uint32_t bq = bk[r] & 0xFFFFu, bn_ = bin_of(bq);
        const uint32_t wv = wcnt[w][bn_ >> 1];
        const uint32_t at = ((wv >> ((bn_ & 1u) * 16u)) & 0xFFFFu) + (bk[r] >> 16);
        bk[r] = bq | ((((vmask >> r) & 1u) ? at : 0xFFFFu) << 16);
    }
#pragma unroll
    for (int h = 0; h < CM_TILE / STG; ++h) {
        const uint32_t lo = h * STG;
        if (h > 0 && tile_valid <= lo) break;              // uniform
        __syncthreads();                                   // (round 0: the last reads of the counters; later: of the staged records)
#pragma unroll
        for (int r = 0; r < CM2_ITEMS; ++r)
            if ((bk[r] >> 16) - lo < static_cast<uint32_t>(STG)) {
                srec[(bk[r] >> 16) - lo] = rec[r];
                sbk[(bk[r] >> 16) - lo] = static_cast<uint16_t>(bk[r]);
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < STG / CM2_BLOCK; ++j) {
            const uint32_t t = lo + j * CM2_BLOCK + threadIdx.x;
            if (t < tile_valid) {
                const uint32_t id = sbk[t - lo];
                const uint32_t pos = gofs[bin_of(id)] + t;
                rec_out[pos] = srec[t - lo];
                if (SUB) dig_out[pos] = static_cast<unsigned char>(id & ((1u << sub_shift) - 1u));
            }
        }
    }
    // The exact bounds of the cloud (pcl::getMinMax3D) for the result and for the next frame's box.
    if (fold && tile == 0) {
        __syncthreads();
        fold_bounds4(reinterpret_cast<float*>(buf), st, records, n_records);
    }
}

}  // namespace

void cmk4_hist(hipStream_t s, const CmFrameDev& f, CmFrameDev* fd, CmTileDev* tiles, bool do_setup, CmFrameState* st,
               const uint32_t* spl, uint32_t* cnt, uint16_t* bid, unsigned long long* tile_state, uint32_t n_tile_state, float* records,
               int grid_mode, int check_box, uint32_t n_tiles, uint32_t n_buckets, uint32_t* big_list, uint32_t sub_shift) {
    // sub_shift (cm_quant_sub_shift(n_buckets)) != 0: shared bins — counted per bucket >> sub_shift (<= CM4_BINS bins)
#define CM4_HIST(L) hipLaunchKernelGGL(k4_hist<L>, dim3(n_tiles), dim3(CM2_BLOCK), 0, s, f, fd, tiles, do_setup ? 1 : 0, st, spl, cnt, bid, \
                                       tile_state, n_tile_state, records, grid_mode, check_box, big_list, sub_shift)
    if (n_buckets <= 2048) CM4_HIST(11);
    else if (n_buckets <= 4096) CM4_HIST(12);
    else CM4_HIST(13);
#undef CM4_HIST
}
void cmk4_colscan(hipStream_t s, CmFrameState* st, uint32_t* host_state, uint32_t* cnt, uint32_t* totals, uint32_t n_tiles,
                  uint32_t cap, uint32_t cap_big, uint32_t* big_list) {
    if (n_tiles <= 12 * 128)
        hipLaunchKernelGGL(k4_colscan<12>, dim3(CM4_BINS / 16), dim3(1024), 0, s, st, host_state, cnt, totals, n_tiles, cap, cap_big, big_list);
    else
        hipLaunchKernelGGL(k4_colscan<32>, dim3(CM4_BINS / 16), dim3(1024), 0, s, st, host_state, cnt, totals, n_tiles, cap, cap_big, big_list);
}
void cmk4_scatter(hipStream_t s, const CmFrameDev* fd, const CmTileDev* tiles, CmFrameState* st, const uint16_t* bid,
                  const uint32_t* cnt, const uint32_t* totals, uint32_t* bofs, uint32_t n_buckets, void* rec_out,
                  const float* records, uint32_t n_records, int fold, uint32_t* tile_kept, uint32_t n_tiles, unsigned char* dig_out,
                  bool ballot, const uint32_t* big_list, uint32_t sub_shift) {
#define CM4_SCATTER(SUB, BAL, DIG) hipLaunchKernelGGL((k4_scatter<SUB, BAL>), dim3(n_tiles), dim3(CM2_BLOCK), 0, s, fd, tiles, st, bid, cnt, totals, \
                                                      bofs, n_buckets, reinterpret_cast<float4*>(rec_out), records, n_records, fold, tile_kept, DIG, big_list, \
                                                      sub_shift)
    if (dig_out) { if (ballot) CM4_SCATTER(true, true, dig_out); else CM4_SCATTER(true, false, dig_out); }
    else { if (ballot) CM4_SCATTER(false, true, nullptr); else CM4_SCATTER(false, false, nullptr); }
#undef CM4_SCATTER
}
