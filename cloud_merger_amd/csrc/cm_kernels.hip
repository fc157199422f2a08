// cm_kernels.hip — gfx950 kernels of the merge → voxel-grid path.
//
// Pipeline per frame (all on one stream, no host round trip):
//   k_minmax     K0  transform + crop, fp32 min/max of the merged cloud (skipped when the crop box
//                    already bounds the grid)                                  [HBM: 16 B/pt read]
//   k_bounds     Kb  PCL's overflow guard, min_b/div_b, key width
//   k_keys       K1  transform + crop + voxel key, first radix histogram      [16 B/pt r, 4 B/pt w]
//   per 8-bit digit: k_hist (not for digit 0), k_colscan, k_scatter           [LDS-tiled LSD radix]
//   k_seg_count  K3a kept voxels per tile of the sorted keys
//   k_finalize   K3s tile offsets, n_out, status; zeroes the next frame's state
//   k_seg_reduce K3b gather + wavefront segmented centroid reduction          [16 B/voxel write]
//
// Arithmetic restates what the reference gets from pcl_ros::transformPointCloud
// (pc_preprocessing_main.cpp:322), pcl::PassThrough in getROI (:20-40) and pcl::VoxelGrid (:171-176);
// semantics in SURVEY.md Appendix A. Every fp32 operation that decides occupancy uses the _rn
// intrinsics so it is rounded once, in the reference's order, never contracted into an FMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cm_device.h"
#include "cm_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t enc_f32(float f) {       // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
    const uint32_t u = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
    return __uint_as_float(u);
}

struct Pt { float x, y, z, i; };

__device__ __forceinline__ float load_f32_unaligned(const unsigned char* p) {
    float f;
    __builtin_memcpy(&f, p, 4);
    return f;
}

__device__ __forceinline__ Pt load_point(const unsigned char* __restrict__ data, uint32_t layout,
                                         uint32_t step, uint32_t ox, uint32_t oy, uint32_t oz,
                                         uint32_t oi, uint32_t idx) {
    Pt p;
    if (layout == CM_LAYOUT_XYZI16) {
        const float4 v = *reinterpret_cast<const float4*>(data + static_cast<size_t>(idx) * 16);
        p.x = v.x; p.y = v.y; p.z = v.z; p.i = v.w;
    } else if (layout == CM_LAYOUT_PCL32) {
        const unsigned char* q = data + static_cast<size_t>(idx) * 32;
        const float4 v = *reinterpret_cast<const float4*>(q);
        p.x = v.x; p.y = v.y; p.z = v.z;
        p.i = *reinterpret_cast<const float*>(q + 16);
    } else {
        const unsigned char* q = data + static_cast<size_t>(idx) * step;
        p.x = load_f32_unaligned(q + ox);
        p.y = load_f32_unaligned(q + oy);
        p.z = load_f32_unaligned(q + oz);
        p.i = (oi == 0xFFFFFFFFu) ? 0.0f : load_f32_unaligned(q + oi);
    }
    return p;
}

// pcl::transformPointCloud scalar form: ((m0*x + m1*y) + m2*z) + m3, each op rounded (A.1).
__device__ __forceinline__ float xf_row(float m0, float m1, float m2, float m3, float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m0, x), __fmul_rn(m1, y)), __fmul_rn(m2, z)), m3);
}

__device__ __forceinline__ bool finite_f32(float v) {
    return (__float_as_uint(v) & 0x7F800000u) != 0x7F800000u;
}

// PassThrough x3 (closed box) + "non-finite points vanish" (A.2, A.3).
__device__ __forceinline__ bool point_valid(float x, float y, float z, uint32_t crop,
                                            const float* __restrict__ cmin, const float* __restrict__ cmax) {
    bool ok = finite_f32(x) && finite_f32(y) && finite_f32(z);
    if (crop) {
        ok = ok && !(x < cmin[0] || x > cmax[0]) && !(y < cmin[1] || y > cmax[1]) &&
             !(z < cmin[2] || z > cmax[2]);
    }
    return ok;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan over the 256 threads of a workgroup. lds: CM_WAVES words. Ends with a barrier.
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* lds, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < CM_WAVES; ++k) {
        const uint32_t c = lds[k];
        if (k < w) woff += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t* lds) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < CM_WAVES; ++k) tot += lds[k];
    __syncthreads();
    return tot;
}

// Which sensor owns padded tile `tile` (wave-uniform).
__device__ __forceinline__ uint32_t sensor_of_tile(const CmFrameDev* __restrict__ fd, uint32_t tile) {
    const uint32_t first = tile * CM_TILE;
    uint32_t s = 0;
    for (uint32_t q = 1; q < fd->n_sensors; ++q) s += (first >= fd->s[q].base) ? 1u : 0u;
    return s;
}

// ------------------------------------------------------------------------------------------------
// k_setup: upload the frame descriptor (passed by value) into HBM.
// ------------------------------------------------------------------------------------------------
__global__ void k_setup(CmFrameDev f, CmFrameDev* __restrict__ dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = f;
}

// ------------------------------------------------------------------------------------------------
// K0: min/max of the transformed, cropped cloud (pcl::getMinMax3D, A.4 step 2)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_minmax(const CmFrameDev* __restrict__ fd,
                                                     CmFrameState* __restrict__ st) {
    __shared__ float s_red[CM_WAVES][6];
    __shared__ uint32_t s_cnt[CM_WAVES];
    const uint32_t tile = blockIdx.x;
    const uint32_t s = sensor_of_tile(fd, tile);
    const CmSensorDev& sd = fd->s[s];
    const unsigned char* data = sd.data;
    const uint32_t n = sd.n, layout = sd.layout, step = sd.point_step;
    const uint32_t ox = sd.off_x, oy = sd.off_y, oz = sd.off_z, oi = sd.off_i;
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
    const uint32_t crop = fd->crop_enable;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t first = tile * CM_TILE - sd.base + w * (64 * CM_ITEMS) + lane;

    const float inf = __uint_as_float(0x7F800000u);
    float mn0 = inf, mn1 = inf, mn2 = inf, mx0 = -inf, mx1 = -inf, mx2 = -inf;
    uint32_t cnt = 0;
#pragma unroll 4
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * 64;
        if (i < n) {
            const Pt p = load_point(data, layout, step, ox, oy, oz, oi, i);
            const float x = xf_row(m[0], m[1], m[2], m[3], p.x, p.y, p.z);
            const float y = xf_row(m[4], m[5], m[6], m[7], p.x, p.y, p.z);
            const float z = xf_row(m[8], m[9], m[10], m[11], p.x, p.y, p.z);
            if (point_valid(x, y, z, crop, fd->crop_min, fd->crop_max)) {
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
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = s_red[0][k];
        for (int q = 1; q < CM_WAVES; ++q) v = (k < 3) ? fminf(v, s_red[q][k]) : fmaxf(v, s_red[q][k]);
        const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (total) {
            const uint32_t e = enc_f32(v);
            atomicMax(&st->mm[k], (k < 3) ? ~e : e);
            if (k == 0) atomicAdd(&st->n_valid_k0, total);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Kb: overflow guard, min_b/max_b/div_b, key width (A.4 steps 3-4). One thread.
// ------------------------------------------------------------------------------------------------
__global__ void k_bounds(const CmFrameDev* __restrict__ fd, CmFrameState* __restrict__ st, int from_crop) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float min_p[3], max_p[3];
    if (from_crop) {
        for (int a = 0; a < 3; ++a) { min_p[a] = fd->crop_min[a]; max_p[a] = fd->crop_max[a]; }
    } else {
        if (st->n_valid_k0 == 0) { st->status = CM_DEV_EMPTY; return; }
        for (int a = 0; a < 3; ++a) { min_p[a] = dec_f32(~st->mm[a]); max_p[a] = dec_f32(st->mm[3 + a]); }
    }
    long long d[3];
    bool overflow = false;
    for (int a = 0; a < 3; ++a) {
        st->min_p[a] = min_p[a];
        st->max_p[a] = max_p[a];
        const float ext = __fmul_rn(__fsub_rn(max_p[a], min_p[a]), fd->inv_leaf[a]);
        if (!(ext < 2147483648.0f)) { overflow = true; d[a] = 0; }
        else d[a] = static_cast<long long>(ext) + 1;       // truncation toward zero
    }
    if (!overflow && d[0] * d[1] * d[2] > 2147483647LL) overflow = true;
    if (overflow) { st->status = CM_DEV_OVERFLOW; return; }
    unsigned long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        const int lo = static_cast<int>(floorf(__fmul_rn(min_p[a], fd->inv_leaf[a])));
        const int hi = static_cast<int>(floorf(__fmul_rn(max_p[a], fd->inv_leaf[a])));
        st->min_b[a] = lo; st->max_b[a] = hi; st->div_b[a] = hi - lo + 1;
        cells *= static_cast<unsigned long long>(hi - lo + 1);
    }
    // div_b can exceed the guard's dx by one per axis; the 32-bit key still has to hold it.
    if (cells > 0xFFFFFFFFull) { st->status = CM_DEV_OVERFLOW; return; }
    uint32_t bits = 1;
    while (bits < 32 && (cells - 1) >> bits) ++bits;
    st->key_bits = bits;
    st->n_passes = (bits + CM_RADIX_BITS - 1) / CM_RADIX_BITS;
}

// ------------------------------------------------------------------------------------------------
// K1: transform + crop + linear voxel index (A.4 step 5) + digit-0 histogram per tile.
// Keys of cropped / non-finite / padding slots are CM_INVALID_KEY and never enter the sort.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_keys(const CmFrameDev* __restrict__ fd,
                                                   const CmFrameState* __restrict__ st,
                                                   uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[CM_RADIX];
    if (st->status != CM_DEV_OK) return;
    const uint32_t tile = blockIdx.x;
    const uint32_t s = sensor_of_tile(fd, tile);
    const CmSensorDev& sd = fd->s[s];
    const unsigned char* data = sd.data;
    const uint32_t n = sd.n, layout = sd.layout, step = sd.point_step;
    const uint32_t ox = sd.off_x, oy = sd.off_y, oz = sd.off_z, oi = sd.off_i;
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
    const uint32_t crop = fd->crop_enable;
    const float inv0 = fd->inv_leaf[0], inv1 = fd->inv_leaf[1], inv2 = fd->inv_leaf[2];
    const float fb0 = static_cast<float>(st->min_b[0]), fb1 = static_cast<float>(st->min_b[1]),
                fb2 = static_cast<float>(st->min_b[2]);
    const uint32_t mul1 = static_cast<uint32_t>(st->div_b[0]);
    const uint32_t mul2 = mul1 * static_cast<uint32_t>(st->div_b[1]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t slot0 = tile * CM_TILE + w * (64 * CM_ITEMS) + lane;   // padded global index
    const uint32_t first = slot0 - sd.base;                                // index in the sensor cloud

    lh[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * 64;
        uint32_t key = CM_INVALID_KEY;
        if (i < n) {
            const Pt p = load_point(data, layout, step, ox, oy, oz, oi, i);
            const float x = xf_row(m[0], m[1], m[2], m[3], p.x, p.y, p.z);
            const float y = xf_row(m[4], m[5], m[6], m[7], p.x, p.y, p.z);
            const float z = xf_row(m[8], m[9], m[10], m[11], p.x, p.y, p.z);
            if (point_valid(x, y, z, crop, fd->crop_min, fd->crop_max)) {
                const int c0 = static_cast<int>(__fsub_rn(floorf(__fmul_rn(x, inv0)), fb0));
                const int c1 = static_cast<int>(__fsub_rn(floorf(__fmul_rn(y, inv1)), fb1));
                const int c2 = static_cast<int>(__fsub_rn(floorf(__fmul_rn(z, inv2)), fb2));
                key = static_cast<uint32_t>(c0) + static_cast<uint32_t>(c1) * mul1 +
                      static_cast<uint32_t>(c2) * mul2;
                atomicAdd(&lh[key & (CM_RADIX - 1)], 1u);
            }
        }
        keys[slot0 + r * 64] = key;
    }
    __syncthreads();
    hist[static_cast<size_t>(threadIdx.x) * fd->n_tiles + tile] = lh[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// LSD radix sort of (key, point index): one 8-bit digit per pass, 4096-item tiles in LDS.
// ------------------------------------------------------------------------------------------------
// Per-tile digit histogram for passes >= 1 (the compacted, partially sorted pairs).
__global__ __launch_bounds__(CM_BLOCK) void k_hist(const CmFrameState* __restrict__ st,
                                                   const uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ hist,
                                                   uint32_t pass, uint32_t n_tiles) {
    __shared__ uint32_t lh[CM_RADIX];
    if (st->status != CM_DEV_OK || pass >= st->n_passes) return;
    const uint32_t n = st->n_valid;
    const uint32_t shift = pass * CM_RADIX_BITS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * CM_TILE + w * (64 * CM_ITEMS) + lane;
    lh[threadIdx.x] = 0;
    __syncthreads();
    if (blockIdx.x * CM_TILE < n) {
#pragma unroll
        for (int r = 0; r < CM_ITEMS; ++r) {
            const uint32_t i = first + r * 64;
            if (i < n) atomicAdd(&lh[(keys[i] >> shift) & (CM_RADIX - 1)], 1u);
        }
    }
    __syncthreads();
    hist[static_cast<size_t>(threadIdx.x) * n_tiles + blockIdx.x] = lh[threadIdx.x];
}

// One workgroup per digit: exclusive scan of that digit's counts over the tiles + digit total.
__global__ __launch_bounds__(CM_BLOCK) void k_colscan(const CmFrameState* __restrict__ st,
                                                      uint32_t* __restrict__ hist,
                                                      uint32_t* __restrict__ totals,
                                                      uint32_t pass, uint32_t n_tiles) {
    __shared__ uint32_t lds[CM_WAVES];
    if (st->status != CM_DEV_OK || pass >= st->n_passes) return;
    uint32_t* row = hist + static_cast<size_t>(blockIdx.x) * n_tiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += CM_BLOCK) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = (t < n_tiles) ? row[t] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(v, lds, &tot);
        if (t < n_tiles) row[t] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
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

// Stable scatter of one tile by one digit. FIRST: values are the padded global indices and
// CM_INVALID_KEY slots are dropped (this is where the concatenated cloud gets compacted).
template <bool FIRST>
__global__ __launch_bounds__(CM_BLOCK) void k_scatter(CmFrameState* __restrict__ st,
                                                      const uint32_t* __restrict__ keys_in,
                                                      const uint32_t* __restrict__ vals_in,
                                                      uint32_t* __restrict__ keys_out,
                                                      uint32_t* __restrict__ vals_out,
                                                      const uint32_t* __restrict__ hist,
                                                      const uint32_t* __restrict__ totals,
                                                      uint32_t pass, uint32_t n_tiles, uint32_t n_padded) {
    __shared__ uint32_t whist[CM_WAVES][CM_RADIX];
    __shared__ uint32_t gofs[CM_RADIX];
    __shared__ uint32_t skey[CM_TILE];
    __shared__ uint32_t sval[CM_TILE];
    __shared__ uint32_t lds[CM_WAVES];
    __shared__ uint32_t s_tile_valid;
    if (st->status != CM_DEV_OK || pass >= st->n_passes) return;
    const uint32_t n = FIRST ? n_padded : st->n_valid;
    const uint32_t tile = blockIdx.x;
    const uint32_t shift = pass * CM_RADIX_BITS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t first = tile * CM_TILE + w * (64 * CM_ITEMS) + lane;

    // Global base of every digit: exclusive scan of the digit totals.
    uint32_t gtot;
    const uint32_t my_total = totals[threadIdx.x];
    const uint32_t gbase = block_excl_scan_u32(my_total, lds, &gtot);
    if (FIRST && tile == 0 && threadIdx.x == 0) st->n_valid = gtot;
    if (tile * CM_TILE >= n) return;                   // uniform: empty tile (after the scan's barriers)

#pragma unroll
    for (int q = 0; q < CM_WAVES; ++q) whist[q][threadIdx.x] = 0;

    uint32_t key[CM_ITEMS];
#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r) {
        const uint32_t i = first + r * 64;
        key[r] = (i < n) ? keys_in[i] : CM_INVALID_KEY;
    }
    __syncthreads();

    // Rank inside the wave, rounds in order, lanes in order: stable.
    volatile uint32_t* wh = whist[w];
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t rank[CM_ITEMS];
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
        gofs[d] = gbase + hist[static_cast<size_t>(d) * n_tiles + tile] - dbase;
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
            sval[pos] = FIRST ? i : vals_in[i];
        }
    }
    __syncthreads();

    const uint32_t tile_valid = s_tile_valid;
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

// ------------------------------------------------------------------------------------------------
// K3a: kept voxels per tile of the sorted keys. A run is owned by the tile holding its head; it is
// kept iff it reaches min_points_per_voxel (A.4 step 7), i.e. keys[head + min_pts - 1] == key.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ const uint32_t* pick(const CmFrameState* st, const uint32_t* a, const uint32_t* b) {
    return (st->n_passes & 1u) ? b : a;      // pass p reads A when p is even and writes the other
}

__global__ __launch_bounds__(CM_BLOCK) void k_seg_count(const CmFrameState* __restrict__ st,
                                                        const uint32_t* __restrict__ keys_a,
                                                        const uint32_t* __restrict__ keys_b,
                                                        uint32_t* __restrict__ tile_counts,
                                                        uint32_t min_pts) {
    __shared__ uint32_t lds[CM_WAVES];
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t* __restrict__ keys = pick(st, keys_a, keys_b);
    const uint32_t base = blockIdx.x * CM_SEG_TILE;
    uint32_t cnt = 0;
    if (base < n) {
#pragma unroll
        for (int j = 0; j < CM_SEG_ITEMS; ++j) {
            const uint32_t i = base + j * CM_BLOCK + threadIdx.x;
            if (i < n) {
                const uint32_t k = keys[i];
                const bool head = (i == 0) || (keys[i - 1] != k);
                bool keep = head;
                if (head && min_pts > 1) {
                    const uint32_t e = i + min_pts - 1;
                    keep = (e >= i) && (e < n) && (keys[e] == k);
                }
                cnt += keep ? 1u : 0u;
            }
        }
    }
    const uint32_t tot = block_sum_u32(cnt, lds);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------
// K3s: exclusive scan of the tile counts, n_out and the frame's final status; prepares the state
// buffer of the NEXT frame (zero) so no memset sits between frames.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_finalize(CmFrameState* __restrict__ st,
                                                       CmFrameState* __restrict__ st_next,
                                                       uint32_t* __restrict__ tile_counts,
                                                       uint32_t n_seg_tiles_max) {
    __shared__ uint32_t lds[CM_WAVES];
    if (st_next && threadIdx.x < sizeof(CmFrameState) / 4)
        reinterpret_cast<uint32_t*>(st_next)[threadIdx.x] = 0;
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    if (n == 0) {
        if (threadIdx.x == 0) { st->status = CM_DEV_EMPTY; st->n_out = 0; }
        return;
    }
    const uint32_t n_tiles = (n + CM_SEG_TILE - 1) / CM_SEG_TILE;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += CM_BLOCK) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = (t < n_tiles && t < n_seg_tiles_max) ? tile_counts[t] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(v, lds, &tot);
        if (t < n_tiles && t < n_seg_tiles_max) tile_counts[t] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { st->n_out = carry; st->n_seg_tiles = n_tiles; }
}

// ------------------------------------------------------------------------------------------------
// K3b: gather the points of a sorted tile, segmented centroid reduction, threshold, compaction.
// Thread t owns 8 consecutive sorted items; runs closed inside a thread are summed sequentially in
// sorted (= stable point) order, runs crossing threads by a wave64 segmented suffix scan (DPP
// shuffles), runs crossing waves through LDS, runs crossing the tile by a cooperative extension
// loop of the owning workgroup. No atomics: sums are deterministic.
// ------------------------------------------------------------------------------------------------
struct Acc { float x, y, z, i; uint32_t c; };

__device__ __forceinline__ void acc_add(Acc& a, const Acc& b) {
    a.x = __fadd_rn(a.x, b.x); a.y = __fadd_rn(a.y, b.y);
    a.z = __fadd_rn(a.z, b.z); a.i = __fadd_rn(a.i, b.i);
    a.c += b.c;
}
__device__ __forceinline__ Acc acc_shfl_down(const Acc& a, int d) {
    Acc r;
    r.x = __shfl_down(a.x, d); r.y = __shfl_down(a.y, d); r.z = __shfl_down(a.z, d);
    r.i = __shfl_down(a.i, d); r.c = __shfl_down(a.c, d);
    return r;
}

struct SensorLds {
    const unsigned char* data;
    uint32_t base, step, ox, oy, oz, oi, layout, _pad;
    float m[12];
};

__device__ __forceinline__ Pt gather_point(const SensorLds* __restrict__ tab, uint32_t n_sensors, uint32_t gidx) {
    uint32_t s = 0;
    for (uint32_t q = 1; q < n_sensors; ++q) s += (gidx >= tab[q].base) ? 1u : 0u;
    const SensorLds& sd = tab[s];
    const Pt p = load_point(sd.data, sd.layout, sd.step, sd.ox, sd.oy, sd.oz, sd.oi, gidx - sd.base);
    Pt o;
    o.x = xf_row(sd.m[0], sd.m[1], sd.m[2], sd.m[3], p.x, p.y, p.z);
    o.y = xf_row(sd.m[4], sd.m[5], sd.m[6], sd.m[7], p.x, p.y, p.z);
    o.z = xf_row(sd.m[8], sd.m[9], sd.m[10], sd.m[11], p.x, p.y, p.z);
    o.i = p.i;
    return o;
}

__global__ __launch_bounds__(CM_BLOCK) void k_seg_reduce(const CmFrameDev* __restrict__ fd,
                                                         const CmFrameState* __restrict__ st,
                                                         const uint32_t* __restrict__ keys_a,
                                                         const uint32_t* __restrict__ vals_a,
                                                         const uint32_t* __restrict__ keys_b,
                                                         const uint32_t* __restrict__ vals_b,
                                                         const uint32_t* __restrict__ tile_offs,
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
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t base = blockIdx.x * CM_SEG_TILE;
    if (base >= n) return;
    const uint32_t* __restrict__ keys = pick(st, keys_a, keys_b);
    const uint32_t* __restrict__ vals = pick(st, vals_a, vals_b);
    const uint32_t n_sensors = fd->n_sensors;
    const uint32_t min_pts = fd->min_pts > 1 ? fd->min_pts : 1u;
    const bool all_fields = fd->downsample_all != 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

    if (threadIdx.x < n_sensors) {
        const CmSensorDev& g = fd->s[threadIdx.x];
        SensorLds& t = tab[threadIdx.x];
        t.data = g.data; t.base = g.base; t.step = g.point_step;
        t.ox = g.off_x; t.oy = g.off_y; t.oz = g.off_z; t.oi = g.off_i; t.layout = g.layout;
        for (int k = 0; k < 12; ++k) t.m[k] = g.m[k];
    }
    __syncthreads();

    const uint32_t tile_n = min(static_cast<uint32_t>(CM_SEG_TILE), n - base);
    const uint32_t i0 = base + threadIdx.x * CM_SEG_ITEMS;

    // keys + previous key
    uint32_t k[CM_SEG_ITEMS];
    if (i0 + CM_SEG_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(keys + i0);
        const uint4 b = *reinterpret_cast<const uint4*>(keys + i0 + 4);
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < CM_SEG_ITEMS; ++j) k[j] = (i0 + j < n) ? keys[i0 + j] : 0u;
    }
    const uint32_t kprev = (i0 > 0 && i0 < n) ? keys[i0 - 1] : 0u;

    // gather + transform
    Pt p[CM_SEG_ITEMS];
    if (i0 + CM_SEG_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(vals + i0);
        const uint4 b = *reinterpret_cast<const uint4*>(vals + i0 + 4);
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < CM_SEG_ITEMS; ++j) p[j] = gather_point(tab, n_sensors, v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < CM_SEG_ITEMS; ++j) {
            if (i0 + j < n) p[j] = gather_point(tab, n_sensors, vals[i0 + j]);
            else { p[j].x = p[j].y = p[j].z = p[j].i = 0.f; }
        }
    }
    if (!all_fields) {
#pragma unroll
        for (int j = 0; j < CM_SEG_ITEMS; ++j) p[j].i = 0.f;
    }

    // head flags
    uint32_t heads = 0;          // bit j: item j starts a run
    uint32_t live = 0;           // bit j: item j exists
#pragma unroll
    for (int j = 0; j < CM_SEG_ITEMS; ++j) {
        const uint32_t i = i0 + j;
        if (i < n) {
            live |= 1u << j;
            const uint32_t pk = (j == 0) ? kprev : k[j - 1];
            if (i == 0 || pk != k[j]) heads |= 1u << j;
        }
    }

    // Extension: items after the tile that continue its last run (owned by this workgroup).
    Acc ext = {0.f, 0.f, 0.f, 0.f, 0u};
    {
        const uint32_t end = base + tile_n;
        const uint32_t klast = keys[end - 1];
        bool any = false;
        for (uint32_t off = 0;; off += CM_BLOCK) {
            const uint32_t j = end + off + threadIdx.x;
            const bool ok = (j < n) && (keys[j] == klast);
            if (ok) {
                Pt q = gather_point(tab, n_sensors, vals[j]);
                if (!all_fields) q.i = 0.f;
                ext.x = __fadd_rn(ext.x, q.x); ext.y = __fadd_rn(ext.y, q.y);
                ext.z = __fadd_rn(ext.z, q.z); ext.i = __fadd_rn(ext.i, q.i);
                ext.c += 1;
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
            const Acc it = {p[j].x, p[j].y, p[j].z, p[j].i, 1u};
            if (heads >> j & 1u) {
                if (open) { fin[j] = run; fkey[j] = run_key; fmask |= 1u << j; }
                run = it; run_key = k[j]; open = true;
            } else if (open) {
                acc_add(run, it);
            } else if (pre.c == 0) {
                pre = it;
            } else {
                acc_add(pre, it);
            }
        }
    }
    const bool has_head = open;

    // Wave64 segmented suffix scan of `pre`: S[t] = pre[t] + (has_head[t] ? 0 : S[t+1]).
    Acc S = pre;
    uint32_t flag = has_head ? 1u : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Acc o = acc_shfl_down(S, d);
        const uint32_t of = __shfl_down(flag, d);
        if (lane + d < 64 && !flag) {
            if (S.c == 0) S = o; else if (o.c) acc_add(S, o);
            flag |= of;
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
    uint32_t tot;
    uint32_t slot = tile_offs[blockIdx.x] + block_excl_scan_u32(nkeep, lds, &tot);
#pragma unroll
    for (int j = 0; j <= CM_SEG_ITEMS; ++j) {
        const bool is_last = (j == CM_SEG_ITEMS);
        const bool emit = is_last ? (has_head && run.c >= min_pts)
                                  : ((fmask >> (j & 7) & 1u) && fin[j & 7].c >= min_pts);
        if (emit) {
            const Acc a = is_last ? run : fin[j & 7];
            const uint32_t ak = is_last ? run_key : fkey[j & 7];
            const float c = static_cast<float>(a.c);
            float4 o;
            o.x = __fdiv_rn(a.x, c); o.y = __fdiv_rn(a.y, c);
            o.z = __fdiv_rn(a.z, c); o.w = __fdiv_rn(a.i, c);
            out[slot] = o;
            if (out_key) { out_key[slot] = ak; out_cnt[slot] = a.c; }
            ++slot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Merged-cloud materialisation (the reference's fused cloud, :137-142): stable compaction of the
// valid transformed points into 16-byte records. Used for CM_GRID_OVERFLOW (output = input) and
// by parity tests; not on the timed path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_merged_count(const CmFrameDev* __restrict__ fd,
                                                           uint32_t* __restrict__ tile_counts) {
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
            cnt += point_valid(x, y, z, fd->crop_enable, fd->crop_min, fd->crop_max) ? 1u : 0u;
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
                                                           float4* __restrict__ out) {
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
            ok = point_valid(o.x, o.y, o.z, fd->crop_enable, fd->crop_min, fd->crop_max);
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32(ok ? 1u : 0u, lds, &tot);
        if (ok) out[slot + ex] = o;
        slot += tot;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from cm_api.cpp)
// ------------------------------------------------------------------------------------------------
#define CM_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)

void cmk_setup(hipStream_t s, const CmFrameDev& f, CmFrameDev* d_frame) {
    CM_LAUNCH(k_setup, 1, 64, s, f, d_frame);
}
void cmk_minmax(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, uint32_t n_tiles) {
    CM_LAUNCH(k_minmax, n_tiles, CM_BLOCK, s, fd, st);
}
void cmk_bounds(hipStream_t s, const CmFrameDev* fd, CmFrameState* st, int from_crop) {
    CM_LAUNCH(k_bounds, 1, 64, s, fd, st, from_crop);
}
void cmk_keys(hipStream_t s, const CmFrameDev* fd, const CmFrameState* st, uint32_t* keys, uint32_t* hist,
              uint32_t n_tiles) {
    CM_LAUNCH(k_keys, n_tiles, CM_BLOCK, s, fd, st, keys, hist);
}
void cmk_hist(hipStream_t s, const CmFrameState* st, const uint32_t* keys, uint32_t* hist, uint32_t pass,
              uint32_t n_tiles) {
    CM_LAUNCH(k_hist, n_tiles, CM_BLOCK, s, st, keys, hist, pass, n_tiles);
}
void cmk_colscan(hipStream_t s, const CmFrameState* st, uint32_t* hist, uint32_t* totals, uint32_t pass,
                 uint32_t n_tiles) {
    CM_LAUNCH(k_colscan, CM_RADIX, CM_BLOCK, s, st, hist, totals, pass, n_tiles);
}
void cmk_scatter(hipStream_t s, CmFrameState* st, const uint32_t* keys_in, const uint32_t* vals_in,
                 uint32_t* keys_out, uint32_t* vals_out, const uint32_t* hist, const uint32_t* totals,
                 uint32_t pass, uint32_t n_tiles, uint32_t n_padded) {
    if (pass == 0)
        CM_LAUNCH(k_scatter<true>, n_tiles, CM_BLOCK, s, st, keys_in, vals_in, keys_out, vals_out, hist,
                  totals, pass, n_tiles, n_padded);
    else
        CM_LAUNCH(k_scatter<false>, n_tiles, CM_BLOCK, s, st, keys_in, vals_in, keys_out, vals_out, hist,
                  totals, pass, n_tiles, n_padded);
}
void cmk_seg_count(hipStream_t s, const CmFrameState* st, const uint32_t* keys_a, const uint32_t* keys_b,
                   uint32_t* tile_counts, uint32_t min_pts, uint32_t n_seg_tiles) {
    CM_LAUNCH(k_seg_count, n_seg_tiles, CM_BLOCK, s, st, keys_a, keys_b, tile_counts, min_pts);
}
void cmk_finalize(hipStream_t s, CmFrameState* st, CmFrameState* st_next, uint32_t* tile_counts,
                  uint32_t n_seg_tiles) {
    CM_LAUNCH(k_finalize, 1, CM_BLOCK, s, st, st_next, tile_counts, n_seg_tiles);
}
void cmk_seg_reduce(hipStream_t s, const CmFrameDev* fd, const CmFrameState* st, const uint32_t* keys_a,
                    const uint32_t* vals_a, const uint32_t* keys_b, const uint32_t* vals_b,
                    const uint32_t* tile_offs, void* out, uint32_t* out_key, uint32_t* out_cnt,
                    uint32_t n_seg_tiles) {
    CM_LAUNCH(k_seg_reduce, n_seg_tiles, CM_BLOCK, s, fd, st, keys_a, vals_a, keys_b, vals_b, tile_offs,
              reinterpret_cast<float4*>(out), out_key, out_cnt);
}
void cmk_merged(hipStream_t s, const CmFrameDev* fd, uint32_t* tile_counts, uint32_t* total, void* out,
                uint32_t n_tiles) {
    CM_LAUNCH(k_merged_count, n_tiles, CM_BLOCK, s, fd, tile_counts);
    CM_LAUNCH(k_scan_counts, 1, CM_BLOCK, s, tile_counts, n_tiles, total);
    CM_LAUNCH(k_merged_write, n_tiles, CM_BLOCK, s, fd, tile_counts, reinterpret_cast<float4*>(out));
}
