// cm_common.hpp — device helpers shared by the kernel files (cm_kernels.hip, cm_kernels_v2.hip):
// point loaders, the reference's fp32 transform / crop arithmetic, PCL's grid set-up, workgroup scans,
// the centroid accumulator and the frame-state report. Semantics in SURVEY.md Appendix A.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cm_device.h"

namespace {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
struct Pt { float x, y, z, i; };

// Sensor payloads live in HBM: say so, so the loads are global_load (vmcnt only), not flat_load.
#if defined(__HIP_DEVICE_COMPILE__)
#define CM_GLOBAL_AS __attribute__((address_space(1)))
#else
#define CM_GLOBAL_AS
#endif
typedef const CM_GLOBAL_AS unsigned char* cm_gptr;
typedef float cm_v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float load_f32_unaligned(cm_gptr p) {
    float f;
    __builtin_memcpy(&f, (const void*)p, 4);
    return f;
}

__device__ __forceinline__ Pt load_point(const unsigned char* __restrict__ data_generic, uint32_t layout,
                                         uint32_t step, uint32_t ox, uint32_t oy, uint32_t oz,
                                         uint32_t oi, uint32_t idx) {
    cm_gptr data = (cm_gptr)data_generic;
    typedef const CM_GLOBAL_AS cm_v4f* f4ptr;
    typedef const CM_GLOBAL_AS float* f1ptr;
    Pt p;
    if (layout == CM_LAYOUT_XYZI16) {
        const cm_v4f v = *(f4ptr)(data + static_cast<size_t>(idx) * 16);
        p.x = v.x; p.y = v.y; p.z = v.z; p.i = v.w;
    } else if (layout == CM_LAYOUT_PCL32) {
        cm_gptr q = data + static_cast<size_t>(idx) * 32;
        const cm_v4f v = *(f4ptr)q;
        p.x = v.x; p.y = v.y; p.z = v.z;
        p.i = *(f1ptr)(q + 16);
    } else {
        cm_gptr q = data + static_cast<size_t>(idx) * step;
        p.x = load_f32_unaligned(q + ox);
        p.y = load_f32_unaligned(q + oy);
        p.z = load_f32_unaligned(q + oz);
        p.i = (oi == 0xFFFFFFFFu) ? 0.0f : load_f32_unaligned(q + ((oi == 0xFFFFFFFFu) ? 0u : oi));
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

// Inclusive wave64 scan on the DPP datapath (no LDS traffic, 6 VALU adds): shifts by 1, 2, 4, 8 lanes
// inside each row of 16, then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2-3.
// A lane without a source reads the first operand (the identity).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int /*lane*/) {
    int x = static_cast<int>(v);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return static_cast<uint32_t>(x);
}
// Sum over the wave, in every lane.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_incl_scan_u32(v, 0)), 63));
}
// min / max over the wave, valid in lane 63 (same DPP ladder; a lane without a source keeps its own value).
__device__ __forceinline__ float wave_min_f32_l63(float v) {
#define CM_DPP_F(ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false))
    v = fminf(v, CM_DPP_F(0x111, 0xf)); v = fminf(v, CM_DPP_F(0x112, 0xf)); v = fminf(v, CM_DPP_F(0x114, 0xf));
    v = fminf(v, CM_DPP_F(0x118, 0xf)); v = fminf(v, CM_DPP_F(0x142, 0xa)); v = fminf(v, CM_DPP_F(0x143, 0xc));
    return v;
}
__device__ __forceinline__ float wave_max_f32_l63(float v) {
    v = fmaxf(v, CM_DPP_F(0x111, 0xf)); v = fmaxf(v, CM_DPP_F(0x112, 0xf)); v = fmaxf(v, CM_DPP_F(0x114, 0xf));
    v = fmaxf(v, CM_DPP_F(0x118, 0xf)); v = fmaxf(v, CM_DPP_F(0x142, 0xa)); v = fmaxf(v, CM_DPP_F(0x143, 0xc));
    return v;
#undef CM_DPP_F
}

// Rank of a record among the records of its wave with the same digit — this instruction's lower lanes and everything the wave
// ranked before — WITHOUT relying on the order in which the LDS executes the returning adds of one instruction (the default
// ranking does, after the device probe k_probe_lds_order has found them lane-ordered; this is what a context falls back to when
// the probe fails, CM_LDS_RANK=0, or the finish finds a pass mis-ranked). `row`: the wave's own counters, two 16-bit counters per
// word. nbits ballots find the lanes that share my digit; every lane reads its counter, then the lowest lane of each group
// adds the group's size (an atomic add only because two groups may share a word; LDS operations of one wave execute in order).
__device__ __forceinline__ uint32_t wave_rank_ballot(uint32_t* __restrict__ row, uint32_t digit, uint32_t nbits, bool has, int lane) {
    unsigned long long m = __ballot(has);
    for (uint32_t b = 0; b < nbits; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    const uint32_t sh = (digit & 1u) * 16u;
    uint32_t old = 0;
    if (has) old = (row[digit >> 1] >> sh) & 0xFFFFu;
    if (has && lane == __ffsll(static_cast<long long>(m)) - 1) atomicAdd(&row[digit >> 1], static_cast<uint32_t>(__popcll(m)) << sh);
    return old + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
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
    v = wave_sum_u32(v);
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
// Tile loader: the 16 points a thread owns in its tile (wave-striped), every load issued before
// the first use. Slots past the end of the cloud read as NaN and are never valid.
// ------------------------------------------------------------------------------------------------
// Branch-free: slots past the end of the cloud load the last point (a valid address) and are turned
// into NaN afterwards, so the N loads go out back to back and the compute waits for them one by one.
template <int LAYOUT, int N>
__device__ __forceinline__ void load_tile_points(const CmSensorDev& sd, uint32_t first, Pt (&p)[N]) {
    const unsigned char* __restrict__ data = sd.data;
    const uint32_t n = sd.n, step = sd.point_step;
    const uint32_t ox = sd.off_x, oy = sd.off_y, oz = sd.off_z, oi = sd.off_i;
    const float nan = __uint_as_float(0x7FC00000u);
    if (n == 0) {                                         // uniform; no tile belongs to an empty cloud anyway
#pragma unroll
        for (int r = 0; r < N; ++r) { p[r].x = nan; p[r].y = nan; p[r].z = nan; p[r].i = 0.f; }
        return;
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const uint32_t i = first + r * 64;
        p[r] = load_point(data, LAYOUT, step, ox, oy, oz, oi, i < n ? i : n - 1);
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const bool ok = first + r * 64 < n;
        p[r].x = ok ? p[r].x : nan; p[r].y = ok ? p[r].y : nan; p[r].z = ok ? p[r].z : nan;
        p[r].i = ok ? p[r].i : 0.f;
    }
}

// A point of the two aligned layouts at a 32-bit byte offset from a wave-uniform base (the tile's first point: the
// offset stays below CM_TILE * 32 bytes), so that the address is scalar base + one VGPR, not a 64-bit multiply-add per load.
// NT: non-temporal load (the last read of the bytes: do not keep them in the caches).
template <int LAYOUT, bool NT = false>
__device__ __forceinline__ Pt load_point_near(const unsigned char* __restrict__ data_generic, uint32_t idx) {
    cm_gptr data = (cm_gptr)data_generic;
    typedef const CM_GLOBAL_AS cm_v4f* f4ptr;
    typedef const CM_GLOBAL_AS float* f1ptr;
    Pt p;
    const uint32_t off = idx * (LAYOUT == CM_LAYOUT_XYZI16 ? 16u : 32u);
    const cm_v4f v = NT ? __builtin_nontemporal_load((f4ptr)(data + off)) : *(f4ptr)(data + off);
    p.x = v.x; p.y = v.y; p.z = v.z; p.i = v.w;
    if (LAYOUT == CM_LAYOUT_PCL32) p.i = *(f1ptr)(data + off + 16u);
    return p;
}

// The same from a tile entry (k_setup): `first` counts from the tile's first point.
template <int LAYOUT, int N, bool NT = false>
__device__ __forceinline__ void load_tile_raw(const unsigned char* __restrict__ data, uint32_t n, uint32_t step, uint32_t ox,
                                              uint32_t oy, uint32_t oz, uint32_t oi, uint32_t first, Pt (&p)[N]) {
    const float nan = __uint_as_float(0x7FC00000u);
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const uint32_t i = first + r * 64;
        if (LAYOUT == CM_LAYOUT_GENERIC) p[r] = load_point(data, LAYOUT, step, ox, oy, oz, oi, i < n ? i : n - 1);
        else p[r] = load_point_near<LAYOUT, NT>(data, i < n ? i : n - 1);
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const bool ok = first + r * 64 < n;
        p[r].x = ok ? p[r].x : nan; p[r].y = ok ? p[r].y : nan; p[r].z = ok ? p[r].z : nan;
        p[r].i = ok ? p[r].i : 0.f;
    }
}
template <int N, bool NT = false>
__device__ __forceinline__ void load_tile_te(const CmTileDev& te, const CmSensorDev& sd, uint32_t first, Pt (&p)[N]) {
    const uint32_t layout = te.info >> 8;
    if (layout == CM_LAYOUT_XYZI16) load_tile_raw<CM_LAYOUT_XYZI16, N, NT>(te.data, te.n_left, 16u, 0u, 4u, 8u, 12u, first, p);
    else if (layout == CM_LAYOUT_PCL32) load_tile_raw<CM_LAYOUT_PCL32, N, NT>(te.data, te.n_left, 32u, 0u, 4u, 8u, 16u, first, p);
    else load_tile_raw<CM_LAYOUT_GENERIC, N>(te.data, te.n_left, sd.point_step, sd.off_x, sd.off_y, sd.off_z, sd.off_i, first, p);
}

// N points of one lane, 64 apart (wave-striped), starting at index `first` of the sensor's cloud.
template <int N>
__device__ __forceinline__ void load_tile(const CmSensorDev& sd, uint32_t first, Pt (&p)[N]) {
    if (sd.layout == CM_LAYOUT_XYZI16) load_tile_points<CM_LAYOUT_XYZI16, N>(sd, first, p);
    else if (sd.layout == CM_LAYOUT_PCL32) load_tile_points<CM_LAYOUT_PCL32, N>(sd, first, p);
    else load_tile_points<CM_LAYOUT_GENERIC, N>(sd, first, p);
}

struct Grid {
    int status;
    uint32_t n_valid_k0;
    float min_p[3], max_p[3];
    int min_b[3], max_b[3], div_b[3];
    uint32_t key_bits, n_passes;
};

__device__ __forceinline__ void compute_grid(const CmFrameDev* __restrict__ fd,
                                             const float* __restrict__ partials, uint32_t n_partials,
                                             int from_crop, const float* __restrict__ inv, float (*s_red)[8], Grid& g) {
    g.status = CM_DEV_OK;
    g.key_bits = 0; g.n_passes = 0; g.n_valid_k0 = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { g.min_b[a] = 0; g.max_b[a] = 0; g.div_b[a] = 1; g.min_p[a] = 0.f; g.max_p[a] = 0.f; }
    if (from_crop == 2) {                               // bounds of the whole fused cloud, from the host
#pragma unroll
        for (int a = 0; a < 3; ++a) { g.min_p[a] = fd->ext_min[a]; g.max_p[a] = fd->ext_max[a]; }
    } else if (from_crop) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { g.min_p[a] = fd->crop_min[a]; g.max_p[a] = fd->crop_max[a]; }
    } else {
        // getMinMax3D: fold the per-workgroup records of k_minmax (min/max are exact and
        // order-independent, so every workgroup gets the same answer).
        const float inf = __uint_as_float(0x7F800000u);
        float v[6] = {inf, inf, inf, -inf, -inf, -inf};
        uint32_t cnt = 0;
        for (uint32_t r = threadIdx.x; r < n_partials; r += CM_BLOCK) {
            const float4 lo = *reinterpret_cast<const float4*>(partials + r * 8);
            const float4 hi = *reinterpret_cast<const float4*>(partials + r * 8 + 4);
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
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) s_red[w][k] = v[k];
            s_red[w][6] = __uint_as_float(cnt);
        }
        __syncthreads();
        cnt = 0;
#pragma unroll
        for (int q = 0; q < CM_WAVES; ++q) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = (q == 0) ? s_red[0][k] : fminf(v[k], s_red[q][k]);
#pragma unroll
            for (int k = 3; k < 6; ++k) v[k] = (q == 0) ? s_red[0][k] : fmaxf(v[k], s_red[q][k]);
            cnt += __float_as_uint(s_red[q][6]);
        }
        g.n_valid_k0 = cnt;
        if (cnt == 0) { g.status = CM_DEV_EMPTY; return; }
#pragma unroll
        for (int a = 0; a < 3; ++a) { g.min_p[a] = v[a]; g.max_p[a] = v[3 + a]; }
    }
    long long d[3];
    bool overflow = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = __fmul_rn(__fsub_rn(g.max_p[a], g.min_p[a]), inv[a]);
        if (!(ext < 2147483648.0f)) { overflow = true; d[a] = 0; }
        else d[a] = static_cast<long long>(ext) + 1;       // truncation toward zero
    }
    if (!overflow && d[0] * d[1] * d[2] > 2147483647LL) overflow = true;
    if (overflow) { g.status = CM_DEV_OVERFLOW; return; }
    unsigned long long cells = 1;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int lo = static_cast<int>(floorf(__fmul_rn(g.min_p[a], inv[a])));
        const int hi = static_cast<int>(floorf(__fmul_rn(g.max_p[a], inv[a])));
        g.min_b[a] = lo; g.max_b[a] = hi; g.div_b[a] = hi - lo + 1;
        cells *= static_cast<unsigned long long>(hi - lo + 1);
    }
    // div_b can exceed the guard's dx by one per axis; the 32-bit key still has to hold it.
    if (cells > 0xFFFFFFFFull) { g.status = CM_DEV_OVERFLOW; return; }
    uint32_t bits = 1;
    while (bits < 32 && (cells - 1) >> bits) ++bits;
    g.key_bits = bits;
    g.n_passes = (bits + CM_RADIX_BITS - 1) / CM_RADIX_BITS;
}

// ------------------------------------------------------------------------------------------------
// The box grid of the bucket path (crop box / predicted box; cm_api.cpp box_grid) and the cell of a point in it.
// ------------------------------------------------------------------------------------------------
struct BoxGrid {
    float inv0, inv1, inv2, fb0, fb1, fb2;
    uint32_t d0, d1, d2;
    uint32_t mul1, mul2l, mul2h;      // cells per x row; cells per z level = mul2h << 12 | mul2l
};

// use_cell: the radius grid of the outlier stage instead of the voxel grid
__device__ __forceinline__ BoxGrid box_grid_of(const CmFrameDev* __restrict__ fd, int use_cell = 0) {
    BoxGrid b;
    const int32_t* min_b = use_cell ? fd->cell_min_b : fd->box_min_b;
    const int32_t* div_b = use_cell ? fd->cell_div_b : fd->box_div_b;
    const float* inv = use_cell ? fd->inv_cell : fd->inv_leaf;
    b.inv0 = inv[0]; b.inv1 = inv[1]; b.inv2 = inv[2];
    b.fb0 = static_cast<float>(min_b[0]); b.fb1 = static_cast<float>(min_b[1]); b.fb2 = static_cast<float>(min_b[2]);
    b.d0 = static_cast<uint32_t>(div_b[0]); b.d1 = static_cast<uint32_t>(div_b[1]); b.d2 = static_cast<uint32_t>(div_b[2]);
    b.mul1 = b.d0;
    const uint32_t mul2 = b.d0 * b.d1;
    b.mul2l = mul2 & 0xFFFu; b.mul2h = mul2 >> 12;
    return b;
}

// Linear index c0 + c1 * d0 + c2 * d0 * d1 (mod 2^32; exact for a cell of the box: the host only takes the bucket
// path when the box has fewer than 2^32 cells and fewer than 2^24 per axis — cm_api.cpp) on the full-rate 24-bit
// multiplier: v_mul_lo_u32 / v_mad_u64_u32 run at a quarter of the rate, and the index is formed several times per point.
__device__ __forceinline__ uint32_t box_index(const BoxGrid& b, uint32_t c0, uint32_t c1, uint32_t c2) {
    uint32_t hi = __umul24(c2, b.mul2h);
    asm("" : "+v"(hi));               // (keeps the shift behind the product: folded into the factor it would need the 32-bit multiplier again)
    return c0 + __umul24(c1, b.mul1) + __umul24(c2, b.mul2l) + (hi << 12);
}

// Cell of a transformed point, PCL's arithmetic (A.4 step 5). `inside`: the cell lies in the box (false for a
// non-finite coordinate as well: NaN converts to 0, so finiteness is asked separately by the callers that need it).
__device__ __forceinline__ uint32_t key_of(const BoxGrid& b, float x, float y, float z, bool* inside) {
    const uint32_t c0 = static_cast<uint32_t>(static_cast<int>(__fsub_rn(floorf(__fmul_rn(x, b.inv0)), b.fb0)));
    const uint32_t c1 = static_cast<uint32_t>(static_cast<int>(__fsub_rn(floorf(__fmul_rn(y, b.inv1)), b.fb1)));
    const uint32_t c2 = static_cast<uint32_t>(static_cast<int>(__fsub_rn(floorf(__fmul_rn(z, b.inv2)), b.fb2)));
    *inside = (c0 < b.d0) & (c1 < b.d1) & (c2 < b.d2);          // (unsigned: a negative cell is a huge one)
    return box_index(b, c0, c1, c2);
}
// ... of a record known to lie in the box
__device__ __forceinline__ uint32_t key_of(const BoxGrid& b, const float4& r) {
    const uint32_t c0 = static_cast<uint32_t>(static_cast<int>(__fsub_rn(floorf(__fmul_rn(r.x, b.inv0)), b.fb0)));
    const uint32_t c1 = static_cast<uint32_t>(static_cast<int>(__fsub_rn(floorf(__fmul_rn(r.y, b.inv1)), b.fb1)));
    const uint32_t c2 = static_cast<uint32_t>(static_cast<int>(__fsub_rn(floorf(__fmul_rn(r.z, b.inv2)), b.fb2)));
    return box_index(b, c0, c1, c2);
}

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

// The workgroup that knows the frame's final numbers writes the whole state record straight into
// pinned host memory (visible to the host when the kernel completes): no copy after the frame.
__device__ __forceinline__ void report_state(uint32_t* __restrict__ host, const CmFrameState* __restrict__ st,
                                             int status, uint32_t n_out, bool skip_err = false) {
    if (threadIdx.x < sizeof(CmFrameState) / 4 && !(skip_err && threadIdx.x == offsetof(CmFrameState, err) / 4)) {
        uint32_t wv = reinterpret_cast<const uint32_t*>(st)[threadIdx.x];
        if (threadIdx.x == offsetof(CmFrameState, status) / 4) wv = static_cast<uint32_t>(status);
        if (threadIdx.x == offsetof(CmFrameState, n_out) / 4) wv = n_out;
        host[threadIdx.x] = wv;
    }
}

}  // namespace
