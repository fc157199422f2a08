// cm_kernels_ground.hip — zone-wise ground removal before the fuse for gfx950.
//
// What the reference does per sensor between getROI and the fuse (proceedFront / proceedRear,
// pc_preprocessing_main.cpp:228-312; top-middle and Livox :436-497): cut the cropped cloud into x-slabs
// (getCloudPart :49-59) and in each slab fit one plane to the points of a z band with RANSAC (removeGround
// :71-122); the plane's inliers are "ground", the rest of the band plus the part above it "no ground".
// Here all slabs of all sensors are handled in one go, in the padded point index space of the frame:
//   kg_classify  transform + crop + slab/band of every point; band points get their slab number as a sort
//                key (one 8-bit digit: counts per tile like k_keys), points kept without a plane fit are
//                marked in the keep-mask                                         [16 B/pt read, 4 B/pt write]
//   k_scatter    (cm_kernels.hip, pass 0) groups the band points by slab, stable: (slab, point index) pairs
//   kg_gather    slab offsets + transformed xyz of the band points in that order
//   kg_ransac    one workgroup per slab: hypotheses in rounds of 32 (three sample points each, plane in fp32
//                as SampleConsensusModelPlane does), inlier counts by wave ballots, PCL's accept/stop loop
//                run over the counts, least-squares refit of the inliers (fp64 sums in a fixed blocked order,
//                Jacobi), final inlier set -> ground mask / keep mask
// The masks then drive the voxel path (k_minmax / k_keys honour the keep-mask) and the fused clouds
// (k_merged_*). The arithmetic is specified in oracle/cm_oracle.h (orc_ransac_plane) and restated here.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cm_common.hpp"
#include "cm_device.h"
#include "cm_kernels.h"

namespace {

__global__ void kg_setup(CmGroundDev g, CmGroundDev* __restrict__ dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = g;
}

// ------------------------------------------------------------------------------------------------
// kg_classify: k_keys' role for the slab sort (grid bookkeeping, clears, digit-0 counts), with the
// slab number as key.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void kg_classify(const CmFrameDev* __restrict__ fd,
                                                        const CmGroundDev* __restrict__ gd,
                                                        CmFrameState* __restrict__ st,
                                                        uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ hist,
                                                        uint32_t* __restrict__ grp_acc,
                                                        uint32_t* __restrict__ grp_clear_a,
                                                        uint32_t* __restrict__ grp_clear_b,
                                                        uint32_t n_group_words, uint32_t n_clear_a_words,
                                                        unsigned char* __restrict__ keep_mask,
                                                        unsigned char* __restrict__ zcode) {
    __shared__ uint32_t lh[CM_RADIX];
    const uint32_t tile = blockIdx.x;
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < 3 * n_group_words; k += gridDim.x * CM_BLOCK) grp_clear_b[k] = 0;
    for (uint32_t k = tile * CM_BLOCK + threadIdx.x; k < n_clear_a_words; k += gridDim.x * CM_BLOCK) grp_clear_a[k] = 0;
    if (tile == 0 && threadIdx.x == 0) {
        st->status = CM_DEV_OK;
        st->key_bits = 8;
        st->n_passes = 1;
    }
    const CmSensorDev& sd = fd->s[sensor_of_tile(fd, tile)];
    const uint32_t s = sd.slot;                          // slab tables, zone keys and planes go by the caller's sensor number
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = sd.m[k];
    const uint32_t crop = fd->crop_enable;
    const uint32_t nz = gd->n_zones[s];
    const float zkeep = gd->z_keep_max;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t slot0 = tile * CM_TILE + w * (64 * CM_ITEMS) + lane;
    Pt p[CM_ITEMS];
    load_tile<CM_ITEMS>(sd, slot0 - sd.base, p);
    lh[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CM_ITEMS; ++r) {
        uint32_t key = CM_INVALID_KEY;
        bool keep = false;
        const float x = xf_row(m[0], m[1], m[2], m[3], p[r].x, p[r].y, p[r].z);
        const float y = xf_row(m[4], m[5], m[6], m[7], p[r].x, p[r].y, p[r].z);
        const float z = xf_row(m[8], m[9], m[10], m[11], p[r].x, p[r].y, p[r].z);
        if (point_valid(x, y, z, crop, fd->crop_min, fd->crop_max)) {
            for (uint32_t q = 0; q < nz; ++q) {                 // first slab that holds the point (PassThrough: closed interval)
                if (!(x < gd->x0[s][q] || x > gd->x1[s][q])) {
                    const float zm = gd->zmax[s][q];
                    if (zm < 0.0f) keep = true;                                       // slab without ground removal
                    else if (!(z < -zm || z > zm)) key = s * CM_DEV_MAX_ZONES + q;    // band: the plane decides
                    else if (!(z < gd->zlo[s][q] || z > zkeep)) keep = true;          // part above the band
                    break;
                }
            }
        }
        if (key != CM_INVALID_KEY) atomicAdd(&lh[key & (CM_RADIX - 1)], 1u);
        keys[slot0 + r * 64] = key;
        if (keep) keep_mask[slot0 + r * 64] = 1;                                      // masks are zeroed before the stage
        if (zcode && key != CM_INVALID_KEY) zcode[slot0 + r * 64] = static_cast<unsigned char>(key);   // slab of a band point
    }
    __syncthreads();
    const uint32_t c = lh[threadIdx.x];
    hist[static_cast<size_t>(tile) * CM_RADIX + threadIdx.x] = c;
    if (c) atomicAdd(&grp_acc[static_cast<size_t>(tile / CM_GROUP) * CM_RADIX + threadIdx.x], c);
}

// ------------------------------------------------------------------------------------------------
// kg_gather: band points in slab order (transformed xyz, w = padded index), and where each slab starts.
// ------------------------------------------------------------------------------------------------
struct SensorLdsG {
    const unsigned char* data;
    uint32_t base, step, ox, oy, oz, oi, layout, _pad;
    float m[12];
};

__global__ __launch_bounds__(CM_BLOCK) void kg_gather(const CmFrameDev* __restrict__ fd,
                                                      const CmFrameState* __restrict__ st,
                                                      const uint32_t* __restrict__ keys_sorted,
                                                      const uint32_t* __restrict__ vals_sorted,
                                                      float4* __restrict__ band_pts,
                                                      uint32_t* __restrict__ zone_off /* 129 */) {
    __shared__ SensorLdsG tab[CM_DEV_MAX_SENSORS];
    const uint32_t n_sensors = fd->n_sensors;
    if (threadIdx.x < n_sensors) {
        const CmSensorDev& g = fd->s[threadIdx.x];
        SensorLdsG& t = tab[threadIdx.x];
        t.data = g.data; t.base = g.base; t.step = g.point_step;
        t.ox = g.off_x; t.oy = g.off_y; t.oz = g.off_z; t.oi = g.off_i; t.layout = g.layout;
        for (int k = 0; k < 12; ++k) t.m[k] = g.m[k];
    }
    __syncthreads();
    const uint32_t n = st->n_valid;
    if (blockIdx.x == 0 && threadIdx.x <= CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES) {
        // first sorted position whose slab number is >= threadIdx.x
        uint32_t a = 0, b = n;
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if (keys_sorted[mid] < threadIdx.x) a = mid + 1; else b = mid;
        }
        zone_off[threadIdx.x] = a;
    }
    for (uint32_t p = blockIdx.x * CM_BLOCK + threadIdx.x; p < n; p += gridDim.x * CM_BLOCK) {
        const uint32_t idx = vals_sorted[p];
        uint32_t s = 0;
        for (uint32_t q = 1; q < n_sensors; ++q) s += (idx >= tab[q].base) ? 1u : 0u;
        const SensorLdsG& sd = tab[s];
        const Pt pt = load_point(sd.data, sd.layout, sd.step, sd.ox, sd.oy, sd.oz, sd.oi, idx - sd.base);
        band_pts[p] = make_float4(xf_row(sd.m[0], sd.m[1], sd.m[2], sd.m[3], pt.x, pt.y, pt.z),
                                  xf_row(sd.m[4], sd.m[5], sd.m[6], sd.m[7], pt.x, pt.y, pt.z),
                                  xf_row(sd.m[8], sd.m[9], sd.m[10], sd.m[11], pt.x, pt.y, pt.z), __uint_as_float(idx));
    }
}

// ------------------------------------------------------------------------------------------------
// kg_ransac: one workgroup of 256 threads per slab. Arithmetic: oracle/cm_oracle.h, orc_ransac_plane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ void sample3(unsigned long long seed, uint32_t zone_key, uint32_t j, unsigned long long n,
                                        uint32_t idx[3]) {
    const unsigned long long base = seed ^ (static_cast<unsigned long long>(zone_key) << 40) ^ (static_cast<unsigned long long>(j) << 2);
    const unsigned long long i0 = __umul64hi(splitmix64(base + 0), n);
    unsigned long long i1 = __umul64hi(splitmix64(base + 1), n - 1);
    if (i1 >= i0) ++i1;
    unsigned long long i2 = __umul64hi(splitmix64(base + 2), n - 2);
    const unsigned long long lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
    if (i2 >= lo) ++i2;
    if (i2 >= hi) ++i2;
    idx[0] = static_cast<uint32_t>(i0); idx[1] = static_cast<uint32_t>(i1); idx[2] = static_cast<uint32_t>(i2);
}

// SampleConsensusModelPlane::computeModelCoefficients in fp32 (every operation rounded once)
__device__ __forceinline__ bool plane_from_3(const float4& p0, const float4& p1, const float4& p2, float pl[4]) {
    const float ax = __fsub_rn(p1.x, p0.x), ay = __fsub_rn(p1.y, p0.y), az = __fsub_rn(p1.z, p0.z);
    const float bx = __fsub_rn(p2.x, p0.x), by = __fsub_rn(p2.y, p0.y), bz = __fsub_rn(p2.z, p0.z);
    const float r0 = __fdiv_rn(ax, bx), r1 = __fdiv_rn(ay, by), r2 = __fdiv_rn(az, bz);
    if (r0 == r1 && r2 == r1) return false;
    float nx = __fsub_rn(__fmul_rn(ay, bz), __fmul_rn(az, by));
    float ny = __fsub_rn(__fmul_rn(az, bx), __fmul_rn(ax, bz));
    float nz = __fsub_rn(__fmul_rn(ax, by), __fmul_rn(ay, bx));
    // (__builtin_sqrtf: the correctly rounded square root; HIP's __fsqrt_rn is the 1-ulp native instruction)
    const float len = __builtin_sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)));
    if (!(len > 0.0f)) return false;
    nx = __fdiv_rn(nx, len); ny = __fdiv_rn(ny, len); nz = __fdiv_rn(nz, len);
    pl[0] = nx; pl[1] = ny; pl[2] = nz;
    pl[3] = __fmul_rn(-1.0f, __fadd_rn(__fadd_rn(__fmul_rn(nx, p0.x), __fmul_rn(ny, p0.y)), __fmul_rn(nz, p0.z)));
    return finite_f32(pl[0]) && finite_f32(pl[1]) && finite_f32(pl[2]) && finite_f32(pl[3]);
}
__device__ __forceinline__ bool plane_inlier(float a, float b, float c, float d, const float4& p, float thr) {
    return fabsf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, p.x), __fmul_rn(b, p.y)), __fmul_rn(c, p.z)), d)) < thr;
}

__device__ void jacobi3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int r = 0; r < 3; ++r) {
            const int p = P[r], q = Q[r];
            const double apq = a[p][q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = __ddiv_rn(__dsub_rn(a[q][q], a[p][p]), __dmul_rn(2.0, apq));
            const double t = __ddiv_rn(theta >= 0.0 ? 1.0 : -1.0,
                                       __dadd_rn(fabs(theta), __dsqrt_rn(__dadd_rn(__dmul_rn(theta, theta), 1.0))));
            const double c = __ddiv_rn(1.0, __dsqrt_rn(__dadd_rn(__dmul_rn(t, t), 1.0))), s = __dmul_rn(t, c);
            for (int k = 0; k < 3; ++k) {
                const double akp = a[k][p], akq = a[k][q];
                a[k][p] = __dsub_rn(__dmul_rn(c, akp), __dmul_rn(s, akq));
                a[k][q] = __dadd_rn(__dmul_rn(s, akp), __dmul_rn(c, akq));
            }
            for (int k = 0; k < 3; ++k) {
                const double apk = a[p][k], aqk = a[q][k];
                a[p][k] = __dsub_rn(__dmul_rn(c, apk), __dmul_rn(s, aqk));
                a[q][k] = __dadd_rn(__dmul_rn(s, apk), __dmul_rn(c, aqk));
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = v[k][p], vkq = v[k][q];
                v[k][p] = __dsub_rn(__dmul_rn(c, vkp), __dmul_rn(s, vkq));
                v[k][q] = __dadd_rn(__dmul_rn(s, vkp), __dmul_rn(c, vkq));
            }
        }
}

__global__ __launch_bounds__(256) void kg_ransac(const CmGroundDev* __restrict__ gd,
                                                 const CmFrameState* __restrict__ st,
                                                 const float4* __restrict__ band_pts,
                                                 const uint32_t* __restrict__ zone_off,
                                                 const float4* __restrict__ hyp0,
                                                 const uint32_t* __restrict__ valid0,
                                                 const uint32_t* __restrict__ counts0,
                                                 CmGroundPlaneDev* __restrict__ planes) {
    __shared__ float s_pl[CM_GROUND_BATCH][4];
    __shared__ uint32_t s_valid[CM_GROUND_BATCH];
    __shared__ uint32_t s_cnt[CM_GROUND_BATCH];
    __shared__ float s_best[4];
    __shared__ uint32_t s_go, s_found, s_iter;
    const uint32_t zone = blockIdx.x;
    const uint32_t z0 = zone_off[zone], z1 = zone_off[zone + 1];
    const uint32_t n = (st->status == CM_DEV_OK) ? z1 - z0 : 0u;
    const float4* __restrict__ pts = band_pts + z0;
    const int lane = threadIdx.x & 63;
    const float thr = gd->threshold;
    if (threadIdx.x == 0) {
        CmGroundPlaneDev& o = planes[zone];
        o.plane[0] = o.plane[1] = o.plane[2] = o.plane[3] = 0.f;
        o.band_points = n; o.inliers = 0; o.iterations = 0; o.found = 0;
    }
    if (n < 3) return;                                    // PCL: not enough points for a model — found stays 0
    const uint32_t max_iter = gd->max_iterations;
    const uint32_t J = max_iter + CM_GROUND_SPARE;
    // PCL's loop state (thread 0)
    uint32_t iterations = 0, skipped = 0, best_j = 0;
    long long best = -1;
    double pno = 1.0, pw = 1.0;
    const double stop = __dsub_rn(1.0, static_cast<double>(gd->probability));
    for (uint32_t j0 = 0; j0 < J; j0 += CM_GROUND_BATCH) {
        if (j0 == 0) {
            // the first round was scored by kg_score0, all slabs at once
            if (threadIdx.x < CM_GROUND_BATCH) {
                const float4 h = hyp0[zone * CM_GROUND_BATCH + threadIdx.x];
                s_pl[threadIdx.x][0] = h.x; s_pl[threadIdx.x][1] = h.y; s_pl[threadIdx.x][2] = h.z; s_pl[threadIdx.x][3] = h.w;
                s_valid[threadIdx.x] = valid0[zone * CM_GROUND_BATCH + threadIdx.x];
                s_cnt[threadIdx.x] = counts0[zone * CM_GROUND_BATCH + threadIdx.x];
            }
            __syncthreads();
        } else {
        if (threadIdx.x < CM_GROUND_BATCH) {
            const uint32_t j = j0 + threadIdx.x;
            uint32_t idx[3];
            float pl[4] = {0.f, 0.f, 0.f, 0.f};
            bool ok = false;
            if (j < J) {
                sample3(gd->seed, zone, j, n, idx);
                ok = plane_from_3(pts[idx[0]], pts[idx[1]], pts[idx[2]], pl);
            }
            s_pl[threadIdx.x][0] = pl[0]; s_pl[threadIdx.x][1] = pl[1]; s_pl[threadIdx.x][2] = pl[2]; s_pl[threadIdx.x][3] = pl[3];
            s_valid[threadIdx.x] = ok ? 1u : 0u;
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        // inlier counts of the 32 hypotheses: every wave walks its share of the points, one ballot per plane
        uint32_t mine = 0;                                // lane h (< 32) of each wave collects plane h
        for (uint32_t i0 = (threadIdx.x >> 6) * 64; i0 < n; i0 += 256) {
            const uint32_t i = i0 + lane;
            const bool have = i < n;
            const float4 p = have ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int h = 0; h < CM_GROUND_BATCH; ++h) {
                const bool in = have && plane_inlier(s_pl[h][0], s_pl[h][1], s_pl[h][2], s_pl[h][3], p, thr);
                const uint32_t c = static_cast<uint32_t>(__popcll(__ballot(in)));
                if (lane == h) mine += c;
            }
        }
        if (lane < CM_GROUND_BATCH && mine) atomicAdd(&s_cnt[lane], mine);
        __syncthreads();
        }
        if (threadIdx.x == 0) {
            bool go = true;
            for (int h = 0; h < CM_GROUND_BATCH && go; ++h) {
                const uint32_t j = j0 + h;
                if (j != iterations + skipped) break;      // (always equal: the loop consumes hypotheses in order)
                if (j >= J) { go = false; break; }
                if (!s_valid[h]) { ++skipped; continue; }
                const long long c = s_cnt[h];
                bool updated = false;
                if (c > best) {
                    best = c; best_j = j;
                    s_best[0] = s_pl[h][0]; s_best[1] = s_pl[h][1]; s_best[2] = s_pl[h][2]; s_best[3] = s_pl[h][3];
                    const double wq = __ddiv_rn(static_cast<double>(c), static_cast<double>(n));
                    pno = __dsub_rn(1.0, __dmul_rn(__dmul_rn(wq, wq), wq));
                    if (pno < 2.220446049250313e-16) pno = 2.220446049250313e-16;
                    if (pno > 1.0 - 2.220446049250313e-16) pno = 1.0 - 2.220446049250313e-16;
                    updated = true;
                }
                ++iterations;
                if (updated) { pw = 1.0; for (uint32_t t = 0; t < iterations; ++t) pw = __dmul_rn(pw, pno); }
                else pw = __dmul_rn(pw, pno);
                if (iterations > max_iter) go = false;
                else if (!(pw > stop)) go = false;
            }
            if (iterations + skipped >= J) go = false;
            s_go = go ? 1u : 0u;
            s_found = best >= 0 ? 1u : 0u;
            s_iter = iterations;
        }
        __syncthreads();
        if (!s_go) break;
    }
    if (!s_found) {                                        // no valid sample at all: no plane
        if (threadIdx.x == 0) planes[zone].iterations = s_iter;
        return;
    }
    if (threadIdx.x == 0) {                                // the best sample's plane; kg_refit_* refines it
        CmGroundPlaneDev& o = planes[zone];
        o.plane[0] = s_best[0]; o.plane[1] = s_best[1]; o.plane[2] = s_best[2]; o.plane[3] = s_best[3];
        o.iterations = s_iter; o.found = 1;
        (void)best_j;
    }
}

// optimizeModelCoefficients, part 1: sums over the inliers of the best sample's plane, one workgroup per chunk
// of CM_GROUND_CHUNK band points of a slab, in the order oracle/cm_oracle.h fixes (element i of the chunk ->
// partial i mod 256, partials added pairwise). Workgroup b finds its (slab, chunk) from the slab offsets.
__device__ __forceinline__ bool chunk_of_block(const uint32_t* __restrict__ zone_off, uint32_t blk, uint32_t* zone,
                                               uint32_t* lo, uint32_t* hi) {
    uint32_t first = 0;
    for (uint32_t z = 0; z < CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES; ++z) {
        const uint32_t z0 = zone_off[z], z1 = zone_off[z + 1];
        const uint32_t nch = (z1 - z0 + CM_GROUND_CHUNK - 1) / CM_GROUND_CHUNK;
        if (blk < first + nch) {
            *zone = z;
            *lo = z0 + (blk - first) * CM_GROUND_CHUNK;
            *hi = min(z1, *lo + CM_GROUND_CHUNK);
            return true;
        }
        first += nch;
    }
    return false;
}

__global__ __launch_bounds__(256) void kg_refit_part(const CmGroundDev* __restrict__ gd, const CmFrameState* __restrict__ st,
                                                     const float4* __restrict__ band_pts, const uint32_t* __restrict__ zone_off,
                                                     const CmGroundPlaneDev* __restrict__ planes, double* __restrict__ chunk_sums) {
    __shared__ double s_part[256][10];
    __shared__ uint32_t s_zone, s_lo, s_hi, s_ok;
    if (st->status != CM_DEV_OK || !gd->optimize) return;
    if (threadIdx.x == 0) {
        uint32_t z = 0, lo = 0, hi = 0;
        s_ok = chunk_of_block(zone_off, blockIdx.x, &z, &lo, &hi) ? 1u : 0u;
        s_zone = z; s_lo = lo; s_hi = hi;
    }
    __syncthreads();
    if (!s_ok) return;
    const CmGroundPlaneDev& pl = planes[s_zone];
    if (!pl.found) return;
    const float a = pl.plane[0], b = pl.plane[1], c = pl.plane[2], d = pl.plane[3], thr = gd->threshold;
    const uint32_t lo = s_lo, hi = s_hi;
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i4 = lo + threadIdx.x; i4 < hi; i4 += 1024) {       // four loads in flight; order inside a partial kept
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = (i4 + u * 256 < hi) ? band_pts[i4 + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 p = q[u];
            if (i4 + u * 256 < hi && plane_inlier(a, b, c, d, p, thr)) {
                const double x = p.x, y = p.y, z = p.z;
                acc[0] = __dadd_rn(acc[0], x); acc[1] = __dadd_rn(acc[1], y); acc[2] = __dadd_rn(acc[2], z);
                acc[3] = __dadd_rn(acc[3], __dmul_rn(x, x)); acc[4] = __dadd_rn(acc[4], __dmul_rn(x, y));
                acc[5] = __dadd_rn(acc[5], __dmul_rn(x, z)); acc[6] = __dadd_rn(acc[6], __dmul_rn(y, y));
                acc[7] = __dadd_rn(acc[7], __dmul_rn(y, z)); acc[8] = __dadd_rn(acc[8], __dmul_rn(z, z));
                acc[9] = __dadd_rn(acc[9], 1.0);
            }
        }
    }
    for (int k = 0; k < 10; ++k) s_part[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if (static_cast<int>(threadIdx.x) < stride)
            for (int k = 0; k < 10; ++k) s_part[threadIdx.x][k] = __dadd_rn(s_part[threadIdx.x][k], s_part[threadIdx.x + stride][k]);
        __syncthreads();
    }
    if (threadIdx.x < 10) chunk_sums[static_cast<size_t>(blockIdx.x) * 10 + threadIdx.x] = s_part[0][threadIdx.x];
}

// part 2, one thread per slab: chunk sums added in chunk order, covariance, smallest eigenvector (Jacobi).
__global__ __launch_bounds__(128) void kg_refit_final(const CmGroundDev* __restrict__ gd, const CmFrameState* __restrict__ st,
                                                      const uint32_t* __restrict__ zone_off, const double* __restrict__ chunk_sums,
                                                      CmGroundPlaneDev* __restrict__ planes) {
    if (st->status != CM_DEV_OK || !gd->optimize) return;
    const uint32_t zone = threadIdx.x;
    if (zone >= CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES || !planes[zone].found) return;
    uint32_t first = 0;
    for (uint32_t z = 0; z < zone; ++z) first += (zone_off[z + 1] - zone_off[z] + CM_GROUND_CHUNK - 1) / CM_GROUND_CHUNK;
    const uint32_t nch = (zone_off[zone + 1] - zone_off[zone] + CM_GROUND_CHUNK - 1) / CM_GROUND_CHUNK;
    double S[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t q = 0; q < nch; ++q)
        for (int k = 0; k < 10; ++k) S[k] = __dadd_rn(S[k], chunk_sums[static_cast<size_t>(first + q) * 10 + k]);
    if (!(S[9] > 3.0)) return;
    const double cnt = S[9];
    const double mx = __ddiv_rn(S[0], cnt), my = __ddiv_rn(S[1], cnt), mz = __ddiv_rn(S[2], cnt);
    double A[3][3], V[3][3];
    A[0][0] = __dsub_rn(__ddiv_rn(S[3], cnt), __dmul_rn(mx, mx)); A[0][1] = __dsub_rn(__ddiv_rn(S[4], cnt), __dmul_rn(mx, my));
    A[0][2] = __dsub_rn(__ddiv_rn(S[5], cnt), __dmul_rn(mx, mz)); A[1][1] = __dsub_rn(__ddiv_rn(S[6], cnt), __dmul_rn(my, my));
    A[1][2] = __dsub_rn(__ddiv_rn(S[7], cnt), __dmul_rn(my, mz)); A[2][2] = __dsub_rn(__ddiv_rn(S[8], cnt), __dmul_rn(mz, mz));
    A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
    jacobi3(A, V);
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    double nx = V[0][m], ny = V[1][m], nz = V[2][m];
    const double len = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(nx, nx), __dmul_rn(ny, ny)), __dmul_rn(nz, nz)));
    nx = __ddiv_rn(nx, len); ny = __ddiv_rn(ny, len); nz = __ddiv_rn(nz, len);
    const double dd = __dmul_rn(-1.0, __dadd_rn(__dadd_rn(__dmul_rn(nx, mx), __dmul_rn(ny, my)), __dmul_rn(nz, mz)));
    if (isfinite(nx) && isfinite(ny) && isfinite(nz) && isfinite(dd)) {
        planes[zone].plane[0] = static_cast<float>(nx); planes[zone].plane[1] = static_cast<float>(ny);
        planes[zone].plane[2] = static_cast<float>(nz); planes[zone].plane[3] = static_cast<float>(dd);
    }
}

// First round of hypotheses of every slab (thread = hypothesis), so that their scoring runs over all band
// points at once instead of slab by slab.
__global__ __launch_bounds__(64) void kg_hyp0(const CmGroundDev* __restrict__ gd, const CmFrameState* __restrict__ st,
                                              const float4* __restrict__ band_pts, const uint32_t* __restrict__ zone_off,
                                              float4* __restrict__ hyp0, uint32_t* __restrict__ valid0,
                                              uint32_t* __restrict__ counts0) {
    const uint32_t zone = blockIdx.x;
    if (threadIdx.x >= CM_GROUND_BATCH) return;
    const uint32_t z0 = zone_off[zone], z1 = zone_off[zone + 1];
    const uint32_t n = (st->status == CM_DEV_OK) ? z1 - z0 : 0u;
    float pl[4] = {0.f, 0.f, 0.f, 0.f};
    bool ok = false;
    const uint32_t j = threadIdx.x;
    if (n >= 3 && j < gd->max_iterations + CM_GROUND_SPARE) {
        uint32_t idx[3];
        sample3(gd->seed, zone, j, n, idx);
        ok = plane_from_3(band_pts[z0 + idx[0]], band_pts[z0 + idx[1]], band_pts[z0 + idx[2]], pl);
    }
    hyp0[zone * CM_GROUND_BATCH + j] = make_float4(pl[0], pl[1], pl[2], pl[3]);
    valid0[zone * CM_GROUND_BATCH + j] = ok ? 1u : 0u;
    counts0[zone * CM_GROUND_BATCH + j] = 0;
}

// Inlier counts of those hypotheses: a workgroup takes 1024 consecutive band points (slab order), which belong
// to one slab or a few; per slab it loads the 32 planes and counts by wave ballots.
__global__ __launch_bounds__(CM_BLOCK) void kg_score0(const CmGroundDev* __restrict__ gd, const CmFrameState* __restrict__ st,
                                                      const float4* __restrict__ band_pts, const uint32_t* __restrict__ keys_sorted,
                                                      const float4* __restrict__ hyp0, uint32_t* __restrict__ counts0) {
    __shared__ float s_pl[CM_GROUND_BATCH][4];
    __shared__ uint32_t s_cnt[CM_GROUND_BATCH];
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t base = blockIdx.x * 1024u;
    if (base >= n) return;
    const uint32_t end = min(base + 1024u, n);
    const int lane = threadIdx.x & 63;
    const float thr = gd->threshold;
    float4 p[4];
    uint32_t zk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t i = base + u * 256 + threadIdx.x;
        p[u] = i < end ? band_pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        zk[u] = i < end ? keys_sorted[i] : 0xFFFFFFFFu;
    }
    const uint32_t z_first = keys_sorted[base], z_last = keys_sorted[end - 1];
    for (uint32_t zone = z_first; zone <= z_last; ++zone) {       // slabs present in this chunk (ascending keys)
        __syncthreads();
        if (threadIdx.x < CM_GROUND_BATCH) {
            const float4 h = hyp0[zone * CM_GROUND_BATCH + threadIdx.x];
            s_pl[threadIdx.x][0] = h.x; s_pl[threadIdx.x][1] = h.y; s_pl[threadIdx.x][2] = h.z; s_pl[threadIdx.x][3] = h.w;
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool have = zk[u] == zone;
            if (__ballot(have) == 0ull) continue;                 // wave-uniform
#pragma unroll 8
            for (int h = 0; h < CM_GROUND_BATCH; ++h) {
                const bool in = have && plane_inlier(s_pl[h][0], s_pl[h][1], s_pl[h][2], s_pl[h][3], p[u], thr);
                const uint32_t c = static_cast<uint32_t>(__popcll(__ballot(in)));
                if (lane == h) mine += c;
            }
        }
        if (lane < CM_GROUND_BATCH && mine) atomicAdd(&s_cnt[lane], mine);
        __syncthreads();
        if (threadIdx.x < CM_GROUND_BATCH && s_cnt[threadIdx.x])
            atomicAdd(&counts0[zone * CM_GROUND_BATCH + threadIdx.x], s_cnt[threadIdx.x]);
    }
}

// Every band point against its slab's plane: inliers are ground, the rest of the band is kept. A workgroup
// takes 1024 consecutive band points; the inlier count of the slab its first point belongs to is collected in
// LDS (one global add per workgroup), points of further slabs in the same chunk add on their own.
__global__ __launch_bounds__(CM_BLOCK) void kg_apply(const CmGroundDev* __restrict__ gd, const CmFrameState* __restrict__ st,
                                                     const float4* __restrict__ band_pts, const uint32_t* __restrict__ keys_sorted,
                                                     CmGroundPlaneDev* __restrict__ planes,
                                                     unsigned char* __restrict__ keep_mask, unsigned char* __restrict__ ground_mask) {
    __shared__ uint32_t s_cnt;
    if (st->status != CM_DEV_OK) return;
    const uint32_t n = st->n_valid;
    const uint32_t base = blockIdx.x * 1024u;
    if (base >= n) return;
    const float thr = gd->threshold;
    const uint32_t z_lead = keys_sorted[base];
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t i = base + u * 256 + threadIdx.x;
        if (i < n) {
            const float4 p = band_pts[i];
            const uint32_t zone = keys_sorted[i];
            const CmGroundPlaneDev& pl = planes[zone];
            const bool in = pl.found && plane_inlier(pl.plane[0], pl.plane[1], pl.plane[2], pl.plane[3], p, thr);
            const uint32_t slot = __float_as_uint(p.w);
            if (in) ground_mask[slot] = 1; else keep_mask[slot] = 1;
            if (in) { if (zone == z_lead) ++mine; else atomicAdd(&planes[zone].inliers, 1u); }
        }
    }
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&planes[z_lead].inliers, s_cnt);
}

// What five hipMemsetAsync calls did in front of every frame of the live node: up to four byte arrays of n bytes set to
// a value each (n a multiple of 16: the padded index space), two state records zeroed. One launch instead of five
// (4-5 us each on a 0.5 ms tick, and five host calls).
__global__ __launch_bounds__(256) void kg_clear(uint32_t n16, uint4* __restrict__ p0, uint32_t v0, uint4* __restrict__ p1, uint32_t v1,
                                                uint4* __restrict__ p2, uint32_t v2, uint4* __restrict__ p3, uint32_t v3,
                                                CmFrameState* __restrict__ st_a, CmFrameState* __restrict__ st_b) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
        if (p0) p0[i] = make_uint4(v0, v0, v0, v0);
        if (p1) p1[i] = make_uint4(v1, v1, v1, v1);
        if (p2) p2[i] = make_uint4(v2, v2, v2, v2);
        if (p3) p3[i] = make_uint4(v3, v3, v3, v3);
    }
    if (blockIdx.x == 0 && threadIdx.x < sizeof(CmFrameState) / 4) {
        if (st_a) reinterpret_cast<uint32_t*>(st_a)[threadIdx.x] = 0;
        if (st_b) reinterpret_cast<uint32_t*>(st_b)[threadIdx.x] = 0;
    }
}

}  // namespace

void cmkg_clear(hipStream_t s, uint32_t n_bytes, void* p0, unsigned char v0, void* p1, unsigned char v1, void* p2, unsigned char v2,
                void* p3, unsigned char v3, CmFrameState* st_a, CmFrameState* st_b) {
    const uint32_t n16 = n_bytes / 16;
    const uint32_t grid = n16 ? (n16 + 256 * 8 - 1) / (256 * 8) : 1;
    hipLaunchKernelGGL(kg_clear, dim3(grid ? grid : 1), dim3(256), 0, s, n16, reinterpret_cast<uint4*>(p0), 0x01010101u * v0,
                       reinterpret_cast<uint4*>(p1), 0x01010101u * v1, reinterpret_cast<uint4*>(p2), 0x01010101u * v2,
                       reinterpret_cast<uint4*>(p3), 0x01010101u * v3, st_a, st_b);
}
void cmkg_setup(hipStream_t s, const CmGroundDev& g, CmGroundDev* d_ground) {
    hipLaunchKernelGGL(kg_setup, dim3(1), dim3(64), 0, s, g, d_ground);
}
void cmkg_classify(hipStream_t s, const CmFrameDev* fd, const CmGroundDev* gd, CmFrameState* st, uint32_t* keys,
                   uint32_t* hist, uint32_t* grp_acc, uint32_t* grp_clear_a, uint32_t* grp_clear_b,
                   uint32_t n_group_words, uint32_t n_clear_a_words, unsigned char* keep_mask, unsigned char* zcode,
                   uint32_t n_tiles) {
    hipLaunchKernelGGL(kg_classify, dim3(n_tiles), dim3(CM_BLOCK), 0, s, fd, gd, st, keys, hist, grp_acc, grp_clear_a,
                       grp_clear_b, n_group_words, n_clear_a_words, keep_mask, zcode);
}
void cmkg_planes(hipStream_t s, const CmFrameDev* fd, const CmGroundDev* gd, const CmFrameState* st,
                 const uint32_t* keys_sorted, const uint32_t* vals_sorted, void* band_pts, uint32_t* zone_off,
                 void* hyp0, uint32_t* valid0, uint32_t* counts0, double* chunk_sums, CmGroundPlaneDev* planes,
                 unsigned char* keep_mask, unsigned char* ground_mask, uint32_t n_padded) {
    const uint32_t blocks = (n_padded + CM_BLOCK * 4 - 1) / (CM_BLOCK * 4);
    const uint32_t zones = CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES;
    float4* bp = reinterpret_cast<float4*>(band_pts);
    float4* h0 = reinterpret_cast<float4*>(hyp0);
    hipLaunchKernelGGL(kg_gather, dim3(blocks), dim3(CM_BLOCK), 0, s, fd, st, keys_sorted, vals_sorted, bp, zone_off);
    hipLaunchKernelGGL(kg_hyp0, dim3(zones), dim3(64), 0, s, gd, st, bp, zone_off, h0, valid0, counts0);
    hipLaunchKernelGGL(kg_score0, dim3(blocks), dim3(CM_BLOCK), 0, s, gd, st, bp, keys_sorted, h0, counts0);
    hipLaunchKernelGGL(kg_ransac, dim3(zones), dim3(256), 0, s, gd, st, bp, zone_off, h0, valid0, counts0, planes);
    hipLaunchKernelGGL(kg_refit_part, dim3(n_padded / CM_GROUND_CHUNK + zones), dim3(256), 0, s, gd, st, bp, zone_off, planes, chunk_sums);
    hipLaunchKernelGGL(kg_refit_final, dim3(1), dim3(128), 0, s, gd, st, zone_off, chunk_sums, planes);
    hipLaunchKernelGGL(kg_apply, dim3(blocks), dim3(CM_BLOCK), 0, s, gd, st, bp, keys_sorted, planes, keep_mask, ground_mask);
}
