/*
 * cm_oracle.cpp — CPU oracle for the merge → voxel-grid hot path.  TEST INFRASTRUCTURE ONLY.
 * See cm_oracle.h for provenance ("parity unpinned") and who may use it.
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * No FMA contraction and no fast-math: every fp32 operation below rounds exactly once, in the
 * order written, like a baseline x86-64 build of the libraries it restates.
 */
#include "cm_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) {
    return std::chrono::duration<double>(b - a).count();
}

inline orc_point default_point() {          // pcl::PointXYZI() — A.0
    orc_point p;
    std::memset(&p, 0, sizeof p);
    p.pad = 1.0f;
    return p;
}

inline bool finite3(const orc_point& p) {
    return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z);
}

// pcl::PassThrough on one float field at byte offset `off` of the 32-B point — A.3.
// PCL builds an index list first and then copies the survivors into a fresh vector.
void passthrough(const std::vector<orc_point>& in, size_t off, float lo, float hi,
                 std::vector<orc_point>& out) {
    std::vector<int> keep(in.size());
    size_t k = 0;
    for (size_t i = 0; i < in.size(); ++i) {
        const orc_point& p = in[i];
        if (!finite3(p)) continue;
        float v;
        std::memcpy(&v, reinterpret_cast<const uint8_t*>(&p) + off, sizeof v);
        if (!std::isfinite(v)) continue;
        if (v < lo || v > hi) continue;      // closed interval [lo, hi]
        keep[k++] = static_cast<int>(i);
    }
    std::vector<orc_point> tmp(k);
    for (size_t i = 0; i < k; ++i) tmp[i] = in[keep[i]];
    out.swap(tmp);
}

struct key_idx {                              // PCL's cloud_point_index_idx
    unsigned int idx;
    unsigned int pt;
    bool operator<(const key_idx& o) const { return idx < o.idx; }
};

struct stage_out {
    std::vector<orc_point> cloud;
    bool is_dense = true;
    double t_ingest = 0, t_xf_crop = 0;
};

// One subscriber callback of the reference: deserialise, by-value copy, transform, (ROI).
void run_sensor(const orc_sensor& s, const orc_params& p, stage_out& o) {
    auto t0 = clk::now();
    std::vector<orc_point> wire(s.n);
    orc_ingest(&s, wire.data());
    std::vector<orc_point> input(wire);       // callbackX(const PointCloud input): by value (:318)
    auto t1 = clk::now();

    float m[12];
    orc_quat_to_matrix(s.q_xyzw, s.t_xyz, m);
    std::vector<orc_point> cloud(input.size());
    orc_transform(input.data(), input.size(), m, s.is_dense, cloud.data());
    o.is_dense = s.is_dense != 0;
    if (p.crop_enable) {                       // getROI: z, then y, then x (:23-39)
        std::vector<orc_point> roi;
        passthrough(cloud, offsetof(orc_point, z), p.crop_min[2], p.crop_max[2], roi);
        passthrough(roi, offsetof(orc_point, y), p.crop_min[1], p.crop_max[1], roi);
        passthrough(roi, offsetof(orc_point, x), p.crop_min[0], p.crop_max[0], roi);
        cloud.swap(roi);
        o.is_dense = true;
    }
    auto t2 = clk::now();
    o.cloud.swap(cloud);
    o.t_ingest = secs(t0, t1);
    o.t_xf_crop = secs(t1, t2);
}

}  // namespace

extern "C" {

void orc_quat_to_matrix(const double q[4], const double t[3], float m[12]) {
    // tf doubles -> Eigen::Quaternionf / Vector3f: per-component double->float rounding.
    const float x = static_cast<float>(q[0]), y = static_cast<float>(q[1]);
    const float z = static_cast<float>(q[2]), w = static_cast<float>(q[3]);
    // Eigen::Quaternion::toRotationMatrix, fp32, in this exact order.
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    m[0] = 1.0f - (tyy + tzz); m[1] = txy - twz;          m[2]  = txz + twy;
    m[4] = txy + twz;          m[5] = 1.0f - (txx + tzz); m[6]  = tyz - twx;
    m[8] = txz - twy;          m[9] = tyz + twx;          m[10] = 1.0f - (txx + tyy);
    m[3] = static_cast<float>(t[0]);
    m[7] = static_cast<float>(t[1]);
    m[11] = static_cast<float>(t[2]);
}

void orc_ingest(const orc_sensor* s, orc_point* out) {
    const uint8_t* base = static_cast<const uint8_t*>(s->data);
    for (uint32_t i = 0; i < s->n; ++i) {
        const uint8_t* r = base + static_cast<size_t>(i) * s->point_step;
        orc_point p = default_point();
        std::memcpy(&p.x, r + s->off_x, 4);
        std::memcpy(&p.y, r + s->off_y, 4);
        std::memcpy(&p.z, r + s->off_z, 4);
        if (s->off_i != ORC_NO_FIELD) std::memcpy(&p.intensity, r + s->off_i, 4);
        out[i] = p;
    }
}

void orc_transform(const orc_point* in, size_t n, const float m[12], int is_dense, orc_point* out) {
    for (size_t i = 0; i < n; ++i) {
        orc_point p = in[i];
        if (is_dense || finite3(p)) {
            const float x = p.x, y = p.y, z = p.z;
            p.x = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
            p.y = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
            p.z = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
        }
        out[i] = p;
    }
}

size_t orc_crop(const orc_point* in, size_t n, const float mn[3], const float mx[3], orc_point* out) {
    std::vector<orc_point> c(in, in + n), r;
    passthrough(c, offsetof(orc_point, z), mn[2], mx[2], r);
    passthrough(r, offsetof(orc_point, y), mn[1], mx[1], r);
    passthrough(r, offsetof(orc_point, x), mn[0], mx[0], r);
    std::copy(r.begin(), r.end(), out);
    return r.size();
}

size_t orc_radius_outlier_removal(const orc_point* in, size_t n, float radius, uint32_t min_neighbors,
                                  orc_point* out, uint8_t* keep_mask) {
    // Uniform hash grid with cells a little wider than the radius: every point within the radius
    // lies in the 27 cells around the query. The grid only finds candidates; the test itself is
    // the fp32 squared distance below.
    const float r2 = static_cast<float>(static_cast<double>(radius) * static_cast<double>(radius));
    const double cell = static_cast<double>(radius) * 1.001;
    auto cid = [&](float v) { return static_cast<int64_t>(std::floor(static_cast<double>(v) / cell)); };
    auto pack = [](int64_t i, int64_t j, int64_t k) {
        return static_cast<uint64_t>((i + (1 << 20)) & 0x1FFFFF) | (static_cast<uint64_t>((j + (1 << 20)) & 0x1FFFFF) << 21) |
               (static_cast<uint64_t>((k + (1 << 20)) & 0x1FFFFF) << 42);
    };
    std::unordered_map<uint64_t, std::vector<uint32_t>> grid;
    grid.reserve(n);
    for (size_t i = 0; i < n; ++i)
        if (finite3(in[i])) grid[pack(cid(in[i].x), cid(in[i].y), cid(in[i].z))].push_back(static_cast<uint32_t>(i));
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        bool keep = false;
        if (finite3(in[i])) {
            const int64_t ci = cid(in[i].x), cj = cid(in[i].y), ck = cid(in[i].z);
            uint32_t k = 0;                                    // includes the query point itself
            for (int dk = -1; dk <= 1 && k <= min_neighbors; ++dk)
                for (int dj = -1; dj <= 1 && k <= min_neighbors; ++dj)
                    for (int di = -1; di <= 1 && k <= min_neighbors; ++di) {
                        auto it = grid.find(pack(ci + di, cj + dj, ck + dk));
                        if (it == grid.end()) continue;
                        for (uint32_t q : it->second) {
                            const float dx = in[i].x - in[q].x, dy = in[i].y - in[q].y, dz = in[i].z - in[q].z;
                            const float d2 = (dx * dx + dy * dy) + dz * dz;
                            if (d2 < r2 && ++k > min_neighbors) break;
                        }
                    }
            keep = k > min_neighbors;
        }
        if (keep_mask) keep_mask[i] = keep ? 1 : 0;
        if (keep) out[o++] = in[i];
    }
    return o;
}

void orc_voxel_cells(const orc_point* in, size_t n, const float leaf[3], int32_t* ijk) {
    const float inv[3] = {1.0f / leaf[0], 1.0f / leaf[1], 1.0f / leaf[2]};
    for (size_t i = 0; i < n; ++i) {
        ijk[3 * i + 0] = static_cast<int32_t>(std::floor(in[i].x * inv[0]));
        ijk[3 * i + 1] = static_cast<int32_t>(std::floor(in[i].y * inv[1]));
        ijk[3 * i + 2] = static_cast<int32_t>(std::floor(in[i].z * inv[2]));
    }
}

int orc_voxelgrid(const orc_point* in, size_t n, const float leaf[3], uint32_t min_pts,
                  int downsample_all, int is_dense, int stable_ties,
                  orc_point* out, size_t* n_out, orc_report* rep,
                  int32_t* out_cells, uint32_t* out_counts) {
    *n_out = 0;
    if (n == 0) return ORC_EMPTY_INPUT;                       // A.4 step 1

    // inverse_leaf_size_ = Array4f::Ones() / leaf_size_.array(): fp32 division.
    const float inv[3] = {1.0f / leaf[0], 1.0f / leaf[1], 1.0f / leaf[2]};

    // getMinMax3D — A.4 step 2.
    float min_p[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                      std::numeric_limits<float>::max()};
    float max_p[3] = {-min_p[0], -min_p[1], -min_p[2]};
    for (size_t i = 0; i < n; ++i) {
        const orc_point& p = in[i];
        if (!is_dense && !finite3(p)) continue;
        min_p[0] = std::min(min_p[0], p.x); max_p[0] = std::max(max_p[0], p.x);
        min_p[1] = std::min(min_p[1], p.y); max_p[1] = std::max(max_p[1], p.y);
        min_p[2] = std::min(min_p[2], p.z); max_p[2] = std::max(max_p[2], p.z);
    }

    // Overflow guard — A.4 step 3 (fp32 product, truncation toward zero, +1).
    const int64_t dx = static_cast<int64_t>((max_p[0] - min_p[0]) * inv[0]) + 1;
    const int64_t dy = static_cast<int64_t>((max_p[1] - min_p[1]) * inv[1]) + 1;
    const int64_t dz = static_cast<int64_t>((max_p[2] - min_p[2]) * inv[2]) + 1;
    if (rep) for (int a = 0; a < 3; ++a) { rep->min_p[a] = min_p[a]; rep->max_p[a] = max_p[a]; }
    if (dx * dy * dz > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) {
        std::copy(in, in + n, out);                           // output = *input_
        *n_out = n;
        return ORC_GRID_OVERFLOW;
    }

    // Step 4.
    int min_b[3], max_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = static_cast<int>(std::floor(min_p[a] * inv[a]));
        max_b[a] = static_cast<int>(std::floor(max_p[a] * inv[a]));
        div_b[a] = max_b[a] - min_b[a] + 1;
        if (rep) { rep->min_b[a] = min_b[a]; rep->max_b[a] = max_b[a]; rep->div_b[a] = div_b[a]; }
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];

    // Step 5: (idx, point) pairs.
    std::vector<key_idx> iv;
    iv.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const orc_point& p = in[i];
        if (!is_dense && !finite3(p)) continue;
        const int i0 = static_cast<int>(std::floor(p.x * inv[0]) - static_cast<float>(min_b[0]));
        const int i1 = static_cast<int>(std::floor(p.y * inv[1]) - static_cast<float>(min_b[1]));
        const int i2 = static_cast<int>(std::floor(p.z * inv[2]) - static_cast<float>(min_b[2]));
        const int idx = i0 + i1 * mul1 + i2 * mul2;
        iv.push_back({static_cast<unsigned int>(idx), static_cast<unsigned int>(i)});
    }

    // Step 6.
    if (stable_ties) std::stable_sort(iv.begin(), iv.end());
    else std::sort(iv.begin(), iv.end());

    // Step 7: runs of equal idx that reach min_points_per_voxel.
    std::vector<std::pair<unsigned, unsigned>> runs;
    runs.reserve(iv.size());
    for (size_t a = 0; a < iv.size();) {
        size_t b = a + 1;
        while (b < iv.size() && iv[b].idx == iv[a].idx) ++b;
        if (b - a >= min_pts) runs.emplace_back(static_cast<unsigned>(a), static_cast<unsigned>(b));
        a = b;
    }

    // Step 8: fp32 accumulators in sorted order.
    size_t o = 0;
    for (const auto& r : runs) {
        orc_point c = default_point();
        const float cnt = static_cast<float>(r.second - r.first);
        if (downsample_all) {                                  // CentroidPoint<PointXYZI>
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
            for (unsigned k = r.first; k < r.second; ++k) {
                const orc_point& p = in[iv[k].pt];
                sx += p.x; sy += p.y; sz += p.z; si += p.intensity;
            }
            c.x = sx / cnt; c.y = sy / cnt; c.z = sz / cnt; c.intensity = si / cnt;
        } else {                                               // Vector4f centroid of xyz+pad
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            for (unsigned k = r.first; k < r.second; ++k) {
                const orc_point& p = in[iv[k].pt];
                s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.pad;
            }
            c.x = s[0] / cnt; c.y = s[1] / cnt; c.z = s[2] / cnt; c.pad = s[3] / cnt;
        }
        if (out_counts) out_counts[o] = r.second - r.first;
        if (out_cells) {                                       // idx -> absolute (i,j,k)
            const unsigned idx = iv[r.first].idx;
            out_cells[3 * o + 0] = static_cast<int>(idx % mul1) + min_b[0];
            out_cells[3 * o + 1] = static_cast<int>((idx / mul1) % div_b[1]) + min_b[1];
            out_cells[3 * o + 2] = static_cast<int>(idx / mul2) + min_b[2];
        }
        out[o++] = c;
    }
    *n_out = o;
    return ORC_OK;
}

int orc_merge_voxelize(const orc_sensor* sensors, int n_sensors, const orc_params* p,
                       int threads, int stable_ties,
                       orc_point* merged_out, orc_point* out, orc_report* rep,
                       int32_t* out_cells, uint32_t* out_counts) {
    if (!sensors || n_sensors <= 0 || !p || !out || !rep) return ORC_BAD_ARG;
    std::memset(rep, 0, sizeof *rep);
    auto T0 = clk::now();

    std::vector<stage_out> st(n_sensors);
    const int nt = std::max(1, std::min(threads, n_sensors));
    rep->threads_used = nt;
    auto t0 = clk::now();
    if (nt == 1) {
        for (int s = 0; s < n_sensors; ++s) run_sensor(sensors[s], *p, st[s]);
    } else {
        std::vector<std::thread> th;
        for (int w = 0; w < nt; ++w)
            th.emplace_back([&, w] { for (int s = w; s < n_sensors; s += nt) run_sensor(sensors[s], *p, st[s]); });
        for (auto& t : th) t.join();
    }
    auto t1 = clk::now();
    double ingest_sum = 0;
    for (int s = 0; s < n_sensors; ++s) { rep->n_in += sensors[s].n; ingest_sum += st[s].t_ingest; }
    rep->t_ingest_s = ingest_sum / nt;                         // share of the stage's wall time
    rep->t_transform_crop_s = secs(t0, t1) - rep->t_ingest_s;

    // fusePointclouds: '=' then '+=' in sensor order (:137-142); vector append with regrowth.
    std::vector<orc_point> merged;
    bool dense = true;
    for (int s = 0; s < n_sensors; ++s) {
        if (s == 0) merged = st[s].cloud;
        else merged.insert(merged.end(), st[s].cloud.begin(), st[s].cloud.end());
        dense = dense && st[s].is_dense;
    }
    if (p->outlier_enable) {                                   // remove_outliers before voxelgrid
        std::vector<orc_point> kept(merged.size());
        kept.resize(orc_radius_outlier_removal(merged.data(), merged.size(), p->outlier_radius,
                                               p->outlier_min_neighbors, kept.data(), nullptr));
        merged.swap(kept);
        dense = true;
    }
    auto t2 = clk::now();
    rep->t_concat_s = secs(t1, t2);
    rep->n_merged = merged.size();
    if (merged_out) std::copy(merged.begin(), merged.end(), merged_out);

    size_t n_out = 0;
    const int status = orc_voxelgrid(merged.data(), merged.size(), p->leaf, p->min_points_per_voxel,
                                     p->downsample_all_data, dense ? 1 : 0, stable_ties,
                                     out, &n_out, rep, out_cells, out_counts);
    auto t3 = clk::now();
    rep->t_voxel_s = secs(t2, t3);
    rep->t_total_s = secs(T0, t3);
    rep->n_out = n_out;
    rep->status = status;
    return status;
}

}  // extern "C"

// ---- ground plane of one zone (header: orc_ransac_plane) -----------------------------------------
namespace {

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t mulhi64(uint64_t a, uint64_t b) { return static_cast<uint64_t>((static_cast<unsigned __int128>(a) * b) >> 64); }

// three distinct indices in [0, n), n >= 3
inline void sample3(uint64_t seed, uint32_t zone_key, uint32_t j, uint64_t n, uint32_t idx[3]) {
    const uint64_t base = seed ^ (static_cast<uint64_t>(zone_key) << 40) ^ (static_cast<uint64_t>(j) << 2);
    const uint64_t i0 = mulhi64(splitmix64(base + 0), n);
    uint64_t i1 = mulhi64(splitmix64(base + 1), n - 1);
    if (i1 >= i0) ++i1;
    uint64_t i2 = mulhi64(splitmix64(base + 2), n - 2);
    const uint64_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
    if (i2 >= lo) ++i2;
    if (i2 >= hi) ++i2;
    idx[0] = static_cast<uint32_t>(i0); idx[1] = static_cast<uint32_t>(i1); idx[2] = static_cast<uint32_t>(i2);
}

// SampleConsensusModelPlane::computeModelCoefficients, fp32
inline bool plane_from_3(const orc_point& p0, const orc_point& p1, const orc_point& p2, float pl[4]) {
    const float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
    const float bx = p2.x - p0.x, by = p2.y - p0.y, bz = p2.z - p0.z;
    const float r0 = ax / bx, r1 = ay / by, r2 = az / bz;          // collinear: all three ratios equal
    if (r0 == r1 && r2 == r1) return false;
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float len = std::sqrt((nx * nx + ny * ny) + nz * nz);
    if (!(len > 0.0f)) return false;
    nx = nx / len; ny = ny / len; nz = nz / len;
    pl[0] = nx; pl[1] = ny; pl[2] = nz;
    pl[3] = -1.0f * ((nx * p0.x + ny * p0.y) + nz * p0.z);
    return std::isfinite(pl[0]) && std::isfinite(pl[1]) && std::isfinite(pl[2]) && std::isfinite(pl[3]);
}
inline bool plane_inlier(const float pl[4], const orc_point& p, float thr) {
    return std::fabs(((pl[0] * p.x + pl[1] * p.y) + pl[2] * p.z) + pl[3]) < thr;
}

// Symmetric 3x3 eigen-decomposition, cyclic Jacobi, 12 sweeps, fp64; a: xx xy xz yy yz zz. v: columns.
inline void jacobi3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int r = 0; r < 3; ++r) {
            const int p = P[r], q = Q[r];
            const double apq = a[p][q];
            if (std::fabs(apq) < 1e-300) continue;
            const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) {                 // columns p, q of a
                const double akp = a[k][p], akq = a[k][q];
                a[k][p] = c * akp - s * akq;
                a[k][q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {                 // rows p, q of a
                const double apk = a[p][k], aqk = a[q][k];
                a[p][k] = c * apk - s * aqk;
                a[q][k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = v[k][p], vkq = v[k][q];
                v[k][p] = c * vkp - s * vkq;
                v[k][q] = s * vkp + c * vkq;
            }
        }
}

}  // namespace

extern "C" void orc_ransac_plane(const orc_point* pts, size_t n, uint32_t max_iterations, float threshold, float probability,
                                 int optimize, uint64_t seed, uint32_t zone_key, orc_plane_result* res,
                                 uint8_t* inlier_mask) {
    std::memset(res, 0, sizeof *res);
    if (inlier_mask) std::memset(inlier_mask, 0, n);
    if (n < 3) return;
    const uint32_t J = max_iterations + 24;
    // PCL's loop over the hypothesis sequence
    uint32_t iterations = 0, skipped = 0;
    long long best = -1;
    uint32_t best_j = 0;
    float best_pl[4] = {0, 0, 0, 0};
    double pno = 1.0, pw = 1.0;
    const double stop = 1.0 - static_cast<double>(probability);
    for (;;) {
        const uint32_t j = iterations + skipped;
        if (j >= J) break;
        uint32_t idx[3];
        sample3(seed, zone_key, j, n, idx);
        float pl[4];
        if (!plane_from_3(pts[idx[0]], pts[idx[1]], pts[idx[2]], pl)) { ++skipped; continue; }
        long long c = 0;
        for (size_t i = 0; i < n; ++i) c += plane_inlier(pl, pts[i], threshold) ? 1 : 0;
        bool updated = false;
        if (c > best) {
            best = c; best_j = j; std::memcpy(best_pl, pl, sizeof pl);
            const double w = static_cast<double>(c) / static_cast<double>(n);
            pno = 1.0 - (w * w) * w;
            if (pno < 2.220446049250313e-16) pno = 2.220446049250313e-16;
            if (pno > 1.0 - 2.220446049250313e-16) pno = 1.0 - 2.220446049250313e-16;
            updated = true;
        }
        ++iterations;
        if (updated) { pw = 1.0; for (uint32_t t = 0; t < iterations; ++t) pw *= pno; }
        else pw *= pno;
        if (iterations > max_iterations) break;
        if (!(pw > stop)) break;
    }
    res->iterations = iterations;
    if (best < 0) return;
    res->found = 1;
    res->best_hypothesis = best_j;
    float pl[4];
    std::memcpy(pl, best_pl, sizeof pl);
    if (optimize && best > 3) {
        // sums over the inliers in the fixed order of the header: chunks of 8192 points; inside a chunk element i
        // goes to partial i mod 256 (in order), the partials are added pairwise (128, 64, ... 1); the chunk sums
        // are added one after the other
        double total[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t c0 = 0; c0 < n; c0 += 8192) {
            double part[256][10];
            std::memset(part, 0, sizeof part);
            const size_t c1 = std::min(n, c0 + 8192);
            for (size_t i = c0; i < c1; ++i) {
                if (!plane_inlier(pl, pts[i], threshold)) continue;
                double* a = part[(i - c0) & 255];
                const double x = pts[i].x, y = pts[i].y, z = pts[i].z;
                a[0] += x; a[1] += y; a[2] += z;
                a[3] += x * x; a[4] += x * y; a[5] += x * z; a[6] += y * y; a[7] += y * z; a[8] += z * z;
                a[9] += 1.0;
            }
            for (int stride = 128; stride > 0; stride >>= 1)
                for (int t = 0; t < stride; ++t)
                    for (int k = 0; k < 10; ++k) part[t][k] += part[t + stride][k];
            for (int k = 0; k < 10; ++k) total[k] += part[0][k];
        }
        double part[1][10];
        for (int k = 0; k < 10; ++k) part[0][k] = total[k];
        const double* S = part[0];
        const double cnt = S[9];
        const double mx = S[0] / cnt, my = S[1] / cnt, mz = S[2] / cnt;
        double a[3][3], v[3][3];
        a[0][0] = S[3] / cnt - mx * mx; a[0][1] = S[4] / cnt - mx * my; a[0][2] = S[5] / cnt - mx * mz;
        a[1][1] = S[6] / cnt - my * my; a[1][2] = S[7] / cnt - my * mz; a[2][2] = S[8] / cnt - mz * mz;
        a[1][0] = a[0][1]; a[2][0] = a[0][2]; a[2][1] = a[1][2];
        jacobi3(a, v);
        int m = 0;
        if (a[1][1] < a[m][m]) m = 1;
        if (a[2][2] < a[m][m]) m = 2;
        double nx = v[0][m], ny = v[1][m], nz = v[2][m];
        const double len = std::sqrt((nx * nx + ny * ny) + nz * nz);
        nx = nx / len; ny = ny / len; nz = nz / len;
        const double d = -1.0 * ((nx * mx + ny * my) + nz * mz);
        if (std::isfinite(nx) && std::isfinite(ny) && std::isfinite(nz) && std::isfinite(d)) {
            pl[0] = static_cast<float>(nx); pl[1] = static_cast<float>(ny); pl[2] = static_cast<float>(nz);
            pl[3] = static_cast<float>(d);
        }
    }
    std::memcpy(res->plane, pl, sizeof pl);
    uint32_t ni = 0;
    for (size_t i = 0; i < n; ++i) {
        const bool in = plane_inlier(pl, pts[i], threshold);
        if (inlier_mask) inlier_mask[i] = in ? 1 : 0;
        ni += in ? 1u : 0u;
    }
    res->n_inliers = ni;
}
