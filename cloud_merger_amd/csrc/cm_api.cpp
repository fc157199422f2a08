// cm_api.cpp — host side of the C-ABI declared in include/cloudmerge.h.
//
// Owns the HBM layout and the launch sequence; no arithmetic on points happens here. There is no
// CPU fallback of any kind: without a gfx950 device cm_create fails.
//
// HBM layout per context (N = padded point capacity, multiples of CM_TILE per sensor):
//   sensor slots      raw PointCloud2 payloads as submitted (or caller-owned device pointers)
//   keys_a/b, vals_a/b  4 x N x u32   radix ping-pong: voxel index, padded point index
//   hist              (N/4096) x 256 u32   digit counts per tile (one coalesced row each)
//   grp               5 x (N/4096/32) x 256 u32   digit counts per group of 32 tiles, per pass
//   seg_tile_counts   N/2048 u32      kept voxels per sorted tile (+ totals per 256 tiles)
//   out               N x 16 B        centroids x,y,z,intensity (ascending voxel index = PCL order)
//   out_key/out_cnt   N x u32 each    only with CM_FLAG_OCCUPANCY
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/cloudmerge.h"
#include "cm_device.h"
#include "cm_kernels.h"

namespace {

// A cloud as a frame sees it: where its payload lies in HBM and how its points are laid out.
struct SlotCloud {
    const void* dptr = nullptr;      // an owned buffer of the slot or a caller-owned device pointer
    uint32_t n = 0, step = 0, ox = 0, oy = 0, oz = 0, oi = 0;
};

// One sensor. Two owned HBM buffers: the frame that was enqueued last reads `active` (and so do its by-products:
// cm_merged_copy, cm_ground_copy, the overflow fallback, a hand-back's redo) until the NEXT frame is enqueued;
// a submit meanwhile always goes to the other buffer and becomes `staged`. Nothing a subscriber thread does can
// therefore touch what a frame in flight — or its by-products afterwards — read, and cm_submit_cloud never waits for a
// merge (the reference's callbacks run beside its 10 Hz loop: pc_preprocessing_main.cpp:513, :318-337, :549-584).
struct Slot {
    std::mutex mu;
    void* buf[2] = {nullptr, nullptr};   // owned HBM buffers (host submits)
    size_t cap[2] = {0, 0};
    int active_buf = -1;                 // which of them `active` lives in (-1: none / a caller-owned pointer)
    SlotCloud active, staged;
    bool has_data = false;               // `active` (or, while fresh, `staged`) holds a cloud
    bool fresh = false;                  // `staged` holds a cloud no frame has consumed yet
    bool copy_pending = false;           // its H2D copy was enqueued without waiting (cm_submit_cloud_async): ev_copy tells
    float m[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy = nullptr;
    uint64_t bytes_h2d = 0;              // payload bytes of the staged cloud that crossed PCIe (0: device submit)
    uint64_t active_bytes_h2d = 0;
    uint64_t gen = 0, active_gen = 0;    // accepted submits so far; the one `active` came from
};

}  // namespace

struct cm_ctx {
    int device = 0;
    uint32_t flags = 0, max_sensors = 0;
    uint64_t max_points = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    Slot slots[CM_MAX_SENSORS];

    uint32_t cap_padded = 0, cap_tiles = 0, cap_seg_tiles = 0;
    uint32_t *keys_a = nullptr, *keys_b = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    uint32_t *hist = nullptr, *grp = nullptr, *totals = nullptr, *seg_counts = nullptr, *seg_tile_counts = nullptr, *seg_groups = nullptr;
    uint32_t cap_groups = 0, frame_seq = 0;
    bool lds_rank = false;               // k_probe_lds_order found lane-ordered LDS adds on this device
    int finish_mode = 0;                 // CM_FINISH: 0 k3_local + k3_compact, 2 (CM_FINISH=v2) k2_local with its look-back
    void* stage32 = nullptr;             // k3_local's staging for partial tables (32-byte entries)
    void* out32 = nullptr;               // the result as pcl::PointXYZI images (cm_result_copy with point_step_out 32)
    int debug_misrank = 0;               // test build (CM_TEST_HOOKS) + CM_DEBUG_MISRANK=1: the last global pass swaps two records of tile 0
    float* partials = nullptr;
    uint32_t *out_key = nullptr, *out_cnt = nullptr, *merged_total = nullptr;
    void* out = nullptr;
    void* merged = nullptr;
    unsigned char* mask = nullptr;       // outlier stage: keep-mask over the padded point indices
    void* sorted_pts = nullptr;          // outlier stage: points in radius-grid order
    void* rows = nullptr;                // outlier stage: (y,z)-row ranges
    CmFrameState* d_state_o = nullptr;   // outlier stage: its grid and counts
    const unsigned char* frame_mask = nullptr;   // mask of the last frame (nullptr: stage off)
    void* partial = nullptr;             // cm_partial_entry table of the last cm_merge_partial
    void* table_entries = nullptr;       // merged entries inside cm_merge_tables
    int last_mode = 0;
    CmFrameDev* d_frame = nullptr;
    CmTileDev* d_tiles = nullptr;        // per-tile entries of the uploaded descriptor (k_setup)
    CmFrameDev frame_uploaded;
    bool frame_uploaded_valid = false;
    CmFrameState* d_state[2] = {nullptr, nullptr};
    int cur = 0;
    CmFrameState* h_state = nullptr;     // pinned, written by the last kernel of a frame
    uint32_t* h_state_dev = nullptr;     // device view of h_state
    hipEvent_t ev_done = nullptr;

    std::mutex merge_mu;
    std::atomic<bool> in_flight{false};
    bool pending = false;                // an enqueued frame has not been waited for
    bool pending_trivial = false;        // ... and it had no kernels (nothing submitted)
    bool trivial_grid = false;           // ... but, as an empty share of a fused cloud, it has the shared grid
    float trivial_box[6] = {0, 0, 0, 0, 0, 0};
    CmFrameDev frame;                    // descriptor of the last enqueued frame
    bool from_crop = false;
    uint64_t n_in = 0;
    uint32_t n_sensors_used = 0;
    cm_result result;
    bool have_result = false;
    bool out_is_merged = false;

    // zone-wise ground removal (cm_kernels_ground.hip)
    bool ground_on = false;
    CmGroundDev ground;                  // host copy of the slab table
    bool ground_uploaded = false;
    CmGroundDev* d_ground = nullptr;
    CmFrameState* d_state_g = nullptr;   // state of the slab sort
    unsigned char* gmask = nullptr;      // ground points of the last frame (padded index space)
    uint32_t* zone_off = nullptr;
    CmGroundPlaneDev* d_planes = nullptr;
    void* hyp0 = nullptr;                // first round of hypotheses of every slab: planes, validity, inlier counts
    uint32_t *valid0 = nullptr, *counts0 = nullptr;
    double* chunk_sums = nullptr;        // least-squares sums per chunk of band points
    bool frame_had_ground = false;
    float ground_outlier_radius = 0.f;   // > 0: radius filter on every slab's band points that are not ground (:119)
    uint32_t ground_outlier_min_nb = 0;
    unsigned char* bmask = nullptr;      // those points (input of that filter)
    unsigned char* zcode = nullptr;      // slab of every band point

    // bucket path (cm_kernels_v2.hip)
    int path_mode = 0;                   // CM_PATH: 0 auto (bucket path when it applies), 1 classic only, 2 bucket only where it applies
    void *rec_a = nullptr, *rec_b = nullptr;      // 16-byte point records, ping-pong
    unsigned char* dig = nullptr;        // next digit of every record
    unsigned long long* tile_state = nullptr;     // published kept-voxel counts of the local finish
    uint32_t* wave_cnt = nullptr;                 // records k2_hist0 packed per wave (frames whose crop box drops most points)
    float* records = nullptr;            // min/max/count per tile
    bool pred_ok = false;                // a box predicted from an earlier frame's bounds
    float pred_min[3] = {0, 0, 0}, pred_max[3] = {0, 0, 0};
    uint32_t v2_extra_passes = 0;        // buckets overflowed LDS: sort more bits globally
    uint32_t v2_good_frames = 0;         // frames since the last overflow (on whichever path they ran)
    uint32_t v2_retry_after = 256;       // ... after this many, try one global pass fewer again (doubles on failure)
    uint32_t v2_off_frames = 0;          // ... or give the path a rest
    uint32_t pre_bucket_off = 0;         // frames for which the outlier stage sorts with the general kernels (a bucket overflowed)
    uint32_t pre_bucket_backoff = 16;
    bool pre_bucket = false;             // this frame's outlier stage may sort with the bucket kernels
    bool last_packed = false;            // the voxel stage's k2_hist0 packed the survivors (CM_PATH_PACKED)
    uint32_t grid_shrink_off = 0;        // frames for which the kernels behind pass 0 get whole grids again (after CM_DEV_ERR_GRID)
    bool last_v2 = false, last_predicted = false, last_k3 = false;
    bool post_bucket = false;            // the frame's pre-stages (ground / outlier removal) run first, then the bucket path
    uint32_t post_g = 0, post_low = 0;
    uint64_t last_n_merged = 0;          // points that entered the voxel grid in the last finished frame (0: none yet)
    bool last_outl = false;
    int last_gm_o = 0;
    uint32_t last_kb_o = 0;
    cm_params last_params;
    int last_grid_mode = 0;
    uint32_t last_key_bits = 0;
    int cell_min_b[3] = {0, 0, 0}, cell_div_b[3] = {1, 1, 1};   // grid the cells in out_key are relative to
    uint64_t n_redone = 0;               // frames the bucket path handed back to the classic one

    // pipelined publish (cm_result_publish_async): the result buffers exist twice, so that the copy-out of frame n runs on
    // its own stream beside the kernels of frame n + 1
    void* out_other = nullptr;           // the result buffer the frame in flight does NOT write
    void* out32_other = nullptr;
    hipStream_t pub_stream = nullptr;
    hipEvent_t ev_pub[2] = {nullptr, nullptr};   // [0]: the last copy-out that read `out`, [1]: ... `out_other`
    bool pub_pending[2] = {false, false};

    // quantile passes (cm_kernels_v4.hip): one global pass into buckets cut at the last frame's quantiles
    bool quant_sub = true;               // CM_QUANT_SUB=0: frames above 2048 buckets take the fixed-grid passes
    int quant_mode = 0;                  // CM_QUANT: 0 auto, 1 never
    uint32_t* spl[2] = {nullptr, nullptr};   // splitters: a frame reads spl[spl_cur]; its finish writes spl[spl_cur ^ 1]
    int spl_cur = 0;
    bool spl_valid = false;              // spl[spl_cur] holds the quantiles of the last finished frame
    uint32_t spl_n = 0;                  // ... which sorted this many records
    int32_t spl_min_b[3] = {0, 0, 0}, spl_div_b[3] = {0, 0, 0};   // ... as indices of this grid
    float spl_inv_leaf[3] = {0, 0, 0};
    uint32_t *qcnt = nullptr, *qtot = nullptr, *qbofs = nullptr;  // per-tile bucket counts / prefixes, bucket totals, bucket starts
    uint32_t* qbig = nullptr;                                     // buckets beyond CM4_CAP records: count, then their numbers
    uint32_t quant_big_arm = 0;          // quantile frames for which the large finish shape is still launched (armed by a hand-back or a listed bucket)
    uint16_t* qbid = nullptr;            // the bucket of every padded slot
    bool last_quant = false;             // the frame in flight runs the quantile passes
    bool wrote_spl = false;              // ... and its finish leaves splitters in spl[spl_cur ^ 1]
    uint32_t quant_off_frames = 0;       // frames for which the fixed-grid passes run although splitters are at hand
    uint32_t quant_hist = 0;             // the last eight attempts, newest in bit 0: 1 = handed back
    uint32_t quant_rest = 8;             // how long the next rest is (doubles while rests keep being needed, back to 8 after 16 good frames)
    uint32_t quant_good = 0;             // good attempts in a row
    int lb_grid_mode = 0, lb_mode = 0;   // the last launch_bucket's arguments (a quantile frame that is handed back is
    uint32_t lb_g = 0, lb_low = 0;       // redone with the fixed-grid passes in the same box)

    // per-sensor figures of the last enqueued frame (cm_frame_stats)
    uint32_t stats_n_sensors = 0;
    uint32_t stats_sensor[CM_MAX_SENSORS] = {0}, stats_n[CM_MAX_SENSORS] = {0}, stats_fresh[CM_MAX_SENSORS] = {0};
    uint64_t stats_bytes[CM_MAX_SENSORS] = {0}, stats_gen[CM_MAX_SENSORS] = {0};
    uint32_t* d_tile_kept = nullptr;     // per 4096-slot tile: points that passed crop / masks and entered the sort — the first scatter
    uint32_t* h_tile_kept = nullptr;     // writes them straight into pinned host memory (d_tile_kept is its device view)
    uint64_t bytes_d2h = 0;              // result / merged / ground bytes copied to the host since the frame was enqueued

    std::vector<hipEvent_t> prof_ev;
    std::vector<std::string> prof_names;
    size_t prof_used = 0;
    cm_stage_times stage_times;

    std::mutex err_mu;
    std::string err;
};

namespace {

const char* k_status_names(int s) {
    switch (s) {
        case CM_OK: return "CM_OK";
        case CM_EMPTY_INPUT: return "CM_EMPTY_INPUT";
        case CM_GRID_OVERFLOW: return "CM_GRID_OVERFLOW";
        case CM_NOT_READY: return "CM_NOT_READY";
        case CM_SKIPPED: return "CM_SKIPPED";
        case CM_BAD_ARG: return "CM_BAD_ARG";
        case CM_HIP_ERROR: return "CM_HIP_ERROR";
        case CM_NO_DEVICE: return "CM_NO_DEVICE";
        case CM_CAPACITY: return "CM_CAPACITY";
        case CM_INTERNAL: return "CM_INTERNAL";
        default: return "CM_UNKNOWN";
    }
}

// (subscriber threads and the loop thread may fail at the same time — an oversize cloud beside a refused merge — and a third
// thread may be reading the text: the string is only touched under its own lock, and cm_last_error hands out a copy)
int fail(cm_ctx* c, int code, const std::string& what) {
    if (c) { std::lock_guard<std::mutex> lk(c->err_mu); c->err = what; }
    return code;
}

#define HIP_TRY(c, call)                                                                        \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail((c), CM_HIP_ERROR, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Eigen::Quaternionf(w,x,y,z).toRotationMatrix() in fp32, tf doubles rounded per component
// (reference call site pc_preprocessing_main.cpp:320-322).
void quat_to_rows(const double q[4], const double t[3], float m[12]) {
    const float x = static_cast<float>(q[0]), y = static_cast<float>(q[1]);
    const float z = static_cast<float>(q[2]), w = static_cast<float>(q[3]);
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    m[0] = 1.0f - (tyy + tzz); m[1] = txy - twz;          m[2] = txz + twy;           m[3] = static_cast<float>(t[0]);
    m[4] = txy + twz;          m[5] = 1.0f - (txx + tzz); m[6] = tyz - twx;           m[7] = static_cast<float>(t[1]);
    m[8] = txz - twy;          m[9] = tyz + twx;          m[10] = 1.0f - (txx + tyy); m[11] = static_cast<float>(t[2]);
}

// Host copy of the kernels' grid guard for a box (the crop box, or bounds handed in): true when the box itself fits PCL's int32 index,
// in which case the data min/max pass can be skipped (box-relative indices give the same
// occupancy and the same order). Also returns the key width.
bool box_grid(const float bmin[3], const float bmax[3], const float inv[3], uint32_t* key_bits,
              int32_t* min_b = nullptr, int32_t* div_b = nullptr) {
    long long d[3];
    unsigned long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        const float ext = (bmax[a] - bmin[a]) * inv[a];
        if (!(ext < 2147483648.0f) || ext < 0.0f) return false;
        d[a] = static_cast<long long>(ext) + 1;
        const int lo = static_cast<int>(std::floor(bmin[a] * inv[a]));
        const int hi = static_cast<int>(std::floor(bmax[a] * inv[a]));
        if (hi < lo) return false;
        cells *= static_cast<unsigned long long>(hi - lo + 1);
        if (min_b) { min_b[a] = lo; div_b[a] = hi - lo + 1; }
    }
    if (d[0] * d[1] * d[2] > 2147483647LL || cells > 0xFFFFFFFFull) return false;
    uint32_t bits = 1;
    while (bits < 32 && (cells - 1) >> bits) ++bits;
    *key_bits = bits;
    return true;
}

void prof_mark(cm_ctx* c, const char* name) {
    if (!(c->flags & CM_FLAG_PROFILE)) return;
    if (c->prof_used >= c->prof_ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->prof_ev.push_back(e);
        c->prof_names.emplace_back();
    }
    c->prof_names[c->prof_used] = name;
    (void)hipEventRecord(c->prof_ev[c->prof_used], c->stream);
    ++c->prof_used;
}

void free_all(cm_ctx* c) {
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    F(c->keys_a); F(c->keys_b); F(c->vals_a); F(c->vals_b); F(c->hist); F(c->totals);
    F(c->seg_counts); F(c->seg_tile_counts); F(c->seg_groups); F(c->grp); F(c->partials); F(c->out_key); F(c->out_cnt); F(c->merged_total); F(c->out); F(c->merged); F(c->partial); F(c->table_entries); F(c->mask); F(c->sorted_pts); F(c->rows); F(c->d_state_o);
    F(c->stage32); F(c->out32); F(c->rec_a); F(c->rec_b); F(c->dig); F(c->tile_state); F(c->wave_cnt); F(c->records);
    F(c->spl[0]); F(c->spl[1]); F(c->qcnt); F(c->qtot); F(c->qbofs); F(c->qbid); F(c->qbig);
    F(c->out_other); F(c->out32_other);
    if (c->pub_stream) (void)hipStreamDestroy(c->pub_stream);
    for (auto e : c->ev_pub) if (e) (void)hipEventDestroy(e);
    F(c->d_ground); F(c->d_state_g); F(c->gmask); F(c->zone_off); F(c->d_planes); F(c->hyp0); F(c->valid0); F(c->counts0); F(c->chunk_sums); F(c->bmask); F(c->zcode);
    F(c->d_frame); F(c->d_tiles); F(c->d_state[0]); F(c->d_state[1]);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_tile_kept) (void)hipHostFree(c->h_tile_kept);
    for (auto& s : c->slots) {
        F(s.buf[0]); F(s.buf[1]);
        if (s.copy_stream) (void)hipStreamDestroy(s.copy_stream);
        if (s.ev_copy) (void)hipEventDestroy(s.ev_copy);
    }
    for (auto e : c->prof_ev) (void)hipEventDestroy(e);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
}

int set_slot_cloud(cm_ctx* c, uint32_t sensor, const void* data, bool on_device, uint32_t n,
                   uint32_t step, uint32_t ox, uint32_t oy, uint32_t oz, uint32_t oi, bool wait_copy = true) {
    if (!c) return CM_BAD_ARG;
    if (sensor >= c->max_sensors) return fail(c, CM_BAD_ARG, "sensor index out of range");
    if (n && !data) return fail(c, CM_BAD_ARG, "null payload");
    auto fits = [step](uint32_t off) { return static_cast<uint64_t>(off) + 4u <= step; };   // (no 32-bit wrap-around)
    if (n && (step < 12 || !fits(ox) || !fits(oy) || !fits(oz) || (oi != CM_NO_FIELD && !fits(oi))))
        return fail(c, CM_BAD_ARG, "field offsets do not fit point_step");
    if (n > c->max_points) return fail(c, CM_CAPACITY, "cloud larger than cm_limits.max_points_total");
    HIP_TRY(c, hipSetDevice(c->device));
    Slot& s = c->slots[sensor];
    std::lock_guard<std::mutex> lk(s.mu);
    if (s.fresh && !(c->flags & CM_FLAG_LATEST_WINS)) return CM_SKIPPED;   // first since last fuse wins
    const size_t bytes = static_cast<size_t>(n) * step;
    SlotCloud sc;
    sc.n = n; sc.step = step; sc.ox = ox; sc.oy = oy; sc.oz = oz; sc.oi = oi;
    if (on_device) {
        sc.dptr = data;
        s.bytes_h2d = 0;
    } else {
        // the buffer no enqueued frame reads (a replaced staged cloud lived there too: same copy stream, in order)
        const int w = s.active_buf == 0 ? 1 : 0;
        if (bytes > s.cap[w]) {
            if (s.buf[w]) HIP_TRY(c, hipFree(s.buf[w]));
            s.buf[w] = nullptr; s.cap[w] = 0;
            const size_t cap = bytes + bytes / 4 + 256;
            HIP_TRY(c, hipMalloc(&s.buf[w], cap));
            s.cap[w] = cap;
        }
        if (bytes) {
            HIP_TRY(c, hipMemcpyAsync(s.buf[w], data, bytes, hipMemcpyHostToDevice, s.copy_stream));
            if (wait_copy) {
                HIP_TRY(c, hipStreamSynchronize(s.copy_stream));     // the caller may release `data` when this returns
                s.copy_pending = false;
            } else {
                HIP_TRY(c, hipEventRecord(s.ev_copy, s.copy_stream));  // the frame that consumes the cloud waits for it
                s.copy_pending = true;
            }
        }
        sc.dptr = s.buf[w];
        s.bytes_h2d = bytes;
    }
    s.staged = sc;
    s.has_data = true;
    s.fresh = true;
    ++s.gen;
    return CM_OK;
}

// Builds the frame descriptor and enqueues every kernel of the frame on c->stream.
int build_frame(cm_ctx* c, const cm_params* p, bool consume, std::vector<std::unique_lock<std::mutex>>& locks,
                bool gate = true) {
    for (uint32_t s = 0; s < c->max_sensors; ++s) locks.emplace_back(c->slots[s].mu);

    // Frame assembly policy (pc_preprocessing_main.cpp:134-157).
    uint32_t have = 0, fresh = 0;
    for (uint32_t s = 0; s < c->max_sensors; ++s) {
        if (c->slots[s].has_data) have |= 1u << s;
        if (c->slots[s].fresh) fresh |= 1u << s;
    }
    const uint32_t required = p->required_sensor_mask ? p->required_sensor_mask : have;
    if (have == 0 || (gate && (required & ~fresh) != 0)) return CM_NOT_READY;   // gate off: redoing a fused frame

    CmFrameDev& f = c->frame;
    std::memset(&f, 0, sizeof f);
    uint32_t base = 0, k = 0;
    uint64_t n_in = 0;
    // the clouds the frame will read: a slot's staged cloud if it has a fresh one, else the one its last frame read
    // (a stale optional sensor rides along like :141)
    for (uint32_t s = 0; s < c->max_sensors; ++s) {
        Slot& sl = c->slots[s];
        if (!sl.has_data) continue;
        const uint64_t nb = static_cast<uint64_t>(base) + round_up(sl.fresh ? sl.staged.n : sl.active.n, CM_TILE);
        if (nb > c->cap_padded) return fail(c, CM_CAPACITY, "frame exceeds cm_limits.max_points_total");
        base = static_cast<uint32_t>(nb);
    }
    base = 0;
    c->stats_n_sensors = 0;
    for (uint32_t s = 0; s < c->max_sensors; ++s) {
        Slot& sl = c->slots[s];
        if (!sl.has_data) continue;
        if (sl.fresh) {
            // an H2D copy enqueued without waiting (cm_submit_cloud_async): the frame's stream waits for it, not the host
            if (sl.copy_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, sl.ev_copy, 0));
            if (consume) {                   // the frame takes the staged cloud over; submits now go to the other buffer
                sl.active = sl.staged;
                sl.active_buf = sl.staged.dptr == sl.buf[0] ? 0 : sl.staged.dptr == sl.buf[1] ? 1 : -1;
                sl.active_bytes_h2d = sl.bytes_h2d;
                sl.active_gen = sl.gen;
                sl.copy_pending = false;
            }
        }
        const SlotCloud& sc = (sl.fresh && !consume) ? sl.staged : sl.active;   // (!consume: cm_local_bounds' peek)
        CmSensorDev& d = f.s[k];
        d.data = static_cast<const unsigned char*>(sc.dptr);
        d.n = sc.n; d.base = base; d.point_step = sc.step; d.slot = s;
        d.off_x = sc.ox; d.off_y = sc.oy; d.off_z = sc.oz; d.off_i = sc.oi;
        const bool al16 = (reinterpret_cast<uintptr_t>(sc.dptr) & 15u) == 0;
        if (al16 && sc.step == 16 && sc.ox == 0 && sc.oy == 4 && sc.oz == 8 && sc.oi == 12) d.layout = CM_LAYOUT_XYZI16;
        else if (al16 && sc.step == 32 && sc.ox == 0 && sc.oy == 4 && sc.oz == 8 && sc.oi == 16) d.layout = CM_LAYOUT_PCL32;
        else d.layout = CM_LAYOUT_GENERIC;
        std::memcpy(d.m, sl.m, sizeof d.m);
        n_in += sc.n;
        if (consume) {
            c->stats_sensor[k] = s; c->stats_n[k] = sc.n;
            c->stats_fresh[k] = sl.fresh ? 1u : 0u;
            c->stats_bytes[k] = sl.fresh ? sl.active_bytes_h2d : 0u;
            c->stats_gen[k] = sl.active_gen;
            c->stats_n_sensors = k + 1;
        }
        ++k;
        base += round_up(sc.n, CM_TILE);
    }
    f.n_sensors = k;
    f.n_padded = base;
    f.n_tiles = base / CM_TILE;
    f.crop_enable = p->crop_enable ? 1u : 0u;
    for (int a = 0; a < 3; ++a) {
        f.crop_min[a] = p->crop_min[a];
        f.crop_max[a] = p->crop_max[a];
        f.inv_leaf[a] = 1.0f / p->leaf[a];          // Array4f::Ones() / leaf_size_: fp32 division
    }
    f.min_pts = p->min_points_per_voxel;
    f.downsample_all = p->downsample_all_data ? 1u : 0u;
    c->n_in = n_in;
    c->n_sensors_used = k;
    if (consume)
        for (auto& sl : c->slots) sl.fresh = false;   // flag reset, :151-157
    return CM_OK;
}

int launch_classic(cm_ctx* c, const cm_params* p, int mode, int grid_mode, uint32_t key_bits, bool outl, int gm_o,
                   uint32_t kb_o);
uint32_t bucket_passes(uint32_t kb, uint64_t est, uint32_t extra);

// Bounds of the merged cloud by one k_minmax pass and a host round trip: only when the bucket path
// has no box yet (first frame of a context without a crop box, or after a point left the predicted box).
void set_predicted_box(cm_ctx* c, const float mn[3], const float mx[3], const float leaf[3]) {
    // A cloud near the limit of PCL's 32-bit index leaves no room for an eighth of its extent on every side: take what
    // fits (a frame right behind one that reached far out would otherwise lose its box, and with it the bucket path).
    float inv[3];
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / leaf[a];
    for (float part = 8.0f; part <= 1024.0f; part *= 2.0f) {
        for (int a = 0; a < 3; ++a) {
            const float ext = mx[a] - mn[a];
            const float margin = std::max(ext / part, (part <= 8.0f ? 8.0f : 2.0f) * leaf[a]);
            c->pred_min[a] = mn[a] - margin;
            c->pred_max[a] = mx[a] + margin;
        }
        uint32_t kb = 0;
        if (box_grid(c->pred_min, c->pred_max, inv, &kb)) break;
    }
    c->pred_ok = true;
}

// Keeps the predicted box while the cloud stays comfortably inside it and the box is not wastefully
// large (so the frame descriptor, and with it the key width, stays put from frame to frame).
void update_predicted_box(cm_ctx* c, const float mn[3], const float mx[3], const float leaf[3]) {
    bool redo = !c->pred_ok;
    for (int a = 0; a < 3 && !redo; ++a) {
        const float margin = std::max((mx[a] - mn[a]) / 8.0f, 8.0f * leaf[a]);
        const float lo = mn[a] - c->pred_min[a], hi = c->pred_max[a] - mx[a];
        redo = !(lo >= margin / 4.0f && lo <= 3.0f * margin && hi >= margin / 4.0f && hi <= 3.0f * margin);
    }
    if (!redo) {
        // ... and not a box so much larger than the cloud needs that it costs a global pass: after a frame that reached
        // far out the box would otherwise stay wide — and the index one digit longer — for as long as the cloud fits it
        float inv[3], tmin[3], tmax[3];
        for (int a = 0; a < 3; ++a) {
            inv[a] = 1.0f / leaf[a];
            const float margin = std::max((mx[a] - mn[a]) / 8.0f, 8.0f * leaf[a]);
            tmin[a] = mn[a] - margin; tmax[a] = mx[a] + margin;
        }
        uint32_t kb_now = 0, kb_tight = 0;
        if (box_grid(c->pred_min, c->pred_max, inv, &kb_now) && box_grid(tmin, tmax, inv, &kb_tight))
            redo = bucket_passes(kb_tight, 0, 0) < bucket_passes(kb_now, 0, 0);
    }
    if (redo) set_predicted_box(c, mn, mx, leaf);
}

int bootstrap_box(cm_ctx* c) {
    const CmFrameDev& f = c->frame;
    hipStream_t st = c->stream;
    if (!c->frame_uploaded_valid || std::memcmp(&f, &c->frame_uploaded, sizeof f) != 0) {
        cmk_setup(st, f, c->d_frame, c->d_tiles);
        c->frame_uploaded = f;
        c->frame_uploaded_valid = true;
    }
    const uint32_t n_partials = f.n_tiles < CM_MINMAX_BLOCKS ? f.n_tiles : CM_MINMAX_BLOCKS;
    cmk_minmax(st, c->d_frame, c->partials, n_partials, nullptr);
    std::vector<float> rec(static_cast<size_t>(n_partials) * 8);
    HIP_TRY(c, hipMemcpyAsync(rec.data(), c->partials, rec.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    const float inf = std::numeric_limits<float>::infinity();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    uint64_t cnt = 0;
    for (uint32_t r = 0; r < n_partials; ++r) {
        uint32_t k;
        std::memcpy(&k, &rec[r * 8 + 6], 4);
        if (!k) continue;
        cnt += k;
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], rec[r * 8 + a]);
            mx[a] = std::max(mx[a], rec[r * 8 + 3 + a]);
        }
    }
    c->pred_ok = false;
    if (cnt) {
        float leaf[3];
        for (int a = 0; a < 3; ++a) leaf[a] = 1.0f / f.inv_leaf[a];
        set_predicted_box(c, mn, mx, leaf);
    }
    return CM_OK;
}

// The launch sequence of cm_kernels_v2.hip for the frame in c->frame: n_global 8-bit passes over the
// key bits above `low_bits`, then the local finish.
int bucket_buffers(cm_ctx* c) {
    const size_t npad = c->cap_padded;
    if (!c->rec_a) HIP_TRY(c, hipMalloc(&c->rec_a, npad * 16));
    if (!c->rec_b) HIP_TRY(c, hipMalloc(&c->rec_b, npad * 16));
    if (!c->dig) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->dig), npad));
    if (!c->tile_state) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->tile_state), (npad / 1024 + 2) * 8));
    if (!c->records) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->records), static_cast<size_t>(c->cap_tiles) * 32));
    if (!c->wave_cnt) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->wave_cnt), static_cast<size_t>(c->cap_tiles) * CM2_WAVES * 4));
    for (auto& p : c->spl)
        if (!p) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p), (CM4_MAX_BUCKETS + 4) * 4));
    if (!c->qcnt) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->qcnt), static_cast<size_t>(std::min<uint32_t>(c->cap_tiles, CM4_MAX_TILES)) * (CM4_BINS / 2) * 4));
    if (!c->qtot) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->qtot), CM4_BINS * 4));
    if (!c->qbid) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->qbid), static_cast<size_t>(std::min<uint32_t>(c->cap_tiles, CM4_MAX_TILES)) * CM_TILE * 2));
    if (!c->qbofs) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->qbofs), (CM4_BINS + 4) * 4));
    if (!c->qbig) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->qbig), (CM4_MAX_BIG + 4) * 4));
    return CM_OK;
}

// Global passes for keys of kb bits over about `est` points: enough that at most CM2_MAX_LOW_BITS index bits are
// left to the local finish, and enough that an average bucket (points / 2^(8 g)) stays well inside its LDS
// capacity. 0: the bucket kernels do not fit this grid.
uint32_t bucket_passes(uint32_t kb, uint64_t est, uint32_t extra) {
    // Dense frames — on average a point or more per cell of the box (the reference's own 10 cm grid on its ROI, or any
    // coarse leaf): sort the whole index globally. The finish then has nothing left to sort, a "bucket" is one voxel, and
    // a voxel of any size is summed by the long-run jobs of k3_local: no bucket can be too large, nothing is handed back.
    if (kb <= 8 * CM_MAX_PASSES && (est >> kb) >= 1) return (kb + 7) / 8;
    uint32_t g = 1 + (kb > CM2_MAX_LOW_BITS + 8 ? (kb - CM2_MAX_LOW_BITS - 1) / 8 : 0);
    while (g < CM_MAX_PASSES && (est >> (8 * g)) > 256) ++g;
    g += extra;
    if (g > 1 && 8 * (g - 1) >= kb) return 0;         // nothing left for the local finish to add
    return g <= CM_MAX_PASSES ? g : 0;
}

// A crop box that dropped more than half of the last frame's points: k2_hist0 then also packs the survivors' records
// (into the record buffer the first scatter does not write), and the first scatter reads those instead of going
// through every raw point a second time — the raw clouds are read once, not twice.
bool pack_survivors(const cm_ctx* c) {
    return c->frame.crop_enable && c->last_n_merged && 2 * c->last_n_merged < c->n_in;
}

int launch_bucket(cm_ctx* c, int grid_mode, uint32_t n_global, uint32_t low_bits, const unsigned char* mask,
                  const CmFrameState* st_outlier, int mode = 0, bool quant = false) {
    CmFrameDev& f = c->frame;
    hipStream_t st = c->stream;
    { const int e = bucket_buffers(c); if (e != CM_OK) return e; }
    c->lb_grid_mode = grid_mode; c->lb_g = n_global; c->lb_low = low_bits; c->lb_mode = mode;
    c->last_quant = quant;
    c->wrote_spl = false;
    if (quant) {
        // One global pass into the buckets the last frame's quantiles cut (cm_kernels_v4.hip), one finish workgroup per bucket.
        const bool do_setup_q = !c->frame_uploaded_valid || std::memcmp(&f, &c->frame_uploaded, sizeof f) != 0;
        if (do_setup_q) { c->frame_uploaded = f; c->frame_uploaded_valid = true; }
        CmFrameState* state = c->d_state[c->cur];
        CmFrameState* state_next = c->d_state[c->cur ^ 1];
        const bool predicted = grid_mode == 2;
        c->from_crop = grid_mode == 1;
        c->last_v2 = true; c->last_predicted = predicted; c->last_packed = false; c->last_k3 = true;
        c->frame_mask = nullptr;
        const uint32_t nt = f.n_tiles;
        const uint32_t nb = cm_quant_buckets(c->spl_n);
        // more buckets than the pass has bins: 2^sub neighbouring buckets share a bin, the pass leaves the low bits of every
        // record's bucket number as a byte beside it (c->dig) and the finish picks its records out of the bin (k3_local<SUB>)
        const uint32_t sub = cm_quant_sub_shift(nb);
        const uint32_t nbins = (nb + (1u << sub) - 1u) >> sub;
        const uint32_t* spl = c->spl[c->spl_cur];
        uint32_t* spl_next = c->spl[c->spl_cur ^ 1];
        // tile_info (one word pair per bucket) and, behind it, the group totals of the kept voxels: zeroed by k4_hist. The
        // number of buckets comes from the LAST frame's size — a frame of a twentieth of its predecessor's points has fewer
        // slots / 1024 than buckets (found by scripts/fuzz_shared_bins.py: the totals were then left as the last frame had them)
        const uint32_t n_tile_state = std::max<uint32_t>(f.n_padded / 1024 + 2, nb + nb / 64 + 2);
        prof_mark(c, "k4_hist");
        cmk4_hist(st, f, c->d_frame, c->d_tiles, do_setup_q, state, spl, c->qcnt, c->qbid, c->tile_state, n_tile_state, c->records,
                  grid_mode, predicted ? 1 : 0, nt, nb, sub ? nullptr : c->qbig, sub);
        prof_mark(c, "k4_colscan");
        // The large finish shape (buckets of up to CM4_CAP_BIG records, one workgroup per CU) costs a launch of its own — 6 us on a
        // frame alone even when it has nothing to do — so it is only armed for 16 frames behind a hand-back or a frame that used
        // it; unarmed, any bucket beyond the usual shape's capacity hands the frame back (and arms it).
        const bool big_armed = !sub && c->quant_big_arm > 0;
        if (c->quant_big_arm) --c->quant_big_arm;
        // (shared bins: a bin beyond 2^sub finish capacities holds a bucket beyond one; the finish itself checks the buckets)
        if (sub) cmk4_colscan(st, state, c->h_state_dev, c->qcnt, c->qtot, nt, CM4_CAP << sub, CM4_CAP << sub, nullptr);
        else cmk4_colscan(st, state, c->h_state_dev, c->qcnt, c->qtot, nt, CM4_CAP, big_armed ? CM4_CAP_BIG : CM4_CAP, c->qbig);
        prof_mark(c, "k4_scatter");
        const bool ballot = !c->lds_rank;                // ranks by ballots where the returning LDS adds are not (known to be) lane-ordered
        cmk4_scatter(st, c->d_frame, c->d_tiles, state, c->qbid, c->qcnt, c->qtot, c->qbofs, nbins, c->rec_a, c->records, nt,
                     predicted ? 1 : 0, c->d_tile_kept, nt, sub ? c->dig : nullptr, ballot, sub ? nullptr : c->qbig, sub);
        const void* rec_sorted = c->rec_a;
        void* stage = c->rec_b;
        const uint32_t* bofs = c->qbofs;
        // (tile_info: one word pair per bucket; the group totals of the kept voxels behind them — n_tile_state words, see above)
        uint32_t* grp_cnt = reinterpret_cast<uint32_t*>(c->tile_state + nb);
        uint32_t* skey = c->out_key ? c->keys_a : nullptr;
        prof_mark(c, "k3_local");
        cmk3_local(st, c->d_frame, state, c->h_state_dev, rec_sorted, c->tile_state, grp_cnt, stage, skey, c->vals_a, false, 0u,
                   0u, spl, bofs, nb, spl_next, ballot, sub, sub ? c->dig : nullptr);
        if (big_armed) {
            // the few buckets that grew beyond what the usual finish workgroup holds (k4_colscan listed them): the large shape
            prof_mark(c, "k3_local(big)");
            cmk3_local_big(st, c->d_frame, state, c->h_state_dev, rec_sorted, c->tile_state, grp_cnt, stage, skey, c->vals_a, spl, bofs, nb,
                           spl_next, c->qbig, ballot);
        }
        c->wrote_spl = true;
        prof_mark(c, "k3_compact");
        cmk3_compact(st, state, state_next, c->h_state_dev, c->tile_state, grp_cnt, stage, skey, c->vals_a, c->out, c->out_key,
                     c->out_cnt, false, 0u, nb);
        prof_mark(c, "end");
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev_done, st));
        c->cur ^= 1;
        c->in_flight.store(true);
        c->pending = true;
        c->pending_trivial = false;
        return CM_OK;
    }
    // (a descriptor that changed since the last frame — new clouds, new poses — goes to HBM with k2_hist0 itself)
    const bool do_setup = !c->frame_uploaded_valid || std::memcmp(&f, &c->frame_uploaded, sizeof f) != 0;
    if (do_setup) {
        c->frame_uploaded = f;
        c->frame_uploaded_valid = true;
    }
    CmFrameState* state = c->d_state[c->cur];
    CmFrameState* state_next = c->d_state[c->cur ^ 1];
    const bool predicted = grid_mode == 2 && mode == 0;   // mode 1: the bounds handed in are the fused cloud's own
    c->from_crop = grid_mode == 1 || (grid_mode == 2 && !predicted);
    c->last_v2 = true;
    c->last_predicted = predicted;
    if (mode == 1 && !c->partial) HIP_TRY(c, hipMalloc(&c->partial, static_cast<size_t>(c->cap_padded) * 32));
    const uint32_t nt = f.n_tiles;
    const uint32_t n_groups = (nt + CM_GROUP - 1) / CM_GROUP;
    const uint32_t gw = n_groups * CM_RADIX;
    const size_t gstride = static_cast<size_t>(c->cap_groups) * CM_RADIX;
    uint32_t* grp0 = c->grp + gstride * (c->frame_seq & 1u);
    uint32_t* grp0_next = c->grp + gstride * ((c->frame_seq & 1u) ^ 1u);
    ++c->frame_seq;
    c->frame_mask = mask;
    const bool pack = !predicted && pack_survivors(c);
    c->last_packed = pack;
    prof_mark(c, "k2_hist0");
    cmk2_hist0(st, f, c->d_frame, c->d_tiles, do_setup, state, c->hist, grp0, grp0_next, c->grp + 2 * gstride, gw, static_cast<uint32_t>(gstride),
               c->tile_state, f.n_padded / 1024 + 2, c->records, grid_mode, predicted ? 1 : 0, low_bits, n_global, nt, mask,
               st_outlier, 0, pack ? c->rec_b : nullptr, c->wave_cnt);
    // The passes behind the first, and the finish, work on the records pass 0 kept. When a crop box dropped most points of
    // the last frame their grids are sized for what that frame kept (+ 50 % + two tiles), not for the padded frame — most of
    // those workgroups would only find out that they have nothing to do (cfg3: 3906 / 7812 of them for 157 / 313 tiles of
    // records). Verified on the device: k3_compact raises CM_DEV_ERR_GRID when the records need more, the frame is redone
    // and the next frames use whole grids again.
    uint32_t nt_later = nt;
    if (mode == 0 && !predicted && f.crop_enable && c->last_n_merged && !c->grid_shrink_off && c->finish_mode != 2) {
        const uint64_t est = static_cast<uint64_t>(c->last_n_merged) + c->last_n_merged / 2 + 2 * CM_TILE;
        nt_later = static_cast<uint32_t>(std::min<uint64_t>(nt, (est + CM_TILE - 1) / CM_TILE));
    }
    if (c->grid_shrink_off) --c->grid_shrink_off;
    const uint32_t n_groups_later = (nt_later + CM_GROUP - 1) / CM_GROUP;
    for (uint32_t pass = 0; pass < n_global; ++pass) {
        uint32_t* grp = pass == 0 ? grp0 : c->grp + 2 * gstride + static_cast<size_t>(pass - 1) * gw;
        const uint32_t nt_p = pass == 0 ? nt : nt_later, n_groups_p = pass == 0 ? n_groups : n_groups_later;
        const bool big_p = n_groups_p > CM_DIRECT_GROUPS;
        if (pass > 0) { prof_mark(c, "k2_hist"); cmk2_hist(st, state, c->dig, c->hist, grp, nt_p); }
        if (big_p) { prof_mark(c, "k_gscan"); cmk_gscan(st, state, grp, c->totals, pass, n_groups_p); }
        prof_mark(c, "k2_scatter");
        const void* in = (pass & 1u) ? c->rec_a : c->rec_b;
        void* out = (pass & 1u) ? c->rec_b : c->rec_a;
        cmk2_scatter(st, pass == 0, c->d_frame, c->d_tiles, state, in, out, c->dig, c->hist, grp, big_p ? c->totals : nullptr,
                     low_bits + 8 * pass, pass + 1 < n_global ? low_bits + 8 * (pass + 1) : 32u, nt_p, n_groups_p,
                     f.n_padded, c->records, nt, predicted ? 1 : 0, mask, 0, (pack && pass == 0) ? c->rec_b : nullptr, c->wave_cnt,
                     (c->debug_misrank && pass + 1 == n_global) ? 1 : 0, c->d_tile_kept,
                     // (fewer than a sixteenth of the points survived the last frame's crop: eight tiles per workgroup)
                     // (... and few enough that a wave's share of a tile is one load: k2_scatter_sparse takes a chunk of more
                     // than 64 records through a loop — at 20 % survivors, the live node's ROI, 40 us against 7)
                     pack && pass == 0 && !c->debug_misrank && 16ull * c->last_n_merged < c->n_in, !c->lds_rank);
    }
    const void* rec_sorted = ((n_global - 1) & 1u) ? c->rec_b : c->rec_a;
    c->last_k3 = c->finish_mode != 2 || !c->lds_rank;     // (k2_local ranks by returning LDS adds only)
    if (c->last_k3) {
        // k3_local stages every tile's centroids in the record buffer the last pass read from (dead by now), at the
        // tile's own place; k3_compact moves them to `out`. Cells and counts (CM_FLAG_OCCUPANCY) ride in the general
        // path's key / value arrays, which the bucket path does not use.
        void* stage = ((n_global - 1) & 1u) ? c->rec_a : c->rec_b;
        if (mode == 1) {
            if (!c->stage32) HIP_TRY(c, hipMalloc(&c->stage32, static_cast<size_t>(c->cap_padded) * 32));
            stage = c->stage32;
        }
        uint32_t* grp_cnt = reinterpret_cast<uint32_t*>(c->tile_state + f.n_padded / 2048);
        uint32_t* skey = c->out_key ? c->keys_a : nullptr;
        prof_mark(c, "k3_local");
        // (the finish also leaves the quantiles of its sorted records: the next frame's splitters, cm_kernels_v4.hip; with
        // L = 0 a tile's sorted range may reach beyond what it holds in LDS — the frame then says so: CmFrameState.spl_incomplete)
        uint32_t* spl_next = mode == 0 ? c->spl[c->spl_cur ^ 1] : nullptr;
        c->wrote_spl = spl_next != nullptr;
        cmk3_local(st, c->d_frame, state, c->h_state_dev, rec_sorted, c->tile_state, grp_cnt, stage, skey, c->vals_a, mode == 1,
                   low_bits, nt_later * CM_TILE, nullptr, nullptr, 0u, spl_next, !c->lds_rank);
        prof_mark(c, "k3_compact");
        cmk3_compact(st, state, state_next, c->h_state_dev, c->tile_state, grp_cnt, stage, skey, c->vals_a,
                     mode == 1 ? c->partial : c->out, c->out_key, c->out_cnt, mode == 1, nt_later * CM_TILE);
    } else {
        prof_mark(c, "k2_local");
        cmk2_local(st, c->d_frame, state, state_next, c->h_state_dev, rec_sorted,
                   c->tile_state, reinterpret_cast<uint32_t*>(c->tile_state + (f.n_padded / 1024 + 1)), c->out, c->out_key,
                   c->out_cnt, mode == 1 ? c->partial : nullptr, low_bits, f.n_padded);
    }
    prof_mark(c, "end");
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_done, st));
    c->cur ^= 1;
    c->in_flight.store(true);
    c->pending = true;
    c->pending_trivial = false;
    return CM_OK;
}

// mode 0: the path (centroids). mode 1: partial table of per-voxel sums (fused cloud across GPUs);
// `bounds` (min xyz, max xyz of the whole fused cloud) then fixes the grid unless the crop box does.
// consume: reset the "fresh" flags (a frame was fused); false for cm_local_bounds' peek.
int enqueue(cm_ctx* c, const cm_params* p, int mode = 0, const float* bounds = nullptr) {
    if (!c || !p) return CM_BAD_ARG;
    for (int a = 0; a < 3; ++a)
        if (!(p->leaf[a] > 0.0f) || !std::isfinite(p->leaf[a])) return fail(c, CM_BAD_ARG, "leaf must be > 0");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->pending) return fail(c, CM_BAD_ARG, "previous frame not waited for (cm_wait)");

    // Everything that can reject the call is checked before the frame is assembled: assembling
    // consumes the sensors' "fresh" flags (:151-157), and a rejected call must not lose a frame.
    float inv_leaf[3], inv_cell[3] = {0.f, 0.f, 0.f};
    for (int a = 0; a < 3; ++a) inv_leaf[a] = 1.0f / p->leaf[a];
    const bool outl = p->outlier_enable != 0;
    if (outl && mode != 0) return fail(c, CM_BAD_ARG, "outlier removal needs the whole fused cloud on one GPU (not with partial tables)");
    if (c->ground_on && (outl || mode != 0)) return fail(c, CM_BAD_ARG, "ground removal is not combined with outlier_enable or partial tables");
    if (outl && (!(p->outlier_radius > 0.0f) || !std::isfinite(p->outlier_radius))) return fail(c, CM_BAD_ARG, "outlier_radius must be > 0");
    const bool outl_g = c->ground_on && c->ground_outlier_radius > 0.0f;      // the ground stage's own radius filter
    const bool any_outl = outl || outl_g;
    const float o_radius = outl ? p->outlier_radius : c->ground_outlier_radius;
    const uint32_t o_min_nb = outl ? p->outlier_min_neighbors : c->ground_outlier_min_nb;
    if (any_outl) for (int a = 0; a < 3; ++a) inv_cell[a] = 1.0f / (o_radius * 1.01f);   // candidate grid a little wider than r
    uint32_t key_bits = 0, kb_o = 0;
    int grid_mode = 0;                               // 0: data min/max (k_minmax), 1: crop box, 2: bounds handed in
    if (p->crop_enable && box_grid(p->crop_min, p->crop_max, inv_leaf, &key_bits)) grid_mode = 1;
    else if (mode == 1 && bounds && box_grid(bounds, bounds + 3, inv_leaf, &key_bits)) grid_mode = 2;
    else if (mode == 1) return fail(c, CM_BAD_ARG, "partial table needs the crop box or the fused cloud's bounds to fix the grid");
    int gm_o = 0;                                    // grid of the outlier stage: crop box or data min/max
    if (any_outl && p->crop_enable) {
        if (!box_grid(p->crop_min, p->crop_max, inv_cell, &kb_o))
            return fail(c, CM_CAPACITY, "outlier radius too small for the crop box (radius grid exceeds 32 bits)");
        gm_o = 1;
    }

    std::vector<std::unique_lock<std::mutex>> locks;
    const int bf = build_frame(c, p, true, locks);
    if (bf != CM_OK) return bf;
    CmFrameDev& f = c->frame;
    if (mode == 1 && bounds) {
        for (int a = 0; a < 3; ++a) { f.ext_min[a] = bounds[a]; f.ext_max[a] = bounds[3 + a]; }
    }
    if (any_outl) {
        for (int a = 0; a < 3; ++a) f.inv_cell[a] = inv_cell[a];
        f.outlier_r2 = static_cast<float>(static_cast<double>(o_radius) * static_cast<double>(o_radius));
        f.outlier_min_nb = o_min_nb;
    }
    c->have_result = false;
    c->out_is_merged = false;
    c->prof_used = 0;
    c->last_mode = mode;
    c->bytes_d2h = 0;
    if (c->pub_pending[0]) {
        // A copy-out (cm_result_publish_async) may still be reading the last frame's result: this frame writes the other
        // pair of buffers, and waits ON THE DEVICE for whatever copy-out read those (two frames ago: long done).
        std::swap(c->out, c->out_other);
        std::swap(c->out32, c->out32_other);
        std::swap(c->ev_pub[0], c->ev_pub[1]);
        std::swap(c->pub_pending[0], c->pub_pending[1]);
        if (c->pub_pending[0]) {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_pub[0], 0));
            c->pub_pending[0] = false;
        }
    }

    if (f.n_padded == 0) {                             // every submitted cloud is empty
        c->frame_had_ground = c->ground_on && mode == 0;   // ... so are the ground cloud and every slab (no stale planes)
        c->trivial_grid = false;
        if (mode == 1) {
            // An empty share of a fused cloud still belongs to the shared grid: cm_merge_tables on this context
            // reports and decodes cells with it.
            const float* lo = grid_mode == 1 ? p->crop_min : bounds;
            const float* hi = grid_mode == 1 ? p->crop_max : bounds + 3;
            uint32_t kb = 0;
            if (box_grid(lo, hi, inv_leaf, &kb, c->cell_min_b, c->cell_div_b)) {
                c->trivial_grid = true;
                for (int a = 0; a < 3; ++a) { c->trivial_box[a] = lo[a]; c->trivial_box[3 + a] = hi[a]; }
            }
        }
        c->pending = true;
        c->pending_trivial = true;
        return CM_OK;
    }

    c->last_params = *p;
    c->last_grid_mode = grid_mode;
    c->last_key_bits = key_bits;
    c->last_v2 = false;
    c->last_predicted = false;
    const bool spl_ok = c->spl_valid;        // (valid again once this frame has finished and left its own splitters)
    c->spl_valid = false;
    c->wrote_spl = false;
    c->last_quant = false;

    // Bucket path: centroids of one GPU's whole frame, with a box known before the first point is
    // read — the crop box, or the last frame's bounds plus a margin (verified on the device).
    // Frames with pre-stages (ground / outlier removal, which leave a keep-mask) can use it too when the crop box
    // fixes the grid: the pre-stages run first (launch_classic), then the bucket path takes the voxel stage.
    const bool pre = outl || c->ground_on;
    c->h_state->err = 0;                     // the bucket kernels write error words straight into the host record
    c->post_bucket = false;
    c->pre_bucket = false;
    c->last_outl = outl; c->last_gm_o = gm_o; c->last_kb_o = kb_o;
    // (without lane-ordered LDS adds — probe failed, CM_LDS_RANK=0, or a pass found mis-ranked — the bucket kernels rank by
    // ballots: same results, more instructions; the outlier stage's bucket sort, which builds on k2_local, then stays off)
    bool want_v2 = c->path_mode != 1 && (mode == 0 || mode == 1) && (!pre || grid_mode == 1);
    if (want_v2 && c->v2_off_frames) { --c->v2_off_frames; want_v2 = false; }
    if (want_v2) {
        int gm = grid_mode;
        uint32_t kb = key_bits;
        if (gm == 0) {
            if (!c->pred_ok) {
                const int e = bootstrap_box(c);
                if (e < 0) return e;
            }
            if (c->pred_ok && box_grid(c->pred_min, c->pred_max, inv_leaf, &kb, f.box_min_b, f.box_div_b)) {
                gm = 2;
                for (int a = 0; a < 3; ++a) { f.ext_min[a] = c->pred_min[a]; f.ext_max[a] = c->pred_max[a]; }
            } else {
                c->pred_ok = false;
            }
        }
        if (gm == 1 && !box_grid(p->crop_min, p->crop_max, inv_leaf, &kb, f.box_min_b, f.box_div_b)) gm = 0;
        if (gm == 2 && mode == 1 && !box_grid(bounds, bounds + 3, inv_leaf, &kb, f.box_min_b, f.box_div_b)) gm = 0;
        // (the bucket kernels form the linear index on the 24-bit multiplier: fewer than 2^24 cells per axis)
        for (int a = 0; a < 3 && gm != 0; ++a)
            if (f.box_div_b[a] >= (1 << 24)) gm = 0;
        if (gm != 0) {
            f.box_key_bits = kb;
            f.box_predicted = (gm == 2 && mode == 0) ? 1u : 0u;
            // (points: what the last frame kept after crop and masks, plus a quarter, when there was one; a frame
            // that overflows anyway is handed back and v2_extra_passes adds a pass for the frames after it)
            const uint64_t est = c->last_n_merged ? std::min<uint64_t>(c->n_in, c->last_n_merged + c->last_n_merged / 4) : c->n_in;
            const uint32_t g = bucket_passes(kb, est, c->v2_extra_passes);
            if (g) {
                const uint32_t low = kb > 8 * g ? kb - 8 * g : 0;
                // Quantile passes (one global pass instead of g): the last frame of this context left the quantiles of its
                // sorted records, as indices of this very grid, and this frame is about as large.
                bool quant = mode == 0 && !pre && c->quant_mode != 1 && c->finish_mode != 2 && spl_ok && g >= 2 &&
                             kb < 32 && f.n_tiles <= CM4_MAX_TILES && !(gm == 1 && pack_survivors(c)) &&
                             std::memcmp(c->spl_min_b, f.box_min_b, sizeof c->spl_min_b) == 0 &&
                             std::memcmp(c->spl_div_b, f.box_div_b, sizeof c->spl_div_b) == 0 &&
                             std::memcmp(c->spl_inv_leaf, f.inv_leaf, sizeof c->spl_inv_leaf) == 0;
                if (quant) {
                    const uint32_t nb = cm_quant_buckets(c->spl_n);
                    quant = nb != 0 && c->spl_n / nb <= CM4_MAX_AVG && est <= 2ull * c->spl_n + CM_TILE &&
                            nb + nb / 64 + 2 <= c->cap_padded / 1024 + 2 &&      // (tile_info + group totals fit their array)
                            // Above 2048 buckets: still one pass, 2 or 4 neighbouring buckets to a bin (cm_device.h cm_quant_sub_shift;
                            // cfg3's dense variant, 13.7 M records: 0.38-0.41 against 0.46-0.50 ms per frame for three fixed-grid passes).
                            (nb <= CM4_BINS || c->quant_sub);
                }
                if (quant && c->quant_off_frames) { --c->quant_off_frames; quant = false; }
                if (!pre) return launch_bucket(c, gm, g, low, nullptr, nullptr, mode, quant);
                c->post_bucket = true; c->post_g = g; c->post_low = low;
                // the outlier stage's own sort can use the bucket kernels as well: the crop box fixes its grid too
                c->pre_bucket = c->lds_rank && gm_o == 1 && box_grid(p->crop_min, p->crop_max, inv_cell, &kb_o, f.cell_min_b, f.cell_div_b) &&
                                static_cast<uint64_t>(f.cell_div_b[1]) * static_cast<uint64_t>(f.cell_div_b[2]) <= CM_ROW_TABLE_CAP &&
                                f.cell_div_b[0] < (1 << 24) && f.cell_div_b[1] < (1 << 24) && f.cell_div_b[2] < (1 << 24);
                f.cell_key_bits = kb_o; f._pad_cell = 0;
                if (c->pre_bucket && c->pre_bucket_off) { --c->pre_bucket_off; c->pre_bucket = false; }
            }
        }
    }
    return launch_classic(c, p, mode, grid_mode, key_bits, outl, gm_o, kb_o);
}

// The launch sequence of cm_kernels.hip for the frame in c->frame.
int launch_classic(cm_ctx* c, const cm_params* p, int mode, int grid_mode, uint32_t key_bits, bool outl, int gm_o,
                   uint32_t kb_o) {
    CmFrameDev& f = c->frame;
    hipStream_t st = c->stream;
    if (!c->frame_uploaded_valid || std::memcmp(&f, &c->frame_uploaded, sizeof f) != 0) {
        prof_mark(c, "k_setup");
        cmk_setup(st, f, c->d_frame, c->d_tiles);
        c->frame_uploaded = f;
        c->frame_uploaded_valid = true;
    }
    CmFrameState* state = c->d_state[c->cur];
    CmFrameState* state_next = c->d_state[c->cur ^ 1];
    c->from_crop = grid_mode != 0;
    const uint32_t passes = c->from_crop ? (key_bits + CM_RADIX_BITS - 1) / CM_RADIX_BITS : CM_MAX_PASSES;
    const uint32_t nt = f.n_tiles;
    const uint32_t nseg = f.n_padded / CM_SEG_TILE;

    const uint32_t n_partials = nt < CM_MINMAX_BLOCKS ? nt : CM_MINMAX_BLOCKS;
    const uint32_t n_groups = (nt + CM_GROUP - 1) / CM_GROUP;
    const uint32_t gw = n_groups * CM_RADIX;                     // words of one group-total array
    // grp: [0],[1] pass-0 arrays (alternate per k_keys launch: it accumulates into one and clears the
    // other for the next launch), [2..4] passes 1..3 (cleared by k_keys, filled by k_hist).
    const size_t gstride = static_cast<size_t>(c->cap_groups) * CM_RADIX;
    const bool big = n_groups > CM_DIRECT_GROUPS;
    const uint32_t n_seg_groups = (nseg + CM_SEG_GROUP - 1) / CM_SEG_GROUP + 1;

    // keys + radix sort of one stage (the voxel grid, or the outlier stage's radius grid)
    auto keys_and_sort = [&](CmFrameState* stg, int gmode, int use_cell, const unsigned char* mask,
                             const CmFrameState* st_outlier, uint32_t n_pass) {
        uint32_t* grp0 = c->grp + gstride * (c->frame_seq & 1u);
        uint32_t* grp0_next = c->grp + gstride * ((c->frame_seq & 1u) ^ 1u);
        ++c->frame_seq;
        prof_mark(c, use_cell ? "k_keys(outlier)" : "k_keys");
        cmk_keys(st, c->d_frame, stg, c->keys_a, c->hist, grp0, grp0_next, c->grp + 2 * gstride, gw,
                 static_cast<uint32_t>(gstride), c->seg_groups, n_seg_groups, c->partials, n_partials, gmode,
                 use_cell, mask, st_outlier, nt);
        // k_keys clears 3*gw words starting at grp[2]; passes 1..3 therefore live at stride gw.
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
            const bool even = (pass & 1u) == 0;
            const uint32_t* kin = even ? c->keys_a : c->keys_b;
            const uint32_t* vin = even ? c->vals_a : c->vals_b;
            uint32_t* kout = even ? c->keys_b : c->keys_a;
            uint32_t* vout = even ? c->vals_b : c->vals_a;
            uint32_t* grp = pass == 0 ? grp0 : c->grp + 2 * gstride + static_cast<size_t>(pass - 1) * gw;
            if (pass > 0) { prof_mark(c, "k_hist"); cmk_hist(st, stg, kin, c->hist, grp, pass, nt); }
            if (big) { prof_mark(c, "k_gscan"); cmk_gscan(st, stg, grp, c->totals, pass, n_groups); }
            prof_mark(c, "k_scatter");
            cmk_scatter(st, stg, kin, vin, kout, vout, c->hist, grp, big ? c->totals : nullptr, pass, nt,
                        n_groups, f.n_padded, c->lds_rank, use_cell ? nullptr : c->d_tile_kept);
        }
    };

    // Radius outlier filter over the points `in` marks (nullptr: every valid point), neighbours counted inside a
    // point's class only when `cls` is given; survivors are marked in `out`.
    auto radius_filter = [&](const unsigned char* in, const unsigned char* cls, unsigned char* out) -> int {
        if (!c->sorted_pts) HIP_TRY(c, hipMalloc(&c->sorted_pts, static_cast<size_t>(c->cap_padded) * 16));
        if (!c->rows) HIP_TRY(c, hipMalloc(&c->rows, static_cast<size_t>(CM_ROW_TABLE_CAP) * 8));
        if (!c->d_state_o) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_state_o), sizeof(CmFrameState)));
        const uint32_t passes_o = gm_o ? (kb_o + CM_RADIX_BITS - 1) / CM_RADIX_BITS : CM_MAX_PASSES;
        HIP_TRY(c, hipMemsetAsync(c->d_state_o, 0, sizeof(CmFrameState), st));
        const uint32_t g = c->pre_bucket ? bucket_passes(kb_o, c->n_in, 0) : 0;
        if (g) {
            // Bucket kernels on the radius grid: records (x, y, z, padded index) grouped by the high key bits in g
            // passes, then sorted tile by tile in LDS and written back in order — what the general path's (key, index)
            // sort + gather produce, in fewer passes over less data. A radius cell too full for a tile hands the
            // frame back (CM_DEV_ERR_BUCKET_PRE).
            { const int e = bucket_buffers(c); if (e != CM_OK) return e; }
            const uint32_t low = kb_o > 8 * g ? kb_o - 8 * g : 0;
            uint32_t* grp0 = c->grp + gstride * (c->frame_seq & 1u);
            uint32_t* grp0_next = c->grp + gstride * ((c->frame_seq & 1u) ^ 1u);
            ++c->frame_seq;
            const bool pack = pack_survivors(c);
            prof_mark(c, "k2_hist0(outlier)");
            cmk2_hist0(st, f, c->d_frame, c->d_tiles, false, c->d_state_o, c->hist, grp0, grp0_next, c->grp + 2 * gstride, gw, static_cast<uint32_t>(gstride),
                       c->tile_state, f.n_padded / 1024 + 2, c->records, 1, 0, low, g, nt, in, nullptr, 1,
                       pack ? c->rec_b : nullptr, c->wave_cnt);
            for (uint32_t pass = 0; pass < g; ++pass) {
                uint32_t* grp = pass == 0 ? grp0 : c->grp + 2 * gstride + static_cast<size_t>(pass - 1) * gw;
                if (pass > 0) { prof_mark(c, "k2_hist"); cmk2_hist(st, c->d_state_o, c->dig, c->hist, grp, nt); }
                if (big) { prof_mark(c, "k_gscan"); cmk_gscan(st, c->d_state_o, grp, c->totals, pass, n_groups); }
                prof_mark(c, "k2_scatter(outlier)");
                cmk2_scatter(st, pass == 0, c->d_frame, c->d_tiles, c->d_state_o, (pass & 1u) ? c->rec_a : c->rec_b, (pass & 1u) ? c->rec_b : c->rec_a,
                             c->dig, c->hist, grp, big ? c->totals : nullptr, low + 8 * pass, pass + 1 < g ? low + 8 * (pass + 1) : 32u,
                             nt, n_groups, f.n_padded, c->records, nt, 0, in, 1, (pack && pass == 0) ? c->rec_b : nullptr, c->wave_cnt);
            }
            prof_mark(c, "k2_local(sort)");
            cmk2_local_sort(st, c->d_frame, c->d_state_o, c->h_state_dev, ((g - 1) & 1u) ? c->rec_b : c->rec_a,
                            (g & 1u) ? c->keys_b : c->keys_a, c->sorted_pts, low, f.n_padded);
        } else {
            if (!gm_o) { prof_mark(c, "k_minmax"); cmk_minmax(st, c->d_frame, c->partials, n_partials, in); }
            keys_and_sort(c->d_state_o, gm_o, 1, in, nullptr, passes_o);
        }
        prof_mark(c, "outlier_mask");
        cmk_outlier_mask(st, c->d_frame, c->d_state_o, c->keys_a, c->vals_a, c->keys_b, c->vals_b, c->sorted_pts,
                         c->rows, out, f.n_padded, cls, c->merged_total + 8, g != 0);
        return CM_OK;
    };
    const bool ground_outl = c->ground_on && mode == 0 && c->ground_outlier_radius > 0.0f;
    c->frame_mask = nullptr;
    c->frame_had_ground = false;
    if (c->ground_on && mode == 0) {
        // Zone-wise ground removal first (per sensor, before the fuse): it leaves a keep-mask for the voxel
        // grid and the fused no-ground cloud, and a ground mask for the fused ground cloud.
        if (!c->mask) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->mask), c->cap_padded));
        if (!c->gmask) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->gmask), c->cap_padded));
        if (!c->sorted_pts) HIP_TRY(c, hipMalloc(&c->sorted_pts, static_cast<size_t>(c->cap_padded) * 16));
        if (!c->d_ground) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_ground), sizeof(CmGroundDev)));
        if (!c->d_state_g) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_state_g), sizeof(CmFrameState)));
        if (!c->zone_off) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->zone_off), (CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES + 1) * 4));
        if (!c->d_planes) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_planes), CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES * sizeof(CmGroundPlaneDev)));
        {
            const size_t nh = static_cast<size_t>(CM_DEV_MAX_SENSORS) * CM_DEV_MAX_ZONES * CM_GROUND_BATCH;
            if (!c->hyp0) HIP_TRY(c, hipMalloc(&c->hyp0, nh * 16));
            if (!c->valid0) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->valid0), nh * 4));
            if (!c->counts0) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->counts0), nh * 4));
            if (!c->chunk_sums) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->chunk_sums), (static_cast<size_t>(c->cap_padded) / CM_GROUND_CHUNK + CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES + 1) * 10 * sizeof(double)));
        }
        if (ground_outl) {
            if (!c->bmask) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->bmask), c->cap_padded));
            if (!c->zcode) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->zcode), c->cap_padded));
        }
        if (!c->ground_uploaded) { cmkg_setup(st, c->ground, c->d_ground); c->ground_uploaded = true; }
        // (masks and the slab sort's state cleared by ONE launch: they were five hipMemsetAsync calls per tick)
        prof_mark(c, "kg_clear");
        cmkg_clear(st, f.n_padded, c->mask, 0, c->gmask, 0, ground_outl ? c->bmask : nullptr, 0, ground_outl ? c->zcode : nullptr, 0xFF,
                   c->d_state_g, nullptr);
        uint32_t* grp0 = c->grp + gstride * (c->frame_seq & 1u);
        uint32_t* grp0_next = c->grp + gstride * ((c->frame_seq & 1u) ^ 1u);
        ++c->frame_seq;
        prof_mark(c, "kg_classify");
        cmkg_classify(st, c->d_frame, c->d_ground, c->d_state_g, c->keys_a, c->hist, grp0, grp0_next, c->grp + 2 * gstride,
                      gw, static_cast<uint32_t>(gstride), c->mask, ground_outl ? c->zcode : nullptr, nt);
        if (big) { prof_mark(c, "k_gscan"); cmk_gscan(st, c->d_state_g, grp0, c->totals, 0, n_groups); }
        prof_mark(c, "k_scatter(slabs)");
        cmk_scatter(st, c->d_state_g, c->keys_a, c->vals_a, c->keys_b, c->vals_b, c->hist, grp0, big ? c->totals : nullptr, 0, nt,
                    n_groups, f.n_padded, c->lds_rank);
        prof_mark(c, "kg_ransac");
        cmkg_planes(st, c->d_frame, c->d_ground, c->d_state_g, c->keys_b, c->vals_b, c->sorted_pts, c->zone_off, c->hyp0,
                    c->valid0, c->counts0, c->chunk_sums, c->d_planes, ground_outl ? c->bmask : c->mask, c->gmask, f.n_padded);
        if (ground_outl) {
            // removeGround's outlierRemoval(no_ground_cloud_ptr) (:119): among the band points of a slab that are
            // not ground, those with no neighbour within the radius go; survivors join the keep-mask
            const int e = radius_filter(c->bmask, c->zcode, c->mask);
            if (e != CM_OK) return e;
        }
        c->frame_mask = c->mask;
        c->frame_had_ground = true;
    }
    if (outl) {
        // Radius outlier removal first: it decides which points the voxel grid sees at all.
        if (!c->mask) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->mask), c->cap_padded));
        HIP_TRY(c, hipMemsetAsync(c->mask, 0, f.n_padded, st));
        const int e = radius_filter(nullptr, nullptr, c->mask);
        if (e != CM_OK) return e;
        c->frame_mask = c->mask;
    }
    if (c->post_bucket) {                                  // the voxel stage goes to the bucket path, with the keep-mask
        c->post_bucket = false;
        return launch_bucket(c, 1, c->post_g, c->post_low, c->frame_mask, (outl || ground_outl) ? c->d_state_o : nullptr);
    }
    if (!c->from_crop) { prof_mark(c, "k_minmax"); cmk_minmax(st, c->d_frame, c->partials, n_partials, c->frame_mask); }
    keys_and_sort(state, grid_mode, 0, c->frame_mask, (outl || ground_outl) ? c->d_state_o : nullptr, passes);
    prof_mark(c, "k_seg_count");
    uint32_t* seg_groups = nseg > CM_SEG_DIRECT_TILES ? c->seg_groups : nullptr;
    cmk_seg_count(st, state, c->keys_a, c->keys_b, c->seg_tile_counts, seg_groups, mode == 1 ? 1u : f.min_pts, nseg);
    prof_mark(c, "k_seg_reduce");
    if (mode == 1 && !c->partial)
        HIP_TRY(c, hipMalloc(&c->partial, static_cast<size_t>(c->cap_padded) * 32));
    cmk_seg_reduce(st, mode, c->d_frame, state, state_next, c->h_state_dev, c->keys_a, c->vals_a, c->keys_b,
                   c->vals_b, c->seg_tile_counts, seg_groups, mode == 1 ? c->partial : c->out, c->out_key,
                   c->out_cnt, nseg);
    prof_mark(c, "end");
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_done, st));     // k_seg_reduce wrote the state record to h_state
    c->cur ^= 1;
    c->in_flight.store(true);
    c->pending = true;
    c->pending_trivial = false;
    return CM_OK;
}

// A frame of the bucket path whose predicted box a point left: k2_hist0 measured the cloud's exact bounds all the same
// (its per-tile records, folded by the first scatter's workgroup 0 before it left), so the frame is redone at once in a box
// around those — on the bucket path again, without the general path's min/max pass. false: not applicable (the caller
// redoes the frame on the general path).
bool redo_in_measured_box(cm_ctx* c, const CmFrameState& h0) {
    if (!h0.outside || h0.err || !c->last_predicted || c->last_mode != 0 || c->frame_mask || c->last_outl || c->ground_on ||
        c->path_mode == 1 || h0.n_valid_k0 == 0 || c->frame.n_padded == 0)
        return false;
    // Everything that can still say "no" works on copies: the frame's descriptor and the predicted box only change once the
    // redo is certain to be launched (a refusal leaves c->frame as the general path's redo expects it — ADVICE r2).
    CmFrameDev f = c->frame;
    const bool pred_ok0 = c->pred_ok;
    float pred_min0[3], pred_max0[3];
    std::memcpy(pred_min0, c->pred_min, sizeof pred_min0);
    std::memcpy(pred_max0, c->pred_max, sizeof pred_max0);
    auto refuse = [&]() {
        c->pred_ok = pred_ok0;
        std::memcpy(c->pred_min, pred_min0, sizeof pred_min0);
        std::memcpy(c->pred_max, pred_max0, sizeof pred_max0);
        return false;
    };
    float leaf[3], inv_leaf[3];
    for (int a = 0; a < 3; ++a) {
        if (!std::isfinite(h0.min_p[a]) || !std::isfinite(h0.max_p[a]) || h0.min_p[a] > h0.max_p[a]) return false;
        leaf[a] = c->last_params.leaf[a];
        inv_leaf[a] = f.inv_leaf[a];
    }
    set_predicted_box(c, h0.min_p, h0.max_p, leaf);
    uint32_t kb = 0;
    if (!box_grid(c->pred_min, c->pred_max, inv_leaf, &kb, f.box_min_b, f.box_div_b)) { refuse(); c->pred_ok = false; return false; }
    for (int a = 0; a < 3; ++a) {
        if (f.box_div_b[a] >= (1 << 24)) return refuse();
        f.ext_min[a] = c->pred_min[a]; f.ext_max[a] = c->pred_max[a];
    }
    f.box_key_bits = kb;
    f.box_predicted = 1u;
    const uint32_t g = bucket_passes(kb, h0.n_valid_k0, c->v2_extra_passes);
    if (!g) return refuse();
    const CmFrameDev f0 = c->frame;
    c->frame = f;
    c->h_state->err = 0;
    c->prof_used = 0;
    if (launch_bucket(c, 2, g, kb > 8 * g ? kb - 8 * g : 0, nullptr, nullptr, 0) == CM_OK) return true;
    c->frame = f0;
    return refuse();
}

int wait_frame(cm_ctx* c, cm_result* res) {
    if (!c) return CM_BAD_ARG;
    if (!c->pending) return fail(c, CM_BAD_ARG, "no frame enqueued");
    HIP_TRY(c, hipSetDevice(c->device));
    cm_result r;
    std::memset(&r, 0, sizeof r);
    r.n_sensors = c->n_sensors_used;
    r.n_in = c->n_in;
    if (c->pending_trivial) {
        r.status = CM_EMPTY_INPUT;
        if (c->last_mode == 1 && c->trivial_grid) {
            r.bounds_from_crop = 1;
            for (int a = 0; a < 3; ++a) {
                r.min_b[a] = c->cell_min_b[a]; r.div_b[a] = c->cell_div_b[a]; r.max_b[a] = r.min_b[a] + r.div_b[a] - 1;
                r.min_p[a] = c->trivial_box[a]; r.max_p[a] = c->trivial_box[3 + a];
            }
        }
    } else {
        HIP_TRY(c, hipEventSynchronize(c->ev_done));
        c->in_flight.store(false);
        bool redone = false;
        if (c->last_v2) {
            // The bucket path hands a frame back when a point lay outside the predicted box, when a
            // bucket did not fit LDS, or when a workgroup gave up waiting for its predecessors: the
            // classic path redoes it (the sensors' clouds are still in place) and the cause is dealt with.
            const CmFrameState& h0 = *c->h_state;
            // A frame of the quantile passes whose buckets did not come out as predicted (one too large for the finish, or —
            // never seen — an index outside its bucket's range): the splitters are stale. Redone at once with the fixed-grid
            // passes in the same box, which leave fresh splitters; the quantile passes rest for a few frames.
            const bool quant_fail = c->last_quant && !h0.outside &&
                                    (h0.err == CM_DEV_ERR_QUANT || h0.err == CM_DEV_ERR_UNSORTED || h0.err == CM_DEV_ERR_BUCKET);
            if (quant_fail) {
                if (getenv("CM_VERBOSE")) std::fprintf(stderr, "[cloudmerge] quantile frame handed back: err %u, n_valid %u, spl_n %u\n", h0.err, h0.n_valid, c->spl_n);
                // The redone frame leaves the splitters of THIS scene, so the next frame may try at once: an abrupt change costs
                // one hand-back. A hand-back costs about a quarter of a frame more than the fixed-grid passes alone and a good
                // attempt saves a sixth, so attempts pay while fewer than one in three fail: the quantile passes only rest —
                // 8, 16, ... 128 frames — once three of the last eight attempts were handed back (a scene whose dense surfaces
                // keep moving across voxel layers: the index is z-major, so a ground plane that tilts by half a voxel at range
                // moves its points to other buckets).
                c->quant_hist = ((c->quant_hist << 1) | 1u) & 0xFFu;
                c->quant_good = 0;
                c->quant_big_arm = 16;                     // (the next frames may have buckets of two to four times the usual size: the large shape takes them)
                if (__builtin_popcount(c->quant_hist) >= 3) {
                    c->quant_off_frames = c->quant_rest;
                    if (c->quant_rest < 128) c->quant_rest *= 2;
                    c->quant_hist = 0;
                }
                ++c->n_redone;
                redone = true;
                c->h_state->err = 0;
                c->prof_used = 0;
                const int e = launch_bucket(c, c->lb_grid_mode, c->lb_g, c->lb_low, nullptr, nullptr, c->lb_mode, false);
                if (e != CM_OK) { c->pending = false; return e; }
                HIP_TRY(c, hipEventSynchronize(c->ev_done));
                c->in_flight.store(false);
            }
            // (h0 is the host record: by now the redone frame's, should there have been one)
            if (h0.outside || h0.err == CM_DEV_ERR_BUCKET || h0.err == CM_DEV_ERR_BUCKET_PRE || h0.err == CM_DEV_ERR_LOOKBACK ||
                h0.err == CM_DEV_ERR_UNSORTED || h0.err == CM_DEV_ERR_GRID) {
                if (h0.err == CM_DEV_ERR_GRID) c->grid_shrink_off = 64;   // more records than the last frame promised: whole grids for a while
                if (h0.outside) c->pred_ok = false;
                // The finish found records out of bucket order: a global pass mis-ranked. Stop trusting lane-ordered LDS adds
                // on this device: from here on every kernel of the context ranks by ballots (this frame is redone on the
                // general path; the next ones take the bucket path again, ballot-ranked).
                if (h0.err == CM_DEV_ERR_UNSORTED) { c->lds_rank = false; c->h_state->err = 0; c->debug_misrank = 0; }   // (the test hook fires once)
                if (h0.err == CM_DEV_ERR_BUCKET) {
                    if (c->v2_extra_passes < CM_MAX_PASSES) ++c->v2_extra_passes;
                    if (c->v2_good_frames < 8 && c->v2_retry_after < (1u << 20)) c->v2_retry_after *= 2;   // the retry failed at once
                    c->v2_good_frames = 0;
                }
                if (h0.err == CM_DEV_ERR_BUCKET_PRE) {        // a radius cell too full for a tile: more passes would not help
                    c->pre_bucket_off = c->pre_bucket_backoff;
                    if (c->pre_bucket_backoff < (1u << 20)) c->pre_bucket_backoff *= 2;
                }
                if (h0.err == CM_DEV_ERR_LOOKBACK) c->v2_off_frames = 0xFFFFFFFFu;
                if (!redone) ++c->n_redone;
                redone = true;
                bool settled = false;
                if (redo_in_measured_box(c, h0)) {            // (a box miss and nothing else: the same path, in a box that fits)
                    HIP_TRY(c, hipEventSynchronize(c->ev_done));
                    c->in_flight.store(false);
                    const CmFrameState& h1 = *c->h_state;     // (the record the redone frame has written by now)
                    settled = !h1.outside && !h1.err;
                    if (h1.outside) c->pred_ok = false;
                }
                if (!settled) {                               // every other cause, or a second hand-back: the general path
                c->prof_used = 0;
                c->last_v2 = false;
                c->last_predicted = false;
                {
                    // The frame's clouds are still where they were: a slot's active buffer is not written to before the
                    // next frame is enqueued, whatever the subscriber threads submit meanwhile. Same descriptor, general kernels.
                    const cm_params pr = c->last_params;
                    c->post_bucket = false;
                    c->pre_bucket = false;
                    int e = c->frame.n_padded ? launch_classic(c, &pr, c->last_mode, c->last_grid_mode, c->last_key_bits, c->last_outl,
                                                               c->last_gm_o, c->last_kb_o)
                                              : CM_INTERNAL;
                    if (e != CM_OK) { c->pending = false; return e == CM_INTERNAL ? fail(c, CM_INTERNAL, "frame could not be redone") : e; }
                }
                HIP_TRY(c, hipEventSynchronize(c->ev_done));
                c->in_flight.store(false);
                }
                r.n_sensors = c->n_sensors_used;
                r.n_in = c->n_in;
            }
        }
        // (counted on the general path too: extra passes can add up to "no bucket path at all", and that must not be for ever)
        if (!redone && c->v2_extra_passes && ++c->v2_good_frames >= c->v2_retry_after) {
            --c->v2_extra_passes;                          // the scene may have thinned out: try with less global sorting
            c->v2_good_frames = 0;
        }
        const CmFrameState& h = *c->h_state;
        if (h.err) {
            c->pending = false;
            if (h.err == CM_DEV_ERR_UNSORTED && c->lds_rank) {
                // The sorted keys were not sorted: stop trusting lane-ordered LDS adds on this device.
                c->lds_rank = false;
                return fail(c, CM_INTERNAL, "radix sort check failed with LDS-add ranking; switched to ballot ranking, resubmit the frame");
            }
            return fail(c, CM_INTERNAL, "device reported an internal error");
        }
        if (h.status == CM_DEV_ABORTED) {                  // (a stage gave up and nobody redid the frame: cannot happen)
            c->pending = false;
            return fail(c, CM_INTERNAL, "a device stage aborted the frame");
        }
        if (h.status == CM_DEV_OUTLIER_GRID) {
            c->pending = false;
            return fail(c, CM_CAPACITY, "outlier radius too small for the cloud's extent (radius grid exceeds its limits)");
        }
        r.status = h.status;
        r.bounds_from_crop = c->from_crop ? 1u : 0u;
        for (int a = 0; a < 3; ++a) {
            r.min_b[a] = h.min_b[a]; r.max_b[a] = h.max_b[a]; r.div_b[a] = h.div_b[a];
            r.min_p[a] = h.min_p[a]; r.max_p[a] = h.max_p[a];
            c->cell_min_b[a] = h.min_b[a]; c->cell_div_b[a] = h.div_b[a];
        }
        r.key_bits = h.key_bits;
        r.sort_passes = h.n_passes;
        r.path_flags = (c->lds_rank ? 1u : 0u) | (c->last_v2 ? 2u : 0u) | (c->last_predicted ? 4u : 0u) | (redone ? 8u : 0u) |
                       ((c->last_v2 && c->last_packed) ? 16u : 0u) | ((c->last_v2 && c->last_k3) ? 32u : 0u) |
                       ((c->last_v2 && c->last_quant) ? 64u : 0u);
        if (c->last_predicted && h.status == CM_OK) {
            // The device sorted by cells of the predicted box (same order); the grid PCL itself would
            // report comes from the cloud's exact bounds, which the frame also produced (A.4 steps 2, 4).
            unsigned long long cells = 1;
            for (int a = 0; a < 3; ++a) {
                const float lo = h.min_p[a] * c->frame.inv_leaf[a], hi = h.max_p[a] * c->frame.inv_leaf[a];
                r.min_b[a] = static_cast<int32_t>(std::floor(lo));
                r.max_b[a] = static_cast<int32_t>(std::floor(hi));
                r.div_b[a] = r.max_b[a] - r.min_b[a] + 1;
                cells *= static_cast<unsigned long long>(r.div_b[a]);
            }
            uint32_t bits = 1;
            while (bits < 32 && (cells - 1) >> bits) ++bits;
            r.key_bits = bits;
        }
        if (h.status == CM_OK && c->last_mode == 0 && !c->frame_mask &&
            (c->last_predicted || (!c->last_v2 && c->last_grid_mode == 0))) {
            float leaf[3];
            for (int a = 0; a < 3; ++a) leaf[a] = 1.0f / c->frame.inv_leaf[a];
            update_predicted_box(c, h.min_p, h.max_p, leaf);
        }
        if (h.status == CM_OK) c->last_n_merged = h.n_valid;
        if (h.status == CM_OK && c->last_v2 && c->wrote_spl && h.n_valid && !h.spl_incomplete) {
            // the finish left the quantiles of this frame's sorted records: the next frame's splitters (cm_kernels_v4.hip)
            c->spl_cur ^= 1;
            c->spl_valid = true;
            c->spl_n = h.n_valid;
            std::memcpy(c->spl_min_b, c->frame.box_min_b, sizeof c->spl_min_b);
            std::memcpy(c->spl_div_b, c->frame.box_div_b, sizeof c->spl_div_b);
            std::memcpy(c->spl_inv_leaf, c->frame.inv_leaf, sizeof c->spl_inv_leaf);
            if (c->last_quant && h.quant_big) c->quant_big_arm = 16;     // (still needed: stays armed)
            if (c->last_quant && !redone) {
                c->quant_hist = (c->quant_hist << 1) & 0xFFu;
                if (++c->quant_good >= 16) c->quant_rest = 8;
            }
        }
        if (h.status == CM_OK) {
            r.n_merged = h.n_valid;
            r.n_out = h.n_out;
        } else if (h.status == CM_GRID_OVERFLOW) {
            // PCL: "output = *input_" — hand back the merged cloud, unvoxelised (A.4 step 3).
            cmk_merged(c->stream, c->d_frame, c->seg_counts, c->merged_total, c->out, c->frame.n_tiles, c->frame_mask);
            uint32_t total = 0;
            HIP_TRY(c, hipMemcpyAsync(&total, c->merged_total, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            r.n_merged = total;
            r.n_out = total;
            c->out_is_merged = true;
        }
        if (c->flags & CM_FLAG_PROFILE) {
            cm_stage_times& t = c->stage_times;
            std::memset(&t, 0, sizeof t);
            const size_t n = c->prof_used ? c->prof_used - 1 : 0;
            for (size_t i = 0; i < n && i < CM_MAX_STAGES; ++i) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]);
                std::snprintf(t.name[i], sizeof t.name[i], "%s", c->prof_names[i].c_str());
                t.ms[i] = ms;
                t.n_stages = static_cast<uint32_t>(i + 1);
            }
            if (c->prof_used >= 2)
                (void)hipEventElapsedTime(&r.device_ms, c->prof_ev[0], c->prof_ev[c->prof_used - 1]);
        }
    }
    c->pending = false;
    c->result = r;
    c->have_result = true;
    if (res) *res = r;
    return r.status;
}

}  // namespace

extern "C" {

int cm_version(void) { return CM_VERSION; }
const char* cm_status_string(int status) { return k_status_names(status); }
const char* cm_last_error(cm_ctx* ctx) {
    if (!ctx) return "null context";
    static thread_local std::string copy;              // valid until the calling thread asks again
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    copy = ctx->err;
    return copy.c_str();
}

int cm_create(cm_ctx** out, int device, const cm_limits* lim) {
    if (!out || !lim) return CM_BAD_ARG;
    *out = nullptr;
    if (lim->max_sensors < 1 || lim->max_sensors > CM_MAX_SENSORS) return CM_BAD_ARG;
    if (lim->max_points_total < 1 || lim->max_points_total >= (1ull << 30)) return CM_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CM_NO_DEVICE;
    if (device < 0 || device >= ndev) return CM_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CM_HIP_ERROR;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CM_HIP_ERROR;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CM_NO_DEVICE;   // kernels are gfx950-only

    cm_ctx* c = new (std::nothrow) cm_ctx();
    if (!c) return CM_INTERNAL;
    c->device = device;
    c->flags = lim->flags;
    c->max_sensors = lim->max_sensors;
    c->max_points = lim->max_points_total;
    const uint64_t padded = static_cast<uint64_t>(round_up(static_cast<uint32_t>(lim->max_points_total), CM_TILE)) +
                            static_cast<uint64_t>(lim->max_sensors) * CM_TILE;
    c->cap_padded = static_cast<uint32_t>(padded);
    c->cap_tiles = c->cap_padded / CM_TILE;
    c->cap_seg_tiles = c->cap_padded / CM_SEG_TILE;

    auto A = [&](void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; };
    const size_t n4 = static_cast<size_t>(c->cap_padded) * 4;
    bool ok = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess;
    c->stream = c->own_stream;
    ok = ok && hipEventCreate(&c->ev_done) == hipSuccess;
    ok = ok && A(reinterpret_cast<void**>(&c->keys_a), n4) && A(reinterpret_cast<void**>(&c->keys_b), n4);
    ok = ok && A(reinterpret_cast<void**>(&c->vals_a), n4) && A(reinterpret_cast<void**>(&c->vals_b), n4);
    ok = ok && A(reinterpret_cast<void**>(&c->hist), static_cast<size_t>(c->cap_tiles) * CM_RADIX * 4);
    c->cap_groups = (c->cap_tiles + CM_GROUP - 1) / CM_GROUP;
    {
        const size_t gbytes = static_cast<size_t>(c->cap_groups) * CM_RADIX * 4 * 5;
        ok = ok && A(reinterpret_cast<void**>(&c->grp), gbytes);
        ok = ok && hipMemset(c->grp, 0, gbytes) == hipSuccess;
    }
    ok = ok && A(reinterpret_cast<void**>(&c->seg_tile_counts), static_cast<size_t>(c->cap_seg_tiles + 1) * 4);
    ok = ok && A(reinterpret_cast<void**>(&c->seg_groups), static_cast<size_t>(c->cap_seg_tiles / CM_SEG_GROUP + 2) * 32 * 4);
    ok = ok && A(reinterpret_cast<void**>(&c->partials), CM_MINMAX_BLOCKS * 8 * sizeof(float));
    ok = ok && A(reinterpret_cast<void**>(&c->totals), CM_RADIX * 4);
    ok = ok && A(reinterpret_cast<void**>(&c->seg_counts), static_cast<size_t>(c->cap_seg_tiles + c->cap_tiles) * 4);
    ok = ok && A(reinterpret_cast<void**>(&c->merged_total), 256);
    ok = ok && A(&c->out, static_cast<size_t>(c->cap_padded) * 16);
    if (c->flags & CM_FLAG_OCCUPANCY)
        ok = ok && A(reinterpret_cast<void**>(&c->out_key), n4) && A(reinterpret_cast<void**>(&c->out_cnt), n4);
    ok = ok && A(reinterpret_cast<void**>(&c->d_frame), sizeof(CmFrameDev));
    ok = ok && A(reinterpret_cast<void**>(&c->d_tiles), static_cast<size_t>(c->cap_tiles) * sizeof(CmTileDev));
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&c->h_tile_kept), static_cast<size_t>(c->cap_tiles) * 4, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_tile_kept), c->h_tile_kept, 0) == hipSuccess;
    ok = ok && A(reinterpret_cast<void**>(&c->d_state[0]), sizeof(CmFrameState));
    ok = ok && A(reinterpret_cast<void**>(&c->d_state[1]), sizeof(CmFrameState));
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&c->h_state), sizeof(CmFrameState), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_state_dev), c->h_state, 0) == hipSuccess;
    for (uint32_t s = 0; ok && s < c->max_sensors; ++s)
        ok = hipStreamCreateWithFlags(&c->slots[s].copy_stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&c->slots[s].ev_copy, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMemset(c->d_state[0], 0, sizeof(CmFrameState)) == hipSuccess;
    ok = ok && hipMemset(c->d_state[1], 0, sizeof(CmFrameState)) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (const char* pm = getenv("CM_PATH")) c->path_mode = std::strcmp(pm, "classic") == 0 ? 1 : 0;
    if (ok) {
        // Probe the device once: lane-ordered returning LDS adds allow the cheap stable ranking.
        // CM_LDS_RANK=0 forces the ballot-match ranking, CM_LDS_RANK=1 skips the probe.
        const char* env = getenv("CM_LDS_RANK");
        if (env && env[0] == '0') c->lds_rank = false;
        else if (env && env[0] == '1') c->lds_rank = true;
        else {
            uint32_t violations = 1;
            ok = hipMemset(c->merged_total, 0, 4) == hipSuccess;
            if (ok) {
                cmk_probe_lds_order(c->stream, c->merged_total, 64);
                ok = hipMemcpyAsync(&violations, c->merged_total, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                     hipStreamSynchronize(c->stream) == hipSuccess;
            }
            c->lds_rank = ok && violations == 0;
        }
    }
    if (const char* fm = getenv("CM_FINISH")) c->finish_mode = std::strcmp(fm, "v2") == 0 ? 2 : 0;
#ifdef CM_TEST_HOOKS                     // (the test build only: python -m cloud_merger_amd.build --test-hooks; never the shipped library)
    if (const char* dm = getenv("CM_DEBUG_MISRANK")) c->debug_misrank = dm[0] == '1' ? 1 : 0;
#endif
    if (const char* qm = getenv("CM_QUANT")) c->quant_mode = qm[0] == '0' ? 1 : 0;     // CM_QUANT=0: fixed-grid passes only
    if (const char* qs = getenv("CM_QUANT_SUB")) c->quant_sub = qs[0] != '0';
    if (!ok) {
        free_all(c);
        delete c;
        return CM_HIP_ERROR;
    }
    std::memset(&c->result, 0, sizeof c->result);
    std::memset(&c->stage_times, 0, sizeof c->stage_times);
    *out = c;
    return CM_OK;
}

int cm_destroy(cm_ctx* c) {
    if (!c) return CM_BAD_ARG;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_all(c);
    delete c;
    return CM_OK;
}

int cm_set_stream(cm_ctx* c, void* hip_stream) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (c->pending) return fail(c, CM_BAD_ARG, "cannot change stream with a frame in flight");
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return CM_OK;
}

int cm_set_sensor_transform(cm_ctx* c, uint32_t sensor, const double q[4], const double t[3]) {
    if (!c || !q || !t) return CM_BAD_ARG;
    if (sensor >= c->max_sensors) return fail(c, CM_BAD_ARG, "sensor index out of range");
    float m[12];
    quat_to_rows(q, t, m);
    std::lock_guard<std::mutex> lk(c->slots[sensor].mu);
    std::memcpy(c->slots[sensor].m, m, sizeof m);
    return CM_OK;
}

int cm_set_sensor_matrix(cm_ctx* c, uint32_t sensor, const float m[12]) {
    if (!c || !m) return CM_BAD_ARG;
    if (sensor >= c->max_sensors) return fail(c, CM_BAD_ARG, "sensor index out of range");
    std::lock_guard<std::mutex> lk(c->slots[sensor].mu);
    std::memcpy(c->slots[sensor].m, m, 12 * sizeof(float));
    return CM_OK;
}

int cm_get_sensor_matrix(cm_ctx* c, uint32_t sensor, float m[12]) {
    if (!c || !m) return CM_BAD_ARG;
    if (sensor >= c->max_sensors) return fail(c, CM_BAD_ARG, "sensor index out of range");
    std::lock_guard<std::mutex> lk(c->slots[sensor].mu);
    std::memcpy(m, c->slots[sensor].m, 12 * sizeof(float));
    return CM_OK;
}

int cm_submit_cloud(cm_ctx* c, uint32_t sensor, const void* host_data, uint32_t n, uint32_t point_step,
                    uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_i) {
    return set_slot_cloud(c, sensor, host_data, false, n, point_step, off_x, off_y, off_z, off_i);
}

int cm_submit_cloud_async(cm_ctx* c, uint32_t sensor, const void* host_data, uint32_t n, uint32_t point_step,
                          uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_i) {
    return set_slot_cloud(c, sensor, host_data, false, n, point_step, off_x, off_y, off_z, off_i, false);
}

int cm_submit_cloud_device(cm_ctx* c, uint32_t sensor, const void* dev_data, uint32_t n, uint32_t point_step,
                           uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_i) {
    return set_slot_cloud(c, sensor, dev_data, true, n, point_step, off_x, off_y, off_z, off_i);
}

int cm_clear_sensor(cm_ctx* c, uint32_t sensor) {
    if (!c) return CM_BAD_ARG;
    if (sensor >= c->max_sensors) return fail(c, CM_BAD_ARG, "sensor index out of range");
    std::lock_guard<std::mutex> lk(c->slots[sensor].mu);
    c->slots[sensor].has_data = false;
    c->slots[sensor].fresh = false;
    c->slots[sensor].staged = SlotCloud();       // (the buffers stay: a frame in flight may still read the active one)
    return CM_OK;
}

int cm_merge_voxelize_async(cm_ctx* c, const cm_params* p) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    return enqueue(c, p);
}

int cm_wait(cm_ctx* c, cm_result* res) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    return wait_frame(c, res);
}

int cm_merge_voxelize(cm_ctx* c, const cm_params* p, cm_result* res) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    const int e = enqueue(c, p);
    if (e != CM_OK) {
        if (res) { std::memset(res, 0, sizeof *res); res->status = e; }
        return e;
    }
    return wait_frame(c, res);
}

int cm_result_device(cm_ctx* c, const void** dev_ptr, uint64_t* n_points) {
    if (!c || !dev_ptr || !n_points) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    *dev_ptr = c->out;
    *n_points = c->result.n_out;
    return CM_OK;
}

int cm_result_copy(cm_ctx* c, void* host_dst, uint64_t capacity_points, uint32_t step_out) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    if (step_out == 0) step_out = 16;
    if (step_out != 16 && step_out != 32) return fail(c, CM_BAD_ARG, "point_step_out must be 16 or 32");
    const uint64_t n = c->result.n_out;
    if (n > capacity_points) return fail(c, CM_CAPACITY, "destination too small");
    if (n == 0) return CM_OK;
    if (!host_dst) return CM_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    c->bytes_d2h += n * 16;
    if (step_out == 16) {
        HIP_TRY(c, hipMemcpyAsync(host_dst, c->out, n * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return CM_OK;
    }
    // pcl::PointXYZI images (A.0) are laid out on the device and cross PCIe as they go on the wire
    if (!c->out32) HIP_TRY(c, hipMalloc(&c->out32, static_cast<size_t>(c->cap_padded) * 32));
    cmk_to_pcl32(c->stream, c->out, c->out32, static_cast<uint32_t>(n));
    HIP_TRY(c, hipMemcpyAsync(host_dst, c->out32, n * 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->bytes_d2h += n * 16;
    return CM_OK;
}

int cm_result_copy_async(cm_ctx* c, void* host_dst, uint64_t capacity_points) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    const uint64_t n = c->result.n_out;
    if (n > capacity_points) return fail(c, CM_CAPACITY, "destination too small");
    if (n == 0) return CM_OK;
    if (!host_dst) return CM_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(host_dst, c->out, n * 16, hipMemcpyDeviceToHost, c->stream));
    c->bytes_d2h += n * 16;
    return CM_OK;
}

int cm_result_publish_async(cm_ctx* c, void* host_dst, uint64_t capacity_points, uint32_t step_out) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    if (c->pending) return fail(c, CM_BAD_ARG, "a frame is in flight: publish its predecessor before enqueueing it");
    if (step_out == 0) step_out = 16;
    if (step_out != 16 && step_out != 32) return fail(c, CM_BAD_ARG, "point_step_out must be 16 or 32");
    const uint64_t n = c->result.n_out;
    if (n > capacity_points) return fail(c, CM_CAPACITY, "destination too small");
    if (n == 0) return CM_OK;
    if (!host_dst) return CM_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->pub_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->pub_stream, hipStreamNonBlocking));
        for (auto& e : c->ev_pub) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(c, hipMalloc(&c->out_other, static_cast<size_t>(c->cap_padded) * 16));
    }
    // (the frame has been waited for — cm_wait returned its counts — so its result is complete; an overflow fallback's
    // merged cloud was written on c->stream and synchronised as well)
    if (step_out == 16) {
        HIP_TRY(c, hipMemcpyAsync(host_dst, c->out, n * 16, hipMemcpyDeviceToHost, c->pub_stream));
    } else {
        if (!c->out32) HIP_TRY(c, hipMalloc(&c->out32, static_cast<size_t>(c->cap_padded) * 32));
        if (!c->out32_other) HIP_TRY(c, hipMalloc(&c->out32_other, static_cast<size_t>(c->cap_padded) * 32));
        cmk_to_pcl32(c->pub_stream, c->out, c->out32, static_cast<uint32_t>(n));
        HIP_TRY(c, hipMemcpyAsync(host_dst, c->out32, n * 32, hipMemcpyDeviceToHost, c->pub_stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev_pub[0], c->pub_stream));
    c->pub_pending[0] = true;
    c->bytes_d2h += n * step_out;
    return CM_OK;
}

int cm_publish_wait(cm_ctx* c) {
    if (!c) return CM_BAD_ARG;
    if (!c->pub_stream) return CM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->pub_stream));
    // (pub_pending stays as it is: it steers which buffers the next frame writes, and a finished event costs nothing to wait for)
    return CM_OK;
}

int cm_sync(cm_ctx* c) {
    if (!c) return CM_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->pub_stream) HIP_TRY(c, hipStreamSynchronize(c->pub_stream));
    return CM_OK;
}

int cm_get_frame_stats(cm_ctx* c, cm_frame_stats* out) {
    if (!c || !out) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    std::memset(out, 0, sizeof *out);
    out->n_sensors = c->stats_n_sensors;
    const uint32_t* kept = c->h_tile_kept;                 // (written by the frame's first scatter; the frame has been waited for)
    const bool have_kept = c->result.status == CM_OK && c->frame.n_tiles && c->last_mode != 2;
    for (uint32_t k = 0; k < c->stats_n_sensors; ++k) {
        out->sensor[k] = c->stats_sensor[k];
        out->n_in[k] = c->stats_n[k];
        out->fresh[k] = c->stats_fresh[k];
        out->bytes_h2d[k] = c->stats_bytes[k];
        out->generation[k] = c->stats_gen[k];
        out->bytes_h2d_total += c->stats_bytes[k];
        if (have_kept) {
            const uint32_t t0 = c->frame.s[k].base / CM_TILE, t1 = t0 + (c->frame.s[k].n + CM_TILE - 1) / CM_TILE;
            uint64_t sum = 0;
            for (uint32_t t = t0; t < t1 && t < c->frame.n_tiles; ++t) sum += kept[t];
            out->n_kept[k] = static_cast<uint32_t>(sum);
        }
    }
    out->bytes_d2h_total = c->bytes_d2h;
    out->bytes_algorithmic = 16ull * c->result.n_in + 16ull * c->result.n_out;
    return CM_OK;
}

int cm_result_copy_cells(cm_ctx* c, int32_t* ijk, uint32_t* counts, uint64_t capacity) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    if (!(c->flags & CM_FLAG_OCCUPANCY)) return fail(c, CM_BAD_ARG, "context created without CM_FLAG_OCCUPANCY");
    if (c->result.status != CM_OK) return fail(c, CM_BAD_ARG, "last frame has no voxel grid");
    const uint64_t n = c->result.n_out;
    if (n > capacity) return fail(c, CM_CAPACITY, "destination too small");
    if (n == 0) return CM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (counts) HIP_TRY(c, hipMemcpyAsync(counts, c->out_cnt, n * 4, hipMemcpyDeviceToHost, c->stream));
    std::vector<uint32_t> keys;
    if (ijk) {
        keys.resize(n);
        HIP_TRY(c, hipMemcpyAsync(keys.data(), c->out_key, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (ijk) {
        // out_key holds linear indices in the grid the frame was sorted in (the crop box, the predicted
        // box or the cloud's own bounds); the caller gets absolute cells floor(p / leaf).
        const uint32_t d0 = static_cast<uint32_t>(c->cell_div_b[0]), d1 = static_cast<uint32_t>(c->cell_div_b[1]);
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t k = keys[i];
            ijk[3 * i + 0] = static_cast<int32_t>(k % d0) + c->cell_min_b[0];
            ijk[3 * i + 1] = static_cast<int32_t>((k / d0) % d1) + c->cell_min_b[1];
            ijk[3 * i + 2] = static_cast<int32_t>(k / (d0 * d1)) + c->cell_min_b[2];
        }
    }
    return CM_OK;
}

int cm_merged_copy(cm_ctx* c, void* host_dst, uint64_t capacity, uint64_t* n_points) {
    if (!c || !n_points) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    *n_points = 0;
    if (c->frame.n_padded == 0) return CM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->merged) HIP_TRY(c, hipMalloc(&c->merged, static_cast<size_t>(c->cap_padded) * 16));
    // seg_counts holds this frame's output offsets in its first cap_seg_tiles words; use the tail.
    uint32_t* counts = c->seg_counts + c->cap_seg_tiles;
    cmk_merged(c->stream, c->d_frame, counts, c->merged_total, c->merged, c->frame.n_tiles, c->frame_mask);
    uint32_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->merged_total, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_points = total;
    if (total > capacity) return fail(c, CM_CAPACITY, "destination too small");
    if (total && host_dst) {
        HIP_TRY(c, hipMemcpyAsync(host_dst, c->merged, static_cast<size_t>(total) * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->bytes_d2h += static_cast<uint64_t>(total) * 16;
    }
    return CM_OK;
}

int cm_set_ground_removal(cm_ctx* c, const cm_ground_params* g) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (c->pending) return fail(c, CM_BAD_ARG, "cannot change ground removal with a frame in flight");
    if (!g) { c->ground_on = false; return CM_OK; }
    if (g->max_iterations < 1 || g->max_iterations > 100000) return fail(c, CM_BAD_ARG, "max_iterations must be in 1..100000");
    if (!(g->distance_threshold > 0.0f) || !std::isfinite(g->distance_threshold)) return fail(c, CM_BAD_ARG, "distance_threshold must be > 0");
    if (!(g->probability > 0.0f && g->probability < 1.0f)) return fail(c, CM_BAD_ARG, "probability must be in (0, 1)");
    if (g->outlier_radius < 0.0f || !std::isfinite(g->outlier_radius)) return fail(c, CM_BAD_ARG, "outlier_radius must be >= 0");
    CmGroundDev d;
    std::memset(&d, 0, sizeof d);
    for (uint32_t s = 0; s < CM_MAX_SENSORS; ++s) {
        if (g->n_zones[s] > CM_MAX_ZONES) return fail(c, CM_BAD_ARG, "more than CM_MAX_ZONES zones for a sensor");
        d.n_zones[s] = g->n_zones[s];
        for (uint32_t z = 0; z < g->n_zones[s]; ++z) {
            const cm_zone& zn = g->zones[s][z];
            if (!std::isfinite(zn.x_min) || !(zn.x_length >= 0.0f) || !std::isfinite(zn.x_length) || !std::isfinite(zn.z_max_ground))
                return fail(c, CM_BAD_ARG, "zone limits must be finite, x_length >= 0");
            d.x0[s][z] = zn.x_min;
            d.x1[s][z] = zn.x_min + zn.x_length;                          // setFilterLimits(deviation, deviation + length), :57
            d.zmax[s][z] = zn.z_max_ground;
            d.zlo[s][z] = static_cast<float>(static_cast<double>(zn.z_max_ground) + 0.01);   // z_max_ground + 0.01, :91
        }
    }
    d.z_keep_max = g->z_keep_max;
    d.threshold = g->distance_threshold;
    d.probability = g->probability;
    d.max_iterations = g->max_iterations;
    d.optimize = g->optimize_coefficients ? 1u : 0u;
    d.seed = g->seed;
    c->ground = d;
    c->ground_outlier_radius = g->outlier_radius;
    c->ground_outlier_min_nb = g->outlier_min_neighbors;
    c->ground_uploaded = false;
    c->ground_on = true;
    return CM_OK;
}

int cm_ground_copy(cm_ctx* c, void* host_dst, uint64_t capacity, uint64_t* n_points) {
    if (!c || !n_points) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result) return fail(c, CM_BAD_ARG, "no result");
    *n_points = 0;
    if (!c->frame_had_ground) return fail(c, CM_BAD_ARG, "the last frame ran without ground removal");
    if (c->frame.n_padded == 0) return CM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->merged) HIP_TRY(c, hipMalloc(&c->merged, static_cast<size_t>(c->cap_padded) * 16));
    uint32_t* counts = c->seg_counts + c->cap_seg_tiles;
    cmk_merged(c->stream, c->d_frame, counts, c->merged_total, c->merged, c->frame.n_tiles, c->gmask);
    uint32_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->merged_total, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_points = total;
    if (total > capacity) return fail(c, CM_CAPACITY, "destination too small");
    if (total && host_dst) {
        HIP_TRY(c, hipMemcpyAsync(host_dst, c->merged, static_cast<size_t>(total) * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return CM_OK;
}

int cm_ground_planes(cm_ctx* c, cm_ground_plane* planes, uint32_t capacity) {
    if (!c || !planes) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result || !c->frame_had_ground) return fail(c, CM_BAD_ARG, "the last frame ran without ground removal");
    static_assert(sizeof(cm_ground_plane) == sizeof(CmGroundPlaneDev), "plane record layout");
    const uint32_t n = std::min<uint32_t>(capacity, CM_DEV_MAX_SENSORS * CM_DEV_MAX_ZONES);
    if (c->frame.n_padded == 0) {                      // an empty frame: no band points anywhere
        std::memset(planes, 0, static_cast<size_t>(n) * sizeof(cm_ground_plane));
        return CM_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(planes, c->d_planes, static_cast<size_t>(n) * sizeof(cm_ground_plane), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CM_OK;
}

int cm_local_bounds(cm_ctx* c, const cm_params* p, float min_xyz[3], float max_xyz[3], uint64_t* n_valid) {
    if (!c || !p || !min_xyz || !max_xyz) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->pending) return fail(c, CM_BAD_ARG, "previous frame not waited for (cm_wait)");
    for (int a = 0; a < 3; ++a)
        if (!(p->leaf[a] > 0.0f)) return fail(c, CM_BAD_ARG, "leaf must be > 0");
    std::vector<std::unique_lock<std::mutex>> locks;
    const int bf = build_frame(c, p, false, locks);          // a peek: the clouds stay fresh
    if (bf != CM_OK) return bf;
    const float inf = std::numeric_limits<float>::infinity();
    for (int a = 0; a < 3; ++a) { min_xyz[a] = inf; max_xyz[a] = -inf; }
    if (n_valid) *n_valid = 0;
    const CmFrameDev& f = c->frame;
    if (f.n_padded == 0) return CM_OK;
    cmk_setup(c->stream, f, c->d_frame, c->d_tiles);
    c->frame_uploaded = f;
    c->frame_uploaded_valid = true;
    const uint32_t n_partials = f.n_tiles < CM_MINMAX_BLOCKS ? f.n_tiles : CM_MINMAX_BLOCKS;
    cmk_minmax(c->stream, c->d_frame, c->partials, n_partials, nullptr);
    std::vector<float> rec(static_cast<size_t>(n_partials) * 8);
    HIP_TRY(c, hipMemcpyAsync(rec.data(), c->partials, rec.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t cnt = 0;
    for (uint32_t r = 0; r < n_partials; ++r) {
        uint32_t k;
        std::memcpy(&k, &rec[r * 8 + 6], 4);
        if (!k) continue;
        cnt += k;
        for (int a = 0; a < 3; ++a) {
            min_xyz[a] = std::min(min_xyz[a], rec[r * 8 + a]);
            max_xyz[a] = std::max(max_xyz[a], rec[r * 8 + 3 + a]);
        }
    }
    if (n_valid) *n_valid = cnt;
    return CM_OK;
}

int cm_merge_partial(cm_ctx* c, const cm_params* p, const float* global_min_max, cm_result* res) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    const int e = enqueue(c, p, 1, global_min_max);
    if (e != CM_OK) {
        if (res) { std::memset(res, 0, sizeof *res); res->status = e; }
        return e;
    }
    return wait_frame(c, res);
}

int cm_partial_device(cm_ctx* c, const void** dev_entries, uint64_t* n_entries) {
    if (!c || !dev_entries || !n_entries) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result || c->last_mode != 1) return fail(c, CM_BAD_ARG, "no partial table (cm_merge_partial)");
    *dev_entries = c->partial;
    *n_entries = c->result.status == CM_OK ? c->result.n_out : 0;
    return CM_OK;
}

int cm_partial_copy(cm_ctx* c, cm_partial_entry* host_dst, uint64_t capacity) {
    if (!c) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!c->have_result || c->last_mode != 1) return fail(c, CM_BAD_ARG, "no partial table (cm_merge_partial)");
    const uint64_t n = c->result.status == CM_OK ? c->result.n_out : 0;
    if (n > capacity) return fail(c, CM_CAPACITY, "destination too small");
    if (n == 0) return CM_OK;
    if (!host_dst) return CM_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(host_dst, c->partial, n * 32, hipMemcpyDefault, c->stream));   // host or device destination
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CM_OK;
}

int cm_merge_tables(cm_ctx* c, const void* const* dev_tables, const uint64_t* n_entries, uint32_t n_tables,
                    const cm_params* p, cm_result* res) {
    if (!c || !p || !dev_tables || !n_entries || n_tables < 1 || n_tables > CM_MAX_SENSORS) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->pending) return fail(c, CM_BAD_ARG, "previous frame not waited for (cm_wait)");
    // The tables take the place of the sensor clouds: table t owns a tile-aligned range of the
    // padded index space, so the point index a key carries maps back to (table, entry).
    CmFrameDev& f = c->frame;
    std::memset(&f, 0, sizeof f);
    uint32_t base = 0;
    uint64_t total = 0;
    for (uint32_t t = 0; t < n_tables; ++t) {
        if (n_entries[t] && (!dev_tables[t] || (reinterpret_cast<uintptr_t>(dev_tables[t]) & 15u)))
            return fail(c, CM_BAD_ARG, "table pointers must be 16-byte aligned device memory");
        CmSensorDev& d = f.s[t];
        d.data = static_cast<const unsigned char*>(dev_tables[t]);
        d.n = static_cast<uint32_t>(n_entries[t]);
        d.base = base;
        d.slot = t;
        d.point_step = 32;
        const uint64_t nb = static_cast<uint64_t>(base) + round_up(d.n, CM_TILE);
        if (nb > c->cap_padded) return fail(c, CM_CAPACITY, "tables exceed cm_limits.max_points_total");
        base = static_cast<uint32_t>(nb);
        total += n_entries[t];
    }
    f.n_sensors = n_tables;
    f.n_padded = base;
    f.n_tiles = base / CM_TILE;
    f.min_pts = p->min_points_per_voxel;
    f.downsample_all = 1;
    cm_result r;
    std::memset(&r, 0, sizeof r);
    if (c->have_result && c->last_mode == 1) {       // keep the shared grid of this rank's partial table
        for (int a = 0; a < 3; ++a) {
            r.min_b[a] = c->result.min_b[a]; r.max_b[a] = c->result.max_b[a]; r.div_b[a] = c->result.div_b[a];
            r.min_p[a] = c->result.min_p[a]; r.max_p[a] = c->result.max_p[a];
        }
        r.bounds_from_crop = c->result.bounds_from_crop;
    }
    r.n_sensors = n_tables;
    r.n_in = total;
    c->have_result = false;
    c->last_mode = 2;
    c->prof_used = 0;
    if (f.n_padded == 0) {
        r.status = CM_EMPTY_INPUT;
        c->result = r; c->have_result = true;
        if (res) *res = r;
        return r.status;
    }
    hipStream_t st = c->stream;
    cmk_setup(st, f, c->d_frame, c->d_tiles);
    c->frame_uploaded = f;
    c->frame_uploaded_valid = true;
    CmFrameState* state = c->d_state[c->cur];
    CmFrameState* state_next = c->d_state[c->cur ^ 1];
    const uint32_t nt = f.n_tiles, nseg = f.n_padded / CM_SEG_TILE;
    const uint32_t n_groups = (nt + CM_GROUP - 1) / CM_GROUP, gw = n_groups * CM_RADIX;
    const size_t gstride = static_cast<size_t>(c->cap_groups) * CM_RADIX;
    uint32_t* grp0 = c->grp + gstride * (c->frame_seq & 1u);
    uint32_t* grp0_next = c->grp + gstride * ((c->frame_seq & 1u) ^ 1u);
    ++c->frame_seq;
    const bool big = n_groups > CM_DIRECT_GROUPS;
    if (!c->table_entries) HIP_TRY(c, hipMalloc(&c->table_entries, static_cast<size_t>(c->cap_padded) * 32));
    cmk_table_keys(st, c->d_frame, state, c->keys_a, c->hist, grp0, grp0_next, c->grp + 2 * gstride, gw,
                   static_cast<uint32_t>(gstride), c->seg_groups, (nseg + CM_SEG_GROUP - 1) / CM_SEG_GROUP + 1, 32u, nt);
    for (uint32_t pass = 0; pass < CM_MAX_PASSES; ++pass) {
        const bool even = (pass & 1u) == 0;
        const uint32_t* kin = even ? c->keys_a : c->keys_b;
        const uint32_t* vin = even ? c->vals_a : c->vals_b;
        uint32_t* kout = even ? c->keys_b : c->keys_a;
        uint32_t* vout = even ? c->vals_b : c->vals_a;
        uint32_t* grp = pass == 0 ? grp0 : c->grp + 2 * gstride + static_cast<size_t>(pass - 1) * gw;
        if (pass > 0) cmk_hist(st, state, kin, c->hist, grp, pass, nt);
        if (big) cmk_gscan(st, state, grp, c->totals, pass, n_groups);
        cmk_scatter(st, state, kin, vin, kout, vout, c->hist, grp, big ? c->totals : nullptr, pass, nt, n_groups,
                    f.n_padded, c->lds_rank);
    }
    uint32_t* seg_groups = nseg > CM_SEG_DIRECT_TILES ? c->seg_groups : nullptr;
    cmk_seg_count(st, state, c->keys_a, c->keys_b, c->seg_tile_counts, seg_groups, 1u, nseg);
    cmk_seg_reduce(st, 2, c->d_frame, state, state_next, c->h_state_dev, c->keys_a, c->vals_a, c->keys_b, c->vals_b,
                   c->seg_tile_counts, seg_groups, c->table_entries, nullptr, nullptr, nseg);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(st));
    c->cur ^= 1;
    const CmFrameState& h = *c->h_state;
    if (h.err) {
        if (h.err == 2 && c->lds_rank) c->lds_rank = false;
        return fail(c, CM_INTERNAL, "device reported an internal error while merging tables");
    }
    const uint32_t n_merged = h.status == CM_OK ? h.n_out : 0;       // distinct voxels over all tables
    uint32_t n_out = 0;
    if (n_merged) {
        cmk_table_finish(st, c->table_entries, n_merged, p->min_points_per_voxel, c->seg_counts, c->merged_total,
                         c->out, c->out_key, c->out_cnt);
        HIP_TRY(c, hipMemcpyAsync(&n_out, c->merged_total, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
    }
    r.status = n_merged ? CM_OK : CM_EMPTY_INPUT;
    r.n_merged = n_merged;
    r.n_out = n_out;
    r.key_bits = 32; r.sort_passes = CM_MAX_PASSES;
    r.path_flags = c->lds_rank ? 1u : 0u;
    c->result = r;
    c->have_result = true;
    c->out_is_merged = false;
    if (res) *res = r;
    return r.status;
}

int cm_get_stage_times(cm_ctx* c, cm_stage_times* out) {
    if (!c || !out) return CM_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->merge_mu);
    if (!(c->flags & CM_FLAG_PROFILE)) return fail(c, CM_BAD_ARG, "context created without CM_FLAG_PROFILE");
    *out = c->stage_times;
    return CM_OK;
}

int cm_host_alloc(void** ptr, size_t bytes) {
    if (!ptr) return CM_BAD_ARG;
    return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? CM_OK : CM_HIP_ERROR;
}

int cm_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) return CM_BAD_ARG;
    return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? CM_OK : CM_INTERNAL;
}
int cm_host_unregister(void* ptr) {
    if (!ptr) return CM_BAD_ARG;
    return hipHostUnregister(ptr) == hipSuccess ? CM_OK : CM_INTERNAL;
}
int cm_host_free(void* ptr) {
    return hipHostFree(ptr) == hipSuccess ? CM_OK : CM_HIP_ERROR;
}

}  // extern "C"
