// merger_node.cpp — see merger_node.hpp.
#include "merger_node.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <sstream>
#include <thread>

namespace cloudmerge {

NodeConfig reference_config() {
    NodeConfig c;
    // Topics :520-525, TF frames :556-561, fuse order :137-142, gate :134-136.
    c.sensors = {
        {"front_right", "/velodyne/front_right/velodyne_points", "/velodyne_front_right", true},
        {"front_left", "/velodyne/front_left/velodyne_points", "/velodyne_front_left", true},
        {"rear_right", "/velodyne/rear_right/velodyne_points", "/velodyne_rear_right", true},
        {"rear_left", "/velodyne/rear_left/velodyne_points", "/velodyne_rear_left", true},
        {"top_middle", "/velodyne/top_middle/velodyne_points", "/velodyne_top_middle", false},
        {"front_middle", "/livoxfront/livox/lidar", "/livox_front", true},
    };
    cm_params& p = c.params;
    p.leaf[0] = p.leaf[1] = p.leaf[2] = 0.1f;       // voxel_size, Parameter.h:28
    p.min_points_per_voxel = 2;                     // points_per_voxel, Parameter.h:27
    p.downsample_all_data = 1;                      // :174
    p.crop_enable = 1;                              // getROI, :20-40
    const float roi_width = 10.0f, roi_length = 75.0f, roi_mid = 15.0f;   // Parameter.h:31-33
    p.crop_min[0] = -roi_mid;        p.crop_max[0] = roi_length - roi_mid;
    p.crop_min[1] = -roi_width / 2;  p.crop_max[1] = roi_width / 2;
    p.crop_min[2] = -0.5f;           p.crop_max[2] = 3.0f;                 // Parameter.h:34-35
    // Radius outlier removal is off here: the live node applies it per ground-removal zone
    // (:119, out of scope). fusion_config() below is the class-based node's chain, which runs it
    // on the fused cloud.
    return c;
}

NodeConfig live_node_config() {
    NodeConfig c = reference_config();
    c.ground_enable = true;
    cm_ground_params& g = c.ground;
    g.max_iterations = 1000;                       // Parameter.h:38
    g.distance_threshold = 0.3f;                   // :40
    g.probability = 0.99f;                         // :41
    g.optimize_coefficients = 1;                   // :95
    g.z_keep_max = 3.0f;                           // roi_z_max, :35
    g.outlier_radius = 0.15f;                      // removeGround's outlierRemoval(no_ground_cloud_ptr), :119 — radius_search, Parameter.h:23
    g.outlier_min_neighbors = 1;                   // min_neighbor, Parameter.h:24
    g.seed = 12345;
    const float roi_mid = 15.0f;
    // proceedFront (:228-269), in its processing order: front, mid2, mid, vehicle, rear (Parameter.h:45-55)
    const float vf_front = 30.0f, vf_mid = 15.0f, vf_mid2 = 11.0f, vf_veh = 8.0f, vf_rear = 11.0f;
    const cm_zone front[5] = {
        {-roi_mid + vf_rear + vf_veh + vf_mid + vf_mid2, vf_front, 2.5f},
        {-roi_mid + vf_rear + vf_veh + vf_mid, vf_mid2, 2.0f},
        {-roi_mid + vf_rear + vf_veh, vf_mid, 1.5f},
        {-roi_mid + vf_rear, vf_veh, 0.3f},
        {-roi_mid, vf_rear, 0.5f}};
    for (int s = 0; s < 4; ++s) {                  // all four corner Velodynes go through proceedFront (:326,:352,:378,:404)
        g.n_zones[s] = 5;
        for (int z = 0; z < 5; ++z) g.zones[s][z] = front[z];
    }
    // top middle (:436-444): one slab with a plane, the part behind it kept whole (Parameter.h:66-68)
    g.n_zones[4] = 2;
    g.zones[4][0] = {20.0f, 40.0f, 1.0f};
    g.zones[4][1] = {-roi_mid, roi_mid + 20.0f, -1.0f};
    // Livox (:475-497): front, mid2, mid, rear (Parameter.h:71-80)
    const float l_front = 26.0f, l_mid = 10.0f, l_mid2 = 10.0f, l_rear = 10.0f, l_dev = 4.0f;
    g.n_zones[5] = 4;
    g.zones[5][0] = {l_dev + l_rear + l_mid + l_mid2, l_front, 1.5f};
    g.zones[5][1] = {l_dev + l_rear + l_mid, l_mid2, 1.2f};
    g.zones[5][2] = {l_dev + l_rear, l_mid, 0.8f};
    g.zones[5][3] = {l_dev, l_rear, 0.5f};
    return c;
}

NodeConfig fusion_config() {
    // my_cloud_fusion/src/cloud_fusion_node.cpp:72-75: remove_outliers, then voxelgrid, on the
    // fused cloud; constants from my_cloud_fusion/src/Parameter.h:15-16,109-110.
    NodeConfig c = reference_config();
    c.voxel_topic = "/cloud_fusion_node/points_voxel";
    c.params.outlier_enable = 1;
    c.params.outlier_radius = 0.1f;
    c.params.outlier_min_neighbors = 1;
    return c;
}

bool load_config(const std::string& path, NodeConfig* cfg, std::string* err) {
    std::ifstream f(path);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    NodeConfig c = reference_config();
    bool own_sensors = false, ground_z_from_crop = false;
    int lineno = 0;
    for (std::string line; std::getline(f, line);) {
        ++lineno;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream is(line);
        std::string key;
        if (!(is >> key)) continue;
        bool ok = true;
        if (key == "sensor") {
            SensorSpec s;
            std::string req;
            ok = static_cast<bool>(is >> s.name >> s.topic >> s.frame >> req) && (req == "required" || req == "optional");
            s.required = req == "required";
            if (ok) { if (!own_sensors) { c.sensors.clear(); own_sensors = true; } c.sensors.push_back(s); }
        } else if (key == "base_frame") ok = static_cast<bool>(is >> c.base_frame);
        else if (key == "voxel_topic") ok = static_cast<bool>(is >> c.voxel_topic);
        else if (key == "rate_hz") ok = static_cast<bool>(is >> c.rate_hz) && c.rate_hz > 0;
        else if (key == "leaf") { float v; ok = static_cast<bool>(is >> v) && v > 0; if (ok) c.params.leaf[0] = c.params.leaf[1] = c.params.leaf[2] = v; }
        else if (key == "min_points_per_voxel") ok = static_cast<bool>(is >> c.params.min_points_per_voxel);
        else if (key == "crop") {
            ok = static_cast<bool>(is >> c.params.crop_min[0] >> c.params.crop_min[1] >> c.params.crop_min[2] >>
                                   c.params.crop_max[0] >> c.params.crop_max[1] >> c.params.crop_max[2]);
            c.params.crop_enable = 1;
        } else if (key == "no_crop") c.params.crop_enable = 0;
        else if (key == "outlier") { ok = static_cast<bool>(is >> c.params.outlier_radius >> c.params.outlier_min_neighbors); c.params.outlier_enable = 1; }
        else if (key == "stamp_from_inputs") { int v; ok = static_cast<bool>(is >> v); c.stamp_from_inputs = v != 0; }
        else if (key == "max_stamp_spread_ms") { double v; ok = static_cast<bool>(is >> v) && v >= 0; c.max_stamp_spread_ns = static_cast<uint64_t>(v * 1e6); }
        else if (key == "ground") {
            ok = static_cast<bool>(is >> c.ground.max_iterations >> c.ground.distance_threshold >> c.ground.probability);
            c.ground.optimize_coefficients = 1; c.ground.seed = 12345;
            ground_z_from_crop = true;                 // roi_z_max: taken from the crop box once the whole file is read
            c.ground_enable = true;
        } else if (key == "ground_outlier") {
            ok = static_cast<bool>(is >> c.ground.outlier_radius >> c.ground.outlier_min_neighbors) && c.ground.outlier_radius >= 0.0f;
        } else if (key == "zone") {
            std::string name; cm_zone z{};
            ok = static_cast<bool>(is >> name >> z.x_min >> z.x_length >> z.z_max_ground);
            size_t sidx = c.sensors.size();
            for (size_t q = 0; q < c.sensors.size(); ++q) if (c.sensors[q].name == name) sidx = q;
            ok = ok && sidx < c.sensors.size() && c.ground.n_zones[sidx] < CM_MAX_ZONES;
            if (ok) c.ground.zones[sidx][c.ground.n_zones[sidx]++] = z;
        }
        else if (key == "max_points_total") ok = static_cast<bool>(is >> c.max_points_total);
        else if (key == "device") ok = static_cast<bool>(is >> c.device);
        else ok = false;
        if (!ok) { if (err) *err = path + ":" + std::to_string(lineno) + ": bad line"; return false; }
    }
    if (c.sensors.empty() || c.sensors.size() > CM_MAX_SENSORS) { if (err) *err = "sensor count must be 1..16"; return false; }
    if (ground_z_from_crop) c.ground.z_keep_max = c.params.crop_max[2];     // (whichever of `crop` and `ground` came first)
    *cfg = c;
    return true;
}

CloudMergerNode::CloudMergerNode(const NodeConfig& cfg)
    : cfg_(cfg), have_tf_(cfg.sensors.size()), stamp_ns_(cfg.sensors.size()), submitted_(cfg.sensors.size()),
      consumed_(cfg.sensors.size()), slot_mu_(cfg.sensors.size()) {
    for (auto& f : have_tf_) f.store(false);
    for (auto& t : stamp_ns_) t.store(0);
    for (auto& f : submitted_) f.store(0);
    for (auto& f : consumed_) f.store(0);
    if (cfg_.sensors.empty() || cfg_.sensors.size() > CM_MAX_SENSORS) {
        error_ = "sensor count must be 1..CM_MAX_SENSORS";
        return;
    }
    uint32_t mask = 0;
    for (size_t s = 0; s < cfg_.sensors.size(); ++s)
        if (cfg_.sensors[s].required) mask |= 1u << s;
    cfg_.params.required_sensor_mask = mask;
    cm_limits lim{};
    lim.max_sensors = static_cast<uint32_t>(cfg_.sensors.size());
    lim.flags = cfg_.flags;
    lim.max_points_total = cfg_.max_points_total;
    const int st = cm_create(&ctx_, cfg_.device, &lim);
    if (st != CM_OK) {
        ctx_ = nullptr;
        error_ = std::string("cm_create: ") + cm_status_string(st);
    } else if (cfg_.ground_enable) {
        const int gs = cm_set_ground_removal(ctx_, &cfg_.ground);
        if (gs != CM_OK) {
            error_ = std::string("cm_set_ground_removal: ") + cm_last_error(ctx_);
            cm_destroy(ctx_);
            ctx_ = nullptr;
        }
    }
    clock_ = [] {
        return static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(
            std::chrono::system_clock::now().time_since_epoch()).count());
    };
}

CloudMergerNode::~CloudMergerNode() {
    if (ctx_) (void)cm_sync(ctx_);
    for (void* p : pipe_registered_) if (p) (void)cm_host_unregister(p);
    if (ctx_) cm_destroy(ctx_);
}

uint64_t CloudMergerNode::newest_stamp() const {
    uint64_t newest = 0;
    for (const auto& t : stamp_ns_) newest = std::max(newest, t.load());
    return newest;
}

void CloudMergerNode::flush_published() {
    if (!pipe_in_flight_ || !ctx_) return;
    (void)cm_publish_wait(ctx_);
    pipe_in_flight_ = false;
    if (publish_) publish_(cfg_.voxel_topic, pipe_msg_[pipe_cur_ ^ 1]);
    frames_.fetch_add(1);
}

void CloudMergerNode::flush() {
    if (frame_pending_ && ctx_) {                                       // deferred_wait: the frame enqueued by the last spin_once
        flush_published();
        frame_pending_ = false;
        (void)collect_and_publish_async(nullptr);
    }
    flush_published();
}

// spin_once with NodeConfig::pipelined_publish: enqueue frame n; while it computes, finish and publish frame n - 1; wait
// for frame n; start its copy-out (cm_result_publish_async: a stream of its own, double-buffered result) and return.
int CloudMergerNode::spin_once_pipelined(cm_result* res) {
    static const bool trace = std::getenv("CM_NODE_TRACE") != nullptr;  // (debugging aid: which call of a slow tick took the time)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    auto t0 = now();
    const uint64_t stamp_enq = newest_stamp();                          // (of the clouds this fuse reads: a callback may deliver the next ones before the wait)
    const int eq = cm_merge_voxelize_async(ctx_, &cfg_.params);        // fusePointclouds + voxelgrid, enqueued
    if (eq == CM_OK) enq_stamp_ = stamp_enq;
    const double t_enq = since(t0); t0 = now();
    flush_published();                                                  // frame n - 1 goes out while frame n runs
    const double t_flush = since(t0);
    if (trace && (t_enq > 3 || t_flush > 3)) std::fprintf(stderr, "[node] slow tick: enqueue %.2f flush %.2f ms\n", t_enq, t_flush);
    if (res) *res = cm_result{};
    if (eq == CM_NOT_READY) return eq;                                  // :575 — nothing fused this tick
    if (eq < 0) { set_error(cm_last_error(ctx_)); return eq; }
    return collect_and_publish_async(res);
}

// spin_once with NodeConfig::deferred_wait (on top of pipelined_publish): the frame enqueued by the LAST call is waited for and
// its copy-out started, then this tick's frame is enqueued and the call returns without waiting for it — the kernels of frame
// n run while the subscriber callbacks of tick n + 1 copy their clouds to the device (a sensor's submit goes to the buffer
// the frame in flight does not read: cm_api.cpp Slot). The return value says whether THIS tick fused a frame (CM_OK /
// CM_NOT_READY, as always); `res` is the PREVIOUS frame's result (zero when there was none to wait for).
int CloudMergerNode::spin_once_deferred(cm_result* res) {
    if (res) *res = cm_result{};
    int st_prev = CM_OK;
    if (frame_pending_) {
        flush_published();                                              // frame n - 2 goes out
        frame_pending_ = false;
        st_prev = collect_and_publish_async(res);                       // frame n - 1: waited for, copy-out started
        if (st_prev < 0) return st_prev;
    }
    const uint64_t stamp_enq = newest_stamp();
    const int eq = cm_merge_voxelize_async(ctx_, &cfg_.params);        // frame n: enqueued, not waited for
    if (eq == CM_OK) { frame_pending_ = true; enq_stamp_ = stamp_enq; }
    else if (eq != CM_NOT_READY) { set_error(cm_last_error(ctx_)); return eq; }
    return eq;                                                          // CM_OK: a frame was fused this tick; CM_NOT_READY: none (:575)
}

// The frame in flight on the context: cm_wait, the consumed generations, its message, the start of its copy-out.
int CloudMergerNode::collect_and_publish_async(cm_result* res) {
    static const bool trace = std::getenv("CM_NODE_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    auto t0 = now();
    struct Report { bool on; double* w; double* p;
                    ~Report() { if (on && (*w > 3 || *p > 3)) std::fprintf(stderr, "[node] slow tick: wait %.2f publish %.2f ms\n", *w, *p); } };
    double t_wait = 0, t_pub = 0;
    Report rep{trace, &t_wait, &t_pub};
    cm_result r{};
    const int st = cm_wait(ctx_, &r);
    t_wait = since(t0); t0 = now();
    if (res) *res = r;
    if (st < 0) { set_error(cm_last_error(ctx_)); return st; }
    if (r.path_flags & CM_PATH_QUANTILE) n_quantile_.fetch_add(1);
    if (r.path_flags & CM_PATH_REDONE) n_redone_.fetch_add(1);
    {
        cm_frame_stats fs;
        if (cm_get_frame_stats(ctx_, &fs) == CM_OK)
            for (uint32_t k = 0; k < fs.n_sensors; ++k)
                if (fs.sensor[k] < consumed_.size()) consumed_[fs.sensor[k]].store(fs.generation[k]);
    }
    PointCloud2& msg = pipe_msg_[pipe_cur_];
    const bool pcl = cfg_.publish_pcl_layout;
    if (msg.fields.empty() || msg.point_step != (pcl ? 32u : 16u)) msg = pcl ? make_pcl_xyzi_message(0) : make_xyzi16_message(0);
    // The payload buffer is reserved once at the largest cloud the context can produce and made DMA-able: the copy-out
    // then is a true asynchronous transfer straight into the message that goes on the wire.
    const size_t cap_bytes = static_cast<size_t>(cfg_.max_points_total) * msg.point_step;
    if (msg.data.capacity() < cap_bytes || pipe_registered_[pipe_cur_] != msg.data.data()) {
        if (pipe_registered_[pipe_cur_]) { (void)cm_host_unregister(pipe_registered_[pipe_cur_]); pipe_registered_[pipe_cur_] = nullptr; }
        msg.data.resize(cap_bytes);                                     // (every page touched once before it is pinned)
        msg.data.resize(1);                                             // (capacity stays; data() of an empty vector need not be its buffer)
        if (cm_host_register(msg.data.data(), cap_bytes) == CM_OK) pipe_registered_[pipe_cur_] = msg.data.data();
    }
    msg.height = 1; msg.width = static_cast<uint32_t>(r.n_out);
    msg.row_step = msg.point_step * msg.width;
    msg.data.resize(static_cast<size_t>(r.n_out) * msg.point_step);    // (within the reserved capacity: the buffer stays put)
    if (st == CM_EMPTY_INPUT) { msg.width = 0; msg.height = 0; msg.row_step = 0; }   // A.4 step 1
    if (r.n_out) {
        const int cs = cm_result_publish_async(ctx_, msg.data.data(), r.n_out, msg.point_step);
        if (cs != CM_OK) { set_error(cm_last_error(ctx_)); return cs; }
    }
    msg.header.seq = seq_++;
    msg.header.stamp_ns = (cfg_.stamp_from_inputs && enq_stamp_) ? enq_stamp_ : clock_();
    msg.header.frame_id = cfg_.base_frame;
    pipe_in_flight_ = true;
    pipe_cur_ ^= 1;
    t_pub = since(t0);
    return st;
}

int CloudMergerNode::sensor_by_topic(const std::string& topic) const {
    for (size_t s = 0; s < cfg_.sensors.size(); ++s)
        if (cfg_.sensors[s].topic == topic) return static_cast<int>(s);
    return -1;
}

int CloudMergerNode::set_transform(size_t sensor, const double q[4], const double t[3]) {
    if (!ctx_ || sensor >= cfg_.sensors.size()) return CM_BAD_ARG;
    const int st = cm_set_sensor_transform(ctx_, static_cast<uint32_t>(sensor), q, t);
    if (st == CM_OK) have_tf_[sensor].store(true);
    return st;
}

bool CloudMergerNode::transforms_ready() const {
    for (const auto& f : have_tf_)
        if (!f.load()) return false;
    return true;
}

int CloudMergerNode::on_cloud(size_t sensor, const PointCloud2& msg, bool* accepted) {
    if (accepted) *accepted = false;
    if (!ctx_ || sensor >= cfg_.sensors.size()) return CM_BAD_ARG;
    if (!transforms_ready()) return CM_NOT_READY;
    const XyziLayout l = find_xyzi(msg);
    if (!l.ok) { set_error(l.error); return CM_BAD_ARG; }
    std::lock_guard<std::mutex> lk(slot_mu_[sensor]);
    const int st = (cfg_.async_submit ? cm_submit_cloud_async : cm_submit_cloud)(ctx_, static_cast<uint32_t>(sensor), msg.data.data(),
                                   static_cast<uint32_t>(msg.num_points()), msg.point_step, l.off_x, l.off_y, l.off_z, l.off_i);
    if (st == CM_OK) { stamp_ns_[sensor].store(msg.header.stamp_ns); submitted_[sensor].fetch_add(1); }   // CM_SKIPPED: the slot keeps its older cloud
    if (accepted) *accepted = st == CM_OK;
    return st == CM_SKIPPED ? CM_OK : st;
}

int CloudMergerNode::spin_once(cm_result* res) {
    if (!ctx_) return CM_NO_DEVICE;
    if (cfg_.max_stamp_spread_ns) {
        // The set the gate (:134) would fuse: every required sensor's unconsumed cloud. Too far apart in time:
        // drop the oldest and wait for that sensor's next cloud.
        bool complete = true;
        uint64_t lo = ~0ull, hi = 0;
        size_t oldest = 0;
        for (size_t s = 0; s < cfg_.sensors.size(); ++s) {
            if (!cfg_.sensors[s].required) continue;
            if (!fresh(s)) { complete = false; break; }
            const uint64_t t = stamp_ns_[s].load();
            if (t < lo) { lo = t; oldest = s; }
            hi = std::max(hi, t);
        }
        if (complete && hi - lo > cfg_.max_stamp_spread_ns) {
            std::lock_guard<std::mutex> lk(slot_mu_[oldest]);          // (no callback of that sensor between the clear and the reset)
            cm_clear_sensor(ctx_, static_cast<uint32_t>(oldest));
            consumed_[oldest].store(submitted_[oldest].load());
            stamp_ns_[oldest].store(0);
            dropped_.fetch_add(1);
            return CM_NOT_READY;
        }
    }
    if (cfg_.pipelined_publish && !cfg_.ground_enable) return (cfg_.deferred_wait && !cfg_.max_stamp_spread_ns) ? spin_once_deferred(res) : spin_once_pipelined(res);   // (the stamp gate above reads what the last frame consumed: known only once it was waited for)
    cm_result r{};
    const int st = cm_merge_voxelize(ctx_, &cfg_.params, &r);      // fusePointclouds + voxelgrid
    if (res) *res = r;
    if (st == CM_NOT_READY) return st;                              // :575 — nothing fused this tick
    if (st < 0) { set_error(cm_last_error(ctx_)); return st; }
    if (r.path_flags & CM_PATH_QUANTILE) n_quantile_.fetch_add(1);
    if (r.path_flags & CM_PATH_REDONE) n_redone_.fetch_add(1);
    {
        // flag reset, :151-157 — for exactly the clouds this fuse read: a callback may have delivered the next one since
        cm_frame_stats fs;
        if (cm_get_frame_stats(ctx_, &fs) == CM_OK)
            for (uint32_t k = 0; k < fs.n_sensors; ++k)
                if (fs.sensor[k] < consumed_.size()) consumed_[fs.sensor[k]].store(fs.generation[k]);
    }
    // publishPointcloud, voxel leg (:215-219): PCL layout, stamp = now, frame = base_footprint.
    PointCloud2& msg = out_msg_;
    {
        const bool pcl = cfg_.publish_pcl_layout;
        if (msg.fields.empty() || msg.point_step != (pcl ? 32u : 16u)) msg = pcl ? make_pcl_xyzi_message(0) : make_xyzi16_message(0);
        msg.height = 1; msg.width = static_cast<uint32_t>(r.n_out);
        msg.row_step = msg.point_step * msg.width;
        msg.data.resize(static_cast<size_t>(r.n_out) * msg.point_step);    // (capacity is kept: no zero-fill in steady state)
    }
    if (st == CM_EMPTY_INPUT) { msg.width = 0; msg.height = 0; msg.row_step = 0; }   // A.4 step 1
    if (r.n_out) {
        const int cs = cm_result_copy(ctx_, msg.data.data(), r.n_out, msg.point_step);
        if (cs != CM_OK) { set_error(cm_last_error(ctx_)); return cs; }
    }
    msg.header.seq = seq_++;
    uint64_t newest = 0;
    for (const auto& t : stamp_ns_) newest = std::max(newest, t.load());
    msg.header.stamp_ns = (cfg_.stamp_from_inputs && newest) ? newest : clock_();
    msg.header.frame_id = cfg_.base_frame;
    if (cfg_.ground_enable && publish_ && st != CM_EMPTY_INPUT) {
        // publishPointcloud's other two legs (:203-213): the fused no-ground and ground clouds
        for (int leg = 0; leg < 2; ++leg) {
            uint64_t n = 0;
            PointCloud2& m2 = side_msg_;
            if (m2.fields.empty()) m2 = make_xyzi16_message(0);
            m2.data.resize(static_cast<size_t>(r.n_in) * m2.point_step);
            const int cs = leg == 0 ? cm_merged_copy(ctx_, m2.data.data(), r.n_in, &n) : cm_ground_copy(ctx_, m2.data.data(), r.n_in, &n);
            if (cs != CM_OK) { set_error(cm_last_error(ctx_)); return cs; }
            m2.width = static_cast<uint32_t>(n); m2.height = n ? 1 : 0; m2.row_step = static_cast<uint32_t>(n) * m2.point_step;
            m2.data.resize(static_cast<size_t>(n) * m2.point_step);
            m2.header = msg.header;
            publish_(leg == 0 ? cfg_.no_ground_topic : cfg_.ground_topic, m2);
        }
    }
    if (publish_) publish_(cfg_.voxel_topic, msg);
    frames_.fetch_add(1);
    return st;
}

void CloudMergerNode::run(const std::atomic<bool>& stop) {
    const auto period = std::chrono::duration<double>(1.0 / cfg_.rate_hz);
    auto next = std::chrono::steady_clock::now();
    while (!stop.load()) {
        spin_once();
        next += std::chrono::duration_cast<std::chrono::steady_clock::duration>(period);
        std::this_thread::sleep_until(next);                        // loop_rate.sleep(), :583
    }
}

}  // namespace cloudmerge
