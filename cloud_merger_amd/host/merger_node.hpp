// merger_node.hpp — C++ host shell with the reference node's surface: subscribe N sensor clouds,
// publish one merged + voxel-filtered cloud. It re-creates the thin L3 layer of
// pcl_preprocessing/src/pc_preprocessing_main.cpp (main :509-587, callbacks :318-508) on top of
// the C-ABI in include/cloudmerge.h; every numeric step runs in libcloudmerge_hip.so.
//
// The SURVEY.md §8f "next" rows ride on the same surface: zone-wise RANSAC ground removal (:49-122, :228-312;
// live_node_config(), NodeConfig::ground*) and radius outlier removal on the fused cloud (cm_params.outlier_*).
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cloudmerge.h"
#include "pointcloud2.hpp"

namespace cloudmerge {

struct SensorSpec {
    std::string name;       // "front_right", ...
    std::string topic;      // input topic (:520-525)
    std::string frame;      // TF frame looked up against the base frame (:556-561)
    bool required = true;   // part of the fuse gate (:134); top_middle is not (:136)
};

struct NodeConfig {
    std::vector<SensorSpec> sensors;
    std::string base_frame = "base_footprint";     // frame_id of everything published (:206,:212,:218)
    std::string voxel_topic = "/points_voxel";      // :518
    double rate_hz = 10.0;                          // ros::Rate loop_rate(10) :527
    cm_params params{};                             // Parameter.h:27-35 by default
    uint64_t max_points_total = 4u << 20;
    int device = 0;
    uint32_t flags = 0;
    bool publish_pcl_layout = true;                 // 32-byte pcl::PointXYZI records like pcl::toROSMsg
    // Pipelined publish: spin_once() enqueues frame n, publishes frame n - 1 (whose copy-out has been running on a stream of
    // its own, into a DMA-able message buffer) while frame n computes, then starts frame n's copy-out and returns — the
    // voxel cloud reaches its subscribers one tick later, the loop no longer waits for PCIe (flush() publishes the last
    // one). Off by default: the reference publishes within the tick (:574-577). Not with ground_enable (three clouds per tick).
    bool pipelined_publish = false;
    // deferred_wait (needs pipelined_publish): spin_once enqueues this tick's frame and returns; the NEXT call waits for it. The
    // frame's kernels then run while the next tick's callbacks copy their clouds to the device — for replays and bulk
    // conversion, where frames per second count; a live 10 Hz node gains nothing and loses a tick of latency. spin_once's
    // return value still says whether THIS tick fused a frame; its cm_result is the previous frame's.
    bool deferred_wait = false;
    // async_submit: on_cloud returns behind the ENQUEUE of the host-to-device copy (cm_submit_cloud_async); the frame that
    // fuses the cloud waits for the copy on the device. The message's payload must then stay valid and unchanged until that
    // frame has been waited for — a transport that keeps its receive buffers (and registers them: cm_host_register) can say
    // so; a callback whose message dies when it returns cannot.
    bool async_submit = false;
    // Stamp of the published cloud. false: ros::Time::now() at publish like the reference (:217).
    // true: the newest stamp among the fused input clouds — what pcl::PointCloud::operator+= leaves in
    // the fused cloud's header and what §8f rank 4 proposes.
    bool stamp_from_inputs = false;
    // Approximate time synchronisation (the reference fuses whatever arrived first since
    // the last fuse, however far apart in time). 0: off. Otherwise a frame is fused only when the stamps of the
    // required sensors' clouds lie within this many nanoseconds of each other; if they do not, the oldest
    // cloud is dropped and the tick is skipped, so that sensor's next cloud can complete the set.
    uint64_t max_stamp_spread_ns = 0;
    // Zone-wise ground removal before the fuse (cm_set_ground_removal). Off in
    // reference_config() (the north-star path); live_node_config() switches it on with the reference's slabs.
    bool ground_enable = false;
    cm_ground_params ground{};
    std::string no_ground_topic = "/points_no_ground";   // :516
    std::string ground_topic = "/points_ground";          // :517
};

// Loads a node description from a text file (an N-sensor configuration instead
// of the reference's string literals). Lines, '#' comments allowed:
//   sensor <name> <topic> <tf_frame> <required|optional>
//   base_frame <id> | voxel_topic <topic> | rate_hz <v> | leaf <v> | min_points_per_voxel <n>
//   crop <x0> <y0> <z0> <x1> <y1> <z1> | outlier <radius> <min_neighbors> | stamp_from_inputs <0|1>
//   max_points_total <n> | device <n> | max_stamp_spread_ms <v>
//   ground <max_iterations> <distance_threshold> <probability> | zone <sensor_name> <x_min> <x_length> <z_max_ground>
//   ground_outlier <radius> <min_neighbors>   (removeGround's outlierRemoval on every slab's band points, :119)
// Starts from reference_config() minus its sensors when the file names any. Returns false + *err.
bool load_config(const std::string& path, NodeConfig* cfg, std::string* err);

// The reference's literals: six sensors in fuse order fr, fl, rr, rl, tm, livox (:137-142), ROI
// crop (Parameter.h:31-35), leaf 0.1 m, min 2 points per voxel (Parameter.h:27-28).
NodeConfig reference_config();
// The live node as a whole (pcl_preprocessing): reference_config() plus the zone-wise ground removal of its
// callbacks — proceedFront for the four corner Velodynes (:228-269, called at :326,:352,:378,:404; proceedRear
// :277-312 is defined but never called), the top-middle slabs (:436-444), the Livox
// slabs (:475-497) — with Parameter.h:38-81's numbers. Publishes /points_no_ground and /points_ground as well.
NodeConfig live_node_config();
// The class-based variant (my_cloud_fusion): same sensors and ROI, plus radius outlier removal on the
// fused cloud before VoxelGrid (cloud_fusion_node.cpp:72-75).
NodeConfig fusion_config();

class CloudMergerNode {
public:
    using Publisher = std::function<void(const std::string& topic, const PointCloud2& msg)>;
    using Clock = std::function<uint64_t()>;        // nanoseconds; stands in for ros::Time::now()

    explicit CloudMergerNode(const NodeConfig& cfg);
    ~CloudMergerNode();
    CloudMergerNode(const CloudMergerNode&) = delete;
    CloudMergerNode& operator=(const CloudMergerNode&) = delete;

    bool ok() const { return ctx_ != nullptr; }
    std::string error() const { std::lock_guard<std::mutex> lk(err_mu_); return error_; }
    const NodeConfig& config() const { return cfg_; }
    int sensor_by_topic(const std::string& topic) const;

    // Result of listener.lookupTransform(base_frame, sensor.frame, Time(0)) (:556-561): the
    // quaternion/origin tf::Transform hands pcl_ros (:320).
    int set_transform(size_t sensor, const double q_xyzw[4], const double t_xyz[3]);
    bool transforms_ready() const;                  // flag_tf (:551,:562)

    // Subscriber callback body (:318-337 and siblings). Callable concurrently for different
    // sensors (AsyncSpinner(6), :513). Returns a cm_status; clouds are ignored until every
    // transform is known, like the reference, whose callbacks run on default transforms before
    // flag_tf but whose fuse gate only opens afterwards.
    // *accepted (optional): false when the slot still held an unconsumed cloud and this one was dropped (:330) — the
    // call still returns CM_OK like the reference's callback; a lossless replay waits and offers the cloud again.
    int on_cloud(size_t sensor, const PointCloud2& msg, bool* accepted = nullptr);

    void set_publisher(Publisher p) { publish_ = std::move(p); }
    void set_clock(Clock c) { clock_ = std::move(c); }

    // One pass of the main loop body (:570-580): fuse when the required sensors are fresh,
    // voxelise, publish. Returns CM_OK (published), CM_NOT_READY (tick skipped, :575) or an error.
    int spin_once(cm_result* res = nullptr);
    // while(ros::ok()) { spin_once(); loop_rate.sleep(); } (:549-584)
    void run(const std::atomic<bool>& stop);
    // pipelined_publish: publishes the frame whose copy-out is still in flight (end of a replay, shutdown)
    void flush();                                   // (deferred_wait: also the frame that was enqueued and not yet waited for)

    uint64_t frames_published() const { return frames_; }
    // how the fused frames were computed (cm_result.path_flags): on the one-pass quantile route / handed back and redone
    uint64_t frames_quantile() const { return n_quantile_; }
    uint64_t frames_redone() const { return n_redone_; }
    uint64_t clouds_dropped_for_sync() const { return dropped_; }

private:
    NodeConfig cfg_;
    cm_ctx* ctx_ = nullptr;
    std::vector<std::atomic<bool>> have_tf_;
    Publisher publish_;
    Clock clock_;
    mutable std::mutex err_mu_;                     // on_cloud runs on several threads at once
    std::string error_;
    void set_error(const std::string& e) { std::lock_guard<std::mutex> lk(err_mu_); error_ = e; }
    std::atomic<uint64_t> frames_{0};
    std::atomic<uint64_t> n_quantile_{0}, n_redone_{0};
    uint32_t seq_ = 0;
    PointCloud2 out_msg_, side_msg_;               // reused from frame to frame (no 3 MB zero-fill per publish)
    PointCloud2 pipe_msg_[2];                       // pipelined publish: the message being filled by the copy-out, and the one before
    void* pipe_registered_[2] = {nullptr, nullptr}; // ... their payload buffers, made DMA-able once (cm_host_register)
    int pipe_cur_ = 0;
    bool pipe_in_flight_ = false;
    bool frame_pending_ = false;                    // deferred_wait: a frame is enqueued on the context and not yet waited for
    uint64_t enq_stamp_ = 0;                        // pipelined modes: newest input stamp when the frame in flight was enqueued
    uint64_t newest_stamp() const;
    void flush_published();
    int spin_once_deferred(cm_result* res);
    int collect_and_publish_async(cm_result* res);
    int spin_once_pipelined(cm_result* res);
    std::vector<std::atomic<uint64_t>> stamp_ns_;   // stamp of the cloud each sensor slot currently holds
    // A slot holds a cloud no fuse has consumed yet when more submits were accepted for it than the last fuse had seen
    // (cm_frame_stats.generation): exact, whichever way a callback and the fuse interleave.
    std::vector<std::atomic<uint64_t>> submitted_, consumed_;
    bool fresh(size_t s) const { return submitted_[s].load() > consumed_[s].load(); }
    std::atomic<uint64_t> dropped_{0};
    // one lock per sensor: a callback's submit + bookkeeping and the loop thread's drop of that sensor's cloud (approximate
    // time synchronisation) exclude each other, so a cloud accepted in between is neither marked consumed nor loses its stamp
    std::vector<std::mutex> slot_mu_;
};

}  // namespace cloudmerge
