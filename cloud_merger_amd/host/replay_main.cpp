// replay_main.cpp — replays a recorded multi-sensor .pcd sequence through CloudMergerNode, the way
// the reference is exercised with `rosbag play` (my_cloud_fusion/launch/bag.launch:7). BASELINE
// config 4: frames are independent, so `--shard r/w` gives rank r every w-th frame (one process
// per GPU, no collective).
//
//   cloudmerge_replay --dir SEQ --sensors 4 --frames 100 [--config NODE.cfg] [--leaf 0.05] [--min-pts 2]
//                     [--crop x0 y0 z0 x1 y1 z1] [--outlier RADIUS MIN_NEIGHBOURS] [--out OUTDIR]
//                     [--device 0] [--shard 0/1]
//                     [--rate 10 --realtime] [--threads] [--pipeline | --defer] [--pin] [--async]
// --threads: the reference's threading — one subscriber thread per sensor (ros::AsyncSpinner(6), :513) hands the clouds to
//   the node while the main thread runs the loop body (:570-580). A sensor's thread offers its next cloud again until the
//   slot has been consumed (lossless, unlike the live node, which drops: :330), so frame k is made of every sensor's
//   cloud k. The copy of frame k+1 into HBM then runs beside frame k's kernels and its publish: the slots are double-buffered.
// SEQ/transforms.txt : one line per sensor "qx qy qz qw tx ty tz" (tf lookup results)
// SEQ/frame_%04d_sensor_%d.pcd : FIELDS x y z [intensity], FLOAT32
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "merger_node.hpp"
#include "pcd_io.hpp"

using namespace cloudmerge;

int main(int argc, char** argv) {
    std::string dir, out_dir;
    int n_sensors = 4, n_frames = 1, device = 0, rank = 0, world = 1;
    double rate = 10.0;
    bool realtime = false, have_config = false, threads = false, pin = false;
    int repeat = 1;
    NodeConfig cfg;
    cfg.params.leaf[0] = cfg.params.leaf[1] = cfg.params.leaf[2] = 0.05f;
    cfg.params.min_points_per_voxel = 2;
    cfg.params.downsample_all_data = 1;
    for (int a = 1; a < argc; ++a) {
        const std::string k = argv[a];
        auto next = [&](int n = 1) { if (a + n >= argc) { std::fprintf(stderr, "missing value for %s\n", k.c_str()); std::exit(2); } return argv[++a]; };
        if (k == "--config") {                                // node description file (merger_node.hpp: load_config)
            std::string e;
            if (!load_config(next(), &cfg, &e)) { std::fprintf(stderr, "%s\n", e.c_str()); return 2; }
            n_sensors = static_cast<int>(cfg.sensors.size());
            have_config = true;
        } else if (k == "--dir") dir = next();
        else if (k == "--out") out_dir = next();
        else if (k == "--sensors") n_sensors = std::atoi(next());
        else if (k == "--frames") n_frames = std::atoi(next());
        else if (k == "--device") device = std::atoi(next());
        else if (k == "--rate") rate = std::atof(next());
        else if (k == "--realtime") realtime = true;
        else if (k == "--threads") threads = true;
        else if (k == "--async") cfg.async_submit = true;           // --threads only (the preloaded clouds never change): on_cloud does not wait for its copy
        else if (k == "--pin") pin = true;                          // one thread: the clouds' host buffers are made DMA-able once (cm_host_register)
        else if (k == "--pipeline") cfg.pipelined_publish = true;   // publish frame n - 1 while frame n computes (merger_node.hpp)
        else if (k == "--defer") cfg.pipelined_publish = cfg.deferred_wait = true;   // ... and wait for frame n during tick n + 1: its kernels run beside that tick's host-to-device copies
        else if (k == "--live") {                                  // the live node's own configuration (pc_preprocessing_main.cpp): six sensors,
            const bool pp = cfg.pipelined_publish, dw = cfg.deferred_wait, as = cfg.async_submit;                  // ROI, 10 cm, min 2 points, zone-wise ground removal + per-slab outlier filter
            cfg = live_node_config();
            cfg.pipelined_publish = pp; cfg.deferred_wait = dw; cfg.async_submit = as;
            n_sensors = static_cast<int>(cfg.sensors.size());
            have_config = true;
        }
        else if (k == "--repeat") repeat = std::max(1, std::atoi(next()));      // --threads: play the sequence this many times (the first pass is the warm-up: not timed)
        else if (k == "--leaf") { const float v = std::strtof(next(), nullptr); cfg.params.leaf[0] = cfg.params.leaf[1] = cfg.params.leaf[2] = v; }
        else if (k == "--min-pts") cfg.params.min_points_per_voxel = static_cast<uint32_t>(std::atoi(next()));
        else if (k == "--crop") {
            cfg.params.crop_enable = 1;
            for (int i = 0; i < 3; ++i) cfg.params.crop_min[i] = std::strtof(next(), nullptr);
            for (int i = 0; i < 3; ++i) cfg.params.crop_max[i] = std::strtof(next(), nullptr);
        } else if (k == "--outlier") {                       // radius, min neighbours (CloudFusionNode.h:74-85)
            cfg.params.outlier_enable = 1;
            cfg.params.outlier_radius = std::strtof(next(), nullptr);
            cfg.params.outlier_min_neighbors = static_cast<uint32_t>(std::atoi(next()));
        } else if (k == "--shard") {
            if (std::sscanf(next(), "%d/%d", &rank, &world) != 2 || world < 1 || rank < 0 || rank >= world) { std::fprintf(stderr, "bad --shard\n"); return 2; }
        } else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (dir.empty() || n_sensors < 1 || n_sensors > CM_MAX_SENSORS) { std::fprintf(stderr, "usage: see header of replay_main.cpp\n"); return 2; }

    if (!have_config)
        for (int s = 0; s < n_sensors; ++s)
            cfg.sensors.push_back({"sensor" + std::to_string(s), "/sensor" + std::to_string(s) + "/points", "/sensor" + std::to_string(s), true});
    cfg.device = device;
    cfg.rate_hz = rate;
    cfg.max_points_total = 0;

    // size the context from the first frame
    std::string err;
    std::vector<PointCloud2> clouds(n_sensors);
    auto frame_path = [&](int f, int s) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "/frame_%04d_sensor_%d.pcd", f, s);
        return dir + buf;
    };
    size_t first_total = 0;
    for (int s = 0; s < n_sensors; ++s) {
        if (!read_pcd(frame_path(rank, s), &clouds[s], &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        first_total += clouds[s].num_points();
    }
    cfg.max_points_total = first_total + first_total / 2 + 1024;

    if (!threads) cfg.async_submit = false;                    // (one thread: the next frame is read into the same buffers)
    CloudMergerNode node(cfg);
    if (!node.ok()) { std::fprintf(stderr, "node: %s\n", node.error().c_str()); return 1; }
    {
        std::ifstream tf(dir + "/transforms.txt");
        for (int s = 0; s < n_sensors; ++s) {
            double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
            if (tf) tf >> q[0] >> q[1] >> q[2] >> q[3] >> t[0] >> t[1] >> t[2];
            node.set_transform(static_cast<size_t>(s), q, t);
        }
    }
    uint64_t voxels = 0, points = 0;
    uint64_t voxels_warm = 0;                                  // --threads --repeat: voxels of the warm-up pass's frames (not timed)
    size_t n_published = 0, n_warm_frames = 0;
    int cur_frame = 0;
    // frames handed to spin_once whose voxel cloud has not been published yet, oldest first (with --pipeline a frame is
    // published during the NEXT spin_once, or by flush())
    std::deque<int> unpublished;
    node.set_publisher([&](const std::string& topic, const PointCloud2& msg) {
        if (topic != cfg.voxel_topic) return;                  // (--live also publishes the no-ground and ground clouds)
        voxels += msg.num_points();
        if (++n_published == n_warm_frames) voxels_warm = voxels;   // (published frames, not fused ones: the pipelined modes publish one or two ticks late)
        const int pub_frame = unpublished.empty() ? cur_frame : unpublished.front();
        if (!unpublished.empty()) unpublished.pop_front();
        if (!out_dir.empty()) {
            char buf[64];
            std::snprintf(buf, sizeof buf, "/voxel_%04d.pcd", pub_frame);
            std::string e;
            if (!write_pcd(out_dir + buf, msg, &e)) std::fprintf(stderr, "%s\n", e.c_str());
        }
    });

    if (threads) {
        // everything is read before the clock starts (rosbag play delivers messages, it does not parse files in the callback)
        std::vector<int> my_frames;
        for (int f = rank; f < n_frames; f += world) my_frames.push_back(f);
        std::vector<std::vector<PointCloud2>> all(my_frames.size(), std::vector<PointCloud2>(n_sensors));
        for (size_t i = 0; i < my_frames.size(); ++i)
            for (int s = 0; s < n_sensors; ++s) {
                if (!read_pcd(frame_path(my_frames[i], s), &all[i][s], &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
                points += all[i][s].num_points();
                // --pin: the payloads become DMA-able (what a transport that receives into registered buffers delivers)
                if (pin && !all[i][s].data.empty()) (void)cm_host_register(all[i][s].data.data(), all[i][s].data.size());
            }
        std::atomic<bool> failed{false};
        // (set once the loop thread has fused its frames: a sensor the gate does not wait for — the reference's top_middle —
        // may have been a cloud late at some tick, and would otherwise offer its last clouds for ever to a slot nobody empties)
        std::atomic<bool> loop_done{false};
        const size_t n_play = all.size() * static_cast<size_t>(repeat);
        const size_t n_warm = repeat > 1 ? all.size() : 0;
        n_warm_frames = n_warm;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> subs;
        for (int s = 0; s < n_sensors; ++s)
            subs.emplace_back([&, s] {
                for (size_t j = 0; j < n_play && !failed.load() && !loop_done.load(); ++j) {
                    const size_t i = j % all.size();
                    bool accepted = false;
                    while (!accepted && !failed.load() && !loop_done.load()) {
                        const int st = node.on_cloud(static_cast<size_t>(s), all[i][s], &accepted);
                        if (st != CM_OK) { std::fprintf(stderr, "on_cloud: %s (%s)\n", cm_status_string(st), node.error().c_str()); failed.store(true); }
                        if (!accepted) std::this_thread::yield();
                    }
                }
            });
        int done = 0;
        while (done < static_cast<int>(n_play) && !failed.load()) {
            if (static_cast<size_t>(done) == n_warm && n_warm) t0 = std::chrono::steady_clock::now();
            cur_frame = my_frames[static_cast<size_t>(done) % all.size()];
            cm_result r{};
            unpublished.push_back(cur_frame);
            const int st = node.spin_once(&r);
            if (st == CM_NOT_READY) { unpublished.pop_back(); std::this_thread::yield(); continue; }
            if (st < 0) { std::fprintf(stderr, "frame %d: %s\n", cur_frame, cm_status_string(st)); failed.store(true); break; }
            ++done;
            if (realtime) std::this_thread::sleep_until(t0 + std::chrono::duration<double>(done / rate));
        }
        loop_done.store(true);
        for (auto& t : subs) t.join();
        if (failed.load()) return 1;
        node.flush();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (pin) for (auto& fr : all) for (auto& m : fr) if (!m.data.empty()) (void)cm_host_unregister(m.data.data());
        const int timed = done - static_cast<int>(n_warm);
        const uint64_t pts_timed = points * static_cast<uint64_t>(repeat > 1 ? repeat - 1 : 1);
        std::printf("{\"rank\": %d, \"world\": %d, \"frames\": %d, \"points_in\": %llu, \"voxels_out\": %llu, "
                    "\"wall_s\": %.6f, \"submit_merge_publish_s\": %.6f, \"frames_per_s\": %.2f, \"points_per_s\": %.3e, "
                    "\"mode\": \"subscriber threads (%d) + loop thread%s\", \"warmup_frames\": %d, "
                    "\"quantile_frames\": %llu, \"redone_frames\": %llu}\n",
                    rank, world, timed, static_cast<unsigned long long>(pts_timed), static_cast<unsigned long long>(voxels - voxels_warm),
                    wall, wall, timed / wall, pts_timed / wall, n_sensors,
                    (std::string(cfg.deferred_wait ? ", pipelined publish, deferred wait" : cfg.pipelined_publish ? ", pipelined publish" : "") + (pin ? ", pinned inputs" : "") + (cfg.async_submit ? ", asynchronous submits" : "")).c_str(),
                    static_cast<int>(n_warm), static_cast<unsigned long long>(node.frames_quantile()),
                    static_cast<unsigned long long>(node.frames_redone()));
        return 0;
    }

    const auto t0 = std::chrono::steady_clock::now();
    double t_gpu = 0;
    int done = 0;
    std::vector<double> tick_ms;                               // callbacks + loop body of every tick (host buffers in, message out)
    double t_steady = 0, slowest_ms = 0;
    int slowest_tick = -1;
    std::vector<void*> pinned(n_sensors, nullptr);             // --pin: what is registered of every sensor's buffer
    std::vector<size_t> pinned_cap(n_sensors, 0);
    for (int f = rank; f < n_frames; f += world) {
        cur_frame = f;
        for (int s = 0; s < n_sensors; ++s) {
            if (f != rank || clouds[s].data.empty())
                if (!read_pcd(frame_path(f, s), &clouds[s], &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            points += clouds[s].num_points();
            if (pin && !clouds[s].data.empty() && (clouds[s].data.data() != pinned[s] || clouds[s].data.capacity() > pinned_cap[s])) {
                if (pinned[s]) (void)cm_host_unregister(pinned[s]);  // (a sequence's clouds have one size: registered once in practice)
                pinned[s] = nullptr;
                if (cm_host_register(clouds[s].data.data(), clouds[s].data.capacity()) == CM_OK) {
                    pinned[s] = clouds[s].data.data(); pinned_cap[s] = clouds[s].data.capacity();
                }
            }
        }
        const auto g0 = std::chrono::steady_clock::now();
        for (int s = 0; s < n_sensors; ++s) {
            const int st = node.on_cloud(static_cast<size_t>(s), clouds[s]);
            if (st != CM_OK) { std::fprintf(stderr, "on_cloud: %s (%s)\n", cm_status_string(st), node.error().c_str()); return 1; }
        }
        cm_result r{};
        unpublished.push_back(f);
        const int st = node.spin_once(&r);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
        t_gpu += dt;
        if (done >= 5) { tick_ms.push_back(1e3 * dt); t_steady += dt; }   // (the first ticks allocate, bootstrap, register buffers)
        if (done >= 5 && 1e3 * dt > slowest_ms) { slowest_ms = 1e3 * dt; slowest_tick = done; }
        if (st < 0 || st == CM_NOT_READY) { std::fprintf(stderr, "frame %d: %s\n", f, cm_status_string(st)); return 1; }
        ++done;
        if (realtime) std::this_thread::sleep_until(t0 + std::chrono::duration<double>(done / rate));
    }
    node.flush();
    for (void* p : pinned) if (p) (void)cm_host_unregister(p);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::sort(tick_ms.begin(), tick_ms.end());
    const double p50 = tick_ms.empty() ? 0.0 : tick_ms[tick_ms.size() / 2];
    const double p99 = tick_ms.empty() ? 0.0 : tick_ms[std::min(tick_ms.size() - 1, static_cast<size_t>(0.99 * tick_ms.size()))];
    std::printf("{\"rank\": %d, \"world\": %d, \"frames\": %d, \"points_in\": %llu, \"voxels_out\": %llu, "
                "\"wall_s\": %.6f, \"submit_merge_publish_s\": %.6f, \"frames_per_s\": %.2f, \"points_per_s\": %.3e, "
                "\"tick_ms_p50\": %.4f, \"tick_ms_p99\": %.4f, \"slowest_tick\": %d, \"steady_frames_per_s\": %.2f, "
                "\"mode\": \"one thread: callbacks, fuse, publish%s\", \"quantile_frames\": %llu, \"redone_frames\": %llu}\n",
                rank, world, done, static_cast<unsigned long long>(points), static_cast<unsigned long long>(voxels), wall,
                t_gpu, done / t_gpu, points / t_gpu, p50, p99, slowest_tick, tick_ms.empty() ? 0.0 : tick_ms.size() / t_steady,
                cfg.deferred_wait ? (pin ? " (pipelined, deferred wait, pinned inputs)" : " (pipelined, deferred wait)")
                                  : cfg.pipelined_publish ? (pin ? " (pipelined, pinned inputs)" : " (pipelined)") : (pin ? " (pinned inputs)" : ""), static_cast<unsigned long long>(node.frames_quantile()),
                static_cast<unsigned long long>(node.frames_redone()));
    return 0;
}
