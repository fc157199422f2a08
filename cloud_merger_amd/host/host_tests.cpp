// host_tests.cpp — CPU-side checks of the host shell (no GPU needed): PointCloud2 field lookup,
// PCD round trip, the reference's literals in reference_config(), and that the node fails loudly
// (no fallback) when no MI355X is present. Exit code 0 = all passed. Run by tests/test_host_shell.py.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

#include "merger_node.hpp"
#include "pcd_io.hpp"

using namespace cloudmerge;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static void test_find_xyzi() {
    PointCloud2 m = make_pcl_xyzi_message(3);
    XyziLayout l = find_xyzi(m);
    CHECK(l.ok && l.off_x == 0 && l.off_y == 4 && l.off_z == 8 && l.off_i == 16 && m.point_step == 32);
    PointCloud2 c = make_xyzi16_message(3);
    l = find_xyzi(c);
    CHECK(l.ok && l.off_i == 12 && c.point_step == 16);
    c.fields.pop_back();                                   // no intensity: treated as 0
    l = find_xyzi(c);
    CHECK(l.ok && l.off_i == 0xFFFFFFFFu);
    c.fields[1].datatype = PointField::FLOAT64;
    CHECK(!find_xyzi(c).ok);
    PointCloud2 shortmsg = make_xyzi16_message(3);
    shortmsg.data.resize(10);
    CHECK(!find_xyzi(shortmsg).ok);
    // Velodyne-like layout with trailing ring/time fields
    PointCloud2 v;
    v.width = 2; v.height = 1; v.point_step = 22; v.row_step = 44; v.data.resize(44);
    v.fields = {{"x", 0, PointField::FLOAT32, 1}, {"y", 4, PointField::FLOAT32, 1}, {"z", 8, PointField::FLOAT32, 1},
                {"intensity", 12, PointField::FLOAT32, 1}, {"ring", 16, PointField::UINT16, 1}, {"time", 18, PointField::FLOAT32, 1}};
    l = find_xyzi(v);
    CHECK(l.ok && l.off_i == 12);
}

static void test_pcd_roundtrip(const char* tmpdir) {
    PointCloud2 m = make_pcl_xyzi_message(5);
    for (int i = 0; i < 5; ++i) {
        float rec[8] = {i + 0.25f, -i * 2.0f, i * 0.5f, 1.0f, 10.0f * i, 0, 0, 0};
        std::memcpy(m.data.data() + i * 32, rec, 32);
    }
    const std::string path = std::string(tmpdir) + "/roundtrip.pcd";
    std::string err;
    CHECK(write_pcd(path, m, &err));
    PointCloud2 r;
    CHECK(read_pcd(path, &r, &err));
    CHECK(r.num_points() == 5 && r.point_step == 16 && r.fields.size() == 4);
    const XyziLayout l = find_xyzi(r);
    CHECK(l.ok && l.off_i == 12);
    for (int i = 0; i < 5 && r.data.size() >= 80; ++i) {
        float rec[4];
        std::memcpy(rec, r.data.data() + i * 16, 16);
        CHECK(rec[0] == i + 0.25f && rec[1] == -i * 2.0f && rec[2] == i * 0.5f && rec[3] == 10.0f * i);
    }
    // ascii variant
    const std::string apath = std::string(tmpdir) + "/ascii.pcd";
    FILE* f = std::fopen(apath.c_str(), "w");
    std::fprintf(f, "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 1\n"
                    "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA ascii\n1.5 2.5 3.5\n-1 -2 nan\n");
    std::fclose(f);
    PointCloud2 a;
    CHECK(read_pcd(apath, &a, &err));
    CHECK(a.num_points() == 2 && a.point_step == 12 && find_xyzi(a).off_i == 0xFFFFFFFFu);
    float z1;
    std::memcpy(&z1, a.data.data() + 12 + 8, 4);
    CHECK(std::isnan(z1));
    CHECK(!read_pcd(std::string(tmpdir) + "/missing.pcd", &a, &err));
    // what pcl_ros pointcloud_to_pcd writes for pcl::PointXYZI (padding fields) and a Velodyne cloud (ring: U2)
    const std::string ppath = std::string(tmpdir) + "/padded.pcd";
    f = std::fopen(ppath.c_str(), "wb");
    std::fprintf(f, "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z _ intensity _ ring\nSIZE 4 4 4 1 4 1 2\nTYPE F F F U F U U\nCOUNT 1 1 1 4 1 10 1\n"
                    "WIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA binary\n");
    for (int i = 0; i < 2; ++i) {
        unsigned char row[32] = {0};
        const float xyz[3] = {1.0f + i, 2.0f + i, 3.0f + i}, inten = 40.0f + i;
        const uint16_t ring = static_cast<uint16_t>(7 + i);
        std::memcpy(row, xyz, 12); std::memcpy(row + 16, &inten, 4); std::memcpy(row + 30, &ring, 2);
        std::fwrite(row, 1, 32, f);
    }
    std::fclose(f);
    PointCloud2 pd;
    CHECK(read_pcd(ppath, &pd, &err));
    const XyziLayout pl = find_xyzi(pd);
    CHECK(pd.num_points() == 2 && pd.point_step == 32 && pl.ok && pl.off_x == 0 && pl.off_i == 16 && pd.fields.size() == 5);
    CHECK(pd.fields.back().name == "ring" && pd.fields.back().offset == 30 && pd.fields.back().datatype == PointField::UINT16);
    float i1; std::memcpy(&i1, pd.data.data() + 32 + 16, 4);
    CHECK(i1 == 41.0f);
    // headers that lie: more points than the file holds, a zero / negative COUNT, an unknown type
    const char* bad[3] = {"FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4000000000\nHEIGHT 1\nPOINTS 4000000000\nDATA binary\n",
                          "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 -1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2 3\n",
                          "FIELDS x y z\nSIZE 4 4 3\nTYPE F F Q\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2 3\n"};
    for (int k = 0; k < 3; ++k) {
        const std::string bpath = std::string(tmpdir) + "/bad" + std::to_string(k) + ".pcd";
        f = std::fopen(bpath.c_str(), "w");
        std::fprintf(f, "VERSION 0.7\n%s", bad[k]);
        std::fclose(f);
        CHECK(!read_pcd(bpath, &pd, &err));
    }
}

static void test_reference_config() {
    const NodeConfig c = reference_config();
    CHECK(c.sensors.size() == 6);
    CHECK(c.sensors[0].topic == "/velodyne/front_right/velodyne_points" && c.sensors[0].frame == "/velodyne_front_right");
    CHECK(c.sensors[5].topic == "/livoxfront/livox/lidar" && c.sensors[5].frame == "/livox_front");
    CHECK(c.sensors[4].name == "top_middle" && !c.sensors[4].required);      // :136
    int required = 0;
    for (const auto& s : c.sensors) required += s.required ? 1 : 0;
    CHECK(required == 5);                                                     // :134
    CHECK(c.voxel_topic == "/points_voxel" && c.base_frame == "base_footprint" && c.rate_hz == 10.0);
    CHECK(c.params.leaf[0] == 0.1f && c.params.min_points_per_voxel == 2 && c.params.downsample_all_data == 1);
    CHECK(c.params.crop_enable == 1);
    CHECK(c.params.crop_min[0] == -15.0f && c.params.crop_max[0] == 60.0f);
    CHECK(c.params.crop_min[1] == -5.0f && c.params.crop_max[1] == 5.0f);
    CHECK(c.params.crop_min[2] == -0.5f && c.params.crop_max[2] == 3.0f);
    CHECK(c.params.outlier_enable == 0);
    const NodeConfig l = live_node_config();
    CHECK(l.ground_enable && l.ground.max_iterations == 1000 && l.ground.distance_threshold == 0.3f && l.ground.probability == 0.99f);
    CHECK(l.ground.n_zones[0] == 5 && l.ground.n_zones[3] == 5 && l.ground.n_zones[4] == 2 && l.ground.n_zones[5] == 4);
    CHECK(l.ground.zones[0][0].x_min == 30.0f && l.ground.zones[0][0].x_length == 30.0f && l.ground.zones[0][0].z_max_ground == 2.5f);   // front slab
    CHECK(l.ground.zones[0][4].x_min == -15.0f && l.ground.zones[0][4].x_length == 11.0f && l.ground.zones[0][4].z_max_ground == 0.5f);  // rear slab
    CHECK(l.ground.zones[0][3].x_min == -4.0f && l.ground.zones[0][2].x_min == 4.0f && l.ground.zones[0][1].x_min == 19.0f);
    CHECK(l.ground.zones[4][1].z_max_ground < 0.0f && l.ground.zones[4][1].x_min == -15.0f && l.ground.zones[4][1].x_length == 35.0f);
    CHECK(l.ground.zones[5][0].x_min == 34.0f && l.ground.zones[5][3].x_min == 4.0f && l.ground.zones[5][3].z_max_ground == 0.5f);
    CHECK(l.no_ground_topic == "/points_no_ground" && l.ground_topic == "/points_ground");
    CHECK(l.ground.outlier_radius == 0.15f && l.ground.outlier_min_neighbors == 1);      // removeGround's outlierRemoval, :119 / Parameter.h:23-24
    const NodeConfig f = fusion_config();
    CHECK(f.params.outlier_enable == 1 && f.params.outlier_radius == 0.1f && f.params.outlier_min_neighbors == 1);
    CHECK(f.sensors.size() == 6 && f.params.crop_enable == 1);
}

static void test_load_config(const char* tmpdir) {
    const std::string path = std::string(tmpdir) + "/node.cfg";
    FILE* f = std::fopen(path.c_str(), "w");
    std::fprintf(f, "# two sensors\nsensor left /left/points /lidar_left required\nsensor right /right/points /lidar_right optional\n"
                    "leaf 0.05\nmin_points_per_voxel 3\ncrop -1 -2 -3 4 5 6\noutlier 0.2 2\nstamp_from_inputs 1\nrate_hz 20\n"
                    "voxel_topic /voxels\nmax_points_total 123456\nmax_stamp_spread_ms 25\n");
    std::fclose(f);
    NodeConfig c;
    std::string err;
    CHECK(load_config(path, &c, &err));
    CHECK(c.sensors.size() == 2 && c.sensors[1].topic == "/right/points" && !c.sensors[1].required && c.sensors[0].required);
    CHECK(c.params.leaf[2] == 0.05f && c.params.min_points_per_voxel == 3 && c.params.crop_enable == 1);
    CHECK(c.params.crop_min[2] == -3.0f && c.params.crop_max[0] == 4.0f);
    CHECK(c.params.outlier_enable == 1 && c.params.outlier_radius == 0.2f && c.params.outlier_min_neighbors == 2);
    CHECK(c.stamp_from_inputs && c.rate_hz == 20.0 && c.voxel_topic == "/voxels" && c.max_points_total == 123456);
    CHECK(c.max_stamp_spread_ns == 25000000ull);
    CHECK(c.base_frame == "base_footprint");                      // untouched keys keep the reference's values
    // ground keys: roi_z_max follows the crop box whichever line comes first; the slab filter's radius has its own key
    for (int order = 0; order < 2; ++order) {
        f = std::fopen(path.c_str(), "w");
        std::fprintf(f, order == 0 ? "ground 500 0.2 0.9\ncrop -1 -2 -3 4 5 6.5\nground_outlier 0.15 1\nzone front_right 0 10 0.5\n"
                                   : "crop -1 -2 -3 4 5 6.5\nground_outlier 0.15 1\nground 500 0.2 0.9\nzone front_right 0 10 0.5\n");
        std::fclose(f);
        NodeConfig g;
        CHECK(load_config(path, &g, &err));
        CHECK(g.ground_enable && g.ground.max_iterations == 500 && g.ground.z_keep_max == 6.5f && g.ground.n_zones[0] == 1);
        CHECK(g.ground.outlier_radius == 0.15f && g.ground.outlier_min_neighbors == 1);
    }
    f = std::fopen(path.c_str(), "w");
    std::fprintf(f, "leaf -1\n");
    std::fclose(f);
    CHECK(!load_config(path, &c, &err) && err.find(":1:") != std::string::npos);
    CHECK(!load_config(std::string(tmpdir) + "/nope.cfg", &c, &err));
}

static void test_node_without_gpu_fails_loudly(bool expect_gpu) {
    NodeConfig c = reference_config();
    c.max_points_total = 1000;
    CloudMergerNode node(c);
    if (expect_gpu) {
        CHECK(node.ok());
        CHECK(node.sensor_by_topic("/livoxfront/livox/lidar") == 5 && node.sensor_by_topic("/nope") == -1);
        CHECK(!node.transforms_ready());
        PointCloud2 m = make_xyzi16_message(4);
        CHECK(node.on_cloud(0, m) == CM_NOT_READY);                        // transforms not looked up yet
        CHECK(node.spin_once() == CM_NOT_READY);
        // stamp policy + "first since last fuse wins" through the node (two sensors, no crop)
        NodeConfig c2;
        c2.sensors = {{"a", "/a", "/fa", true}, {"b", "/b", "/fb", true}};
        c2.params.leaf[0] = c2.params.leaf[1] = c2.params.leaf[2] = 0.5f;
        c2.params.downsample_all_data = 1;
        c2.max_points_total = 1000;
        c2.stamp_from_inputs = true;
        CloudMergerNode n2(c2);
        CHECK(n2.ok());
        const double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
        n2.set_transform(0, q, t); n2.set_transform(1, q, t);
        uint64_t got_stamp = 0; size_t got_pts = 0; std::string got_frame;
        n2.set_publisher([&](const std::string&, const PointCloud2& out) { got_stamp = out.header.stamp_ns; got_pts = out.num_points(); got_frame = out.header.frame_id; });
        PointCloud2 ca = make_xyzi16_message(2), cb = make_xyzi16_message(1), cc = make_xyzi16_message(1);
        const float pa[8] = {0.1f, 0.1f, 0.1f, 1.f, 0.2f, 0.2f, 0.2f, 3.f}, pb[4] = {5.f, 5.f, 5.f, 7.f}, pc[4] = {9.f, 9.f, 9.f, 9.f};
        std::memcpy(ca.data.data(), pa, 32); std::memcpy(cb.data.data(), pb, 16); std::memcpy(cc.data.data(), pc, 16);
        ca.header.stamp_ns = 1000; cb.header.stamp_ns = 2500; cc.header.stamp_ns = 9999;
        CHECK(n2.on_cloud(0, ca) == CM_OK);
        CHECK(n2.spin_once() == CM_NOT_READY);                             // sensor b still missing (:134)
        CHECK(n2.on_cloud(1, cb) == CM_OK);
        CHECK(n2.on_cloud(1, cc) == CM_OK);                                // dropped: b already holds a fresh cloud (:356)
        CHECK(n2.spin_once() == CM_OK);
        CHECK(got_pts == 2 && got_stamp == 2500 && got_frame == "base_footprint");   // stamp of the newest FUSED cloud
        // pipelined publish / deferred wait: the same frames come out, one and two calls later; flush() drains
        for (int mode = 1; mode <= 2; ++mode) {
            NodeConfig cp = c2;
            cp.pipelined_publish = true;
            cp.deferred_wait = mode == 2;
            CloudMergerNode np(cp);
            CHECK(np.ok());
            np.set_transform(0, q, t); np.set_transform(1, q, t);
            std::vector<size_t> pts; std::vector<uint64_t> stamps;
            np.set_publisher([&](const std::string&, const PointCloud2& out) { pts.push_back(out.num_points()); stamps.push_back(out.header.stamp_ns); });
            cm_result r{};
            ca.header.stamp_ns = 1000; cb.header.stamp_ns = 2500;
            CHECK(np.on_cloud(0, ca) == CM_OK && np.on_cloud(1, cb) == CM_OK);
            CHECK(np.spin_once(&r) == CM_OK);                              // frame 1: fused (mode 1: waited for; mode 2: only enqueued)
            CHECK(pts.empty() && r.n_out == (mode == 1 ? 2u : 0u));
            CHECK(np.spin_once(&r) == CM_NOT_READY);                       // nothing new to fuse; mode 2: frame 1 is waited for now
            CHECK(mode == 1 ? pts.size() == 1 : (pts.empty() && r.n_out == 2));
            ca.header.stamp_ns = 5000; cc.header.stamp_ns = 7000;
            CHECK(np.on_cloud(0, ca) == CM_OK && np.on_cloud(1, cc) == CM_OK);
            CHECK(np.spin_once(&r) == CM_OK);                              // frame 2 (points a + c: two voxels again)
            np.flush();
            CHECK(pts.size() == 2 && pts[0] == 2 && pts[1] == 2 && stamps[0] == 2500 && stamps[1] == 7000 && np.frames_published() == 2);
            if (pts.size() != 2) std::printf("  mode %d: %zu messages\n", mode, pts.size());
        }
        // approximate time synchronisation: clouds 1 s apart are not fused; the older one is dropped
        NodeConfig c3 = c2;
        c3.max_stamp_spread_ns = 50ull * 1000 * 1000;                      // 50 ms
        CloudMergerNode n3(c3);
        CHECK(n3.ok());
        n3.set_transform(0, q, t); n3.set_transform(1, q, t);
        size_t pts3 = 0; uint64_t stamp3 = 0;
        n3.set_publisher([&](const std::string&, const PointCloud2& out) { pts3 = out.num_points(); stamp3 = out.header.stamp_ns; });
        ca.header.stamp_ns = 1000000000ull; cb.header.stamp_ns = 2000000000ull;
        CHECK(n3.on_cloud(0, ca) == CM_OK && n3.on_cloud(1, cb) == CM_OK);
        CHECK(n3.spin_once() == CM_NOT_READY && n3.clouds_dropped_for_sync() == 1 && n3.frames_published() == 0);
        ca.header.stamp_ns = 2010000000ull;                                // sensor a's next cloud, 10 ms after b's
        CHECK(n3.on_cloud(0, ca) == CM_OK);
        CHECK(n3.spin_once() == CM_OK && pts3 == 2 && stamp3 == 2010000000ull && n3.frames_published() == 1);
        CHECK(n3.spin_once() == CM_NOT_READY && n3.clouds_dropped_for_sync() == 1);
        // ground removal through the node: a flat ground + a box; three topics come out
        NodeConfig c4;
        c4.sensors = {{"a", "/a", "/fa", true}};
        c4.params.leaf[0] = c4.params.leaf[1] = c4.params.leaf[2] = 0.5f;
        c4.params.downsample_all_data = 1;
        c4.params.crop_enable = 1;
        c4.params.crop_min[0] = -15.f; c4.params.crop_min[1] = -5.f; c4.params.crop_min[2] = -0.5f;
        c4.params.crop_max[0] = 60.f; c4.params.crop_max[1] = 5.f; c4.params.crop_max[2] = 3.f;
        c4.max_points_total = 1000;
        c4.ground_enable = true;
        c4.ground.max_iterations = 100; c4.ground.distance_threshold = 0.3f; c4.ground.probability = 0.99f;
        c4.ground.optimize_coefficients = 1; c4.ground.z_keep_max = 3.0f; c4.ground.seed = 7;
        c4.ground.n_zones[0] = 1; c4.ground.zones[0][0] = {0.0f, 10.0f, 0.5f};
        CloudMergerNode n4(c4);
        CHECK(n4.ok());
        n4.set_transform(0, q, t);
        size_t n_vox = 0, n_ng = 0, n_gr = 0;
        n4.set_publisher([&](const std::string& topic, const PointCloud2& out) {
            if (topic == "/points_voxel") n_vox = out.num_points();
            else if (topic == "/points_no_ground") n_ng = out.num_points();
            else if (topic == "/points_ground") n_gr = out.num_points();
        });
        PointCloud2 cg = make_xyzi16_message(104);
        for (int i = 0; i < 100; ++i) {                                   // 10 x 10 ground points at z = 0.01 * (i % 3)
            const float rec[4] = {0.5f + (i % 10), -4.5f + (i / 10), 0.01f * (i % 3), 1.f};
            std::memcpy(cg.data.data() + i * 16, rec, 16);
        }
        for (int i = 0; i < 4; ++i) {                                     // a box 1.5 m above the ground
            const float rec[4] = {5.0f + 0.1f * i, 0.0f, 1.5f, 2.f};
            std::memcpy(cg.data.data() + (100 + i) * 16, rec, 16);
        }
        CHECK(n4.on_cloud(0, cg) == CM_OK);
        CHECK(n4.spin_once() == CM_OK);
        CHECK(n_gr == 100 && n_ng == 4 && n_vox == 1);
        // a subscriber thread keeps delivering clouds — of two sizes, the larger one forcing the slot to
        // allocate — while the loop fuses and then reads the frame's by-products (cm_merged_copy, cm_ground_copy). Every
        // published triple must belong to ONE of the two clouds: ground + no-ground = that cloud's points.
        {
            c4.flags = CM_FLAG_LATEST_WINS;
            c4.max_points_total = 40000;
            CloudMergerNode n5(c4);
            CHECK(n5.ok());
            n5.set_transform(0, q, t);
            PointCloud2 big = make_xyzi16_message(104 + 30000);
            std::memcpy(big.data.data(), cg.data.data(), 104 * 16);
            for (int i = 0; i < 30000; ++i) {                             // a wall of points above the band: all "no ground"
                const float rec[4] = {1.0f + 0.0002f * i, -3.0f + 0.0001f * i, 2.0f + 0.00001f * i, 3.f};
                std::memcpy(big.data.data() + (104 + i) * 16, rec, 16);
            }
            std::atomic<bool> stop{false};
            std::atomic<int> submits{0};
            std::thread feeder([&] {
                for (int k = 0; !stop.load(); ++k) {
                    if (n5.on_cloud(0, (k & 1) ? big : cg) == CM_OK) submits.fetch_add(1);
                    std::this_thread::sleep_for(std::chrono::microseconds(150));     // (a sensor, not a lock-hogging loop)
                }
            });
            size_t ng = 0, gr = 0; int bad = 0, frames = 0, bigs = 0;
            n5.set_publisher([&](const std::string& topic, const PointCloud2& out) {
                if (topic == "/points_no_ground") ng = out.num_points();
                else if (topic == "/points_ground") gr = out.num_points();
                else { ++frames; const size_t tot = ng + gr; if (tot != 104 && tot != 30104) ++bad; if (tot == 30104) ++bigs; if (gr != 100) ++bad; }
            });
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(10);
            while (frames < 300 && std::chrono::steady_clock::now() < t_end) (void)n5.spin_once();   // (NOT_READY until the feeder delivers)
            stop.store(true);
            feeder.join();
            CHECK(bad == 0 && frames > 50 && bigs > 5 && submits.load() > frames);
            if (bad || frames <= 50 || bigs <= 5) std::printf("  bad %d frames %d bigs %d submits %d last ng %zu gr %zu err '%s'\n", bad, frames, bigs, submits.load(), ng, gr, n5.error().c_str());
        }
    } else {
        CHECK(!node.ok() && !node.error().empty());
        CHECK(node.spin_once() == CM_NO_DEVICE);
    }
}

int main(int argc, char** argv) {
    const char* tmpdir = argc > 1 ? argv[1] : "/tmp";
    const bool expect_gpu = argc > 2 && std::strcmp(argv[2], "gpu") == 0;
    test_find_xyzi();
    test_pcd_roundtrip(tmpdir);
    test_reference_config();
    test_load_config(tmpdir);
    test_node_without_gpu_fails_loudly(expect_gpu);
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "ok", failures);
    return failures ? 1 : 0;
}
