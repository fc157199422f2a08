// san_driver.cpp — drives the oracle's threaded per-sensor stage (ingest + transform + crop on one thread per sensor,
// mirroring AsyncSpinner(6), pc_preprocessing_main.cpp:513) and its serial VoxelGrid under a sanitizer build
// (scripts/host_sanitize.sh). Test infrastructure, like everything under oracle/.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cm_oracle.h"

int main() {
    const int n_sensors = 6;
    const uint32_t n = 20000;
    std::vector<std::vector<float>> data(n_sensors, std::vector<float>(static_cast<size_t>(n) * 4));
    std::vector<orc_sensor> s(n_sensors);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return static_cast<float>((x >> 11) % 100000) / 100000.0f; };
    for (int k = 0; k < n_sensors; ++k) {
        for (auto& v : data[k]) v = 20.0f * rnd() - 10.0f;
        s[k].data = data[k].data(); s[k].n = n; s[k].point_step = 16;
        s[k].off_x = 0; s[k].off_y = 4; s[k].off_z = 8; s[k].off_i = 12;
        s[k].q_xyzw[0] = 0; s[k].q_xyzw[1] = 0; s[k].q_xyzw[2] = 0.1 * k; s[k].q_xyzw[3] = 1;
        s[k].t_xyz[0] = k; s[k].t_xyz[1] = -k; s[k].t_xyz[2] = 0; s[k].is_dense = 1; s[k]._pad = 0;
    }
    orc_params p{};
    p.leaf[0] = p.leaf[1] = p.leaf[2] = 0.1f; p.min_points_per_voxel = 2; p.downsample_all_data = 1;
    p.crop_enable = 1;
    for (int a = 0; a < 3; ++a) { p.crop_min[a] = -8.f; p.crop_max[a] = 8.f; }
    std::vector<orc_point> merged(static_cast<size_t>(n) * n_sensors), out(static_cast<size_t>(n) * n_sensors);
    uint64_t last = 0;
    for (int threads = 1; threads <= 6; ++threads) {
        orc_report rep{};
        const int st = orc_merge_voxelize(s.data(), n_sensors, &p, threads, 1, merged.data(), out.data(), &rep, nullptr, nullptr);
        if (st != ORC_OK || (last && rep.n_out != last)) { std::printf("FAILED: status %d n_out %llu\n", st, (unsigned long long)rep.n_out); return 1; }
        last = rep.n_out;
    }
    std::printf("ok: %llu voxels with 1..6 threads\n", (unsigned long long)last);
    return 0;
}
