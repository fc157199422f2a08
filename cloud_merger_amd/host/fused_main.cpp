// fused_main.cpp — the single fused cloud across GPUs (BASELINE.json configs[4]) from the C++ side: one
// process per GPU, each with a share of the sensors, ONE exchange step over RCCL (xGMI on a node). What
// cloud_merger_amd/fused.py does through torch.distributed, with the collective called directly:
//
//   cm_merge_partial (this rank's sensors -> table of per-voxel sums, threshold deferred)
//   ncclAllGather of the table lengths, then of the tables padded to the longest one
//   cm_merge_tables (every rank: concatenate in rank order, add per voxel, threshold, divide)
//
// Reference: fusePointclouds + voxelgrid, pc_preprocessing_main.cpp:131-177, generalised to clouds that arrive on
// different GPUs. Rendezvous without MPI: rank 0 writes its ncclUniqueId to a file every rank can read (one node).
//
//   cloudmerge_fused --rank R --world W --rendezvous /tmp/cm_fused.id [--device D] [--sensors 16] [--points N]
//                    [--leaf 0.01] [--min-pts 2] [--steps 10] [--seed 5001]
// Synthetic sensors (the reference ships no data): sensor s lives on rank s % W; its points come from splitmix64(seed + s),
// uniform in the cfg5 crop box widened by 10 % (so that the crop has something to do), identity poses. Prints one JSON line
// per rank; "checksum" (sum over the output of x + 2y + 3z + 5 intensity in fp64) and "voxels_out" must agree on all ranks
// and with a single-rank run of the same sensors (tests/test_fused.py compares with the oracle through the same generator).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <ctime>
#include <thread>
#include <vector>

#include "../../include/cloudmerge.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define CMCHECK(x) do { int s_ = (x); if (s_ < 0) { std::fprintf(stderr, "%s: %s (%s)\n", #x, cm_status_string(s_), cm_last_error(ctx)); return 1; } } while (0)

static uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float unit(uint64_t& s) { return static_cast<float>(splitmix64(s) >> 40) * (1.0f / 16777216.0f); }   // [0, 1), 24 bits

int main(int argc, char** argv) {
    int rank = 0, world = 1, device = -1, n_sensors = 16, steps = 10;
    uint32_t n_points = 4000000, min_pts = 2;
    float leaf = 0.01f;
    uint64_t seed = 5001;
    std::string rendezvous = "/tmp/cm_fused.id";
    unsigned long long run_id = 0;                     // --run-id: the same number on every rank of ONE run (0: not checked)
    for (int a = 1; a < argc; ++a) {
        const std::string k = argv[a];
        auto next = [&]() { if (a + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", k.c_str()); std::exit(2); } return argv[++a]; };
        if (k == "--rank") rank = std::atoi(next());
        else if (k == "--world") world = std::atoi(next());
        else if (k == "--device") device = std::atoi(next());
        else if (k == "--sensors") n_sensors = std::atoi(next());
        else if (k == "--points") n_points = static_cast<uint32_t>(std::atoll(next()));
        else if (k == "--leaf") leaf = std::strtof(next(), nullptr);
        else if (k == "--min-pts") min_pts = static_cast<uint32_t>(std::atoi(next()));
        else if (k == "--steps") steps = std::atoi(next());
        else if (k == "--seed") seed = static_cast<uint64_t>(std::atoll(next()));
        else if (k == "--rendezvous") rendezvous = next();
        else if (k == "--run-id") run_id = std::strtoull(next(), nullptr, 10);
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (world < 1 || world > CM_MAX_SENSORS || rank < 0 || rank >= world || n_sensors < 1 || n_sensors > CM_MAX_SENSORS) {
        std::fprintf(stderr, "usage: see the header of fused_main.cpp\n");
        return 2;
    }
    if (device < 0) device = rank;
    HIPCHECK(hipSetDevice(device));

    // ---- rendezvous: rank 0 publishes {run id, communicator id} through a file (written under a temporary name, then
    // renamed). A file left by an earlier run must not be taken for this run's: rank 0 removes whatever is there before it
    // asks for an id (and removes its own file when it leaves); the other ranks only accept a file that carries their --run-id
    // (when one was given) and that was written after they started (less a grace period for ranks started by hand).
    struct Hello { unsigned long long run_id; ncclUniqueId id; };
    Hello hello{};
    const time_t t_start = std::time(nullptr);
    if (rank == 0) {
        std::remove(rendezvous.c_str());
        hello.run_id = run_id;
        NCCLCHECK(ncclGetUniqueId(&hello.id));
        const std::string tmp = rendezvous + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&hello, sizeof hello, 1, f) != 1) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
        std::fclose(f);
        if (std::rename(tmp.c_str(), rendezvous.c_str()) != 0) { std::fprintf(stderr, "cannot publish %s\n", rendezvous.c_str()); return 1; }
    } else {
        bool ok = false;
        for (int tries = 0; tries < 600 && !ok; ++tries) {
            struct stat sb;
            if (stat(rendezvous.c_str(), &sb) == 0 && sb.st_mtime + 30 >= t_start) {
                FILE* f = std::fopen(rendezvous.c_str(), "rb");
                if (f) {
                    ok = std::fread(&hello, sizeof hello, 1, f) == 1 && (run_id == 0 || hello.run_id == run_id);
                    std::fclose(f);
                }
            }
            if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!ok) { std::fprintf(stderr, "rank %d: no communicator id of this run at %s\n", rank, rendezvous.c_str()); return 1; }
    }
    const ncclUniqueId id = hello.id;
    ncclComm_t comm;
    NCCLCHECK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));

    // ---- this rank's sensors
    const float cmin[3] = {-15.0f, -5.0f, -0.5f}, cmax[3] = {45.0f, 5.0f, 3.0f};
    std::vector<int> mine;
    for (int s = 0; s < n_sensors; ++s) if (s % world == rank) mine.push_back(s);
    cm_limits lim{};
    lim.max_sensors = static_cast<uint32_t>(mine.empty() ? 1 : mine.size());
    lim.max_points_total = std::max<uint64_t>(static_cast<uint64_t>(mine.size()) * n_points, 1u << 16);
    cm_ctx* ctx = nullptr;
    { const int st = cm_create(&ctx, device, &lim); if (st != CM_OK) { std::fprintf(stderr, "cm_create: %s\n", cm_status_string(st)); return 1; } }
    CMCHECK(cm_set_stream(ctx, stream));
    std::vector<void*> dev_clouds(mine.size(), nullptr);
    {
        std::vector<float> host(static_cast<size_t>(n_points) * 4);
        for (size_t k = 0; k < mine.size(); ++k) {
            uint64_t st = seed + static_cast<uint64_t>(mine[k]);
            for (uint32_t i = 0; i < n_points; ++i) {
                for (int a = 0; a < 3; ++a) {
                    const float ext = cmax[a] - cmin[a];
                    host[4 * static_cast<size_t>(i) + a] = cmin[a] - 0.05f * ext + 1.1f * ext * unit(st);
                }
                host[4 * static_cast<size_t>(i) + 3] = 255.0f * unit(st);
            }
            HIPCHECK(hipMalloc(&dev_clouds[k], host.size() * 4));
            HIPCHECK(hipMemcpy(dev_clouds[k], host.data(), host.size() * 4, hipMemcpyHostToDevice));
            const double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
            CMCHECK(cm_set_sensor_transform(ctx, static_cast<uint32_t>(k), q, t));
        }
    }
    cm_params p{};
    p.leaf[0] = p.leaf[1] = p.leaf[2] = leaf;
    p.min_points_per_voxel = min_pts;
    p.downsample_all_data = 1;
    p.crop_enable = 1;
    for (int a = 0; a < 3; ++a) { p.crop_min[a] = cmin[a]; p.crop_max[a] = cmax[a]; }

    // Per step every rank sends {table length, error word}: a rank whose cm_merge_partial / cm_merge_tables failed says so in
    // the exchange every rank takes part in anyway, and all ranks leave together instead of the others waiting in an
    // all-gather for ever. A HIP / RCCL failure inside the loop aborts the communicator (the peers' collectives then fail
    // instead of hanging).
#define STEP_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); (void)ncclCommAbort(comm); return 1; } } while (0)
#define STEP_NCCL(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, ncclGetErrorString(e_)); (void)ncclCommAbort(comm); return 1; } } while (0)
    unsigned long long *d_len = nullptr, *d_lens = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_len), 16));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_lens), 16 * static_cast<size_t>(world)));
    void *send = nullptr, *gathered = nullptr;
    size_t send_cap = 0, gathered_cap = 0;
    // entries the library's own table buffer holds at least (one per padded input slot): up to that many the table is sent
    // from where cm_merge_partial left it, without a staging copy
    const uint64_t own_cap = mine.empty() ? 0 : static_cast<uint64_t>(mine.size()) * ((static_cast<uint64_t>(n_points) + 4095) / 4096 * 4096);
    cm_result res{};
    double t_partial = 0, t_exchange = 0, t_merge = 0;
    hipEvent_t ev_x0, ev_x1;
    HIPCHECK(hipEventCreate(&ev_x0));
    HIPCHECK(hipEventCreate(&ev_x1));
    unsigned long long my_err = 0;
    int failed_rank = -1;
    const auto t_all0 = std::chrono::steady_clock::now();
    for (int it = 0; it <= steps; ++it) {                    // (one exchange more than steps: the last merge's error word)
        const auto t0 = std::chrono::steady_clock::now();
        cm_result part{};
        uint64_t n_mine = 0;
        const void* d_table = nullptr;
        if (it < steps && !my_err) {
            for (size_t k = 0; k < mine.size() && !my_err; ++k)
                if (cm_submit_cloud_device(ctx, static_cast<uint32_t>(k), dev_clouds[k], n_points, 16, 0, 4, 8, 12) < 0) my_err = 1;
            if (!mine.empty() && !my_err) {
                if (cm_merge_partial(ctx, &p, nullptr, &part) < 0 || cm_partial_device(ctx, &d_table, &n_mine) < 0) my_err = 2;
            }
            if (my_err) std::fprintf(stderr, "rank %d: %s\n", rank, cm_last_error(ctx));
        }
        const auto t1 = std::chrono::steady_clock::now();
        const unsigned long long hello2[2] = {my_err ? 0ull : n_mine, my_err};
        STEP_HIP(hipMemcpyAsync(d_len, hello2, 16, hipMemcpyHostToDevice, stream));
        STEP_NCCL(ncclAllGather(d_len, d_lens, 2, ncclUint64, comm, stream));
        std::vector<unsigned long long> lens2(2 * static_cast<size_t>(world));
        STEP_HIP(hipMemcpyAsync(lens2.data(), d_lens, 16 * static_cast<size_t>(world), hipMemcpyDeviceToHost, stream));
        STEP_HIP(hipStreamSynchronize(stream));
        for (int r = 0; r < world; ++r) if (lens2[2 * r + 1] && failed_rank < 0) failed_rank = r;
        if (failed_rank >= 0 || it == steps) break;             // every rank sees the same words: all leave together
        unsigned long long max_n = 1;
        std::vector<unsigned long long> lens(static_cast<size_t>(world));
        for (int r = 0; r < world; ++r) { lens[r] = lens2[2 * r]; max_n = std::max(max_n, lens[r]); }
        // the tables padded to the longest, one all-gather (xGMI is a full mesh: every GPU pushes its table on all links at once)
        const size_t row = static_cast<size_t>(max_n) * sizeof(cm_partial_entry);
        if (row * world > gathered_cap) { if (gathered) STEP_HIP(hipFree(gathered)); STEP_HIP(hipMalloc(&gathered, row * world)); gathered_cap = row * world; }
        const void* src = d_table;
        if (!d_table || max_n > own_cap) {                      // (no table here, or a peer's is longer than this rank's buffer: stage)
            if (row > send_cap) { if (send) STEP_HIP(hipFree(send)); STEP_HIP(hipMalloc(&send, row)); send_cap = row; }
            if (n_mine) STEP_HIP(hipMemcpyAsync(send, d_table, static_cast<size_t>(n_mine) * sizeof(cm_partial_entry), hipMemcpyDeviceToDevice, stream));
            src = send;
        }
        STEP_HIP(hipEventRecord(ev_x0, stream));
        STEP_NCCL(ncclAllGather(src, gathered, row, ncclUint8, comm, stream));
        STEP_HIP(hipEventRecord(ev_x1, stream));                 // (no host wait: cm_merge_tables runs behind the all-gather on the same stream)
        const auto t2 = std::chrono::steady_clock::now();
        std::vector<const void*> tables(static_cast<size_t>(world));
        std::vector<uint64_t> counts(static_cast<size_t>(world));
        for (int r = 0; r < world; ++r) { tables[r] = static_cast<const char*>(gathered) + row * r; counts[r] = lens[r]; }
        if (cm_merge_tables(ctx, tables.data(), counts.data(), static_cast<uint32_t>(world), &p, &res) < 0) {
            my_err = 3;
            std::fprintf(stderr, "rank %d: %s\n", rank, cm_last_error(ctx));
        }
        const auto t3 = std::chrono::steady_clock::now();
        if (it) {   // (the first frame allocates)
            float x_ms = 0.f;                                    // the table all-gather on the device (the length exchange is in t2 - t1)
            (void)hipEventElapsedTime(&x_ms, ev_x0, ev_x1);
            t_partial += std::chrono::duration<double>(t1 - t0).count();
            t_exchange += std::chrono::duration<double>(t2 - t1).count() + 1e-3 * x_ms;
            t_merge += std::max(0.0, std::chrono::duration<double>(t3 - t2).count() - 1e-3 * x_ms);
        }
    }
    if (failed_rank >= 0) {
        std::fprintf(stderr, "rank %d: rank %d reported a failure; every rank leaves\n", rank, failed_rank);
        cm_destroy(ctx);
        (void)ncclCommDestroy(comm);
        if (rank == 0) std::remove(rendezvous.c_str());
        return 1;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all0).count();
    double checksum = 0;
    if (res.status == CM_OK && res.n_out) {
        std::vector<float> out(static_cast<size_t>(res.n_out) * 4);
        CMCHECK(cm_result_copy(ctx, out.data(), res.n_out, 16));
        for (uint64_t i = 0; i < res.n_out; ++i)
            checksum += static_cast<double>(out[4 * i]) + 2.0 * out[4 * i + 1] + 3.0 * out[4 * i + 2] + 5.0 * out[4 * i + 3];
    }
    const int timed = steps > 1 ? steps - 1 : 1;
    std::printf("{\"rank\": %d, \"world\": %d, \"sensors\": %d, \"sensors_here\": %zu, \"points_per_sensor\": %u, \"status\": %d, "
                "\"voxels_out\": %llu, \"distinct_voxels\": %llu, \"checksum\": %.9e, \"partial_ms\": %.4f, \"exchange_ms\": %.4f, "
                "\"merge_ms\": %.4f, \"steps\": %d, \"wall_s\": %.4f}\n",
                rank, world, n_sensors, mine.size(), n_points, res.status, static_cast<unsigned long long>(res.n_out),
                static_cast<unsigned long long>(res.n_merged), checksum, 1e3 * t_partial / timed, 1e3 * t_exchange / timed,
                1e3 * t_merge / timed, steps, wall);
    cm_destroy(ctx);
    for (void* d : dev_clouds) if (d) (void)hipFree(d);
    (void)ncclCommDestroy(comm);
    if (rank == 0) std::remove(rendezvous.c_str());
    return res.status < 0 ? 1 : 0;
}
