// lds_fadd_order.hip — does ds_add_f32 (LDS float atomic add, no return) give a SEQUENTIAL fp32 sum in lane
// order (and, for successive instructions of one wave, in program order), bit for bit equal to a chain of
// round-to-nearest adds? Not an architectural promise: the voxel finish (k3_local) only accumulates this way on
// a device where the same check (k3_probe_fadd, run at cm_create) finds no difference.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lds_fadd_order lds_fadd_order.hip && ./lds_fadd_order
// Each wave: R rounds; in round r lane l adds v[r][l] to cell c[r][l], cells non-decreasing along (r, l) the way
// sorted voxel runs are (runs of random length, also runs longer than a wave and runs crossing rounds). Values:
// random magnitudes over 40 binades, both signs, a share of denormals, zeros, -0.0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

constexpr int R = 8, CELLS = 512, WAVES = 4;

__global__ __launch_bounds__(64 * WAVES) void k_probe(const float* __restrict__ v, const uint16_t* __restrict__ c,
                                                     float* __restrict__ out) {
    __shared__ float acc[WAVES][CELLS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = lane; k < CELLS; k += 64) acc[w][k] = 0.0f;
    __syncthreads();
    const size_t b = (static_cast<size_t>(blockIdx.x) * WAVES + w) * R * 64;
    float x[R]; uint32_t cell[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = v[b + r * 64 + lane]; cell[r] = c[b + r * 64 + lane]; }
#pragma unroll
    for (int r = 0; r < R; ++r) __hip_atomic_fetch_add(&acc[w][cell[r]], x[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    for (int k = lane; k < CELLS; k += 64) out[(static_cast<size_t>(blockIdx.x) * WAVES + w) * CELLS + k] = acc[w][k];
}

int main() {
    const int blocks = 2048;
    const size_t n = static_cast<size_t>(blocks) * WAVES * R * 64;
    std::vector<float> v(n); std::vector<uint16_t> c(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t wv = 0; wv < n / (R * 64); ++wv) {
        uint32_t cell = 0; int left = 0;
        const int mode = rnd() % 4;                    // typical run lengths per wave: 1-3, 1-20, 1-200, one giant run
        for (int i = 0; i < R * 64; ++i) {
            if (left == 0) {
                if (i) ++cell;
                left = mode == 0 ? 1 + rnd() % 3 : mode == 1 ? 1 + rnd() % 20 : mode == 2 ? 1 + rnd() % 200 : 1 + rnd() % 600;
            }
            --left;
            c[wv * R * 64 + i] = static_cast<uint16_t>(cell);
            const uint64_t r = rnd();
            float f;
            const int kind = r % 16;
            if (kind == 0) { uint32_t u = (r >> 8) & 0x007FFFFFu; u |= (r >> 40 & 1u) << 31; std::memcpy(&f, &u, 4); }   // denormal
            else if (kind == 1) { uint32_t u = (r >> 40 & 1u) << 31; std::memcpy(&f, &u, 4); }                             // +-0
            else { uint32_t u = ((100u + (r >> 8) % 40u) << 23) | ((r >> 20) & 0x007FFFFFu) | ((r >> 50 & 1u) << 31); std::memcpy(&f, &u, 4); }
            v[wv * R * 64 + i] = f;
        }
    }
    float *dv, *dout; uint16_t* dc;
    hipMalloc(&dv, n * 4); hipMalloc(&dc, n * 2); hipMalloc(&dout, static_cast<size_t>(blocks) * WAVES * CELLS * 4);
    hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64 * WAVES), 0, 0, dv, dc, dout);
    std::vector<float> out(static_cast<size_t>(blocks) * WAVES * CELLS);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, cells = 0, bad_denorm = 0;
    for (size_t wv = 0; wv < n / (R * 64); ++wv) {
        std::vector<float> ref(CELLS, 0.0f);
        std::vector<int> used(CELLS, 0);
        for (int i = 0; i < R * 64; ++i) {
            const uint16_t k = c[wv * R * 64 + i];
            volatile float t = ref[k] + v[wv * R * 64 + i];     // one rounding per add, left to right (x86-64 SSE: RN, denormals kept)
            ref[k] = t; used[k] = 1;
        }
        for (int k = 0; k < CELLS; ++k) {
            if (!used[k]) continue;
            ++cells;
            uint32_t a, b2; std::memcpy(&a, &ref[k], 4); std::memcpy(&b2, &out[wv * CELLS + k], 4);
            if (a != b2) { ++bad; if ((a & 0x7F800000u) == 0 || (b2 & 0x7F800000u) == 0) ++bad_denorm;
                if (bad <= 5) std::printf("mismatch wave %zu cell %d: host %08x device %08x\n", wv, k, a, b2); }
        }
    }
    std::printf("cells %zu  mismatches %zu (of which with a denormal/zero result %zu)\n", cells, bad, bad_denorm);
    return bad ? 1 : 0;
}
