// atomic_or.hip — throughput of returning device-scope atomicOr to random words of a bitmap
// (what a "seen once / seen twice" voxel filter would cost), 4 M updates per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_mark(uint32_t* __restrict__ bm1, uint32_t* __restrict__ bm2, uint32_t mask_bits, uint32_t n, int ret) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    h &= mask_bits;
    const uint32_t bit = 1u << (h & 31);
    if (ret) {
        const uint32_t old = atomicOr(&bm1[h >> 5], bit);
        if (old & bit) atomicOr(&bm2[h >> 5], bit);
    } else {
        atomicOr(&bm1[h >> 5], bit);
    }
}
__global__ __launch_bounds__(256) void k_test(const uint32_t* __restrict__ bm2, uint32_t mask_bits, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    h &= mask_bits;
    if (bm2[h >> 5] >> (h & 31) & 1u) atomicAdd(out, 0u);
}
int main() {
    const uint32_t n = 4u << 20;
    for (int lg : {25, 26, 28}) {
        const size_t words = (1ull << lg) / 32;
        uint32_t *b1, *b2, *o;
        hipMalloc(&b1, words * 4); hipMalloc(&b2, words * 4); hipMalloc(&o, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int ret : {0, 1}) {
            float best = 1e9f;
            for (int it = 0; it < 8; ++it) {
                hipMemsetAsync(b1, 0, words * 4, 0); hipMemsetAsync(b2, 0, words * 4, 0);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_mark, dim3(n / 256), dim3(256), 0, 0, b1, b2, (1u << lg) - 1u, n, ret);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            std::printf("bitmap 2^%d bits (%zu MB): 4M atomicOr %s: %.2f us\n", lg, words * 4 >> 20, ret ? "returning + conditional second" : "no return", best * 1e3);
        }
        float best = 1e9f;
        for (int it = 0; it < 8; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_test, dim3(n / 256), dim3(256), 0, 0, b2, (1u << lg) - 1u, n, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        std::printf("bitmap 2^%d bits: 4M random bit tests: %.2f us\n", lg, best * 1e3);
        float msf;
        hipEventRecord(e0); hipMemsetAsync(b1, 0, words * 4, 0); hipMemsetAsync(b2, 0, words * 4, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&msf, e0, e1);
        std::printf("clearing both: %.2f us\n", msf * 1e3);
        hipFree(b1); hipFree(b2); hipFree(o);
    }
    return 0;
}
