// lds_atomic_rate.hip — what LDS atomics cost on this GPU: ds_add_u32 / ds_add_rtn_u32 / ds_add_f32 with all 64 lanes on
// different words, with runs of 2-3 lanes on one word (sorted voxel runs), and with all lanes on one word.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, int SHARE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ float acc[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned r = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = w * 512 + u * 64 / SHARE + lane / SHARE + (it & 1);
            if (KIND == 0) atomicAdd(reinterpret_cast<unsigned*>(&acc[idx]), 1u);
            else if (KIND == 1) r += atomicAdd(reinterpret_cast<unsigned*>(&acc[idx]), 1u);
            else if (KIND == 2) __hip_atomic_fetch_add(&acc[idx], 1.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc[idx] = 1.5f;
        }
    }
    __syncthreads();
    if (r == 0x12345678u || acc[threadIdx.x] == -1.f) out[0] = 1.f;
}
template <int KIND, int SHARE>
void run(const char* name) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 200, blocks = 768;
    hipLaunchKernelGGL((k<KIND, SHARE>), dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, SHARE>), dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 3 workgroups x 8 waves x iters x 8 wave-instructions
    const double instr_per_cu = 3.0 * 8 * iters * 8;
    std::printf("%-34s share %2d: %7.1f us  -> %6.1f ns per wave-instruction per CU (%.0f cycles at 2.4 GHz)\n", name, SHARE, ms * 1e3,
                ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    hipFree(out);
}
int main() {
    run<3, 1>("ds_write_b32");
    run<0, 1>("ds_add_u32"); run<0, 2>("ds_add_u32"); run<0, 4>("ds_add_u32"); run<0, 64>("ds_add_u32");
    run<1, 1>("ds_add_rtn_u32"); run<1, 2>("ds_add_rtn_u32"); run<1, 64>("ds_add_rtn_u32");
    run<2, 1>("ds_add_f32"); run<2, 2>("ds_add_f32"); run<2, 4>("ds_add_f32"); run<2, 16>("ds_add_f32"); run<2, 64>("ds_add_f32");
    return 0;
}
