// stream_bw.hip — what a plain streaming kernel reaches on this GPU for the sizes of this path
// (64 MB read, 64 MB read + 64 MB write), to put the kernels' GB/s in context.
//   hipcc --offload-arch=gfx950 -O3 -o stream_bw stream_bw.hip && ./stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ITEMS>
__global__ __launch_bounds__(512) void k_read(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t base = static_cast<size_t>(blockIdx.x) * 512 * ITEMS + (threadIdx.x >> 6) * 64 * ITEMS + (threadIdx.x & 63);
    float4 v[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) v[r] = (base + r * 64 < n) ? in[base + r * 64] : make_float4(0, 0, 0, 0);
    float s = 0;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) s += v[r].x + v[r].y + v[r].z + v[r].w;
    if (s == 123.456f) out[0] = s;
}
template <int ITEMS>
__global__ __launch_bounds__(512) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    const size_t base = static_cast<size_t>(blockIdx.x) * 512 * ITEMS + (threadIdx.x >> 6) * 64 * ITEMS + (threadIdx.x & 63);
    float4 v[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) v[r] = (base + r * 64 < n) ? in[base + r * 64] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) if (base + r * 64 < n) out[base + r * 64] = v[r];
}
// persistent grid-stride variant: 256 CUs x 8 workgroups
__global__ __launch_bounds__(256) void k_read_gs(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    float s = 0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256 * 4) {
        float4 a = in[i];
        float4 b = (i + gridDim.x * 256ull < n) ? in[i + gridDim.x * 256ull] : make_float4(0, 0, 0, 0);
        float4 c = (i + gridDim.x * 512ull < n) ? in[i + gridDim.x * 512ull] : make_float4(0, 0, 0, 0);
        float4 d = (i + gridDim.x * 768ull < n) ? in[i + gridDim.x * 768ull] : make_float4(0, 0, 0, 0);
        s += a.x + b.y + c.z + d.w;
    }
    if (s == 123.456f) out[0] = s;
}

int main() {
    for (size_t mb : {64, 256, 1024}) {
        const size_t n = mb * 1024 * 1024 / 16;
        float4 *a, *b; float* o;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, 64);
        hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto timeit = [&](const char* name, auto launch, double bytes) {
            for (int i = 0; i < 3; ++i) launch();
            hipDeviceSynchronize();
            const int reps = 20;
            float best = 1e9f, tot = 0;
            for (int i = 0; i < reps; ++i) {
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot += ms;
            }
            std::printf("%5zu MB %-22s best %7.2f us  avg %7.2f us  -> %6.2f TB/s (best)\n", mb, name, best * 1e3, tot / reps * 1e3, bytes / (best * 1e-3) / 1e12);
        };
        const unsigned g8 = static_cast<unsigned>((n + 512 * 8 - 1) / (512 * 8)), g4 = static_cast<unsigned>((n + 512 * 4 - 1) / (512 * 4));
        timeit("read 8x16B/thread", [&] { hipLaunchKernelGGL(k_read<8>, dim3(g8), dim3(512), 0, 0, a, o, n); }, n * 16.0);
        timeit("read 4x16B/thread", [&] { hipLaunchKernelGGL(k_read<4>, dim3(g4), dim3(512), 0, 0, a, o, n); }, n * 16.0);
        timeit("read grid-stride 2048wg", [&] { hipLaunchKernelGGL(k_read_gs, dim3(2048), dim3(256), 0, 0, a, o, n); }, n * 16.0);
        timeit("copy 8x16B/thread", [&] { hipLaunchKernelGGL(k_copy<8>, dim3(g8), dim3(512), 0, 0, a, b, n); }, n * 32.0);
        timeit("copy 4x16B/thread", [&] { hipLaunchKernelGGL(k_copy<4>, dim3(g4), dim3(512), 0, 0, a, b, n); }, n * 32.0);
        timeit("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); }, n * 32.0);
        timeit("empty launch", [&] { hipLaunchKernelGGL(k_read<8>, dim3(1), dim3(512), 0, 0, a, o, 0); }, 1.0);
        hipFree(a); hipFree(b); hipFree(o);
    }
    return 0;
}
