// wide_scatter.hip — what a ONE-pass scatter of 16-byte records into B bins costs on this GPU as B grows
// (round 3: can the second global pass of the bucket path go?). A 4096-record tile leaves 4096 / B records per
// bin: at B = 2048 every record is its own 32-byte run, and whether the XCD's L2 merges the runs of neighbouring
// tiles into whole lines before they reach the fabric decides the write traffic.
//   hipcc --offload-arch=gfx950 -O3 -o wide_scatter wide_scatter.hip && ./wide_scatter
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- ./wide_scatter   (bytes per variant)
// Bin numbers are given (uniformly random: balanced buckets are the design); offsets come from the host, as the
// hist + column-scan kernels would leave them: toff[t][b] = records of bin b in tiles before t (u16), bbase[b].
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

constexpr int TILE = 4096, BLOCK = 512, WAVES = 8, ITEMS = 8;

template <int B, bool XCD>
__global__ __launch_bounds__(BLOCK) void k_scatter(const float4* __restrict__ in, const uint16_t* __restrict__ bin,
                                                    const uint16_t* __restrict__ toff, const uint32_t* __restrict__ bbase,
                                                    float4* __restrict__ out, uint32_t n_tiles) {
    __shared__ uint32_t wcnt[WAVES][B / 2];      // per-wave counters, two 16-bit per word
    __shared__ uint32_t absb[B];                 // first destination of this tile's records of bin b
    uint32_t tile = blockIdx.x;
    if (XCD) {
        const uint32_t per = gridDim.x / 8;
        if (blockIdx.x < per * 8) tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    if (tile >= n_tiles) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t first = static_cast<size_t>(tile) * TILE + w * (64 * ITEMS) + lane;
    float4 rec[ITEMS];
    uint32_t b[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(in + first + r * 64));
        rec[r] = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) b[r] = bin[first + r * 64];
    for (uint32_t q = threadIdx.x; q < WAVES * B / 2; q += BLOCK) (&wcnt[0][0])[q] = 0;
    for (uint32_t q = threadIdx.x; q < B; q += BLOCK) absb[q] = bbase[q] + toff[static_cast<size_t>(tile) * B + q];
    __syncthreads();
    uint32_t rk[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t sh = (b[r] & 1u) * 16u;
        rk[r] = (atomicAdd(&wcnt[w][b[r] >> 1], 1u << sh) >> sh) & 0xFFFFu;
    }
    __syncthreads();
    // prefix over the waves, per bin (thread t: bins 2t, 2t+1 of word t; B/2 words over BLOCK threads)
    for (uint32_t q = threadIdx.x; q < B / 2; q += BLOCK) {
        uint32_t r0 = 0, r1 = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) {
            const uint32_t c = wcnt[k][q];
            wcnt[k][q] = r0 | (r1 << 16);
            r0 += c & 0xFFFFu; r1 += c >> 16;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t wp = (wcnt[w][b[r] >> 1] >> ((b[r] & 1u) * 16u)) & 0xFFFFu;
        out[absb[b[r]] + wp + rk[r]] = rec[r];
    }
}


// Variants: ITEMS records per thread (16: a workgroup takes 8192 consecutive records, table rows per 8192), STAGE: the
// records go through LDS in rounds of 2048 sorted positions and leave as runs (a wave's store instruction then covers
// neighbouring bins in ascending address order, same-bin records side by side), NT: non-temporal stores; `pad` bytes
// of dynamic LDS limit the workgroups per CU.
template <int B, int ITEMS_, bool STAGE, bool NT>
__global__ __launch_bounds__(BLOCK) void k_scatter2(const float4* __restrict__ in, const uint16_t* __restrict__ bin,
                                                     const uint16_t* __restrict__ toff, const uint32_t* __restrict__ bbase,
                                                     float4* __restrict__ out, uint32_t n_tiles) {
    constexpr int TW = BLOCK * ITEMS_;
    constexpr int WC_WORDS = WAVES * B / 2, ST_WORDS = STAGE ? 2048 * 4 + 2048 / 2 : 0;
    __shared__ uint32_t buf[WC_WORDS > ST_WORDS ? WC_WORDS : ST_WORDS];   // per-wave counters, later the staging round
    __shared__ uint32_t gofs[B];
    __shared__ uint32_t lds[WAVES];
    extern __shared__ uint32_t dyn_pad[];
    uint32_t (*wcnt)[B / 2] = reinterpret_cast<uint32_t (*)[B / 2]>(buf);
    float4* srec = reinterpret_cast<float4*>(buf);
    uint16_t* sbin = reinterpret_cast<uint16_t*>(buf + 2048 * 4);
    uint32_t tile = blockIdx.x;
    {
        const uint32_t per = gridDim.x / 8;
        if (blockIdx.x < per * 8) tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    if (tile >= n_tiles) return;
    if (dyn_pad == nullptr) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t first = static_cast<size_t>(tile) * TW + w * (64 * ITEMS_) + lane;
    float4 rec[ITEMS_];
    uint32_t b[ITEMS_];
#pragma unroll
    for (int r = 0; r < ITEMS_; ++r) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(in + first + r * 64));
        rec[r] = make_float4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int r = 0; r < ITEMS_; ++r) b[r] = bin[first + r * 64];
    for (uint32_t q = threadIdx.x; q < WAVES * B / 2; q += BLOCK) buf[q] = 0;
    __syncthreads();
    uint32_t rk[ITEMS_];
#pragma unroll
    for (int r = 0; r < ITEMS_; ++r) {
        const uint32_t sh = (b[r] & 1u) * 16u;
        rk[r] = (atomicAdd(&wcnt[w][b[r] >> 1], 1u << sh) >> sh) & 0xFFFFu;
    }
    __syncthreads();
    // per bin: prefix over the waves; tile-local first position of the bin (exclusive scan over the bins)
    constexpr int WPT = (B / 2 + BLOCK - 1) / BLOCK;       // counter words per thread (consecutive: the scan runs in bin order)
    uint32_t t0[WPT], t1[WPT], tsum = 0;
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const uint32_t q = threadIdx.x * WPT + k;
        t0[k] = t1[k] = 0;
        if (q < B / 2) {
            uint32_t r0 = 0, r1 = 0;
#pragma unroll
            for (int v = 0; v < WAVES; ++v) {
                const uint32_t c = wcnt[v][q];
                wcnt[v][q] = r0 | (r1 << 16);
                r0 += c & 0xFFFFu; r1 += c >> 16;
            }
            t0[k] = r0; t1[k] = r1; tsum += r0 + r1;
        }
    }
    // block exclusive scan of tsum
    uint32_t incl = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    uint32_t run = incl - tsum;
    for (int v = 0; v < w; ++v) run += lds[v];
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
        const uint32_t q = threadIdx.x * WPT + k;
        if (q < B / 2) {
            const uint32_t a0 = bbase[2 * q] + toff[static_cast<size_t>(tile) * B + 2 * q];
            const uint32_t a1 = bbase[2 * q + 1] + toff[static_cast<size_t>(tile) * B + 2 * q + 1];
            if (STAGE) { gofs[2 * q] = a0 - run; gofs[2 * q + 1] = a1 - (run + t0[k]); }
            else { gofs[2 * q] = a0; gofs[2 * q + 1] = a1; }
            // (STAGE: the wave prefixes become tile-local sorted positions)
            if (STAGE) {
#pragma unroll
                for (int v = 0; v < WAVES; ++v) {
                    const uint32_t c = wcnt[v][q];
                    wcnt[v][q] = ((c & 0xFFFFu) + run) | (((c >> 16) + run + t0[k]) << 16);
                }
            }
            run += t0[k] + t1[k];
        }
    }
    __syncthreads();
    auto store = [&](float4* p, const float4& v) {
        if (NT) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
            __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
        } else *p = v;
    };
    if (!STAGE) {
#pragma unroll
        for (int r = 0; r < ITEMS_; ++r) {
            const uint32_t wp = (wcnt[w][b[r] >> 1] >> ((b[r] & 1u) * 16u)) & 0xFFFFu;
            store(out + gofs[b[r]] + wp + rk[r], rec[r]);
        }
    } else {
        uint32_t pos[ITEMS_];
#pragma unroll
        for (int r = 0; r < ITEMS_; ++r) pos[r] = ((wcnt[w][b[r] >> 1] >> ((b[r] & 1u) * 16u)) & 0xFFFFu) + rk[r];
        for (uint32_t lo = 0; lo < TW; lo += 2048) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS_; ++r)
                if (pos[r] - lo < 2048u) { srec[pos[r] - lo] = rec[r]; sbin[pos[r] - lo] = static_cast<uint16_t>(b[r]); }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2048 / BLOCK; ++j) {
                const uint32_t t = j * BLOCK + threadIdx.x;
                store(out + gofs[sbin[t]] + lo + t, srec[t]);
            }
        }
    }
}

__global__ void k_check(const float4* __restrict__ out, uint32_t n, uint32_t* bad) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i + 1 < n) {
        // records carry (bin, original index) in x, y: bins ascending, stable inside a bin
        const float4 a = out[i], c = out[i + 1];
        const uint32_t ba = __float_as_uint(a.x), bc = __float_as_uint(c.x);
        if (ba > bc || (ba == bc && __float_as_uint(a.y) >= __float_as_uint(c.y))) atomicAdd(bad, 1u);
    }
}

template <int B>
void run(uint32_t n, int nbuf, float skew) {
    const uint32_t n_tiles = n / TILE;
    std::mt19937 rng(1234 + B);
    std::vector<uint16_t> hbin(n);
    std::vector<float4> hin(n);
    // skew > 0: a share `skew` of the records comes in scan order (bin = f(position)), the rest uniformly at random
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t b = rng() % B;
        if (skew > 0.f && (rng() % 1000) < skew * 1000) b = static_cast<uint32_t>((static_cast<uint64_t>(i) * B) / n);
        hbin[i] = static_cast<uint16_t>(b);
        hin[i] = make_float4(0, 0, 0, 0);
        reinterpret_cast<uint32_t*>(&hin[i])[0] = b;
        reinterpret_cast<uint32_t*>(&hin[i])[1] = i;
    }
    std::vector<uint32_t> tot(B, 0), base(B, 0);
    std::vector<uint16_t> toff(static_cast<size_t>(n_tiles) * B);
    for (uint32_t t = 0; t < n_tiles; ++t) {
        for (int b = 0; b < B; ++b) toff[static_cast<size_t>(t) * B + b] = static_cast<uint16_t>(tot[b]);
        for (uint32_t i = t * TILE; i < (t + 1) * TILE; ++i) ++tot[hbin[i]];
    }
    uint32_t mx = 0;
    for (int b = 1; b < B; ++b) base[b] = base[b - 1] + tot[b - 1];
    for (int b = 0; b < B; ++b) mx = tot[b] > mx ? tot[b] : mx;
    if (mx > 65535) { std::printf("B=%d: bin of %u records, u16 offsets do not hold it\n", B, mx); return; }
    std::vector<float4*> din(nbuf);
    float4* dout[2];
    uint16_t *dbin, *dtoff; uint32_t *dbase, *dbad;
    for (auto& p : din) { hipMalloc(&p, n * 16ull); hipMemcpy(p, hin.data(), n * 16ull, hipMemcpyHostToDevice); }
    for (auto& p : dout) hipMalloc(&p, n * 16ull);
    hipMalloc(&dbin, n * 2ull); hipMemcpy(dbin, hbin.data(), n * 2ull, hipMemcpyHostToDevice);
    hipMalloc(&dtoff, toff.size() * 2); hipMemcpy(dtoff, toff.data(), toff.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&dbase, B * 4); hipMemcpy(dbase, base.data(), B * 4, hipMemcpyHostToDevice);
    hipMalloc(&dbad, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int xcd = 0; xcd < 2; ++xcd) {
        auto launch = [&](int i) {
            if (xcd) hipLaunchKernelGGL((k_scatter<B, true>), dim3(n_tiles), dim3(BLOCK), 0, 0, din[i % nbuf], dbin, dtoff, dbase, dout[i & 1], n_tiles);
            else hipLaunchKernelGGL((k_scatter<B, false>), dim3(n_tiles), dim3(BLOCK), 0, 0, din[i % nbuf], dbin, dtoff, dbase, dout[i & 1], n_tiles);
        };
        for (int i = 0; i < 3; ++i) launch(i);
        hipDeviceSynchronize();
        float best = 1e9f, tot_ms = 0;
        const int reps = 20;
        for (int i = 0; i < reps; ++i) {
            hipEventRecord(e0); launch(i); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot_ms += ms;
        }
        hipMemset(dbad, 0, 4);
        hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, dout[(reps - 1) & 1], n, dbad);
        uint32_t bad = 0; hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
        std::printf("B=%5d skew %.2f %s  best %7.2f us  avg %7.2f us  (%5.2f TB/s of 2x%llu MB)  order errors %u\n", B, skew,
                    xcd ? "xcd-ranges" : "linear    ", best * 1e3, tot_ms / reps * 1e3, 2.0 * n * 16 / (best * 1e-3) / 1e12, n * 16ull >> 20, bad);
    }
    for (auto& p : din) hipFree(p);
    for (auto& p : dout) hipFree(p);
    hipFree(dbin); hipFree(dtoff); hipFree(dbase); hipFree(dbad);
}

template <int B, int ITEMS_, bool STAGE, bool NT>
void run2(uint32_t n, int nbuf, int pad_bytes) {
    constexpr int TW = BLOCK * ITEMS_;
    const uint32_t n_tiles = n / TW;
    std::mt19937 rng(1234 + B);
    std::vector<uint16_t> hbin(n);
    std::vector<float4> hin(n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t b = rng() % B;
        hbin[i] = static_cast<uint16_t>(b);
        hin[i] = make_float4(0, 0, 0, 0);
        reinterpret_cast<uint32_t*>(&hin[i])[0] = b;
        reinterpret_cast<uint32_t*>(&hin[i])[1] = i;
    }
    std::vector<uint32_t> tot(B, 0), base(B, 0);
    std::vector<uint16_t> toff(static_cast<size_t>(n_tiles) * B);
    for (uint32_t t = 0; t < n_tiles; ++t) {
        for (int b = 0; b < B; ++b) toff[static_cast<size_t>(t) * B + b] = static_cast<uint16_t>(tot[b]);
        for (uint32_t i = t * TW; i < (t + 1) * TW; ++i) ++tot[hbin[i]];
    }
    for (int b = 1; b < B; ++b) base[b] = base[b - 1] + tot[b - 1];
    std::vector<float4*> din(nbuf);
    float4* dout[2];
    uint16_t *dbin, *dtoff; uint32_t *dbase, *dbad;
    for (auto& p : din) { hipMalloc(&p, n * 16ull); hipMemcpy(p, hin.data(), n * 16ull, hipMemcpyHostToDevice); }
    for (auto& p : dout) hipMalloc(&p, n * 16ull);
    hipMalloc(&dbin, n * 2ull); hipMemcpy(dbin, hbin.data(), n * 2ull, hipMemcpyHostToDevice);
    hipMalloc(&dtoff, toff.size() * 2); hipMemcpy(dtoff, toff.data(), toff.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&dbase, B * 4); hipMemcpy(dbase, base.data(), B * 4, hipMemcpyHostToDevice);
    hipMalloc(&dbad, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = k_scatter2<B, ITEMS_, STAGE, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto launch = [&](int i) { hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(BLOCK), pad_bytes, 0, din[i % nbuf], dbin, dtoff, dbase, dout[i & 1], n_tiles); };
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize();
    float best = 1e9f, tot_ms = 0;
    const int reps = 20;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0); launch(i); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot_ms += ms;
    }
    hipMemset(dbad, 0, 4);
    hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, dout[(reps - 1) & 1], n, dbad);
    uint32_t bad = 0; hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    std::printf("v2 B=%5d items %2d stage %d nt %d pad %6d  best %7.2f us  avg %7.2f us  order errors %u  (%s)\n", B, ITEMS_, STAGE ? 1 : 0, NT ? 1 : 0,
                pad_bytes, best * 1e3, tot_ms / reps * 1e3, bad, hipGetErrorString(hipGetLastError()));
    for (auto& p : din) hipFree(p);
    for (auto& p : dout) hipFree(p);
    hipFree(dbin); hipFree(dtoff); hipFree(dbase); hipFree(dbad);
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? std::atoi(argv[1]) : 4000 * 1024;   // a multiple of 4096 * 8
    const int nbuf = argc > 2 ? std::atoi(argv[2]) : 4;
    if (argc > 3) {
        for (float skew : {0.f, 0.5f}) {
            run<256>(n, nbuf, skew);
            run<1024>(n, nbuf, skew);
            run<2048>(n, nbuf, skew);
            run<4096>(n, nbuf, skew);
        }
        return 0;
    }
    // B = 2048: direct / staged, 4096- / 8192-record workgroups, nt stores, workgroups per CU (LDS padding)
    run2<2048, 8, false, false>(n, nbuf, 0);
    run2<2048, 8, false, false>(n, nbuf, 20 * 1024);    // ~2 per CU
    run2<2048, 8, false, false>(n, nbuf, 60 * 1024);    // 1 per CU
    run2<2048, 8, false, true>(n, nbuf, 0);
    run2<2048, 8, true, false>(n, nbuf, 0);
    run2<2048, 8, true, true>(n, nbuf, 0);
    run2<2048, 16, false, false>(n, nbuf, 0);
    run2<2048, 16, true, false>(n, nbuf, 0);
    run2<1024, 8, true, false>(n, nbuf, 0);
    run2<1024, 16, true, false>(n, nbuf, 0);
    run2<256, 8, true, false>(n, nbuf, 0);
    return 0;
}
