// Developer probe (not product code): how fast can ONE compute unit pull a 121 MB table (the SK-128 spectral key) out of L2 /
// Infinity Cache / HBM?  Variants: register loads (global_load_dwordx4, D loads in flight per wave) and LDS-DMA
// (global_load_lds_dwordx4 into a ring).  Grid = G workgroups of 512 threads walking disjoint or identical streams.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_probe.out fetch_probe.hip && ./fetch_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int D>
__global__ __launch_bounds__(512) void reg_stream(const uint4 *__restrict__ src, size_t chunks_per_wg, int same, unsigned *sink) {
    // a "chunk" = 8 KiB = 512 x 16 B: one 16-B load per thread; D chunks in flight per thread
    const uint4 *p = src + (same ? 0 : (size_t)blockIdx.x * chunks_per_wg * 512) + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (size_t c = 0; c + D <= chunks_per_wg; c += D) {
        uint4 v[D];
#pragma unroll
        for (int d = 0; d < D; d++) v[d] = p[(c + d) * 512];
#pragma unroll
        for (int d = 0; d < D; d++) acc.x ^= v[d].x, acc.y ^= v[d].y, acc.z ^= v[d].z, acc.w ^= v[d].w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int SLOTS>
__global__ __launch_bounds__(512) void dma_stream(const uint4 *__restrict__ src, size_t chunks_per_wg, int same, unsigned *sink) {
    __shared__ uint4 ring[SLOTS][512];
    const uint4 *p = src + (same ? 0 : (size_t)blockIdx.x * chunks_per_wg * 512) + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)&ring[0][0] + wave * 1024u;
    unsigned acc = 0;
    // SLOTS-1 chunks in flight; every wave waits for its own slice only (no barrier: nobody reads other waves' slices here)
    int slot = 0;
    size_t issued = 0;
    auto issue = [&]() {
        const uint4 *g = p + issued * 512;
        const unsigned off = __builtin_amdgcn_readfirstlane(base + (unsigned)slot * 8192u);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(off) : "memory", "m0");
        issued++;
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
    };
    for (int q = 0; q < SLOTS - 1; q++) issue();
    for (size_t c = 0; c + SLOTS <= chunks_per_wg; c++) {
        issue();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS - 1) : "memory");
        acc ^= ring[c % SLOTS][threadIdx.x].x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = 1;
}

int main() {
    const size_t bytes = 121ull << 20;
    const size_t chunks = bytes / 8192;
    uint4 *d;
    unsigned *sink;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kernel, int G, int same) {
        const size_t per = same ? chunks : chunks / G;
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(G), dim3(512), 0, 0, d, per, same, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double gb = (double)per * 8192 / 1e9;
        printf("%-14s G=%3d %s: %8.3f ms  %7.1f GB/s per WG  (%.1f B/clk @2.4GHz)  total %.1f GB/s\n", name, G, same ? "same  " : "disjnt", best, gb / (best * 1e-3),
               gb / (best * 1e-3) / 2.4, gb * G / (best * 1e-3));
    };
    for (int G : {1, 2, 8, 64, 256}) {
        for (int same : {0, 1}) {
            if (G == 1 && same) continue;
            run("reg D=4", reg_stream<4>, G, same);
            run("reg D=8", reg_stream<8>, G, same);
            run("reg D=16", reg_stream<16>, G, same);
            run("dma 4 slots", dma_stream<4>, G, same);
            run("dma 8 slots", dma_stream<8>, G, same);
            run("dma 16 slots", dma_stream<16>, G, same);
        }
    }
    return 0;
}
