// Developer probe (not product code): how do FP64 and 32-bit VALU instructions of ONE or TWO waves share a SIMD of gfx950?
// Every CU runs one workgroup; waves 0-3 land on the four SIMDs, waves 4-7 are their partners.  A role is a straight-line loop body of
// independent (or dependent) instructions; the kernel reports s_memtime cycles per instruction for the wave of role A (lane 0 of wave 0)
// and of role B (wave 4).      hipcc --offload-arch=gfx950 -O3 -o issue_probe.out issue_probe.hip && ./issue_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Role { NONE = 0, F64_IND, F64_DEP, I32_IND, MIX_IND, DPP_IND, F64_DEP2, F64_DEP4 };

template <int ROLE>
__device__ __forceinline__ void body(double (&a)[8], unsigned (&u)[8], double c, int iters) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {   // 64 instructions per iteration
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (ROLE == F64_IND) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[q]) : "v"(c));
                if (ROLE == F64_DEP) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[0]) : "v"(c));
                if (ROLE == F64_DEP2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[q & 1]) : "v"(c));
                if (ROLE == F64_DEP4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[q & 3]) : "v"(c));
                if (ROLE == I32_IND) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[q]) : "v"(u[(q + 1) & 7]));
                if (ROLE == DPP_IND) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(u[q]));
                if (ROLE == MIX_IND) {
                    if (q & 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[q]) : "v"(u[(q + 1) & 7]));
                    else asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[q]) : "v"(c));
                }
            }
        }
    }
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long *out, double *sink) {
    const int wave = threadIdx.x >> 6;
    double a[8];
    unsigned u[8];
    for (int q = 0; q < 8; q++) a[q] = 1.0 + threadIdx.x * 1e-9 + q, u[q] = threadIdx.x * 2654435761u + q;
    const double c = 1.0000001;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) body<RA>(a, u, c, iters);
    else body<RB>(a, u, c, iters);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    unsigned x = 0;
    for (int q = 0; q < 8; q++) s += a[q], x ^= u[q];
    if (s == 123.456 && x == 77) sink[0] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int RA, int RB>
void run(const char *name, int waves_b) {
    unsigned long long *d_out;
    double *d_sink;
    CK(hipMalloc(&d_out, 8 * sizeof(unsigned long long)));
    CK(hipMalloc(&d_sink, 8));
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(waves_b ? 512 : 256), 0, 0, iters, d_out, d_sink);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[8] = {0};
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    const double n = 64.0 * iters;
    printf("%-44s  A: %6.2f cyc/instr", name, h[0] / n);
    if (waves_b) printf("   B: %6.2f cyc/instr", h[4] / n);
    printf("\n");
    CK(hipFree(d_out));
    CK(hipFree(d_sink));
}

int main() {
    run<F64_IND, NONE>("one wave/SIMD: fp64 independent", 0);
    run<F64_DEP, NONE>("one wave/SIMD: fp64 dependent chain", 0);
    run<F64_DEP2, NONE>("one wave/SIMD: fp64 2 interleaved chains", 0);
    run<F64_DEP4, NONE>("one wave/SIMD: fp64 4 interleaved chains", 0);
    run<I32_IND, NONE>("one wave/SIMD: int32 independent", 0);
    run<DPP_IND, NONE>("one wave/SIMD: dpp mov", 0);
    run<MIX_IND, NONE>("one wave/SIMD: fp64 / int32 alternating", 0);
    run<F64_IND, F64_IND>("two waves/SIMD: fp64 | fp64", 1);
    run<F64_IND, I32_IND>("two waves/SIMD: fp64 | int32", 1);
    run<I32_IND, I32_IND>("two waves/SIMD: int32 | int32", 1);
    run<MIX_IND, MIX_IND>("two waves/SIMD: mix | mix", 1);
    run<F64_DEP, F64_DEP>("two waves/SIMD: fp64 dep | fp64 dep", 1);
    run<F64_IND, DPP_IND>("two waves/SIMD: fp64 | dpp", 1);
    run<F64_DEP, I32_IND>("two waves/SIMD: fp64 dep | int32", 1);
    return 0;
}
