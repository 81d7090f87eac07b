// thfhe_mk_shared.h -- kernels shared by the multi-key schemes (3-gen: thfhe_mk.hip, CCS: thfhe_ccs.hip): the gate's linear
// prologue + mod-switch on (n, P) mask matrices, and the per-party key switch with the cross-party combine of b.
#ifndef THFHE_MK_SHARED_H
#define THFHE_MK_SHARED_H

#include <hip/hip_runtime.h>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"
#include "thfhe_lane.h"

namespace {
using namespace thfhe;

// ------------------------------------------------------------------------------------------------------
// prologue: tmp = (0, cb) + cx x + cy y + cz z ; bara[job][P*n], barb[job]
// ------------------------------------------------------------------------------------------------------
struct MKLin {
    int32_t cb, cx, cy, cz;
};
__host__ __device__ inline bool mk_gate_lin(int op, int which, MKLin &L) {
    const int32_t E8 = 1 << 29, E4 = 1 << 30;
    switch (op) {
    case THFHE_NAND: L = MKLin{E8, -1, -1, 0}; return true;   // J/3gen_mk_gates.jl:8-14
    case THFHE_OR: L = MKLin{E8, 1, 1, 0}; return true;       // :24-30
    case THFHE_AND: L = MKLin{-E8, 1, 1, 0}; return true;     // :40-46
    case THFHE_XOR: L = MKLin{E4, 2, 2, 0}; return true;      // :68-74
    case THFHE_AND3: L = MKLin{-E4, 1, 1, 1}; return true;    // :55-64
    case THFHE_MUX: L = which == 0 ? MKLin{-E8, 1, 1, 0} : MKLin{-E8, -1, 0, 1}; return true;  // :133-150 (two ANDs)
    case kOpIdentity: L = MKLin{0, 1, 0, 0}; return true;
    default: return false;
    }
}
__global__ __launch_bounds__(256) void mk_prologue_kernel(const int32_t *__restrict__ in0, const int32_t *__restrict__ in1,
                                                           const int32_t *__restrict__ in2, MKLin L0, MKLin L1, const int32_t *__restrict__ ops,
                                                           int rot_per_gate, int words, int w_pad, int log2_2n, long jobs,
                                                           int32_t *__restrict__ bara, int32_t *__restrict__ barb) {
    const long job = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (job >= jobs || i > words) return;
    const long gate = job / rot_per_gate;
    MKLin L = (job % rot_per_gate) == 0 ? L0 : L1;
    if (ops) mk_gate_lin(ops[gate], 0, L);  // per-gate opcodes of a mixed DAG level (validated on the host)
    const size_t off = (size_t)gate * (words + 1) + i;
    uint32_t v = (uint32_t)L.cx * (uint32_t)in0[off];
    if (L.cy != 0) v += (uint32_t)L.cy * (uint32_t)in1[off];
    if (L.cz != 0) v += (uint32_t)L.cz * (uint32_t)in2[off];
    if (i == words) {
        v += (uint32_t)L.cb;
        barb[job] = modswitch2n((int32_t)v, log2_2n);
    } else {
        bara[job * w_pad + i] = modswitch2n((int32_t)v, log2_2n);
    }
}

// ------------------------------------------------------------------------------------------------------
// key switch: one workgroup per gate, parties in sequence          J/mk_internals.jl:730-744
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mk_ksk_pad_kernel(const int32_t *__restrict__ src, long rows, int n, int row_words,
                                                          int32_t *__restrict__ dst) {
    const long r = blockIdx.x;
    if (r >= rows) return;
    for (int q = threadIdx.x; q < row_words; q += 256) dst[r * row_words + q] = q <= n ? src[r * (n + 1) + q] : 0;
}

struct MKKSArgs {
    const int32_t *ksk;  // [P][N][t][base-1][row_words]
    const int32_t *u;    // [gates][N+1]
    int32_t *out;        // [gates][P*n+1]
    long gates;
    int n, t, basebit, parties, row_words;
    int N;  // ring degree = dimension of the extracted sample
    // extracted-sample layout: record stride u_rec words, party p reads mask words [p * u_pstride, p * u_pstride + N), b is the
    // last word.  3-gen: ONE mask for all parties (u_rec = N + 1, u_pstride = 0); CCS: one mask per party (u_rec = P N + 1, u_pstride = N)
    int u_rec, u_pstride;
};

// grid = (gates, parties, nsplit): block (g, p, s) key-switches coordinates [s*N/nsplit, (s+1)*N/nsplit) of gate g with
// party p's key and adds its partial sum into the zero-initialised output with integer atomics (order-independent: bit-exact)
__global__ __launch_bounds__(256) void mk_keyswitch_kernel(MKKSArgs a, int nsplit) {
    __shared__ uint32_t sA[2048];   // this block's slice of the extracted mask: N / nsplit <= 2048 words (N = 4096 is launched with nsplit >= 2)
    __shared__ uint32_t sRed[4][768];
    const long g = blockIdx.x;
    const int p = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t prec_offset = 1u << (32 - (1 + a.basebit * a.t));
    const int32_t *u = a.u + (size_t)g * a.u_rec + (size_t)p * a.u_pstride;
    const int i_lo = (int)blockIdx.z * (a.N / nsplit), i_hi = i_lo + a.N / nsplit;
    for (int q = i_lo + tid; q < i_hi; q += 256) sA[q - i_lo] = (uint32_t)u[q] + prec_offset;
    __syncthreads();
    const int base1 = (1 << a.basebit) - 1;
    const uint32_t mask = (uint32_t)base1;
    const int wpl = a.row_words / 64;  // words per lane (<= 12)
    unsigned int *out = reinterpret_cast<unsigned int *>(a.out) + (size_t)g * ((size_t)a.parties * a.n + 1);
    uint32_t r[12];
#pragma unroll
    for (int q = 0; q < 12; q++) r[q] = 0;
    const int32_t *kp = a.ksk + (size_t)p * a.N * a.t * base1 * a.row_words;
    for (int i = i_lo + wave; i < i_hi; i += 4) {
        const uint32_t ai = sA[i - i_lo];
        for (int j = 0; j < a.t; j++) {
            const uint32_t d = (ai >> (32 - (j + 1) * a.basebit)) & mask;
            if (d == 0) continue;
            const int32_t *row = kp + (((size_t)i * a.t + j) * base1 + (d - 1)) * a.row_words + 2 * lane;
#pragma unroll
            for (int q = 0; q < 6; q++)
                if (2 * q < wpl) {
                    const uint2 x = *reinterpret_cast<const uint2 *>(row + q * 128);
                    r[2 * q] -= x.x;
                    r[2 * q + 1] -= x.y;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < 6; q++)
        if (2 * q < wpl) {
            sRed[wave][q * 128 + 2 * lane] = r[2 * q];
            sRed[wave][q * 128 + 2 * lane + 1] = r[2 * q + 1];
        }
    __syncthreads();
    for (int q = tid; q <= a.n; q += 256) {
        uint32_t v = sRed[0][q] + sRed[1][q] + sRed[2][q] + sRed[3][q];
        if (q < a.n) {
            atomicAdd(out + (size_t)p * a.n + q, v);
        } else {
            if (p == 0 && blockIdx.z == 0) v += (uint32_t)a.u[(size_t)g * a.u_rec + a.u_rec - 1];  // b = b' + sum over parties of the parts' b
            atomicAdd(out + (size_t)a.parties * a.n, v);
        }
    }
}

}  // namespace

#endif  // THFHE_MK_SHARED_H
