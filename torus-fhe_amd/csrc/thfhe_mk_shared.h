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

// ------------------------------------------------------------------------------------------------------
// key switch, staged variant (from 192 samples on; row of 512, 640 or 768 words, ks_basebit 2 or 3, t * ks_basebit <= 16): the single-key kernel of
// thfhe_sk.hip (sk_keyswitch_staged_kernel) per party.  A workgroup of eight waves takes 32 samples, one party and 128 (basebit 2) or 64
// (basebit 3) coordinates; it copies the rows KS[p][i][j][1 .. base-1] of four (two; with 768-word rows two (one)) consecutive (i, j) at a time into LDS -- contiguous in global
// memory, double buffered through registers -- and every lane reads its part of the row its sample's digit names (digit 0: a row of zeros)
// with ds_read_b128: the digit selects an address, not a branch, and base-1 rows per (i, j) leave L2 once for 32 samples instead of 0.75 .. 0.88
// rows per sample.  A wave takes four samples, one per 16-lane group of the LDS hardware, so a group reads 16 consecutive pieces of ONE row
// = every bank once.  Partial sums of the coordinate ranges and the parties' parts of b meet in the zeroed output with integer atomics.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mk_ks_sub(uint32_t &r, uint32_t x) { asm("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(x)); }   // in place, never re-associated
template <int W, int R, int SJ>   // W: 16-byte pieces per lane (row_words = 64 W); R = 2^basebit - 1 rows per (i, j); SJ: (i, j) pairs per stage
__global__ __launch_bounds__(512) void mk_keyswitch_staged_kernel(MKKSArgs a) {
    constexpr int ROW4 = 16 * W, Q = W, GW = 32;
    constexpr int SPAN = R == 3 ? 128 : 64;      // coordinates per workgroup
    constexpr int STAGE4 = SJ * R * ROW4;
    constexpr int NLD = (STAGE4 + 511) / 512;
    constexpr int KS_CHUNK = 3;
    static_assert(NLD <= 5, "a stage is at most five rounds of 512 pieces");
    __shared__ uint4 sL[ROW4 + 2 * STAGE4];      // [row of zeros][stage 0][stage 1]
    __shared__ uint16_t sDig[GW][SPAN];          // top 16 bits of u + offset: all t digits of a coordinate
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int q5 = lane & 31;
    const int gl = 2 * (lane >> 5) + (int)((0xF00F0FF0u >> q5) & 1u);   // the lane's ds_read_b128 group = its sample within the wave
    const int c = q5 < 4 ? q5 : q5 < 12 ? q5 - 4 : q5 < 20 ? q5 - 8 : q5 < 28 ? q5 - 12 : q5 - 16;   // position in the group: 0 .. 15
    const long g0 = (long)blockIdx.x * GW;
    const int p = blockIdx.y;
    const int first = (int)blockIdx.z * SPAN;
    const uint32_t prec_offset = 1u << (32 - (1 + a.basebit * a.t));
    for (int q = tid; q < GW * SPAN; q += 512) {
        const int g = q / SPAN, ii = q % SPAN;
        uint32_t v = 0;
        if (g0 + g < a.gates) v = (uint32_t)a.u[(size_t)(g0 + g) * a.u_rec + (size_t)p * a.u_pstride + first + ii] + prec_offset;
        sDig[g][ii] = (uint16_t)(v >> 16);  // absent samples: all digits zero
    }
    for (int q = tid; q < ROW4; q += 512) sL[q] = uint4{0u, 0u, 0u, 0u};
    const uint4 *src = reinterpret_cast<const uint4 *>(a.ksk) + ((size_t)p * a.N + first) * a.t * R * ROW4;
    const int NS = SPAN * a.t / SJ;
    uint4 pre0, pre1 = uint4{0u, 0u, 0u, 0u}, pre2 = pre1, pre3 = pre1, pre4 = pre1;
    const bool last_ok = 512 * NLD <= STAGE4 || tid + 512 * (NLD - 1) < STAGE4;
    const int last_idx = last_ok ? tid + 512 * (NLD - 1) : STAGE4 - 1;
#define MK_KS_GLOAD(st)                                           \
    {                                                             \
        const uint4 *p_ = src + (size_t)(st) * STAGE4;            \
        pre0 = p_[NLD == 1 ? last_idx : tid];                     \
        if (NLD > 1) pre1 = p_[NLD == 2 ? last_idx : tid + 512];  \
        if (NLD > 2) pre2 = p_[NLD == 3 ? last_idx : tid + 1024]; \
        if (NLD > 3) pre3 = p_[NLD == 4 ? last_idx : tid + 1536]; \
        if (NLD > 4) pre4 = p_[last_idx];                         \
    }
#define MK_KS_LSTORE(buf)                                         \
    {                                                             \
        uint4 *d_ = sL + ROW4 + (buf) * STAGE4 + tid;             \
        if (NLD > 1 || last_ok) d_[0] = pre0;                     \
        if (NLD > 2 || (NLD == 2 && last_ok)) d_[512] = pre1;     \
        if (NLD > 3 || (NLD == 3 && last_ok)) d_[1024] = pre2;    \
        if (NLD > 4 || (NLD == 4 && last_ok)) d_[1536] = pre3;    \
        if (NLD == 5 && last_ok) d_[2048] = pre4;                 \
    }
    MK_KS_GLOAD(0)
    MK_KS_LSTORE(0)
    __syncthreads();
    uint4 acc[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) acc[k] = uint4{0u, 0u, 0u, 0u};
    const uint16_t *dig = sDig[wave * 4 + gl];
    const uint32_t dmask = (uint32_t)R;
    int ii0 = 0, j0 = 0;   // coordinate and level of the stage's first pair
    for (int st = 0; st < NS; st++) {
        if (st + 1 < NS) {
            MK_KS_GLOAD(st + 1)
        }
        const uint4 *row[SJ];
#pragma unroll
        for (int pp = 0; pp < SJ; pp++) {
            int ii = ii0, j = j0 + pp;
            while (j >= a.t) j -= a.t, ii++;
            const uint32_t d = ((uint32_t)dig[ii] >> (16 - (j + 1) * a.basebit)) & dmask;
            row[pp] = sL + (d ? ROW4 + (st & 1) * STAGE4 + (pp * R + (int)d - 1) * ROW4 : 0) + c;
        }
        j0 += SJ;
        while (j0 >= a.t) j0 -= a.t, ii0++;
        constexpr int NCH = (Q + KS_CHUNK - 1) / KS_CHUNK;
        uint4 x[2][KS_CHUNK];
        auto reads = [&](int ch) {   // ch < SJ * NCH, compile-time after unrolling
            const uint4 *r = row[ch / NCH];
            const int k0 = (ch % NCH) * KS_CHUNK;
#pragma unroll
            for (int k = 0; k < KS_CHUNK; k++)
                if (k0 + k < Q) x[ch & 1][k] = r[16 * (k0 + k)];
        };
        reads(0);
#pragma unroll
        for (int ch = 0; ch < SJ * NCH; ch++) {
            if (ch + 1 < SJ * NCH) reads(ch + 1);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const int k0 = (ch % NCH) * KS_CHUNK;
#pragma unroll
            for (int k = 0; k < KS_CHUNK; k++)
                if (k0 + k < Q) {
                    uint4 &t = acc[k0 + k];
                    const uint4 v = x[ch & 1][k];
                    mk_ks_sub(t.x, v.x), mk_ks_sub(t.y, v.y), mk_ks_sub(t.z, v.z), mk_ks_sub(t.w, v.w);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (st + 1 < NS) {
            MK_KS_LSTORE((st + 1) & 1)
        }
        __syncthreads();
    }
#undef MK_KS_GLOAD
#undef MK_KS_LSTORE
    const long g = g0 + wave * 4 + gl;
    if (g < a.gates) {
        unsigned int *out = reinterpret_cast<unsigned int *>(a.out) + (size_t)g * ((size_t)a.parties * a.n + 1);
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const uint32_t v4[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int col = 4 * (c + 16 * k) + e;
                uint32_t v = v4[e];
                if (col < a.n) {
                    atomicAdd(out + (size_t)p * a.n + col, v);
                } else if (col == a.n) {
                    if (p == 0 && blockIdx.z == 0) v += (uint32_t)a.u[(size_t)g * a.u_rec + a.u_rec - 1];  // b = b' + sum over parties of the parts' b
                    atomicAdd(out + (size_t)a.parties * a.n, v);
                }
            }
        }
    }
}

// the key switch of `k.gates` extracted samples into the ZEROED k.out: staged kernel where its shape allows, else one workgroup per (sample, party, range)
inline bool mk_ks_staged_shape(const MKKSArgs &k) {
    const bool w = k.row_words == 512 || k.row_words == 640 || k.row_words == 768;
    const bool base = k.basebit == 2 || k.basebit == 3;
    return w && base && k.t >= 3 && k.t * k.basebit <= 16 && k.N % (k.basebit == 2 ? 128 : 64) == 0 && k.n < k.row_words;
}
inline void mk_launch_keyswitch(const MKKSArgs &k, int nsplit_plain, hipStream_t stream, long staged_min = 192) {
    if (k.gates >= staged_min && mk_ks_staged_shape(k)) {
        const int span = k.basebit == 2 ? 128 : 64;
        const dim3 grid((unsigned)((k.gates + 31) / 32), (unsigned)k.parties, (unsigned)(k.N / span)), block(512);
        // stage depth by LDS: two stages + the row of zeros + the digits stay under half a CU's LDS (two workgroups per CU)
        if (k.row_words == 512) {
            if (k.basebit == 2) hipLaunchKernelGGL((mk_keyswitch_staged_kernel<8, 3, 4>), grid, block, 0, stream, k);
            else hipLaunchKernelGGL((mk_keyswitch_staged_kernel<8, 7, 2>), grid, block, 0, stream, k);
        } else if (k.row_words == 640) {
            if (k.basebit == 2) hipLaunchKernelGGL((mk_keyswitch_staged_kernel<10, 3, 4>), grid, block, 0, stream, k);
            else hipLaunchKernelGGL((mk_keyswitch_staged_kernel<10, 7, 2>), grid, block, 0, stream, k);
        } else {
            if (k.basebit == 2) hipLaunchKernelGGL((mk_keyswitch_staged_kernel<12, 3, 2>), grid, block, 0, stream, k);
            else hipLaunchKernelGGL((mk_keyswitch_staged_kernel<12, 7, 1>), grid, block, 0, stream, k);
        }
        return;
    }
    hipLaunchKernelGGL(mk_keyswitch_kernel, dim3((unsigned)k.gates, (unsigned)k.parties, (unsigned)nsplit_plain), dim3(256), 0, stream, k, nsplit_plain);
}

}  // namespace

#endif  // THFHE_MK_SHARED_H
