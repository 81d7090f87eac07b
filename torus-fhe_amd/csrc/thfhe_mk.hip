// thfhe_mk.hip -- 3-gen multi-key (Torus64 ring) gate bootstrapping on gfx950: kernels + C ABI.
//
// Reference path: mk_gate_*_3gen (J/3gen_mk_gates.jl:8-150) -> mk_bootstrap_3gen (J/3gen_mk_internals.jl:112-116)
//   = mk_bootstrap_wo_keyswitch_3gen (:99-109): P*n sequential CMuxes on ONE Torus64 accumulator, party-major
//     (mk_blind_rotate_3gen :78-84, mk_mux_rotate_3gen :59-62, tgsw_extern_mul_3gen J/tgsw_3gen.jl:102-113),
//     rlwe_extract_sample_64 (J/rlwe.jl:70-74)
//   + mk_keyswitch_3gen (J/mk_internals.jl:730-744): one key switch per party, b = b' + sum of the parts' b.
//
// Kernels:
//   mk_key_transform_kernel   TransformedBootstrapKeyPart_3gen (J/3gen_mk_internals.jl:45-56): int64 coefficient
//                             polynomials -> four balanced 16-bit limbs -> FP64 spectra in streaming order
//   mk_prologue_kernel        gate linear part + mod-switch of the (n, P) mask matrix and of b
//   mk_blind_rotate_coop_kernel / _pair_kernel   one 512-thread workgroup per gate / per two gates: the eight (output polynomial, limb) spectra on
//                             eight waves; _coop2k / _pair2k on the ring of degree 2048 (two twisted half transforms per polynomial);
//                             thfhe_rot2k.h (any number of digit row parts, N = 2048) and thfhe_rot4k.h (N = 4096) for the large-party sets
//   mk_keyswitch_kernel / mk_keyswitch_staged_kernel (thfhe_mk_shared.h)   P key switches of the extracted sample + the cross-party combine
//                             of b: one workgroup per (sample, party, range), or from 192 samples on the rows staged in LDS for 32 samples
#include <hip/hip_runtime.h>

#include <type_traits>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"
#include "thfhe_dag.h"
#include "thfhe_lane.h"
#include "thfhe_mk_shared.h"

using namespace thfhe;

namespace {

// ------------------------------------------------------------------------------------------------------
// key transform: one wave per (pi, row, output) key polynomial, four limb spectra each
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mk_key_transform_kernel(const int64_t *__restrict__ bk, long PN, int l,
                                                                const cplx *__restrict__ tw, cplx *__restrict__ spec) {
    __shared__ cplx sT1[512];
    __shared__ cplx sX[4][512];
    for (int t = threadIdx.x; t < 512; t += 256) sT1[t] = tw[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const W64 w64{tw[512 + 1 * 8 + (lane & 7)]};
    const int rows = 2 * l;
    const long item = (long)blockIdx.x * 4 + wave;  // (pi, r, o)
    if (item >= PN * rows * 2) return;
    const int o = (int)(item & 1);
    const int r = (int)((item >> 1) % rows);
    const long pi = (item >> 1) / rows;
    const int j = r / l, lv = r % l;
    const int64_t *poly = bk + (((size_t)pi * 4 + mk_part_index(j, o)) * l + lv) * 1024;
    cplx z[4][8];
    key_limbs64_to_z(lane, poly, z);
#pragma unroll
    for (int h = 0; h < 4; h++) {
        wave_fft_fwd_s(lane, z[h], sX[wave], sT1, w64);
        cplx *dst = spec + mk_chunk_index(pi, r, h, o, rows) * 512;
#pragma unroll
        for (int m = 0; m < 8; m++) dst[m * 64 + lane] = cplx{z[h][m].re * (1.0 / 512), z[h][m].im * (1.0 / 512)};
    }
}

// wide gadget base: src [poly = (pi, part_q, level)][2048] -> dst [(pi, part_q, level * parts + w)][2048] = src << (pw w)
__global__ __launch_bounds__(256) void mk_expand_parts_kernel(const int64_t *__restrict__ src, int64_t *__restrict__ dst, int l, int parts, int pw) {
    const long poly = blockIdx.x;   // (pi * 4 + q) * l + level
    const int w = blockIdx.y;
    const long head = poly / l, level = poly % l;
    const int64_t *s = src + poly * 2048;
    int64_t *d = dst + ((head * l + level) * parts + w) * 2048;
    for (int t = threadIdx.x; t < 2048; t += 256) d[t] = (int64_t)((uint64_t)s[t] << (pw * w));
}

// N = 2048: one wave per (pi, row, output, limb); two twisted 512-point spectra (even / odd outputs) per item, scaled by 1/1024
__global__ __launch_bounds__(256) void mk_key_transform_2k_kernel(const int64_t *__restrict__ bk, long PN, int l,
                                                                   const cplx *__restrict__ tw, cplx *__restrict__ spec) {
    __shared__ cplx sT1[2][512];
    __shared__ cplx sX[4][512];
    for (int t = threadIdx.x; t < 1024; t += 256) (&sT1[0][0])[t] = tw[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const W64 w64{tw[1024 + 1 * 8 + (lane & 7)]};
    const int rows = 2 * l;
    const long item = (long)blockIdx.x * 4 + wave;  // (pi, r, o, h)
    if (item >= PN * rows * 8) return;
    const int h = (int)(item & 3), o = (int)((item >> 2) & 1);
    const int r = (int)((item >> 3) % rows);
    const long pi = (item >> 3) / rows;
    const int j = r / l, lv = r % l;
    const int64_t *poly = bk + (((size_t)pi * 4 + mk_part_index(j, o)) * l + lv) * 2048;
    cplx z[16], y0[8], y1[8];
    key_limbs64_to_z16(lane, poly, h, z);
    split2048(z, y0, y1);
    wave_fft_fwd_t<1>(lane, y0, sX[wave], sT1[0], w64);
    wave_fft_fwd_t<5>(lane, y1, sX[wave], sT1[1], w64);
    cplx *dst = spec + mk_chunk_index_2k(pi, r, h, o, rows) * 512;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        dst[m * 64 + lane] = cplx{y0[m].re * (1.0 / 1024), y0[m].im * (1.0 / 1024)};
        dst[512 + m * 64 + lane] = cplx{y1[m].re * (1.0 / 1024), y1[m].im * (1.0 / 1024)};
    }
}

// ------------------------------------------------------------------------------------------------------
// blind rotate + extract
// ------------------------------------------------------------------------------------------------------
struct MKBRArgs {
    const cplx *bk;
    const cplx *tw;
    const int32_t *bara;  // [jobs][w_pad]
    const int32_t *barb;
    int32_t *out;         // [jobs][N+1]
    long jobs;
    int pn;               // parties * n : number of CMuxes
    int w_pad, Bgbit;
    int64_t mu;
    // party-sharded mode (one rank per party): the accumulator enters from / leaves to global memory instead of being
    // initialised / extracted here.  acc_in == nullptr: start from X^{-barb} * mu; acc_out == nullptr: extract into `out`.
    const int64_t *acc_in = nullptr;
    int64_t *acc_out = nullptr;
    // N = 2048, wide gadget base (the 16+-party sets: l = 1, Bgbit 24 .. 26): every digit is cut into `parts` balanced parts of `pw` bits,
    // d = sum_w d_w 2^(pw w); part w multiplies the key row shifted left by pw w bits (a second / third copy in the key table)
    int parts = 1, pw = 0;
};

// ------------------------------------------------------------------------------------------------------
// blind rotate + extract: one 512-thread workgroup = ONE gate.  The eight (output, limb)
// spectra of a 3-gen external product map onto the eight waves:
//   phase 1  waves 0 .. 2l-1: wave r rotates / decomposes / transforms digit row r and publishes the spectrum in LDS;
//            then every wave requests its key chunks (2l x 8 loads of 16 B per lane, into registers);
//   phase 2  all waves, (o, h) = (wave >> 2, wave & 3): S = sum_r spectrum_r * key(r, h, o); inverse transform;
//            round(S) << 16h is added into accumulator polynomial o with 64-bit LDS atomics (integer adds commute: bit-exact).
// Two workgroup barriers per CMux and no redundant forward transforms.  (An LDS-ring variant in the style of
// sk_blind_rotate_ring_kernel -- 4 gates x 2 output waves per workgroup -- was built and measured 1.3x slower at 1024 gates
// and 3.5x slower for a handful: 9 barriers per digit row, redundant forward transforms; see git history / DESIGN.md 4.3.)
// LDS: T1 8 + acc 16 + spectra 2l x 8 + 8 transpose buffers x 8 KiB (l = 3: 136 KiB).
// ------------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(512, 2) void mk_blind_rotate_coop_kernel(MKBRArgs a) {
    constexpr int ROWS = 2 * L;
    __shared__ cplx sT1[512];
    __shared__ int64_t sAcc[2048];
    __shared__ cplx sSpec[ROWS][512];
    __shared__ cplx sX[8][512];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    sT1[threadIdx.x] = a.tw[threadIdx.x];
    const W64 w64{a.tw[512 + 1 * 8 + (lane & 7)]};
    const long job = blockIdx.x;
    const int32_t *bara = a.bara + job * a.w_pad;
    const int Bgbit = a.Bgbit;
    const uint64_t offset = decomp_offset64(L, Bgbit);
    if (a.acc_in) {
        for (int q = threadIdx.x; q < 2048; q += 512) sAcc[q] = a.acc_in[job * 2048 + q];
    } else if (wave == 0) {
        acc_init16_64(lane, sAcc, sAcc + 1024, a.barb[job], a.mu);
    }
    __syncthreads();
    const int o = wave >> 2, h = wave & 3;
    unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc) + o * 1024;

    for (int i = 0; i < a.pn; i++) {  // party-major, key index inner: J/3gen_mk_internals.jl:66-84
        const int ai = bara[i];       // uniform over the workgroup
        if (ai == 0) continue;
        const int a2n = ai & 2047;
        if (wave < ROWS) {
            uint32_t t[16];
            cplx z[8];
            load_rotated16_hi(lane, sAcc + (wave / L) * 1024, a2n, offset, t);
            digits_to_z(t, (wave % L) + 1, Bgbit, z);
            wave_fft_fwd_s(opaque_lane(lane), z, sX[wave], sT1, w64);
#pragma unroll
            for (int m = 0; m < 8; m++) sSpec[wave][m * 64 + lane] = z[m];
        }
        // key rows in registers: all 2l of them up to l = 3 (192 VGPRs); from l = 4 on (the 8-party set: 8 rows = 256 VGPRs, which
        // spilled 332 B/lane) a window of PRE rows, refilled as the multiply walks the rows
        constexpr int PRE = ROWS <= 6 ? ROWS : 4;
        cplx B[PRE][8];
#pragma unroll
        for (int r = 0; r < PRE; r++) load8(lane, B[r], a.bk + mk_chunk_index(i, r, h, o, ROWS) * 512);
        __syncthreads();  // spectra published; every rotated read of the accumulator is done
        cplx S[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S[m] = cplx{0.0, 0.0};
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            cplx z[8];
#pragma unroll
            for (int m = 0; m < 8; m++) z[m] = sSpec[r][m * 64 + lane];
            mac8r(S, z, B[r % PRE]);
            if (r + PRE < ROWS) load8(lane, B[r % PRE], a.bk + mk_chunk_index(i, r + PRE, h, o, ROWS) * 512);
        }
        wave_fft_inv_s(opaque_lane(lane), S, sX[wave], sT1, w64);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int q = lane + 64 * m;
            atomicAdd(accu + q, (unsigned long long)round_i64(S[m].re) << (16 * h));
            atomicAdd(accu + q + 512, (unsigned long long)round_i64(S[m].im) << (16 * h));
        }
        __syncthreads();  // accumulator updated before anybody rotates it again
    }
    if (a.acc_out) {
        for (int q = threadIdx.x; q < 2048; q += 512) a.acc_out[job * 2048 + q] = sAcc[q];
    } else if (wave == 0) {
        extract16_64(lane, sAcc, sAcc + 1024, a.out + job * 1025);
    }
}

// ------------------------------------------------------------------------------------------------------
// Throughput kernel: one 512-thread workgroup = TWO gates.  The cooperative kernel above is bound by the key bytes a CU
// has to pull through its vector-memory path (2l x 4 limbs x 2 outputs x 8 KiB = 256 KiB per CMux at l = 2, ~13 k cycles at the
// measured ~20 B/clk/CU, against ~5 k cycles of FP64 work).  Here every (output, limb) wave multiplies its key chunks into the
// spectra of two gates, so each key byte serves two CMuxes; the 2 x 2l forward transforms of a step keep all eight waves busy in
// phase 1 (l = 2: one each), and the transpose scratch aliases the spectrum area (a third barrier frees it for the inverses), which is
// what lets two accumulators and two sets of spectra fit: acc 32 + spectra 2 x 2l x 8 KiB (l = 3: 128 KiB).  Transforms are variant "qs"
// (first transpose in registers, pass-1 twiddles from per-lane roots: no T1 table, one LDS crossing per transform); measured on MI355X the
// three ways of holding the pass-1 twiddles give MK2 39.5 k (hoisted by the compiler, 88 B/lane scratch) / 45.8 k (LaneTw, 48 B) /
// 52.2 k gates/s (rebuilt per transform, 8 B): with four key rows carried across the loop the registers are worth more than 28 FP64
// instructions per transform.
// ------------------------------------------------------------------------------------------------------
// mk_pin: memory operations do not move across this point (keeps the paced key requests where they are written)
__device__ __forceinline__ void mk_pin2() { asm volatile("" ::: "memory"); }
// inverse transform of one partial spectrum with the requests for two key rows of the NEXT step between its stages
#ifndef THFHE_MK_LEAN_ROOTS
#define THFHE_MK_LEAN_ROOTS 2
#endif
// pass-1 twiddles of the "qs" transforms: 0 = the compiler may hoist all eight products b s^k out of the CMux loop (32 VGPRs),
// 1 = only the even ones live across the loop (LaneTw, 20 VGPRs), 2 = everything is rebuilt per transform from the two roots (8 VGPRs)
#if THFHE_MK_LEAN_ROOTS == 1
typedef LaneTw MK_ROOTS;
__device__ __forceinline__ MK_ROOTS mk_make_roots(const LaneRoots &r) { return make_lane_tw(r); }
__device__ __forceinline__ const MK_ROOTS &mk_use_roots(const MK_ROOTS &r) { return r; }
#elif THFHE_MK_LEAN_ROOTS == 2
typedef LaneRoots MK_ROOTS;
__device__ __forceinline__ MK_ROOTS mk_make_roots(const LaneRoots &r) { return r; }
__device__ __forceinline__ MK_ROOTS mk_use_roots(const MK_ROOTS &r) { return LaneRoots{opaque_cplx(r.b), opaque_cplx(r.s)}; }
#else
typedef LaneRoots MK_ROOTS;
__device__ __forceinline__ MK_ROOTS mk_make_roots(const LaneRoots &r) { return r; }
__device__ __forceinline__ const MK_ROOTS &mk_use_roots(const MK_ROOTS &r) { return r; }
#endif
template <bool PF0, bool PF1>
__device__ __forceinline__ void inv_s_prefetch(int lane, cplx (&S)[8], cplx *xb, const MK_ROOTS &roots, const W64 &w, cplx (&b0)[8], const cplx *src0, cplx (&b1)[8],
                                               const cplx *src1) {   // wave_fft_inv_qs with the requests in between
    wave_sync();
    invs_seg1(lane, S, xb, w);
    mk_pin2();
    if (PF0) load8(lane, b0, src0);
    mk_pin2();
    wave_sync();
    invs_seg2_ld(lane, S, xb);
    dft8<-1>(S);
    mk_pin2();
    if (PF1) load8(lane, b1, src1);
    mk_pin2();
    wave_transpose_hi3(S);
    invq_seg3(S, mk_use_roots(roots));
}
template <int L>
__global__ __launch_bounds__(512, 2) void mk_blind_rotate_pair_kernel(MKBRArgs a) {
    constexpr int ROWS = 2 * L;
    constexpr int FFTS = 2 * ROWS;
    constexpr int SPEC_SLOTS = (FFTS > 8 ? FFTS : 8) * 512;
    constexpr int PRE = ROWS <= 4 ? ROWS : 3;  // key rows in flight (32 VGPRs each); measured at l = 3: 2 rows 19.9 k, 3 rows 20.6 k gates/s (MK4), 4 rows spill 168 B/lane
    __shared__ int64_t sAcc[2][2048];
    __shared__ cplx sSpec[SPEC_SLOTS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const W64 w64{a.tw[512 + 1 * 8 + (lane & 7)]};
    const LaneRoots roots0{a.tw[1088 + 2 * lane], a.tw[1088 + 2 * lane + 1]};
    const MK_ROOTS roots = mk_make_roots(roots0);
    const long job0 = 2 * (long)blockIdx.x;
    const bool has1 = job0 + 1 < a.jobs;
    const int32_t *bara0 = a.bara + job0 * a.w_pad;
    const int32_t *bara1 = bara0 + (has1 ? a.w_pad : 0);
    const int Bgbit = a.Bgbit;
    const uint64_t offset = decomp_offset64(L, Bgbit);
    if (a.acc_in) {   // party-sharded mode: the accumulators of the party block before this one (J/3gen_mk_internals.jl:78-84)
        for (int q = threadIdx.x; q < 4096; q += 512) {
            const int g = q >> 11;
            if (g == 0 || has1) sAcc[g][q & 2047] = a.acc_in[(job0 + g) * 2048 + (q & 2047)];
        }
    } else if (wave < 2 && (wave == 0 || has1)) {
        acc_init16_64(lane, sAcc[wave], sAcc[wave] + 1024, a.barb[job0 + wave], a.mu);
    }
    __syncthreads();
    const int o = wave >> 2, h = wave & 3;

    // The first PRE key rows of a step are requested during the inverse transforms of the step before (the registers that held this
    // step's rows are dead by then), between the stages of the transforms: the 8 PRE loads per wave stream in under ~8 k cycles of
    // compute instead of queueing up in front of the multiply (256 KiB per workgroup and step at l = 2 are ~6 k cycles of the CU's
    // vector-memory path).  Unconditional loads: the last step re-requests its own rows.
    auto active = [&](int i) { return bara0[i] != 0 || (has1 && bara1[i] != 0); };   // uniform over the workgroup
    int i = 0;
    while (i < a.pn && !active(i)) i++;
    cplx B[PRE][8];
    if (i < a.pn) {
#pragma unroll
        for (int r = 0; r < PRE; r++) load8(lane, B[r], a.bk + mk_chunk_index(i, r, h, o, ROWS) * 512);
    }
    while (i < a.pn) {
        const int ai0 = bara0[i], ai1 = has1 ? bara1[i] : 0;
        int inext = i + 1;
        while (inext < a.pn && !active(inext)) inext++;
        const int inl = inext < a.pn ? inext : i;
#pragma unroll
        for (int f0 = 0; f0 < FFTS; f0 += 8) {
            const int f = f0 + wave;  // forward-transform task: gate f / ROWS, digit row f % ROWS
            if (f < FFTS) {
                const int g = f / ROWS, r = f % ROWS;
                const int ai = g ? ai1 : ai0;
                if (ai != 0) {
                    uint32_t t[16];
                    cplx z[8];
                    load_rotated16_hi(lane, sAcc[g] + (r / L) * 1024, ai & 2047, offset, t);
                    digits_to_z(t, (r % L) + 1, Bgbit, z);
                    cplx *slot = sSpec + f * 512;  // transposes run inside the task's own, not yet published, spectrum slot
                    wave_fft_fwd_qs(opaque_lane(lane), z, slot, mk_use_roots(roots), w64);
                    wave_sync();
#pragma unroll
                    for (int m = 0; m < 8; m++) slot[m * 64 + lane] = z[m];
                }
            }
        }
        __syncthreads();  // spectra published; every rotated read of the accumulators is done
        cplx S0[8], S1[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S0[m] = S1[m] = cplx{0.0, 0.0};
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            cplx z[8];
            if (ai0 != 0) {
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[r * 512 + m * 64 + lane];
                mac8r(S0, z, B[r % PRE]);
            }
            if (ai1 != 0) {
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[(ROWS + r) * 512 + m * 64 + lane];
                mac8r(S1, z, B[r % PRE]);
            }
            mk_pin2();
            if (r + PRE < ROWS) load8(lane, B[r % PRE], a.bk + mk_chunk_index(i, r + PRE, h, o, ROWS) * 512);
            mk_pin2();
        }
        __syncthreads();  // spectra consumed: the area is transpose scratch from here on
        cplx *xb = sSpec + wave * 512;
        const int ln = opaque_lane(lane);
        const cplx *nx[4];
#pragma unroll
        for (int r = 0; r < 4; r++) nx[r] = a.bk + mk_chunk_index(inl, r < PRE ? r : 0, h, o, ROWS) * 512;
        // gate 0's transform carries the requests for rows 0 / 1 of the next step, gate 1's those for rows 2 / 3 (PRE = 4); a skipped gate's
        // requests are issued plainly so that B is complete whichever gates are active
        if (ai0 != 0) {
            unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc[0]) + o * 1024;
            inv_s_prefetch<true, (PRE > 1)>(ln, S0, xb, roots, w64, B[0], nx[0], B[PRE > 1 ? 1 : 0], nx[1]);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int q = lane + 64 * m;
                atomicAdd(accu + q, (unsigned long long)round_i64(S0[m].re) << (16 * h));
                atomicAdd(accu + q + 512, (unsigned long long)round_i64(S0[m].im) << (16 * h));
            }
        } else {
            load8(lane, B[0], nx[0]);
            if (PRE > 1) load8(lane, B[PRE > 1 ? 1 : 0], nx[1]);
        }
        if (ai1 != 0) {
            unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc[1]) + o * 1024;
            inv_s_prefetch<(PRE > 2), (PRE > 3)>(ln, S1, xb, roots, w64, B[PRE > 2 ? 2 : 0], nx[2], B[PRE > 3 ? 3 : 0], nx[3]);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int q = lane + 64 * m;
                atomicAdd(accu + q, (unsigned long long)round_i64(S1[m].re) << (16 * h));
                atomicAdd(accu + q + 512, (unsigned long long)round_i64(S1[m].im) << (16 * h));
            }
        } else {
            if (PRE > 2) load8(lane, B[PRE > 2 ? 2 : 0], nx[2]);
            if (PRE > 3) load8(lane, B[PRE > 3 ? 3 : 0], nx[3]);
        }
        __syncthreads();  // accumulators updated and scratch free before the next step
        i = inext;
    }
    if (a.acc_out) {
        for (int q = threadIdx.x; q < 4096; q += 512) {
            const int g = q >> 11;
            if (g == 0 || has1) a.acc_out[(job0 + g) * 2048 + (q & 2047)] = sAcc[g][q & 2047];
        }
    } else if (wave < 2 && (wave == 0 || has1)) {
        extract16_64(lane, sAcc[wave], sAcc[wave] + 1024, a.out + (job0 + wave) * 1025);
    }
}

// ------------------------------------------------------------------------------------------------------
// N = 2048 (BASELINE config 5).  Same wave roles as above; every 1024-point transform is a radix-2 split and two twisted
// 512-point transforms (thfhe_lane.h), so a digit row publishes TWO half spectra (16 KiB) and a phase-2 wave keeps two
// partial spectra S0 / S1.  LDS at l = 3: accumulator 32 + spectra 96 = 128 KiB (table-free "qs" transforms), so the transpose buffers
// alias the spectrum area: in phase 1 a forward unit transposes inside its own (not yet published) half-spectrum slot, in phase 2 a third
// barrier frees the whole area before the eight waves' inverse transform pairs use 8 KiB of it each.
// ------------------------------------------------------------------------------------------------------
THFHE_STAMP_STORAGE
__device__ __forceinline__ void mk_pin() { asm volatile("" ::: "memory"); }  // memory operations do not move across this point
__device__ __forceinline__ void lds_barrier_any() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // workgroup barrier ordering LDS traffic only

template <int LE>   // LE = l x parts: digit rows per accumulator polynomial
__global__ __launch_bounds__(512, 2) void mk_blind_rotate_coop2k_kernel(MKBRArgs a) {
    constexpr int ROWS = 2 * LE;
    const int parts = a.parts > 1 ? a.parts : 1, pw = a.pw;
    const int L = LE / parts;   // decomposition levels
    constexpr int SPEC_SLOTS = ROWS * 1024 > 8 * 512 ? ROWS * 1024 : 8 * 512;
    __shared__ int64_t sAcc[4096];
    __shared__ cplx sSpec[SPEC_SLOTS];
    __shared__ cplx sXb[4 * 512];   // transpose scratch of waves 0 - 3 (the last 32 KiB of the LDS)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // per-lane transform constants are phase-local (L1 / L2 hits at the start of a transform phase), not 10 VGPRs alive across the multiply
    auto tw_w64 = [&](int ln) { return W64{a.tw[1024 + 1 * 8 + (ln & 7)]}; };
    auto tw_roots1 = [&](int ln) { return LaneRoots{opaque_cplx(a.tw[ln]), opaque_cplx(a.tw[1216 + ln])}; };   // b_T = T1_T[0][lane], pass-1 ratio
    auto tw_roots5 = [&](int ln) { return LaneRoots{opaque_cplx(a.tw[512 + ln]), opaque_cplx(a.tw[1216 + ln])}; };
    const long job = blockIdx.x;
    const int32_t *bara = a.bara + job * a.w_pad;
    const int Bgbit = a.Bgbit;
    const uint64_t offset = decomp_offset64(L, Bgbit);
    if (a.acc_in) {
        for (int q = threadIdx.x; q < 4096; q += 512) sAcc[q] = a.acc_in[job * 4096 + q];
    } else if (wave == 0) {
        acc_init_64_n<2048>(lane, sAcc, sAcc + 2048, a.barb[job], a.mu);
    }
    __syncthreads();
    const int o = wave >> 2, h = wave & 3;
    unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc) + o * 2048;

    // The key stream of a step is 2l rows x 2 half spectra = 4l chunks of 8 KiB per wave (l = 3: 96 KiB per wave, 768 KiB per workgroup
    // and CMux -- the CU's vector-memory path carries ~45 B/clk, so it is busy for a good part of every step).  The chunks are
    // requested two ahead of their use (registers bA / bB), and the first two chunks of the NEXT step are requested before this
    // step's inverse transforms, so the memory pipeline never waits for the compute phases and the multiply never waits for a
    // round trip of its own.  Compiler fences (mk_pin) keep the requests where they are written.
    // The step loop exists twice, for waves 0 - 3 (EARLY) and 4 - 7: the same phases with the "spectra consumed" barrier behind / in front of
    // the inverse transforms.  One loop with the barrier under a wave test made the register allocator spill 200 - 2 500 B per lane.
    auto steps = [&](auto early_tag) {
    constexpr bool EARLY = decltype(early_tag)::value;
    int i = 0;
    while (i < a.pn && bara[i] == 0) i++;
    cplx bA[8], bB[8];
    auto chunk = [&](int step, int u) { return a.bk + mk_chunk_index_2k(step, u >> 1, h, o, ROWS) * 512 + (u & 1) * 512; };  // u = 2 r + half
    if (i < a.pn) {
        load8(lane, bA, chunk(i, 0));
        load8(lane, bB, chunk(i, 1));
    }
    STAMP_DECL;
    while (i < a.pn) {
        const int a2n = bara[i] & 4095;
        int inext = i + 1;
        while (inext < a.pn && bara[inext] == 0) inext++;
        const int inl = inext < a.pn ? inext : i;   // the last step re-requests its own chunks (unconditional loads: one register set)
        // forward units (digit row, half): 2 ROWS of them on eight waves.  ROWS = 6: waves 0 - 3 take both halves of rows 0 - 3 (two transforms side
        // by side), waves 4 - 7 one half of rows 4 / 5 each -- three units per SIMD (wave w and w + 4 share one); ROWS < 6: one unit per wave
        const bool two = ROWS == 6 && wave < 4;
        const int row = ROWS == 6 ? (wave < 4 ? wave : 4 + ((wave - 4) >> 1)) : (wave >> 1);
        const int half_of = wave & 1;
        if (ROWS == 6 || wave < 2 * ROWS) {
            cplx y0[8], y1[8];   // a one-unit wave forms its half in y0
            {
                // digits of the four coefficients (j, j + 512, j + 1024, j + 1536) that make one (y0[m], y1[m]) pair: no 32-word t[],
                // no 16-point z[] alive next to the two half transforms
                const int64_t *ap = sAcc + (row / LE) * 2048;
                const int level = (row % LE) / parts, part = (row % LE) % parts;   // uniform per wave
                const int shift = 32 - (level + 1) * Bgbit;
                const uint32_t mask = (1u << Bgbit) - 1u;
                const int32_t half = 1 << (Bgbit - 1);
                const int32_t hp = pw ? 1 << (pw - 1) : 0, mp = (1 << pw) - 1;
                constexpr double R = 0.70710678118654752440;
                const double sg = half_of ? -1.0 : 1.0;
                // all 32 rotated words first (their LDS reads in flight together), then the branch-free digit arithmetic: with the loop over the
                // parts inside the unrolled body every rotated read was waited for on its own (s_waitcnt lgkmcnt(0) + a uniform branch per coefficient)
#pragma unroll
                for (int m0 = 0; m0 < 8; m0 += 4) {   // in two halves: 16 words alive next to the points already formed (all 32 spilled 56 B per lane)
                uint32_t t[4][4];
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int q = 0; q < 4; q++) t[m][q] = (uint32_t)((rot_minus_self64_n<2048>(ap, lane + 64 * (m0 + m) + 512 * q, a2n) + offset) >> 32);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int mm = 0; mm < 4; mm++) {
                    const int m = m0 + mm;
                    double d[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        int32_t dg = (int32_t)((t[mm][q] >> shift) & mask) - half;      // decompose, J/tgsw.jl:112-138
                        if (parts > 1) {   // balanced parts, least significant first; at most three (l x parts <= 3); uniform selects, no branches
                            const int32_t lo0 = ((dg + hp) & mp) - hp, v1 = (dg - lo0) >> pw;
                            const int32_t lo1 = ((v1 + hp) & mp) - hp, v2 = (v1 - lo1) >> pw;
                            const int32_t p1 = parts > 2 ? lo1 : v1;
                            dg = part == 0 ? lo0 : (part == 1 ? p1 : v2);
                        }
                        d[q] = (double)dg;
                    }
                    // z[m] = (d0, d2) (coefficients j, j + 1024), z[m + 8] = (d1, d3); split2048: y0/1 = z[m] +- e^{i pi/4} z[m + 8]
                    const cplx w{(d[1] - d[3]) * R, (d[1] + d[3]) * R};
                    if (two) {
                        y0[m] = cplx{d[0] + w.re, d[2] + w.im};
                        y1[m] = cplx{d[0] - w.re, d[2] - w.im};
                    } else {
                        y0[m] = cplx{d[0] + sg * w.re, d[2] + sg * w.im};   // one rounding, like the sum / difference it stands for
                    }
                }
                }
            }
            cplx *xb = sSpec + row * 1024;
            const int ln = opaque_lane(lane);   // the swizzled LDS slot maps are recomputed here, not hoisted out of the CMux loop and spilled
            const W64 w64 = tw_w64(ln);
            if (two) {
                wave_fft_fwd_tq_two<1, 5>(ln, y0, y1, xb, xb + 512, tw_roots1(ln), tw_roots5(ln), w64);
                wave_sync();
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    xb[m * 64 + ln] = y0[m];
                    xb[512 + m * 64 + ln] = y1[m];
                }
            } else {
                cplx *slot = xb + half_of * 512;
                if (half_of == 0) wave_fft_fwd_tq<1>(ln, y0, slot, tw_roots1(ln), w64);
                else wave_fft_fwd_tq<5>(ln, y0, slot, tw_roots5(ln), w64);
                wave_sync();
#pragma unroll
                for (int m = 0; m < 8; m++) slot[m * 64 + ln] = y0[m];
            }
        }
        STAMP(0);
        __syncthreads();  // spectra published; every rotated read of the accumulator is done
        STAMP(1);
        cplx S0[8], S1[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S0[m] = S1[m] = cplx{0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 2 * ROWS; u += 2) {
            cplx z[8];
#pragma unroll
            for (int m = 0; m < 8; m++) z[m] = sSpec[(u >> 1) * 1024 + m * 64 + lane];
            mac8r(S0, z, bA);
            mk_pin();
            load8(lane, bA, u + 2 < 2 * ROWS ? chunk(i, u + 2) : chunk(inl, 0));
            mk_pin();
#pragma unroll
            for (int m = 0; m < 8; m++) z[m] = sSpec[(u >> 1) * 1024 + 512 + m * 64 + lane];
            mac8r(S1, z, bB);
            mk_pin();
            load8(lane, bB, u + 2 < 2 * ROWS ? chunk(i, u + 3) : chunk(inl, 1));
            mk_pin();
        }
        STAMP(2);
        auto finish = [&](cplx *xb) {   // inverse transform pair, radix-2 merge, round(S) << 16h into accumulator polynomial o
            const int ln = opaque_lane(lane);
            wave_fft_inv_tq_two<1, 5>(ln, S0, S1, xb, tw_roots1(ln), tw_roots5(ln), tw_w64(ln));
            cplx lo[8], hi[8];
            merge2048(S0, S1, lo, hi);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int q = lane + 64 * m;
                atomicAdd(accu + q, (unsigned long long)round_i64(lo[m].re) << (16 * h));
                atomicAdd(accu + q + 512, (unsigned long long)round_i64(hi[m].re) << (16 * h));
                atomicAdd(accu + q + 1024, (unsigned long long)round_i64(lo[m].im) << (16 * h));
                atomicAdd(accu + q + 1536, (unsigned long long)round_i64(hi[m].im) << (16 * h));
            }
        };
        // The older wave of every SIMD (0 - 3) is served first by the key stream and leaves the multiply ~6 k cycles before its partner.  It owns
        // 8 KiB of transpose scratch outside the spectrum area, so it does not wait for the others to finish reading the spectra: it
        // transforms at once and ARRIVES at that barrier afterwards, about when the younger waves get there from their multiply; those then
        // transform with a SIMD to themselves.  (The accumulator atomics touch nothing the multiply reads.)
        if (EARLY) {
            STAMP(4);
            finish(sXb + wave * 512);
            STAMP(3);
            __syncthreads();  // spectra consumed (reached after the transforms)
        } else {
            __syncthreads();  // spectra consumed: the area is transpose scratch from here on
            STAMP(4);
            finish(sSpec + wave * 512);
            STAMP(3);
        }
        __syncthreads();  // accumulator updated and scratch free before the next rotation
        STAMP(5);
        i = inext;
    }
    STAMP_FLUSH(blockIdx.x, wave);
    };
    if (wave < 4) steps(std::true_type{});
    else steps(std::false_type{});
    if (a.acc_out) {
        for (int q = threadIdx.x; q < 4096; q += 512) a.acc_out[job * 4096 + q] = sAcc[q];
    } else if (wave == 0) {
        extract_64_n<2048>(lane, sAcc, sAcc + 2048, a.out + job * 2049);
    }
}

// the batched N = 2048 rotation (any number of row parts, two-part digits) shared with the KMS scheme
#include "thfhe_rot2k.h"
// the ring of degree 4096 (the 64-party "for fft" and the 512-party 3-gen sets)
#include "thfhe_rot4k.h"

// ------------------------------------------------------------------------------------------------------
// N = 2048 throughput kernel: one 512-thread workgroup = TWO gates (batches above one gate per CU).  The one-gate kernel above spends half
// of every step pulling 768 KiB of key spectra (l = 3) through the CU's vector-memory path; here every key chunk multiplies the digit
// spectra of two gates.  Two accumulators (64 KiB) leave 96 KiB for spectra -- the 2 x 2l HALF spectra of one twist -- so a step runs as
// two half passes: forward transforms of the even-output halves (twist 1) of all rows of both gates, multiply into S0, the same for the
// odd-output halves (twist 5, digits extracted again: the accumulators do not change inside a step) into S1, then the four inverse
// transforms, the radix-2 merge and the integer atomics.  Transforms are the "qs" form of the twisted halves (no T1 tables in LDS, one
// LDS crossing); the barriers wait for the LDS only (vmcnt(16)), so the two key chunks a wave has requested stay in flight across them:
// the key stream never stops at a phase boundary.  LDS at l = 3: 64 + 96 = 160 KiB.
// ------------------------------------------------------------------------------------------------------
struct P2KDigits {
    int parts, pw, L, Bgbit;
    uint64_t offset;
};
// Digits of a step, computed ONCE for both half passes and all levels:
//   stage   every wave takes half of one (gate, accumulator polynomial): t = top 32 bits of (X^a acc - acc) + offset for 16 coefficients per
//           lane -> 8 KiB of 32-bit words per (gate, polynomial) in the spectrum area (slots 0..3, free at this point);
//   pack    task f = (gate f / ROWS, digit row f % ROWS) reads the 32 words of the coefficients its lane transforms, cuts out its digit
//           (level, part) and keeps the four digits of a radix-2 group as 16-bit fields of two registers: 16 VGPRs per task carry the
//           digits through both half passes (the rotated 64-bit reads + offset + decomposition cost more than a half transform).
__device__ __forceinline__ void p2k_stage(int wave, int lane, const int64_t (*sAcc)[4096], uint32_t *stage, int ai0, int ai1, uint64_t offset) {
    const int gj = wave >> 1, g = gj >> 1, j = gj & 1;
    const int ai = g ? ai1 : ai0;
    if (ai == 0) return;
    const int a2n = ai & 4095;
    const int64_t *ap = sAcc[g] + j * 2048;
    uint32_t *dst = stage + gj * 2048;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int c = (wave & 1) * 1024 + lane + 64 * k;
        dst[c] = (uint32_t)((rot_minus_self64_n<2048>(ap, c, a2n) + offset) >> 32);
    }
}
// ONE_PART: the gadget digit is used whole (parts = 1: every set with Bgbit <= 10, e.g. BASELINE configs[4]) -- the per-coefficient part loops and
// their 32 uniform branches per task are compiled out; the general form serves the wide-base sets (two / three balanced parts).
template <int LE, bool ONE_PART>
__device__ __forceinline__ void p2k_pack(int f, int lane, const uint32_t *stage, const P2KDigits &dg, uint32_t (&pk)[8][2]) {
    constexpr int ROWS = 2 * LE;
    const int g = f / ROWS, r = f % ROWS;
    const uint32_t *src = stage + (g * 2 + r / LE) * 2048;
    const int level = ONE_PART ? (r % LE) : (r % LE) / dg.parts, part = ONE_PART ? 0 : (r % LE) % dg.parts;   // uniform per wave
    const int shift = 32 - (level + 1) * dg.Bgbit;
    const uint32_t mask = (1u << dg.Bgbit) - 1u;
    const int32_t half = 1 << (dg.Bgbit - 1);
    const int pw = dg.pw;
    const int32_t hp = pw ? 1 << (pw - 1) : 0, mp = (1 << pw) - 1;
    // all 32 staged words first (independent LDS reads in flight together), then the digit arithmetic -- branch-free: with a loop over the parts
    // inside the unrolled body every read was followed by s_waitcnt lgkmcnt(0) and a uniform branch (64 exposed LDS round trips per step)
    uint32_t t[8][4];
#pragma unroll
    for (int m = 0; m < 8; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) t[m][q] = src[lane + 64 * m + 512 * q];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        int32_t d[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int32_t v = (int32_t)((t[m][q] >> shift) & mask) - half;      // decompose, J/tgsw.jl:112-138
            if (!ONE_PART) {   // balanced parts, least significant first; at most three (l x parts <= 3)
                const int32_t lo0 = ((v + hp) & mp) - hp, v1 = (v - lo0) >> pw;
                const int32_t lo1 = ((v1 + hp) & mp) - hp, v2 = (v1 - lo1) >> pw;
                const int32_t p0 = dg.parts > 1 ? lo0 : v, p1 = dg.parts > 2 ? lo1 : v1;
                v = part == 0 ? p0 : (part == 1 ? p1 : v2);
            }
            d[q] = v;
        }
        pk[m][0] = ((uint32_t)d[0] & 0xffffu) | ((uint32_t)d[1] << 16);
        pk[m][1] = ((uint32_t)d[2] & 0xffffu) | ((uint32_t)d[3] << 16);
    }
}
template <int LE>
__global__ __launch_bounds__(512, 2) void mk_blind_rotate_pair2k_kernel(MKBRArgs a) {
    constexpr int ROWS = 2 * LE;
    constexpr int TASKS = 2 * ROWS;
    constexpr int SLOTS = TASKS > 8 ? TASKS : 8;
    constexpr int PRE = 2;   // key chunks in flight per wave (32 VGPRs each); S0 / S1 of both gates hold 128
    __shared__ int64_t sAcc[2][4096];
    __shared__ cplx sSpec[SLOTS * 512];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // per-lane transform constants are phase-local: fetched again (L1 / L2 hits) at the start of every transform phase instead of living in
    // 16 VGPRs across the multiply phases, where the four partial spectra, two key chunks and a digit spectrum leave no room (the allocator
    // spilled them to scratch and reloaded them in front of every use)
    auto tw_w64 = [&](int ln) { return W64{a.tw[1024 + 1 * 8 + (ln & 7)]}; };
    auto tw_roots1 = [&](int ln) { return LaneRoots{a.tw[ln], a.tw[1216 + ln]}; };         // b_T = T1_T[0][lane], ratio
    auto tw_roots5 = [&](int ln) { return LaneRoots{a.tw[512 + ln], a.tw[1216 + ln]}; };
    P2KDigits dg;
    dg.parts = a.parts > 1 ? a.parts : 1;
    dg.pw = a.pw;
    dg.L = LE / dg.parts;
    dg.Bgbit = a.Bgbit;
    dg.offset = decomp_offset64(dg.L, a.Bgbit);
    const long job0 = 2 * (long)blockIdx.x;
    const bool has1 = job0 + 1 < a.jobs;
    const int32_t *bara0 = a.bara + job0 * a.w_pad;
    const int32_t *bara1 = bara0 + (has1 ? a.w_pad : 0);
    if (a.acc_in) {
        for (int q = threadIdx.x; q < 8192; q += 512) {
            const int g = q >> 12;
            if (g == 0 || has1) sAcc[g][q & 4095] = a.acc_in[(job0 + g) * 4096 + (q & 4095)];
        }
    } else if (wave < 2 && (wave == 0 || has1)) {
        acc_init_64_n<2048>(lane, sAcc[wave], sAcc[wave] + 2048, a.barb[job0 + wave], a.mu);
    }
    __syncthreads();
    const int o = wave >> 2, h = wave & 3;
    auto chunk = [&](int step, int r, int half) { return a.bk + mk_chunk_index_2k(step, r, h, o, ROWS) * 512 + half * 512; };
    auto active = [&](int i) { return bara0[i] != 0 || (has1 && bara1[i] != 0); };   // uniform over the workgroup
    int i = 0;
    while (i < a.pn && !active(i)) i++;
    STAMP_DECL;
    while (i < a.pn) {
        const int ai0 = bara0[i], ai1 = has1 ? bara1[i] : 0;
        int inext = i + 1;
        while (inext < a.pn && !active(inext)) inext++;
        // ---- digits of the step
        const bool t0 = wave < TASKS && ((wave / ROWS) ? ai1 : ai0) != 0;                  // this wave's first / second forward task is live
        const bool t1 = wave + 8 < TASKS && (((wave + 8) / ROWS) ? ai1 : ai0) != 0;
        uint32_t pk0[8][2], pk1[8][2];
        p2k_stage(wave, lane, sAcc, reinterpret_cast<uint32_t *>(sSpec), ai0, ai1, dg.offset);
        lds_barrier<8 * PRE>();
        if (dg.parts == 1) {   // uniform over the launch
            if (t0) p2k_pack<LE, true>(wave, lane, reinterpret_cast<const uint32_t *>(sSpec), dg, pk0);
            if (t1) p2k_pack<LE, true>(wave + 8, lane, reinterpret_cast<const uint32_t *>(sSpec), dg, pk1);
        } else {
            if (t0) p2k_pack<LE, false>(wave, lane, reinterpret_cast<const uint32_t *>(sSpec), dg, pk0);
            if (t1) p2k_pack<LE, false>(wave + 8, lane, reinterpret_cast<const uint32_t *>(sSpec), dg, pk1);
        }
        lds_barrier<8 * PRE>();   // staged words consumed: the slots are free for the spectra
        // ---- half pass 0: even outputs (twist 1); its first key chunks are requested ahead of the transforms (no partial spectra alive yet)
        cplx B[PRE][8];
        mk_pin();
#pragma unroll
        for (int r = 0; r < PRE; r++) load8(lane, B[r], chunk(i, r, 0));
        mk_pin();
        if (t0 || t1) {
            const int ln = opaque_lane(lane);
            const W64 w64 = tw_w64(ln);
            const LaneRoots roots1 = tw_roots1(ln);
            if (t0 && t1) {
                r2k_transform_two<0>(wave, wave + 8, lane, sSpec, pk0, pk1, roots1, w64);
            } else {
                if (t0) r2k_transform<0>(wave, lane, sSpec, pk0, roots1, w64);
                if (t1) r2k_transform<0>(wave + 8, lane, sSpec, pk1, roots1, w64);
            }
        }
        STAMP(0);
        lds_barrier<8 * PRE>();   // spectra published
        STAMP(1);
        cplx S0a[8], S0b[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S0a[m] = S0b[m] = cplx{0.0, 0.0};
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            cplx z[8];
            if (ai0 != 0) {
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[r * 512 + m * 64 + lane];
                mac8r(S0a, z, B[r % PRE]);
            }
            if (ai1 != 0) {
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[(ROWS + r) * 512 + m * 64 + lane];
                mac8r(S0b, z, B[r % PRE]);
            }
            mk_pin();
            if (r + PRE < ROWS) load8(lane, B[r % PRE], chunk(i, r + PRE, 0));
            mk_pin();
        }
        STAMP(2);
        lds_barrier<8 * PRE>();   // spectra consumed
        STAMP(4);
        // ---- half pass 1: odd outputs (twist 5), same digits
        if (t0 || t1) {
            const int ln = opaque_lane(lane);
            const W64 w64 = tw_w64(ln);
            const LaneRoots roots5 = tw_roots5(ln);
            if (t0 && t1) {
                r2k_transform_two<1>(wave, wave + 8, lane, sSpec, pk0, pk1, roots5, w64);
            } else {
                if (t0) r2k_transform<1>(wave, lane, sSpec, pk0, roots5, w64);
                if (t1) r2k_transform<1>(wave + 8, lane, sSpec, pk1, roots5, w64);
            }
        }
        mk_pin();
#pragma unroll
        for (int r = 0; r < PRE; r++) load8(lane, B[r], chunk(i, r, 1));
        mk_pin();
        STAMP(0);
        lds_barrier<8 * PRE>();
        STAMP(1);
        cplx S1a[8], S1b[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S1a[m] = S1b[m] = cplx{0.0, 0.0};
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            cplx z[8];
            if (ai0 != 0) {
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[r * 512 + m * 64 + lane];
                mac8r(S1a, z, B[r % PRE]);
            }
            if (ai1 != 0) {
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[(ROWS + r) * 512 + m * 64 + lane];
                mac8r(S1b, z, B[r % PRE]);
            }
            mk_pin();
            if (r + PRE < ROWS) load8(lane, B[r % PRE], chunk(i, r + PRE, 1));
            mk_pin();
        }
        STAMP(2);
        lds_barrier<8 * PRE>();   // spectra consumed: the area is transpose scratch from here on; every rotated read of the accumulators is done
        STAMP(4);
        // ---- inverse transforms, merge, accumulate
        {
            cplx *xb = sSpec + wave * 512;
            const int ln = opaque_lane(lane);
            const W64 w64 = tw_w64(ln);
            const LaneRoots roots1 = tw_roots1(ln), roots5 = tw_roots5(ln);
            const bool both = ai0 != 0 && ai1 != 0;   // all but one step in 4096: both jobs' transforms of a twist run side by side
            if (both) {
                {
                    const LaneRoots r{opaque_cplx(roots1.b), opaque_cplx(roots1.s)};
                    wave_fft_inv_tq_two<1, 1>(ln, S0a, S0b, xb, r, r, w64);
                }
                {
                    const LaneRoots r{opaque_cplx(roots5.b), opaque_cplx(roots5.s)};
                    wave_fft_inv_tq_two<5, 5>(ln, S1a, S1b, xb, r, r, w64);
                }
            }
            if (ai0 != 0) {
                unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc[0]) + o * 2048;
                if (!both) {
                    wave_fft_inv_tq<1>(ln, S0a, xb, LaneRoots{opaque_cplx(roots1.b), opaque_cplx(roots1.s)}, w64);
                    wave_fft_inv_tq<5>(ln, S1a, xb, LaneRoots{opaque_cplx(roots5.b), opaque_cplx(roots5.s)}, w64);
                }
                cplx lo[8], hi[8];
                merge2048(S0a, S1a, lo, hi);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int q = lane + 64 * m;
                    atomicAdd(accu + q, (unsigned long long)round_i64(lo[m].re) << (16 * h));
                    atomicAdd(accu + q + 512, (unsigned long long)round_i64(hi[m].re) << (16 * h));
                    atomicAdd(accu + q + 1024, (unsigned long long)round_i64(lo[m].im) << (16 * h));
                    atomicAdd(accu + q + 1536, (unsigned long long)round_i64(hi[m].im) << (16 * h));
                }
            }
            if (ai1 != 0) {
                unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc[1]) + o * 2048;
                if (!both) {
                    wave_fft_inv_tq<1>(ln, S0b, xb, LaneRoots{opaque_cplx(roots1.b), opaque_cplx(roots1.s)}, w64);
                    wave_fft_inv_tq<5>(ln, S1b, xb, LaneRoots{opaque_cplx(roots5.b), opaque_cplx(roots5.s)}, w64);
                }
                cplx lo[8], hi[8];
                merge2048(S0b, S1b, lo, hi);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int q = lane + 64 * m;
                    atomicAdd(accu + q, (unsigned long long)round_i64(lo[m].re) << (16 * h));
                    atomicAdd(accu + q + 512, (unsigned long long)round_i64(hi[m].re) << (16 * h));
                    atomicAdd(accu + q + 1024, (unsigned long long)round_i64(lo[m].im) << (16 * h));
                    atomicAdd(accu + q + 1536, (unsigned long long)round_i64(hi[m].im) << (16 * h));
                }
            }
        }
        STAMP(3);
        lds_barrier<8 * PRE>();   // accumulators updated and scratch free before the next step
        STAMP(5);
        i = inext;
    }
    STAMP_FLUSH(blockIdx.x, wave);
    __syncthreads();
    if (a.acc_out) {
        for (int q = threadIdx.x; q < 8192; q += 512) {
            const int g = q >> 12;
            if (g == 0 || has1) a.acc_out[(job0 + g) * 4096 + (q & 4095)] = sAcc[g][q & 4095];
        }
    } else if (wave < 2 && (wave == 0 || has1)) {
        extract_64_n<2048>(lane, sAcc[wave], sAcc[wave] + 2048, a.out + (job0 + wave) * 2049);
    }
}


// initial accumulator of the batched path, in global memory: acc = (0, X^{-barb} * (mu, ..., mu))   (J/3gen_mk_internals.jl:91-92)
__global__ __launch_bounds__(256) void mk_acc_init_2k_kernel(const int32_t *__restrict__ barb, int64_t mu, long jobs, int64_t *__restrict__ acc, int N = 2048) {
    const long job = blockIdx.x;
    if (job >= jobs) return;
    const int b = barb[job];
    for (int q = threadIdx.x; q < N; q += 256) {
        acc[job * 2 * N + q] = 0;
        acc[job * 2 * N + N + q] = (((q + b) & (2 * N - 1)) & N) ? (int64_t)(0ull - (uint64_t)mu) : mu;
    }
}

__global__ __launch_bounds__(256) void mk_linear_kernel(const int32_t *__restrict__ x, const int32_t *__restrict__ y, int32_t *__restrict__ out,
                                                         size_t words, size_t rec, int mode) {
    // mode 0: copy, 1: negate, 2: (0, 1/8) + x + y   (the 3-gen MUX epilogue, J/3gen_mk_gates.jl:144-147)
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= words) return;
    uint32_t v = (uint32_t)x[q];
    if (mode == 1) v = 0u - v;
    if (mode == 2) {
        v += (uint32_t)y[q];
        if (q % rec == rec - 1) v += 1u << 29;
    }
    out[q] = (int32_t)v;
}

__global__ __launch_bounds__(256) void mk_mux_combine_kernel(const int32_t *__restrict__ t, int32_t *__restrict__ out, size_t words, size_t rec) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= words) return;
    const size_t g = q / rec, w = q % rec;
    uint32_t v = (uint32_t)t[(2 * g) * rec + w] + (uint32_t)t[(2 * g + 1) * rec + w];
    if (w == rec - 1) v += 1u << 29;
    out[q] = (int32_t)v;
}

}  // namespace

struct thfhe_mk_ctx {
    thfhe_params p;
    int device = 0;
    hipStream_t stream = nullptr;      // the stream every call enqueues on
    hipStream_t own_stream = nullptr;  // created with the context; `stream` differs only after thfhe_mk_set_stream
    cplx *d_bk = nullptr;
    int32_t *d_ksk = nullptr;
    cplx *d_tw = nullptr;
    int row_words = 0, w_pad = 0, words = 0, log2_2n = 11;
    Rot2kPark park;              // batched N = 2048 rotation, two jobs per workgroup: partial spectra between row-part batches
    long pair_threshold = 256;  // batches of more rotations than this run two gates per workgroup (mk_blind_rotate_pair_kernel)
    bool batched = false;       // N = 2048 with l x digit parts > 3: thfhe_rot2k.h (row parts through the LDS in batches), key table in its layout
    int64_t *d_acc = nullptr;   // batched path: accumulators in global memory, int64[jobs][2][2048]
    size_t cap_acc = 0;
    int parts = 1, pw = 0;      // N = 2048 with a wide gadget base: digit parts and their width (MKBRArgs)
    size_t cap_jobs = 0;
    int32_t *d_bara = nullptr, *d_barb = nullptr, *d_u = nullptr, *d_tmp = nullptr;
    size_t cap_stage = 0;
    DagBuffers dag;   // gate-DAG executor tables (thfhe_dag.h)
    size_t dag_slice = 8192;  // gates per launch of a DAG level
    int32_t *d_in[3] = {nullptr, nullptr, nullptr};
    int32_t *d_out = nullptr;
    bool profiling = false, ev_valid = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex mu;
};

namespace {

int mk_ensure_workspace(thfhe_mk_ctx *c, size_t jobs) {
    if (jobs <= c->cap_jobs) return THFHE_OK;
    (void)hipFree(c->d_bara);
    (void)hipFree(c->d_barb);
    (void)hipFree(c->d_u);
    (void)hipFree(c->d_tmp);
    c->d_bara = c->d_barb = c->d_u = c->d_tmp = nullptr;
    c->cap_jobs = 0;
    THFHE_HIP(hipMalloc(&c->d_bara, jobs * c->w_pad * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_barb, jobs * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_u, jobs * ((size_t)c->p.N + 1) * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_tmp, jobs * (c->words + 1) * sizeof(int32_t)));
    c->cap_jobs = jobs;
    return THFHE_OK;
}
int mk_ensure_stage(thfhe_mk_ctx *c, size_t words) {
    if (words <= c->cap_stage) return THFHE_OK;
    for (auto &p : c->d_in) {
        (void)hipFree(p);
        p = nullptr;
    }
    (void)hipFree(c->d_out);
    c->d_out = nullptr;
    c->cap_stage = 0;
    for (auto &p : c->d_in) THFHE_HIP(hipMalloc(&p, words * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_out, words * sizeof(int32_t)));
    c->cap_stage = words;
    return THFHE_OK;
}

__global__ void mk_extract_kernel(const int64_t *__restrict__ acc, int32_t *__restrict__ out, long jobs, int N);
int mk_launch_rotation(thfhe_mk_ctx *c, const MKBRArgs &a);

// bootstrap (prologue + blind rotate + key switch) of `jobs` = gates * rot jobs; results to d_dst[jobs][P*n+1]
int mk_enqueue_bootstraps(thfhe_mk_ctx *c, const int32_t *d0, const int32_t *d1, const int32_t *d2, MKLin L0, MKLin L1, int rot,
                          size_t gates, int64_t mu, int32_t *d_dst, const int32_t *d_ops = nullptr) {
    const size_t jobs = gates * rot;
    if (c->profiling) THFHE_HIP(hipEventRecord(c->ev[0], c->stream));
    dim3 pg((unsigned)((c->words + 1 + 255) / 256), (unsigned)jobs);
    hipLaunchKernelGGL(mk_prologue_kernel, pg, dim3(256), 0, c->stream, d0, d1, d2, L0, L1, d_ops, rot, c->words, c->w_pad, c->log2_2n, (long)jobs, c->d_bara, c->d_barb);
    if (c->profiling) THFHE_HIP(hipEventRecord(c->ev[1], c->stream));
    MKBRArgs a{c->d_bk, c->d_tw, c->d_bara, c->d_barb, c->d_u, (long)jobs, c->p.parties * c->p.n, c->w_pad, c->p.Bgbit, mu};
    {
        int rc = mk_launch_rotation(c, a);
        if (rc) return rc;
    }
    if (c->profiling) THFHE_HIP(hipEventRecord(c->ev[2], c->stream));
    MKKSArgs k{c->d_ksk, c->d_u, d_dst, (long)jobs, c->p.n, c->p.ks_t, c->p.ks_basebit, c->p.parties, c->row_words, c->p.N, c->p.N + 1, 0};
    const int nsplit = jobs * c->p.parties <= 64 ? 16 : (jobs * c->p.parties <= 256 ? 4 : (c->p.N > 2048 ? 2 : 1));  // fill the chip at small batch sizes; a block holds <= 2048 mask words
    THFHE_HIP(hipMemsetAsync(d_dst, 0, jobs * ((size_t)c->words + 1) * sizeof(int32_t), c->stream));
    mk_launch_keyswitch(k, nsplit, c->stream);
    if (c->profiling) {
        THFHE_HIP(hipEventRecord(c->ev[3], c->stream));
        c->ev_valid = true;
    }
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}

int mk_launch_rotation(thfhe_mk_ctx *c, const MKBRArgs &a) {
    const dim3 grid((unsigned)a.jobs), block(512);
    if (c->p.N == 4096) {
        // ring of degree 4096: accumulators in global memory, rotated in place by r4k_rotate_kernel (thfhe_rot4k.h)
        int64_t *acc = a.acc_out;
        if (!acc) {
            if ((size_t)a.jobs * 2 > c->cap_acc) {   // cap_acc counts 4096-word accumulators
                (void)hipFree(c->d_acc);
                c->d_acc = nullptr;
                c->cap_acc = 0;
                THFHE_HIP(hipMalloc(&c->d_acc, (size_t)a.jobs * 8192 * sizeof(int64_t)));
                c->cap_acc = (size_t)a.jobs * 2;
            }
            acc = c->d_acc;
        }
        if (!a.acc_in) hipLaunchKernelGGL(mk_acc_init_2k_kernel, dim3((unsigned)a.jobs), dim3(256), 0, c->stream, a.barb, a.mu, a.jobs, acc, 4096);
        else if (a.acc_in != acc) THFHE_HIP(hipMemcpyAsync(acc, a.acc_in, (size_t)a.jobs * 8192 * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
        R4KArgs k{c->d_bk, c->d_tw, a.bara, acc, a.jobs, a.pn, c->p.l, c->p.Bgbit, c->parts, c->pw, a.w_pad};
        if ((size_t)a.jobs > c->park.cap_wgs) {   // 128 KiB per workgroup for the parked partial spectra (thfhe_rot4k.h)
            THFHE_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->park.buf);
            c->park.buf = nullptr;
            c->park.cap_wgs = 0;
            THFHE_HIP(hipMalloc(&c->park.buf, (size_t)a.jobs * 8 * 2 * 512 * sizeof(cplx)));
            c->park.cap_wgs = (size_t)a.jobs;
        }
        hipLaunchKernelGGL(r4k_rotate_kernel, grid, block, 0, c->stream, k, c->park.buf);
        if (!a.acc_out) hipLaunchKernelGGL(mk_extract_kernel, grid, dim3(256), 0, c->stream, (const int64_t *)acc, a.out, a.jobs, 4096);
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    if (c->batched) {
        // accumulators in global memory: start (unless the caller hands one in), rotate by all P n key bits in place, then extract (unless the caller wants
        // the accumulator itself)
        int64_t *acc = a.acc_out;
        if (!acc) {
            if ((size_t)a.jobs > c->cap_acc) {
                (void)hipFree(c->d_acc);
                c->d_acc = nullptr;
                c->cap_acc = 0;
                THFHE_HIP(hipMalloc(&c->d_acc, (size_t)a.jobs * 4096 * sizeof(int64_t)));
                c->cap_acc = (size_t)a.jobs;
            }
            acc = c->d_acc;
        }
        if (!a.acc_in) hipLaunchKernelGGL(mk_acc_init_2k_kernel, dim3((unsigned)a.jobs), dim3(256), 0, c->stream, a.barb, a.mu, a.jobs, acc);
        KmsBRArgs k{c->d_bk, c->d_tw, a.bara, acc, a.acc_in ? a.acc_in : acc, a.jobs, a.pn, c->p.l, c->p.Bgbit, c->parts, c->pw, 1, 1, a.w_pad};
        {
            int rc = rot2k_launch(k, c->stream, c->pair_threshold, c->park);
            if (rc) return rc;
        }
        if (!a.acc_out) hipLaunchKernelGGL(mk_extract_kernel, grid, dim3(256), 0, c->stream, (const int64_t *)acc, a.out, a.jobs, 2048);
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    if (c->p.N == 2048) {
        MKBRArgs b = a;
        b.parts = c->parts;
        b.pw = c->pw;
        if (a.jobs > c->pair_threshold) {   // more gates than CUs: two gates per workgroup share every key chunk
            const dim3 pgrid((unsigned)((a.jobs + 1) / 2));
            switch (c->p.l * c->parts) {
            case 1: hipLaunchKernelGGL(mk_blind_rotate_pair2k_kernel<1>, pgrid, block, 0, c->stream, b); break;
            case 2: hipLaunchKernelGGL(mk_blind_rotate_pair2k_kernel<2>, pgrid, block, 0, c->stream, b); break;
            case 3: hipLaunchKernelGGL(mk_blind_rotate_pair2k_kernel<3>, pgrid, block, 0, c->stream, b); break;
            default: return thfhe_fail(THFHE_E_UNSUPPORTED, "N = 2048 needs l x digit parts <= 3");
            }
            THFHE_HIP(hipGetLastError());
            return THFHE_OK;
        }
        switch (c->p.l * c->parts) {
        case 1: hipLaunchKernelGGL(mk_blind_rotate_coop2k_kernel<1>, grid, block, 0, c->stream, b); break;
        case 2: hipLaunchKernelGGL(mk_blind_rotate_coop2k_kernel<2>, grid, block, 0, c->stream, b); break;
        case 3: hipLaunchKernelGGL(mk_blind_rotate_coop2k_kernel<3>, grid, block, 0, c->stream, b); break;
        default: return thfhe_fail(THFHE_E_UNSUPPORTED, "N = 2048 needs l x digit parts <= 3");
        }
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    if (c->p.l <= 3 && a.jobs > c->pair_threshold) {  // throughput path: two gates per workgroup (also for the party-sharded pieces)
        const dim3 pgrid((unsigned)((a.jobs + 1) / 2));
        switch (c->p.l) {
        case 1: hipLaunchKernelGGL(mk_blind_rotate_pair_kernel<1>, pgrid, block, 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL(mk_blind_rotate_pair_kernel<2>, pgrid, block, 0, c->stream, a); break;
        default: hipLaunchKernelGGL(mk_blind_rotate_pair_kernel<3>, pgrid, block, 0, c->stream, a); break;
        }
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    switch (c->p.l) {
    case 1: hipLaunchKernelGGL(mk_blind_rotate_coop_kernel<1>, grid, block, 0, c->stream, a); break;
    case 2: hipLaunchKernelGGL(mk_blind_rotate_coop_kernel<2>, grid, block, 0, c->stream, a); break;
    case 3: hipLaunchKernelGGL(mk_blind_rotate_coop_kernel<3>, grid, block, 0, c->stream, a); break;
    case 4: hipLaunchKernelGGL(mk_blind_rotate_coop_kernel<4>, grid, block, 0, c->stream, a); break;
    default: return thfhe_fail(THFHE_E_UNSUPPORTED, "decomposition length l must be 1..4");
    }
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}

__global__ __launch_bounds__(256) void mk_extract_kernel(const int64_t *__restrict__ acc, int32_t *__restrict__ out, long jobs, int N) {
    const long job = blockIdx.x;
    if (job >= jobs) return;
    const int64_t *ap = acc + job * 2 * N;
    for (int q = threadIdx.x; q <= N; q += 256) {
        int64_t v = q == N ? ap[N] : (q == 0 ? ap[0] : (int64_t)(0ull - (uint64_t)ap[N - q]));
        out[job * (N + 1) + q] = t64tot32(v);
    }
}

// party-sharded prologue: the gate's linear part + mod-switch for the n mask words of this context's block of parties (record
// words [first_word, first_word + n), n = parties_of_this_context * n_lwe) and for b.  Records have rec_words = P_total * n_lwe + 1 words.
__global__ __launch_bounds__(256) void mk_prologue_slice_kernel(const int32_t *__restrict__ in0, const int32_t *__restrict__ in1,
                                                                 const int32_t *__restrict__ in2, MKLin L, int rec_words, int first_word, int n,
                                                                 int log2_2n, long jobs, int32_t *__restrict__ bara, int32_t *__restrict__ barb) {
    const long job = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (job >= jobs || i > n) return;
    const size_t off = (size_t)job * rec_words + (i == n ? rec_words - 1 : first_word + i);
    uint32_t v = (uint32_t)L.cx * (uint32_t)in0[off];
    if (L.cy != 0) v += (uint32_t)L.cy * (uint32_t)in1[off];
    if (L.cz != 0) v += (uint32_t)L.cz * (uint32_t)in2[off];
    if (i == n) {
        v += (uint32_t)L.cb;
        barb[job] = modswitch2n((int32_t)v, log2_2n);
    } else {
        bara[job * n + i] = modswitch2n((int32_t)v, log2_2n);
    }
}

int mk_gates_dev_locked(thfhe_mk_ctx *c, int op, const int32_t *d0, const int32_t *d1, const int32_t *d2, int32_t *dout, size_t count) {
    if (count == 0) return THFHE_OK;
    if (count > (size_t)INT32_MAX / 4) return thfhe_fail(THFHE_E_INVALID, "count too large");
    THFHE_HIP(hipSetDevice(c->device));
    const size_t rec = (size_t)c->words + 1, words = count * rec;
    const unsigned lb = (unsigned)((words + 255) / 256);
    if (op == THFHE_NOT || op == THFHE_COPY) {
        hipLaunchKernelGGL(mk_linear_kernel, dim3(lb), dim3(256), 0, c->stream, d0, d0, dout, words, rec, op == THFHE_NOT ? 1 : 0);
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    MKLin L0, L1;
    if (!mk_gate_lin(op, 0, L0) || op == kOpIdentity) return thfhe_fail(THFHE_E_INVALID, "gate not defined for the 3-gen multi-key scheme");
    mk_gate_lin(op, 1, L1);
    if (!d1 || ((op == THFHE_MUX || op == THFHE_AND3) && !d2)) return thfhe_fail(THFHE_E_INVALID, "null operand");
    const int64_t MU = (int64_t)1 << 61;  // encode_message64(1, 8)
    if (op != THFHE_MUX) {
        int rc = mk_ensure_workspace(c, count);
        if (rc) return rc;
        return mk_enqueue_bootstraps(c, d0, d1, d2, L0, L0, 1, count, MU, dout);
    }
    // MUX: t1 = AND(x, y), t2 = AND(-x, z) as two full bootstraps, then (0, 1/8) + t1 + t2 without bootstrapping
    int rc = mk_ensure_workspace(c, 2 * count);
    if (rc) return rc;
    rc = mk_enqueue_bootstraps(c, d0, d1, d2, L0, L1, 2, count, MU, c->d_tmp);  // job 2g: AND(x,y); job 2g+1: AND(-x,z)
    if (rc) return rc;
    // d_tmp holds [t1_0, t2_0, t1_1, t2_1, ...]: out = (0, 1/8) + t1 + t2            J/3gen_mk_gates.jl:144-147
    hipLaunchKernelGGL(mk_mux_combine_kernel, dim3(lb), dim3(256), 0, c->stream, c->d_tmp, dout, words, rec);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}

}  // namespace

extern "C" {

int thfhe_mk_ctx_create(const thfhe_params *p, const int64_t *bk_coeff, const int32_t *ksk, int device, thfhe_mk_ctx **out) {
    if (!p || !bk_coeff || !ksk || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    *out = nullptr;
    if (p->torus_bits != 64) return thfhe_fail(THFHE_E_UNSUPPORTED, "thfhe_mk_ctx_create is the Torus64 3-gen multi-key path");
    if ((p->N != 1024 && p->N != 2048 && p->N != 4096) || p->k != 1) return thfhe_fail(THFHE_E_UNSUPPORTED, "only N = 1024 / 2048 / 4096, k = 1 is implemented");
    if (p->N == 2048 && p->l > 3) return thfhe_fail(THFHE_E_UNSUPPORTED, "N = 2048 needs decomposition length l <= 3");
    if (p->parties < 1 || p->parties > 512) return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= parties <= 512");
    // digits beyond 10 bit are cut into balanced parts of <= 9 bit (N = 2048 only: the sets that use a wide base live on that ring,
    // J/mk_api.jl:214-298); |sum| <= 2 l parts N 2^(pw-1) 2^15 <= 2^36.6 stays inside the N = 2048 exactness bound
    const int parts = p->Bgbit > 10 ? (p->Bgbit + 8) / 9 : 1;
    const int pw = parts > 1 ? (p->Bgbit + parts - 1) / parts : 0;
    // l x parts <= 3: the one-pass N = 2048 kernel; more row parts (the 256-party set: l = 2, Bgbit = 18 -> 2 x 2 parts) go through the batched
    // rotation of thfhe_rot2k.h, which knows one- and two-part digits.  Exactness: 2 l parts N 2^(part bits - 1) 2^15 <= 2^37 (section 4.3).
    const bool batched = p->N == 2048 && p->l * parts > 3;
    const bool ring4k = p->N == 4096;   // thfhe_rot4k.h: at most six row parts, digits from the top 32 bits; |sum| <= 6 x 4096 x 2^8 x 2^15 = 2^37.6 (one level more than N = 2048)
    if (ring4k && (p->l < 1 || 2 * p->l * parts > 6 || p->l * p->Bgbit > 32 || (double)(2 * p->l * parts) * 4096.0 * (double)(1 << ((parts > 1 ? pw : p->Bgbit) - 1)) * 32768.0 > 274877906944.0))
        return thfhe_fail(THFHE_E_UNSUPPORTED, "N = 4096 needs l x ceil(Bgbit / 9) <= 3, l*Bgbit <= 32 and the FP64 exactness bound 2 l parts N 2^(part bits - 1) 2^15 <= 2^38");
    if (p->l < 1 || p->l > 4 || p->Bgbit < 1 || (!batched && p->l * p->Bgbit > 32) || p->l * p->Bgbit > 64 || (parts > 1 && p->N == 1024) || (batched && parts > 2) ||
        (batched && (double)(2 * p->l * parts) * 2048.0 * (double)(1 << ((parts > 1 ? pw : p->Bgbit) - 1)) * 32768.0 > 137438953472.0))
        return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= l <= 4, l*Bgbit <= 32 (64 on the batched path), and Bgbit <= 10 (FP64 exactness bound) unless N = 2048 with l x ceil(Bgbit / 9) <= 3 or two-part digits");
    if (p->n < 1 || p->n > 767) return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= n <= 767");
    if (p->ks_t < 1 || p->ks_basebit < 1 || p->ks_t * p->ks_basebit > 31) return thfhe_fail(THFHE_E_INVALID, "bad key-switch parameters");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return thfhe_fail(THFHE_E_NO_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    THFHE_HIP(hipSetDevice(device));
    thfhe_mk_ctx *c = new (std::nothrow) thfhe_mk_ctx;
    if (!c) return thfhe_fail(THFHE_E_NOMEM, "out of host memory");
    c->p = *p;
    c->device = device;
    c->words = p->parties * p->n;
    c->w_pad = (c->words + 3) & ~3;
    c->row_words = 128 * ((p->n + 1 + 127) / 128);
    c->log2_2n = ilog2(2 * p->N);
    c->parts = parts;
    c->pw = pw;
    c->batched = batched;
    int64_t *d_coeff = nullptr, *d_exp = nullptr;  // upload staging, freed on every path
    int32_t *d_raw = nullptr;
    auto fail = [&](int code) {
        (void)hipFree(d_coeff);
        (void)hipFree(d_exp);
        (void)hipFree(d_raw);
        thfhe_mk_ctx_destroy(c);
        return code;
    };
#define CK(expr)                                                      \
    do {                                                              \
        hipError_t e_ = (expr);                                       \
        if (e_ != hipSuccess) return fail(thfhe_fail_hip(e_, #expr)); \
    } while (0)
    CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (auto &e : c->ev) CK(hipEventCreate(&e));
    std::vector<cplx> tw(1088 + 128 + 64 + 256);   // [1280..): per-lane roots of the four quarter twists (N = 4096)
    make_lane_roots_4096(tw.data() + 1280);
    // N = 1024: T1[512] T2[64]; N = 2048: T1(twist 1)[512] T1(twist 5)[512] T2[64]; [1088..): per-lane roots (N = 1024); [1216..): pass-1 ratio (N = 2048)
    make_lane_ratio_2048(tw.data() + 1216);
    if (p->N >= 2048) {   // (N = 4096 reads only T2, the ratio and its own roots: the table-free transforms)
        std::vector<cplx> unused(512);
        make_twiddles_2048(tw.data(), tw.data() + 512);
        make_twiddles_1024(unused.data(), tw.data() + 1024);
    } else {
        make_twiddles_1024(tw.data(), tw.data() + 512);
    }
    make_lane_roots_1024(tw.data() + 1088);
    CK(hipMalloc(&c->d_tw, tw.size() * sizeof(cplx)));
    CK(hipMemcpyAsync(c->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream));
    if (ring4k) {
        // key table of thfhe_rot4k.h: [party * n + i][row part rp = (j l + level) parts + part][output o][limb][quarter][512]; row part (j, level, part)
        // of output o is part_{mk_part_index(j, o)}[level] shifted left by part * pw bits (wrapping).  Staged party by party.
        const int RP = 2 * p->l * parts, N = 4096;
        const size_t polys_per_party = (size_t)p->n * RP * 2;
        CK(hipMalloc(&c->d_bk, (size_t)p->parties * polys_per_party * 4 * 2048 * sizeof(cplx)));
        CK(hipMalloc(&d_coeff, polys_per_party * N * sizeof(int64_t)));
        std::vector<int64_t> host(polys_per_party * N);
        for (int q = 0; q < p->parties; q++) {
            for (int i = 0; i < p->n; i++)
                for (int j = 0; j < 2; j++)
                    for (int lv = 0; lv < p->l; lv++)
                        for (int part = 0; part < parts; part++)
                            for (int o = 0; o < 2; o++) {
                                const int64_t *src = bk_coeff + ((((size_t)q * p->n + i) * 4 + mk_part_index(j, o)) * p->l + lv) * N;
                                const int rp = (j * p->l + lv) * parts + part;
                                int64_t *dst = host.data() + (((size_t)i * RP + rp) * 2 + o) * N;
                                const int sh = part * pw;
                                for (int t = 0; t < N; t++) dst[t] = (int64_t)((uint64_t)src[t] << sh);
                            }
            CK(hipMemcpyAsync(d_coeff, host.data(), host.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(r4k_key_transform_kernel, dim3((unsigned)((polys_per_party * 4 + 3) / 4)), dim3(256), 0, c->stream, d_coeff, (long)polys_per_party,
                               c->d_tw, c->d_bk + (size_t)q * polys_per_party * 4 * 2048);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(c->stream));   // `host` is reused for the next party
        }
        (void)hipFree(d_coeff);
        d_coeff = nullptr;
    } else if (batched) {
        // key table of thfhe_rot2k.h: [party * n + i][row part rp = (j l + level) parts + part][output o][limb][half][512]; row part (j, level,
        // part) of output o is part_{mk_part_index(j, o)}[level] shifted left by part * pw bits (wrapping): d (*) K = d_lo (*) K + d_hi (*) (K << pw).
        // Staged party by party (the 256-party set: 194 MB of coefficients per party, 185 GB of spectra in all).
        const int RP = 2 * p->l * parts, N = 2048;
        const size_t polys_per_party = (size_t)p->n * RP * 2;
        CK(hipMalloc(&c->d_bk, (size_t)p->parties * polys_per_party * 4 * 1024 * sizeof(cplx)));
        CK(hipMalloc(&d_coeff, polys_per_party * N * sizeof(int64_t)));
        std::vector<int64_t> host(polys_per_party * N);
        for (int q = 0; q < p->parties; q++) {
            for (int i = 0; i < p->n; i++)
                for (int j = 0; j < 2; j++)
                    for (int lv = 0; lv < p->l; lv++)
                        for (int part = 0; part < parts; part++)
                            for (int o = 0; o < 2; o++) {
                                const int64_t *src = bk_coeff + ((((size_t)q * p->n + i) * 4 + mk_part_index(j, o)) * p->l + lv) * N;
                                const int rp = (j * p->l + lv) * parts + part;
                                int64_t *dst = host.data() + (((size_t)i * RP + rp) * 2 + o) * N;
                                const int sh = part * pw;
                                for (int t = 0; t < N; t++) dst[t] = (int64_t)((uint64_t)src[t] << sh);
                            }
            CK(hipMemcpyAsync(d_coeff, host.data(), host.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(kms_key_transform_kernel, dim3((unsigned)((polys_per_party * 4 + 3) / 4)), dim3(256), 0, c->stream, d_coeff, (long)polys_per_party,
                               c->d_tw, c->d_bk + (size_t)q * polys_per_party * 4 * 1024);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(c->stream));   // `host` is reused for the next party
        }
        (void)hipFree(d_coeff);
        d_coeff = nullptr;
    } else {
    const long PN = (long)p->parties * p->n;
    const size_t coeff_words = (size_t)PN * 4 * p->l * p->N;
    CK(hipMalloc(&d_coeff, coeff_words * sizeof(int64_t)));
    CK(hipMemcpyAsync(d_coeff, bk_coeff, coeff_words * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    const int le = p->l * parts;
    const size_t chunks = (size_t)PN * 2 * le * 8;
    CK(hipMalloc(&c->d_bk, chunks * (p->N / 2) * sizeof(cplx)));
    if (p->N == 2048) {
        if (parts > 1) {   // key rows followed by their copies shifted left by pw, 2 pw bits (wrapping): d (*) K = sum_w d_w (*) (K << pw w)
            CK(hipMalloc(&d_exp, coeff_words * parts * sizeof(int64_t)));
            const long polys = PN * 4 * p->l;
            hipLaunchKernelGGL(mk_expand_parts_kernel, dim3((unsigned)polys, (unsigned)parts), dim3(256), 0, c->stream, d_coeff, d_exp, p->l, parts, pw);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(c->stream));
            (void)hipFree(d_coeff);
            d_coeff = d_exp;
            d_exp = nullptr;
        }
        const long items = PN * 2 * le * 8;
        hipLaunchKernelGGL(mk_key_transform_2k_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, c->stream, d_coeff, PN, le, c->d_tw, c->d_bk);
    } else {
        const long items = PN * 2 * p->l * 2;
        hipLaunchKernelGGL(mk_key_transform_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, c->stream, d_coeff, PN, p->l, c->d_tw, c->d_bk);
    }
    CK(hipGetLastError());
    }
    const long rows = (long)p->parties * p->N * p->ks_t * ((1 << p->ks_basebit) - 1);
    CK(hipMalloc(&d_raw, (size_t)rows * (p->n + 1) * sizeof(int32_t)));
    CK(hipMemcpyAsync(d_raw, ksk, (size_t)rows * (p->n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CK(hipMalloc(&c->d_ksk, (size_t)rows * c->row_words * sizeof(int32_t)));
    hipLaunchKernelGGL(mk_ksk_pad_kernel, dim3((unsigned)rows), dim3(256), 0, c->stream, d_raw, rows, p->n, c->row_words, c->d_ksk);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c->stream));
    (void)hipFree(d_coeff);
    (void)hipFree(d_raw);
#undef CK
    *out = c;
    return THFHE_OK;
}

void thfhe_mk_ctx_destroy(thfhe_mk_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    (void)hipFree(c->d_bk);
    (void)hipFree(c->d_ksk);
    (void)hipFree(c->d_tw);
    (void)hipFree(c->park.buf);
    (void)hipFree(c->d_bara);
    (void)hipFree(c->d_barb);
    (void)hipFree(c->d_u);
    (void)hipFree(c->d_tmp);
    (void)hipFree(c->d_acc);
    for (auto &p : c->d_in) (void)hipFree(p);
    (void)hipFree(c->d_out);
    c->dag.release();
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

void *thfhe_mk_dev_alloc(thfhe_mk_ctx *c, size_t bytes) {
    if (!c) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void thfhe_mk_dev_free(thfhe_mk_ctx *c, void *p) {
    if (c) (void)hipSetDevice(c->device);
    (void)hipFree(p);
}
int thfhe_mk_copy_h2d(thfhe_mk_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    THFHE_HIP(hipSetDevice(c->device));
    THFHE_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}
int thfhe_mk_copy_d2h(thfhe_mk_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    THFHE_HIP(hipSetDevice(c->device));
    THFHE_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}
int thfhe_mk_reserve(thfhe_mk_ctx *c, size_t max_count) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    return mk_ensure_workspace(c, max_count * 2);
}
int thfhe_mk_sync(thfhe_mk_ctx *c) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}
int thfhe_mk_set_profiling(thfhe_mk_ctx *c, int enabled) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    c->profiling = enabled != 0;
    c->ev_valid = false;
    return THFHE_OK;
}
int thfhe_mk_last_timings(thfhe_mk_ctx *c, float ms[4]) {
    if (!c || !ms) return thfhe_fail(THFHE_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->ev_valid) return thfhe_fail(THFHE_E_INVALID, "no profiled call recorded");
    THFHE_HIP(hipEventSynchronize(c->ev[3]));
    THFHE_HIP(hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]));
    THFHE_HIP(hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]));
    THFHE_HIP(hipEventElapsedTime(&ms[2], c->ev[2], c->ev[3]));
    THFHE_HIP(hipEventElapsedTime(&ms[3], c->ev[0], c->ev[3]));
    return THFHE_OK;
}

int thfhe_mk_gates_dev(thfhe_mk_ctx *c, int op, const int32_t *d0, const int32_t *d1, const int32_t *d2, int32_t *dout, size_t count) {
    if (!c || !d0 || !dout) return thfhe_fail(THFHE_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    return mk_gates_dev_locked(c, op, d0, d1, d2, dout, count);
}

int thfhe_mk_gates(thfhe_mk_ctx *c, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2, int32_t *out, size_t count) {
    if (!c || !in0 || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t words = count * ((size_t)c->words + 1), bytes = words * sizeof(int32_t);
    int rc = mk_ensure_stage(c, words);
    if (rc) return rc;
    const int32_t *src[3] = {in0, in1, in2};
    for (int q = 0; q < 3; q++)
        if (src[q]) THFHE_HIP(hipMemcpyAsync(c->d_in[q], src[q], bytes, hipMemcpyHostToDevice, c->stream));
    rc = mk_gates_dev_locked(c, op, c->d_in[0], in1 ? c->d_in[1] : nullptr, in2 ? c->d_in[2] : nullptr, c->d_out, count);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

// Gate-DAG evaluation for the 3-gen scheme: the reference's integer circuits (mk_add_3gen ... mk_int_mul_3gen, J/3gen_mk_gates.jl:183-362) as
// ASAP levels, wires resident in HBM (thfhe_dag.h), `instances` independent evaluations side by side.  Classes: two-input gates NAND / OR /
// AND / XOR with per-gate opcodes in one launch, AND3, MUX (two ANDs + linear combine, :133-150), NOT / COPY (mk_gate_not_3gen,
// mk_copy_3gen: no bootstrap).
int thfhe_mk_dag_run_batch(thfhe_mk_ctx *c, const int32_t *inputs, size_t n_inputs, const int32_t *gates, size_t n_gates, size_t instances,
                           const int32_t *out_wires, size_t n_out, int32_t *outputs, int64_t *stats) {
    if (!c || (!inputs && n_inputs) || (!gates && n_gates) || (!outputs && n_gates) || (!out_wires && n_out)) return thfhe_fail(THFHE_E_INVALID, "null argument");
    DagPlan plan;
    int rc = dag_plan(gates, n_inputs, n_gates,
                      [](int op) {
                          return op == THFHE_NOT || op == THFHE_COPY ? 2 : (op == THFHE_MUX ? 1 : (op == THFHE_AND3 ? 3 : (op == THFHE_NAND || op == THFHE_OR || op == THFHE_AND || op == THFHE_XOR ? 0 : -1)));
                      },
                      plan);
    if (rc) return rc;
    if (stats) plan.fill_stats(stats);
    std::lock_guard<std::mutex> lk(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const int words = c->words + 1;
    MKLin L;
    mk_gate_lin(THFHE_NAND, 0, L);
    return dag_execute(
        plan, c->dag, c->stream, words, n_inputs, n_gates, instances, inputs, out_wires, n_out, outputs, c->dag_slice,
        [&](size_t max_gates, int32_t **in, int32_t **out) {
            int r = mk_ensure_workspace(c, 2 * max_gates);
            if (!r) r = mk_ensure_stage(c, max_gates * words);
            in[0] = c->d_in[0], in[1] = c->d_in[1], in[2] = c->d_in[2], *out = c->d_out;
            return r;
        },
        [&](int cls, const int32_t *d_ops, size_t n) {
            if (cls == 0) return mk_enqueue_bootstraps(c, c->d_in[0], c->d_in[1], nullptr, L, L, 1, n, (int64_t)1 << 61, c->d_out, d_ops);
            return mk_gates_dev_locked(c, cls == 1 ? THFHE_MUX : THFHE_AND3, c->d_in[0], c->d_in[1], c->d_in[2], c->d_out, n);
        });
}

int thfhe_mk_set_dag_slice(thfhe_mk_ctx *c, size_t max_gates) {
    if (!c || max_gates < 1 || max_gates > 32767) return thfhe_fail(THFHE_E_INVALID, "slice must be 1 .. 32767 gates");
    std::lock_guard<std::mutex> g(c->mu);
    c->dag_slice = max_gates;
    return THFHE_OK;
}

int thfhe_mk_dag_run(thfhe_mk_ctx *c, int32_t *wires, size_t n_inputs, const int32_t *gates, size_t n_gates, int64_t *stats) {
    if (!wires) return thfhe_fail(THFHE_E_INVALID, "null argument");
    return thfhe_mk_dag_run_batch(c, wires, n_inputs, gates, n_gates, 1, nullptr, 0, wires + n_inputs * (size_t)(c ? c->words + 1 : 0), stats);
}

int thfhe_mk_gates_mixed(thfhe_mk_ctx *c, const int32_t *ops, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count) {
    if (!c || !ops || !in0 || !in1 || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    for (size_t g = 0; g < count; g++)
        if (!(ops[g] == THFHE_NAND || ops[g] == THFHE_OR || ops[g] == THFHE_AND || ops[g] == THFHE_XOR))
            return thfhe_fail(THFHE_E_INVALID, "thfhe_mk_gates_mixed takes the two-input 3-gen gates NAND / OR / AND / XOR only");
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t words = count * ((size_t)c->words + 1), bytes = words * sizeof(int32_t);
    int rc = mk_ensure_stage(c, words);
    if (rc) return rc;
    rc = mk_ensure_workspace(c, count);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_in[0], in0, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_in[1], in1, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_in[2], ops, count * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    MKLin L;
    mk_gate_lin(THFHE_NAND, 0, L);
    rc = mk_enqueue_bootstraps(c, c->d_in[0], c->d_in[1], nullptr, L, L, 1, count, (int64_t)1 << 61, c->d_out, c->d_in[2]);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

// ---- party-sharded building blocks (device pointers; see include/thfhe_hip.h and thfhe/party_sharded.py) -------------------
int thfhe_mk_rotate_partial_dev(thfhe_mk_ctx *c, const int32_t *d_bara, const int32_t *d_barb, int64_t mu, const int64_t *d_acc_in,
                                int64_t *d_acc_out, size_t count) {
    if (!c || !d_bara || !d_acc_out || (!d_acc_in && !d_barb)) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    MKBRArgs a{c->d_bk, c->d_tw, d_bara, d_barb, nullptr, (long)count, c->p.parties * c->p.n, c->words, c->p.Bgbit, mu, d_acc_in, d_acc_out};
    return mk_launch_rotation(c, a);
}
int thfhe_mk_prologue_dev(thfhe_mk_ctx *c, int op, int which, const int32_t *d0, const int32_t *d1, const int32_t *d2, int rec_words,
                          int first_word, int32_t *d_bara, int32_t *d_barb, size_t count) {
    if (!c || !d0 || !d_bara || !d_barb) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    MKLin L;
    if (op == -1) op = kOpIdentity;  // plain bootstrap of in0
    if (!mk_gate_lin(op, which, L)) return thfhe_fail(THFHE_E_INVALID, "gate not defined for the 3-gen multi-key scheme");
    if ((L.cy != 0 && !d1) || (L.cz != 0 && !d2)) return thfhe_fail(THFHE_E_INVALID, "null operand");
    const int nw = c->words;  // this context's parties are contiguous in the record: parties * n mask words
    if (first_word < 0 || first_word + nw > rec_words - 1) return thfhe_fail(THFHE_E_INVALID, "party slice outside the record");
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const dim3 grid((unsigned)((nw + 1 + 255) / 256), (unsigned)count);
    hipLaunchKernelGGL(mk_prologue_slice_kernel, grid, dim3(256), 0, c->stream, d0, d1, d2, L, rec_words, first_word, nw, c->log2_2n, (long)count, d_bara, d_barb);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}
#ifdef THFHE_STAMPS
int thfhe_debug_read_stamps_mk(unsigned long long *dst, size_t count) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
int thfhe_mk_set_pair_threshold(thfhe_mk_ctx *c, long max_single_jobs) {
    if (!c || max_single_jobs < 0) return thfhe_fail(THFHE_E_INVALID, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    c->pair_threshold = max_single_jobs;
    return THFHE_OK;
}
int thfhe_mk_set_stream(thfhe_mk_ctx *c, void *hip_stream) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return THFHE_OK;
}
int thfhe_mk_extract_dev(thfhe_mk_ctx *c, const int64_t *d_acc, int32_t *d_u, size_t count) {
    if (!c || !d_acc || !d_u) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(mk_extract_kernel, dim3((unsigned)count), dim3(256), 0, c->stream, d_acc, d_u, (long)count, c->p.N);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}
int thfhe_mk_keyswitch_dev(thfhe_mk_ctx *c, const int32_t *d_u, int32_t *d_out, size_t count) {
    if (!c || !d_u || !d_out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    MKKSArgs k{c->d_ksk, d_u, d_out, (long)count, c->p.n, c->p.ks_t, c->p.ks_basebit, c->p.parties, c->row_words, c->p.N, c->p.N + 1, 0};
    const int nsplit = count * c->p.parties <= 64 ? 16 : (count * c->p.parties <= 256 ? 4 : (c->p.N > 2048 ? 2 : 1));
    THFHE_HIP(hipMemsetAsync(d_out, 0, count * ((size_t)c->words + 1) * sizeof(int32_t), c->stream));
    mk_launch_keyswitch(k, nsplit, c->stream);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}

int thfhe_mk_bootstrap(thfhe_mk_ctx *c, int64_t mu, const int32_t *x, int32_t *out, size_t count) {
    if (!c || !x || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t words = count * ((size_t)c->words + 1), bytes = words * sizeof(int32_t);
    int rc = mk_ensure_stage(c, words);
    if (rc) return rc;
    rc = mk_ensure_workspace(c, count);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_in[0], x, bytes, hipMemcpyHostToDevice, c->stream));
    MKLin L;
    mk_gate_lin(kOpIdentity, 0, L);
    rc = mk_enqueue_bootstraps(c, c->d_in[0], c->d_in[0], c->d_in[0], L, L, 1, count, mu, c->d_out);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

}  // extern "C"
