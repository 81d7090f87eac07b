// thfhe_sk.hip -- single-key (Torus32) gate bootstrapping on gfx950: kernels + C ABI.
//
// Kernels (one HIP stream per context, no host sync inside a call):
//   sk_key_transform_kernel   BootstrapKey forward_transform step (J/bootstrap.jl:11-12): coefficient-domain TGSW
//                             rows -> two-limb FP64 spectra in the blind-rotate kernel's register order
//   sk_prologue_kernel        gate linear part (J/gates.jl:15-177) + mod-switch decode_message(.,2N)
//                             (J/bootstrap.jl:80-81) -> bara[job][n], barb[job]
//   sk_blind_rotate_ring_kernel / sk_blind_rotate_coop_kernel   blind_rotate_and_extract (J/bootstrap.jl:38-65): accumulator in
//                             LDS for all n CMuxes; throughput (8 gates per workgroup, key through an LDS-DMA ring) and latency
//                             (one workgroup per gate) variants
//   sk_keyswitch_kernel / sk_keyswitch_staged_kernel / sk_keyswitch_multi_kernel   keyswitch (J/keyswitch.jl:45-80) (+ the MUX combine of
//                             J/gates.jl:172-176): one gate per workgroup (small batches); from 192 gates on the rows of a few (i, j) staged in
//                             LDS for 32 gates, the digit selecting an address; rows in registers selected by branches for the remaining shapes
//   sk_linear_kernel          NOT / COPY (J/gates.jl:76-79)
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"
#include "thfhe_dag.h"
#include "thfhe_lane.h"

using namespace thfhe;

namespace {

// ------------------------------------------------------------------------------------------------------
// key transform
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sk_key_transform_kernel(const int32_t *__restrict__ polys, long npolys,
                                                                const cplx *__restrict__ tw, cplx *__restrict__ spec) {
    __shared__ cplx sT1[512];
    __shared__ cplx sT2[64];
    __shared__ cplx sX[4][kXbufSlots];
    for (int t = threadIdx.x; t < 512; t += 256) sT1[t] = tw[t];
    if (threadIdx.x < 64) sT2[threadIdx.x] = tw[512 + threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + wave;
    if (q >= npolys) return;
    cplx zlo[8], zhi[8];
    key_limbs_to_z(lane, polys + q * 1024, zlo, zhi);
    cplx *xb = sX[wave];
    wave_fft_fwd(lane, zlo, xb, sT1, sT2);
    wave_fft_fwd(lane, zhi, xb, sT1, sT2);
    cplx *out = spec + q * 1024;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        out[m * 64 + lane] = cplx{zlo[m].re * (1.0 / 512), zlo[m].im * (1.0 / 512)};
        out[512 + m * 64 + lane] = cplx{zhi[m].re * (1.0 / 512), zhi[m].im * (1.0 / 512)};
    }
}

// ------------------------------------------------------------------------------------------------------
// prologue: tmp = (0, cb) + cx * x + cy * y ; bara = decode_message(tmp.a, 2N) ; barb likewise
// ------------------------------------------------------------------------------------------------------
struct Lin {
    int32_t cb, cx, cy;
    int ysel;  // 1: second operand is in1, 2: in2
};
__host__ __device__ inline bool gate_lin(int op, int which, Lin &L) {
    const int32_t E8 = 1 << 29, E4 = 1 << 30;  // encode_message(1,8), (1,4)   J/numeric-functions.jl:86-89
    switch (op) {
    case THFHE_NAND: L = Lin{E8, -1, -1, 1}; return true;
    case THFHE_OR: L = Lin{E8, 1, 1, 1}; return true;
    case THFHE_AND: L = Lin{-E8, 1, 1, 1}; return true;
    case THFHE_XOR: L = Lin{E4, 2, 2, 1}; return true;
    case THFHE_XNOR: L = Lin{-E4, -2, -2, 1}; return true;
    case THFHE_NOR: L = Lin{-E8, -1, -1, 1}; return true;
    case THFHE_ANDNY: L = Lin{-E8, -1, 1, 1}; return true;
    case THFHE_ANDYN: L = Lin{-E8, 1, -1, 1}; return true;
    case THFHE_ORNY: L = Lin{E8, -1, 1, 1}; return true;
    case THFHE_ORYN: L = Lin{E8, 1, -1, 1}; return true;
    case THFHE_MUX: L = which == 0 ? Lin{-E8, 1, 1, 1} : Lin{-E8, -1, 1, 2}; return true;  // J/gates.jl:166-171
    case kOpIdentity: L = Lin{0, 1, 0, 1}; return true;                                      // plain bootstrap(x)
    default: return false;
    }
}

__global__ __launch_bounds__(256) void sk_prologue_kernel(const int32_t *__restrict__ in0, const int32_t *__restrict__ in1,
                                                           const int32_t *__restrict__ in2, int op, const int32_t *__restrict__ ops,
                                                           int rot_per_gate, int n, int n_pad, int log2_2n, long jobs,
                                                           int32_t *__restrict__ bara, int32_t *__restrict__ barb) {
    const long job = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (job >= jobs || i > n) return;
    const long gate = job / rot_per_gate;
    const int which = (int)(job % rot_per_gate);
    Lin L;
    gate_lin(ops ? ops[gate] : op, which, L);  // ops: per-gate opcodes of a mixed level (validated on the host)
    const size_t off = (size_t)gate * (n + 1) + i;
    uint32_t v = (uint32_t)L.cx * (uint32_t)in0[off];
    if (L.cy != 0) v += (uint32_t)L.cy * (uint32_t)(L.ysel == 2 ? in2[off] : in1[off]);
    if (i == n) {
        v += (uint32_t)L.cb;
        barb[job] = modswitch2n((int32_t)v, log2_2n);
    } else {
        bara[job * n_pad + i] = modswitch2n((int32_t)v, log2_2n);
    }
}

THFHE_STAMP_STORAGE

// arguments of the blind-rotate kernels
struct BRArgs {
    const cplx *bk;        // spectral key
    const cplx *tw;        // T1[512] ++ T2[64]
    const int32_t *bara;   // [jobs][n_pad]
    const int32_t *barb;   // [jobs]
    int32_t *out;          // [jobs][N+1]
    long jobs;
    int n, n_pad, Bgbit;
    int32_t mu;
};

// ------------------------------------------------------------------------------------------------------
// blind rotate + extract, throughput kernel ("LDS ring", second generation).
//
// One 512-thread workgroup = 8 wavefronts = 8 jobs, one workgroup per CU, all 160 KiB of LDS:
//     8 x (accumulator int32[2][1024] 8 KiB + transpose buffer 9 KiB) | key ring 3 x 8 KiB                  = 163 840 B
// Each wave owns one job; its accumulator never leaves LDS during the n CMuxes.  The eight waves walk the key index i, the digit
// rows and the four (column, limb) chunks of a row in lock step.  A chunk is 8 KiB of key spectrum = 8 slices of 1 KiB; wave w brings
// slice w into the ring with ONE global_load_lds_dwordx4 (LDS-DMA, no registers), so the whole key crosses the CU's vector-memory
// path once per workgroup instead of once per wave.  Hand-off of chunk q: every wave waits for its own slice (s_waitcnt vmcnt(N),
// N = younger DMAs in flight), then s_barrier -- after it the chunk is complete AND everybody has finished reading chunk q-1, whose
// slot is refilled at once with chunk q+2.  Per row: 5 barriers, 4 DMA issues per wave.  A wave whose mod-switched mask word is 0
// (J/bootstrap.jl:40) or that has no job still streams and synchronises.
//
// What the second generation changed comes from an in-kernel cycle trace and the PMC counters of the first (tools/ring_stamps.py,
// profiles/r02_ring_generations.md): with two waves per SIMD that kernel was bound by exposed LDS round trips and by the LDS
// instruction pipe (69 % busy, bursts in lock step), with the VALU 55 % busy.  So:
//   * the FIRST transpose of every transform (register index <-> lane bits 3..5) stays in registers: v_permlane32_swap,
//     v_permlane16_swap and row_ror:8 DPP moves (wave_transpose_hi3) instead of 8 ds_write_b128 + 8 ds_read_b128;
//   * pass-1 twiddles are rebuilt from two per-lane roots (thfhe_lane.h, variant "q") instead of read from a T1 table in LDS: the
//     table reads sat between the butterflies and the transpose of every transform;
//   * the second transpose uses the padded 576-slot buffer (the 8 KiB of the T1 table pay for the padding): its slot maps are
//     base + immediate, 2 LDS address registers per wave instead of 16 for the XOR-swizzled maps;
//   * the multiply-accumulate is software-pipelined over half chunks: the four ring reads of one half are in flight under the
//     16 FMAs of the half before (the first generation read one slice, used it, read the next: 8 serial round trips per chunk),
//     and S += z * b is four FMAs (the mul + fma + add form cost 384 more FP64 instructions per CMux);
// LDS pipe instructions per launch 1.74e9 -> 1.38e9, LDS pipe busy 69 % -> 46 %, VALU busy 55 % -> 75 %, 34.8 -> 33.3 ms per 4096 gates.
// Third pass (33.3 -> 29.9 ms):
//   * rotate + decompose in one step on byte offsets (thfhe_lane.h: rotated_word / mixed_digits_z), the index / sign / subtraction
//     work once per accumulator polynomial (fields kept across its l levels), a level = one v_bfe_i32 + one conversion per coefficient;
//   * register-lean pass-1 twiddles (LaneTw): the compiler hoists loop-invariant twiddle products out of the CMux loop and then spills
//     some of them; every reload was followed by s_waitcnt vmcnt(0), which also waits for the ring DMAs issued just before the
//     transform.  Keeping only the even products and forming the odd ones in place (16 more FP64 instructions per transform) removes
//     those reloads from the transforms.
// ------------------------------------------------------------------------------------------------------
#ifndef THFHE_RING_NF
#define THFHE_RING_NF 16
#endif
#ifndef THFHE_RING_LEAN_ROOTS
#define THFHE_RING_LEAN_ROOTS 1   // eight-wave shape: pass-1 twiddles rebuilt per transform (no scratch); 0 = even products kept across the loop
#endif
#ifndef THFHE_RING_READ_FIRST
#define THFHE_RING_READ_FIRST 0   // 1 = round 4's experiment (profiles/r04_ring_multiply_phase.md): measured slower, kept for the record
#endif
// W = waves (= jobs) per workgroup.  W = 8 is the throughput shape described above.  W = 4 (one wave per SIMD, 92 KiB of LDS, each wave
// brings TWO slices of a chunk) is the shape for batches that cannot give every CU eight jobs (<= 1024 rotations): a wave alone on its
// SIMD issues at ~87 % of what a pair reaches together (tools/probes/issue_probe.hip), so four jobs finish much sooner than eight.
template <int L, int V = 1, int W = 8>
__global__ __launch_bounds__(64 * W, W == 8 ? 2 : 1) void sk_blind_rotate_ring_kernel(BRArgs a) {
    __shared__ __attribute__((aligned(4096))) int32_t sAcc[W][2048];   // rotated_digits_z ORs byte offsets into the polynomial base
    __shared__ cplx sX[W][kXbufSlots];
    __shared__ cplx sRing[3][512];
    constexpr int ROWS = 2 * L;
    constexpr int DPC = 8 / W;   // ring DMAs per wave and chunk
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const W64 w64{a.tw[512 + 1 * 8 + (lane & 7)]};
    const LaneRoots roots{a.tw[576 + 2 * lane], a.tw[576 + 2 * lane + 1]};
    const LaneTw tw = make_lane_tw(roots);
    const long job = (long)blockIdx.x * W + wave;
    const bool has_job = job < a.jobs;
    int32_t *acc = sAcc[wave];
    cplx *xb = sX[wave];
    const uniform_i32_ptr bara = as_uniform(a.bara + (has_job ? job : 0) * a.n_pad);   // job is wave-uniform: scalar loads
    const int Bgbit = a.Bgbit;
    if (has_job) acc_init16(lane, acc, acc + 1024, a.barb[job], a.mu);

    const long total_chunks = (long)a.n * ROWS * 4;
    const cplx *gsrc = a.bk + wave * (64 * DPC) + lane;
    long q_issue = 0;
    int slot_issue = 0;
    const uint32_t ring_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)&sRing[0][0] + (uint32_t)wave * (1024u * DPC);
    auto issue = [&]() {
#pragma unroll
        for (int d = 0; d < DPC; d++) ring_dma(gsrc + 64 * d, ring_base + (uint32_t)slot_issue * 8192u + 1024u * d);
        if (q_issue + 1 < total_chunks) {
            gsrc += 512;
            q_issue++;
        }
        slot_issue = slot_issue == 2 ? 0 : slot_issue + 1;
    };
    __syncthreads();
    issue();
    issue();
    issue();
    int slot_use = 0;
    STAMP_DECL;

    for (int i = 0; i < a.n; i++) {
        const int ai = bara[i];
        const bool active = has_job && ai != 0;
        const int a2n = ai & 2047;
        cplx S[2][2][8];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int m = 0; m < 8; m++) S[c][h][m] = cplx{0.0, 0.0};
        constexpr int NF = THFHE_RING_NF;   // rotated fields kept across the levels of a polynomial (register budget)
        uint32_t fld[NF];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            cplx z[8];
            if (active) {
                // index / sign / subtraction once per accumulator polynomial, then one signed bit-field extract + one conversion per level
                int a2n_r = a2n;
                asm volatile("" : "+s"(a2n_r));  // opaque per row: the rotated LDS addresses are recomputed, not kept alive
                if (r % L == 0) rotated_fields_keep<NF, W == 4>(lane, acc + (r / L) * 1024, a2n_r, L, Bgbit, fld);
                asm volatile("" : "+s"(a2n_r));
                mixed_digits_z<NF>(lane, acc + (r / L) * 1024, a2n_r, (r % L) + 1, L, Bgbit, fld, z);
                // pass-1 twiddles: the eight-wave shape rebuilds all eight products b s^k per transform from the two per-lane roots (made opaque so that
                // they are not hoisted out of the CMux loop): 12 fewer registers live across the loop than with the even products kept (LaneTw), the
                // compiler then parks nothing in scratch (44 -> 0 B per lane: no reload in front of a row's digits waits for the ring DMAs any more) --
                // 30.21 -> 29.97 ms per 4096 gates.  The four-wave shape (registers to spare) keeps LaneTw: 9.67 against 9.80 ms per 1024 gates.
                if (!(V & 1)) wave_fft_fwd_r(lane, z, xb, roots, w64);
                else if (THFHE_RING_LEAN_ROOTS && W == 8) wave_fft_fwd_q(lane, z, xb, LaneRoots{opaque_cplx(roots.b), opaque_cplx(roots.s)}, w64);
                else wave_fft_fwd_q(lane, z, xb, tw, w64);
            }
            STAMP(0);
            cplx bA[4], bB[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                if (c4 == 0) ring_barrier<2 * DPC>(); else ring_barrier<DPC>();
                STAMP(1);
                const cplx *B = &sRing[slot_use][0];
#if THFHE_RING_READ_FIRST
                // the chunk's first four reads go out the moment the barrier falls (the eight waves' reads of a chunk keep the LDS array busy for
                // 256 cycles: they, not the arithmetic, are the longest thing between two barriers), then the refill DMA, then the arithmetic
                if (active) {
#pragma unroll
                    for (int m = 0; m < 4; m++) bA[m] = B[m * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (c4 > 0) issue();
                if (active) {
#else
                if (c4 > 0) issue();
                if (active) {
#pragma unroll
                    for (int m = 0; m < 4; m++) bA[m] = B[m * 64 + lane];
#endif
                    if (c4 > 0) {
#pragma unroll
                        for (int m = 0; m < 4; m++) cfma(S[(c4 - 1) >> 1][(c4 - 1) & 1][4 + m], z[4 + m], bB[m]);
                    }
#pragma unroll
                    for (int m = 0; m < 4; m++) bB[m] = B[(4 + m) * 64 + lane];
#pragma unroll
                    for (int m = 0; m < 4; m++) cfma(S[c4 >> 1][c4 & 1][m], z[m], bA[m]);
                }
                slot_use = slot_use == 2 ? 0 : slot_use + 1;
                STAMP(2);
            }
            ring_barrier<2 * DPC>();  // the row's last chunk is read by all (its second half sits in bB): refill its slot before the next transform
            issue();
            STAMP(1);
            if (active) {
#pragma unroll
                for (int m = 0; m < 4; m++) cfma(S[1][1][4 + m], z[4 + m], bB[m]);
            }
            STAMP(2);
        }
        if (active) {
            wave_sync();
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if ((V & 1) && !(V & 2)) {
                    if (THFHE_RING_LEAN_ROOTS && W == 8) {
                        wave_fft_inv_q(lane, S[c][0], xb, LaneRoots{opaque_cplx(roots.b), opaque_cplx(roots.s)}, w64);
                        wave_fft_inv_q(lane, S[c][1], xb, LaneRoots{opaque_cplx(roots.b), opaque_cplx(roots.s)}, w64);
                    } else {
                        wave_fft_inv_q(lane, S[c][0], xb, tw, w64);
                        wave_fft_inv_q(lane, S[c][1], xb, tw, w64);
                    }
                } else {
                    wave_fft_inv_r(lane, S[c][0], xb, roots, w64);
                    wave_fft_inv_r(lane, S[c][1], xb, roots, w64);
                }
                acc_update16(lane, acc + c * 1024, S[c][0], S[c][1]);
            }
            wave_sync();
        }
        STAMP(3);
    }
    STAMP_FLUSH(blockIdx.x, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (has_job) extract16(lane, acc, acc + 1024, a.out + job * 1025);
}

// ------------------------------------------------------------------------------------------------------
// blind rotate + extract, latency kernel ("cooperative", second generation): one 512-thread workgroup = ONE job.  For the small
// batches the reference's gate-at-a-time callers produce (boots* shims, ripple-carry circuits: 855 of the 1 033 levels of the KNN
// decision hold 1-3 gates) the ring kernel leaves 7/8 of a CU idle; here the work of one CMux is spread over the eight waves and
// the n CMuxes are a dependent chain, so what counts is the length of one step's critical path:
//   F  waves 0 .. 2l-1: wave r rotates / decomposes / transforms digit row r and publishes its spectrum in LDS;       barrier
//   M  wave w = (column c, limb h, half): S = sum over its half of the rows of spectrum_r * key(r, c, h) (key chunks in registers);
//      the waves of half 1 hand their partial sums to their partners through LDS;                                     barrier
//   I  waves 0-3 (one per SIMD) add the partner's partial sum, inverse-transform, and add round(S) << 16h into accumulator
//      polynomial c with 32-bit LDS atomics (two limbs per polynomial; integer adds commute -> bit-exact);            barrier
// The key stream is what the first generation tripped over: it requested the 24 chunks of a step (192 KiB per workgroup at l = 3)
// at the top of the step, in front of the forward transforms -- 192 wave-loads queue up on the CU's one vector-memory path
// (64 B/clk: ~3 k cycles) and a wave cannot start its transform before its own loads have been accepted.  A microbenchmark
// (tools/probes/fetch_probe.hip, profiles/r02_fetch_probe.md) shows one CU can pull the 121 MB key at 95-113 GB/s when loads are
// spread out, 2.4x what that kernel reached.  Here the chunks of step i+1 are requested during step i, once the registers that
// held step i's chunks are dead: the idle waves of half 1 right after the hand-off, the transforming waves one row's worth at a
// time between the stages of their inverse transform (compiler fences pin the places) -- nobody's transform waits on the queue.
// Transforms are variant "r" (padded buffer, pass-1 twiddles from per-lane roots): LDS = acc 8 + spectra 2l x 8 + 8 x 9 KiB.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void pin() { asm volatile("" ::: "memory"); }  // memory operations do not move across this point

template <int L, int PACE = 1>
__global__ __launch_bounds__(512, 2) void sk_blind_rotate_coop_kernel(BRArgs a) {
    constexpr int ROWS = 2 * L;
    __shared__ __attribute__((aligned(4096))) int32_t sAcc[2048];
    __shared__ cplx sSpec[ROWS][512];
    __shared__ cplx sX[8][kXbufSlots];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const W64 w64{a.tw[512 + 1 * 8 + (lane & 7)]};
    const LaneRoots roots{a.tw[576 + 2 * lane], a.tw[576 + 2 * lane + 1]};
    const long job = blockIdx.x;
    const uniform_i32_ptr bara = as_uniform(a.bara + job * a.n_pad);
    const int Bgbit = a.Bgbit;
    if (wave == 0) acc_init16(lane, sAcc, sAcc + 1024, a.barb[job], a.mu);
    const int c = (wave >> 1) & 1, h = wave & 1, half = wave >> 2, r0 = half * L;  // role in M: rows r0 .. r0+L-1 of (column c, limb h)
    unsigned int *ap = reinterpret_cast<unsigned int *>(sAcc) + c * 1024;
    cplx *xb = sX[wave];

    int i = 0;
    while (i < a.n && bara[i] == 0) i++;   // J/bootstrap.jl:40: mask words that mod-switch to 0 are skipped (uniform over the workgroup)
    cplx B[L][8];
    if (i < a.n) {
#pragma unroll
        for (int r = 0; r < L; r++) load8(lane, B[r], a.bk + bk_spec_index(i, r0 + r, c, h, ROWS));
    }
    wg_barrier();
    STAMP_DECL;
    while (i < a.n) {
        const int a2n = bara[i] & 2047;
        int inext = i + 1;
        while (inext < a.n && bara[inext] == 0) inext++;
        const int inl = inext < a.n ? inext : i;   // the last step re-requests its own chunks: unconditional loads keep B one set of registers
        // ---- F ----
        if (wave < ROWS) {
            cplx z[8];
            rotated_digits_z(lane, sAcc + (wave / L) * 1024, a2n, (wave % L) + 1, L, Bgbit, z);
            wave_fft_fwd_q(lane, z, xb, roots, w64);
#pragma unroll
            for (int m = 0; m < 8; m++) sSpec[wave][m * 64 + lane] = z[m];
        }
        STAMP(0);
        wg_barrier();  // spectra published; every rotated read of the accumulator is done
        STAMP(1);
        // ---- M ----
        cplx S[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S[m] = cplx{0.0, 0.0};
#pragma unroll
        for (int r = 0; r < L; r++) {
            cplx z[8];
#pragma unroll
            for (int m = 0; m < 8; m++) z[m] = sSpec[r0 + r][m * 64 + lane];
            mac8r(S, z, B[r]);
            pin();   // one row's spectrum in registers at a time (hoisting all 2l x 8 reads costs more registers than there are)
        }
        STAMP(2);
        if (half == 1) {
#pragma unroll
            for (int m = 0; m < 8; m++) xb[m * 64 + lane] = S[m];   // hand-off to wave - 4 (this wave's transpose buffer is idle)
            pin();
            STAMP(3);
            wg_barrier();
            STAMP(4);
            // nothing else to do until the next step: request all of its chunks now (AFTER the barrier: the partners' inverse
            // transforms must not wait for these 8 l loads to be accepted by the memory pipeline)
#pragma unroll
            for (int r = 0; r < L; r++) {
                const cplx *src = a.bk + bk_spec_index(inl, r0 + r, c, h, ROWS);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    B[r][m] = src[m * 64 + lane];
                    pin();
                    __builtin_amdgcn_s_sleep(PACE);   // paced: these waves have the whole inverse phase; a flooded queue stalls the partners' loads
                }
            }
        } else {
            load8(lane, B[0], a.bk + bk_spec_index(inl, r0, c, h, ROWS));
            pin();
            STAMP(3);
            wg_barrier();
            STAMP(4);
            // ---- I ----
            const cplx *px = sX[wave + 4];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const cplx v = px[m * 64 + lane];
                S[m].re += v.re;
                S[m].im += v.im;
            }
            wave_sync();
            invr_seg1(lane, S, xb, w64);
            pin();
            if (L > 1) load8(lane, B[L > 1 ? 1 : 0], a.bk + bk_spec_index(inl, r0 + 1, c, h, ROWS));
            pin();
            wave_sync();
            inv_seg2_ld(lane, S, xb);
            dft8<-1>(S);
            pin();
            if (L > 2) load8(lane, B[L > 2 ? 2 : 0], a.bk + bk_spec_index(inl, r0 + 2, c, h, ROWS));
            if (L > 3) load8(lane, B[L > 3 ? 3 : 0], a.bk + bk_spec_index(inl, r0 + 3, c, h, ROWS));
            pin();
            wave_transpose_hi3(S);
            invq_seg3(S, roots);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int q = lane + 64 * m;
                atomicAdd(ap + q, round_lo32(S[m].re) << (16 * h));
                atomicAdd(ap + q + 512, round_lo32(S[m].im) << (16 * h));
            }
        }
        STAMP(3);
        wg_barrier();  // accumulator updated before anybody rotates it again
        STAMP(5);
        i = inext;
    }
    STAMP_FLUSH(blockIdx.x, wave);
    if (wave == 0) extract16(lane, sAcc, sAcc + 1024, a.out + job * 1025);
}

// ------------------------------------------------------------------------------------------------------
// key switch.  One workgroup (4 waves) per gate; wave w takes input coordinates i = w (mod 4); every lane keeps
// its 4*NX4 + 2*NX2 words of the padded output row in registers.  KSK rows are padded to 64*(4*NX4+2*NX2) words
// (n = 630: 640 words = 2560 B, 16-B aligned): per row each lane issues NX4 16-byte and NX2 8-byte loads.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sk_ksk_pad_kernel(const int32_t *__restrict__ src, long rows, int n, int row_words,
                                                          int32_t *__restrict__ dst) {
    const long r = blockIdx.x;
    if (r >= rows) return;
    for (int q = threadIdx.x; q < row_words; q += 256) dst[r * row_words + q] = q <= n ? src[r * (n + 1) + q] : 0;
}

struct KSArgs {
    const int32_t *ksk;  // [N][t][base-1][row_words]
    const int32_t *u;    // [jobs][N+1]
    int32_t *out;        // [gates][n+1]
    long gates;
    int rot_per_gate;    // 1, or 2 for MUX: input = (0, 2^29) + u1 + u2     (J/gates.jl:172-176)
    int n, t, basebit;
    int nsplit;          // > 1: grid.y workgroups share one gate (small batches) and accumulate into a zeroed output with atomics
};

template <int NX4, int NX2>
__global__ __launch_bounds__(256) void sk_keyswitch_kernel(KSArgs a) {
    constexpr int ROW = 64 * (4 * NX4 + 2 * NX2);
    __shared__ uint32_t sA[1024];
    __shared__ uint32_t sRed[3][ROW];
    const long g = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t prec_offset = 1u << (32 - (1 + a.basebit * a.t));
    const int32_t *u1 = a.u + (size_t)g * a.rot_per_gate * 1025;
    for (int q = tid; q < 1024; q += 256) {
        uint32_t v = (uint32_t)u1[q];
        if (a.rot_per_gate == 2) v += (uint32_t)u1[1025 + q];
        sA[q] = v + prec_offset;
    }
    __syncthreads();
    const int base1 = (1 << a.basebit) - 1;
    const uint32_t mask = (uint32_t)base1;
    uint32_t r4[NX4 > 0 ? NX4 : 1][4];
    uint32_t r2[NX2 > 0 ? NX2 : 1][2];
#pragma unroll
    for (int c = 0; c < NX4; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) r4[c][q] = 0;
    r2[0][0] = r2[0][1] = 0;
    const int i_lo = (int)blockIdx.y * (1024 / a.nsplit), i_hi = i_lo + 1024 / a.nsplit;
    for (int i = i_lo + wave; i < i_hi; i += 4) {
        const uint32_t ai = sA[i];
        const int32_t *rowi = a.ksk + (size_t)i * a.t * base1 * ROW;
        for (int j = 0; j < a.t; j++) {
            const uint32_t d = (ai >> (32 - (j + 1) * a.basebit)) & mask;
            if (d == 0) continue;  // wave-uniform
            const int32_t *row = rowi + ((size_t)j * base1 + (d - 1)) * ROW;
#pragma unroll
            for (int c = 0; c < NX4; c++) {
                const uint4 x = *reinterpret_cast<const uint4 *>(row + c * 256 + 4 * lane);
                r4[c][0] -= x.x; r4[c][1] -= x.y; r4[c][2] -= x.z; r4[c][3] -= x.w;
            }
            if (NX2 > 0) {
                const uint2 x = *reinterpret_cast<const uint2 *>(row + NX4 * 256 + 2 * lane);
                r2[0][0] -= x.x; r2[0][1] -= x.y;
            }
        }
    }
    if (wave > 0) {
        uint32_t *red = sRed[wave - 1];
#pragma unroll
        for (int c = 0; c < NX4; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) red[c * 256 + 4 * lane + q] = r4[c][q];
        if (NX2 > 0) {
            red[NX4 * 256 + 2 * lane] = r2[0][0];
            red[NX4 * 256 + 2 * lane + 1] = r2[0][1];
        }
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t b = (uint32_t)u1[1024];
        if (a.rot_per_gate == 2) b += (uint32_t)u1[1025 + 1024] + (1u << 29);
        int32_t *out = a.out + (size_t)g * (a.n + 1);
        auto emit = [&](int q, uint32_t v) {
            v += sRed[0][q] + sRed[1][q] + sRed[2][q];
            if (q == a.n && blockIdx.y == 0) v += b;
            if (q > a.n) return;
            if (a.nsplit == 1) out[q] = (int32_t)v;
            else atomicAdd(reinterpret_cast<unsigned int *>(out) + q, v);  // integer adds commute: still bit-exact
        };
#pragma unroll
        for (int c = 0; c < NX4; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) emit(c * 256 + 4 * lane + q, r4[c][q]);
        if (NX2 > 0) {
            emit(NX4 * 256 + 2 * lane, r2[0][0]);
            emit(NX4 * 256 + 2 * lane + 1, r2[0][1]);
        }
    }
}

// words per lane of a padded KSK row: smallest even W with 64*W >= n+1
// ------------------------------------------------------------------------------------------------------
// key switch, throughput variant (ks_basebit == 2, large batches).  sk_keyswitch_kernel reads 0.75 rows per gate and (i, j) out of
// L2 -- 15.7 MB per gate, 64 GB per 4096-gate launch: it is bound by L2 bandwidth, not by its subtractions.  Here one workgroup
// takes G gates and walks (i, j) once for all of them: the three rows KS[i][j][1..3] are loaded ONCE into registers (3 x W words per
// lane) and every gate subtracts the row its wave-uniform digit selects (or nothing), so a launch pulls 3/G rows per gate and (i, j)
// (G = 16: 4x less).  Wave w takes coordinates i = w (mod 4); the four partial sums per gate are combined with integer atomics into the
// zeroed output (adds commute: bit-exact).
// ------------------------------------------------------------------------------------------------------
template <int NX4, int NX2, int G>
__global__ __launch_bounds__(256) void sk_keyswitch_multi_kernel(KSArgs a) {
    constexpr int W = 4 * NX4 + 2 * NX2;  // words per lane of a padded row
    constexpr int ROW = 64 * W;
    __shared__ uint32_t sA[G][1024];
    const long g0 = (long)blockIdx.x * G;
    const int ng = (a.gates - g0) < G ? (int)(a.gates - g0) : G;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const uint32_t prec_offset = 1u << (32 - (1 + 2 * a.t));
    const int span = 1024 / a.nsplit, first = (int)blockIdx.y * span;
    for (int q = tid; q < G * span; q += 256) {
        const int g = q / span, i = first + q % span;
        uint32_t v = 0;
        if (g < ng) {
            const int32_t *u1 = a.u + (size_t)(g0 + g) * a.rot_per_gate * 1025;
            v = (uint32_t)u1[i];
            if (a.rot_per_gate == 2) v += (uint32_t)u1[1025 + i];
            v += prec_offset;
        }
        sA[g][i] = v;  // absent gates: all digits zero
    }
    __syncthreads();
    uint32_t r[G][W];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int q = 0; q < W; q++) r[g][q] = 0;
    const int i_lo = (int)blockIdx.y * (1024 / a.nsplit), i_hi = i_lo + 1024 / a.nsplit;
    for (int i = i_lo + wave; i < i_hi; i += 4) {
        uint32_t ai[G];
#pragma unroll
        for (int g = 0; g < G; g++) ai[g] = __builtin_amdgcn_readfirstlane(sA[g][i]);
        const int32_t *rowi = a.ksk + (size_t)i * a.t * 3 * ROW;
        for (int j = 0; j < a.t; j++) {
            const int sh = 32 - 2 * (j + 1);
            uint32_t x[3][W];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const int32_t *row = rowi + ((size_t)j * 3 + d) * ROW;
#pragma unroll
                for (int c = 0; c < NX4; c++) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(row + c * 256 + 4 * lane);
                    x[d][4 * c] = v.x, x[d][4 * c + 1] = v.y, x[d][4 * c + 2] = v.z, x[d][4 * c + 3] = v.w;
                }
                if (NX2 > 0) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(row + NX4 * 256 + 2 * lane);
                    x[d][4 * NX4] = v.x, x[d][4 * NX4 + 1] = v.y;
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                const uint32_t d = (ai[g] >> sh) & 3u;  // wave-uniform
                if (d == 1) {
#pragma unroll
                    for (int q = 0; q < W; q++) r[g][q] -= x[0][q];
                } else if (d == 2) {
#pragma unroll
                    for (int q = 0; q < W; q++) r[g][q] -= x[1][q];
                } else if (d == 3) {
#pragma unroll
                    for (int q = 0; q < W; q++) r[g][q] -= x[2][q];
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g < ng) {  // (no `break`: the loop must unroll completely so that r[][] stays in registers)
            const int32_t *u1 = a.u + (size_t)(g0 + g) * a.rot_per_gate * 1025;
            unsigned int *out = reinterpret_cast<unsigned int *>(a.out) + (size_t)(g0 + g) * (a.n + 1);
            uint32_t b = 0;
            if (wave == 0 && blockIdx.y == 0) {
                b = (uint32_t)u1[1024];
                if (a.rot_per_gate == 2) b += (uint32_t)u1[1025 + 1024] + (1u << 29);
            }
#pragma unroll
            for (int q = 0; q < W; q++) {
                const int col = q < 4 * NX4 ? (q >> 2) * 256 + 4 * lane + (q & 3) : NX4 * 256 + 2 * lane + (q - 4 * NX4);
                uint32_t v = r[g][q];
                if (col == a.n) v += b;
                if (col <= a.n) atomicAdd(out + col, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// key switch, staged variant (ks_basebit == 2, t <= 8 and a multiple of the stage depth, rows of 512 / 640 / 1152 words, >= 192 gates).  The kernel above selects a row per gate with
// wave-uniform branches; the compiler turns that chain into flag-guarded blocks with subtract-into-a-copy + moves, and the loop runs at a
// quarter of its subtraction rate.  Here the digit selects an ADDRESS: a workgroup of eight waves copies the rows KS[i][j][1..3] of four
// (i, j) at a time into LDS (twelve contiguous rows in global memory; double buffered through registers), and a lane reads its part of the
// row its gate's digit names -- digit 0 names a row of zeros -- with ds_read_b128: no branch, no select, 3/32 rows per gate and (i, j)
// out of L2.  A wave takes FOUR gates, one per 16-lane group of the LDS hardware ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
// serve one ds_read_b128 cycle each): the 16 lanes of a group read 16 consecutive pieces of ONE row = all 64 banks once, whatever the four
// digits are (eight lanes per gate met other gates' rows in their group: two-way conflicts, 1.3 ms per 4096 gates).  Partial sums of the
// coordinate ranges meet in the zeroed output with integer atomics (adds commute: bit-exact).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ks_sub(uint32_t &r, uint32_t x) { asm("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(x)); }   // in place, never re-associated
template <int W, int SJ>   // W: 16-byte pieces per lane (row of 64 W words); SJ: (i, j) pairs per stage (4; 2 for the long rows of n = 1024)
__global__ __launch_bounds__(512) void sk_keyswitch_staged_kernel(KSArgs a) {
    constexpr int ROW4 = 16 * W;             // 16-byte pieces of a padded row
    constexpr int Q = W;                     // pieces per lane: sixteen lanes share a row
    constexpr int GW = 32;                   // gates per workgroup
    constexpr int KS_CHUNK = 3;              // reads in flight behind the ones being subtracted (measured: 3 <= 5 < 10; SJ = 2 loses 4 %)
    constexpr int STAGE4 = SJ * 3 * ROW4;
    constexpr int NLD = (STAGE4 + 511) / 512;
    static_assert(NLD <= 4, "a stage is at most four rounds of 512 pieces");
    __shared__ uint4 sL[ROW4 + 2 * STAGE4];     // [row of zeros][stage 0][stage 1]
    __shared__ uint16_t sDig[GW][256];          // the t <= 8 digits of every coordinate of this workgroup's range: top 16 bits of u + offset
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int q5 = lane & 31;
    const int gl = 2 * (lane >> 5) + (int)((0xF00F0FF0u >> q5) & 1u);   // the lane's ds_read_b128 group = its gate within the wave
    const int c = q5 < 4 ? q5 : q5 < 12 ? q5 - 4 : q5 < 20 ? q5 - 8 : q5 < 28 ? q5 - 12 : q5 - 16;   // position in the group: 0 .. 15
    const long g0 = (long)blockIdx.x * GW;
    const int span = 1024 / a.nsplit, first = (int)blockIdx.y * span;   // span <= 256
    const uint32_t prec_offset = 1u << (32 - (1 + 2 * a.t));
    for (int q = tid; q < GW * span; q += 512) {
        const int g = q / span, ii = q % span;
        uint32_t v = 0;
        if (g0 + g < a.gates) {
            const int32_t *u1 = a.u + (size_t)(g0 + g) * a.rot_per_gate * 1025;
            v = (uint32_t)u1[first + ii];
            if (a.rot_per_gate == 2) v += (uint32_t)u1[1025 + first + ii];
            v += prec_offset;
        }
        sDig[g][ii] = (uint16_t)(v >> 16);  // absent gates: all digits zero
    }
    for (int q = tid; q < ROW4; q += 512) sL[q] = uint4{0u, 0u, 0u, 0u};
    const uint4 *src = reinterpret_cast<const uint4 *>(a.ksk) + (size_t)first * a.t * 3 * ROW4;
    const int NS = span * a.t / SJ;
    // stage st of the key: twelve contiguous rows; thread tid moves pieces tid + 512 k (a partial last round reads a clamped index and stores nothing).
    // Named scalars: as arrays behind an unrolled loop the pieces stayed in scratch memory.
    uint4 pre0, pre1 = uint4{0u, 0u, 0u, 0u}, pre2 = pre1, pre3 = pre1;
    const bool last_ok = 512 * NLD <= STAGE4 || tid + 512 * (NLD - 1) < STAGE4;
    const int last_idx = last_ok ? tid + 512 * (NLD - 1) : STAGE4 - 1;
#define KS_GLOAD(st)                                              \
    {                                                             \
        const uint4 *p_ = src + (size_t)(st) * STAGE4;            \
        pre0 = p_[NLD == 1 ? last_idx : tid];                     \
        if (NLD > 1) pre1 = p_[NLD == 2 ? last_idx : tid + 512];  \
        if (NLD > 2) pre2 = p_[NLD == 3 ? last_idx : tid + 1024]; \
        if (NLD > 3) pre3 = p_[last_idx];                         \
    }
#define KS_LSTORE(buf)                                                    \
    {                                                                     \
        uint4 *d_ = sL + ROW4 + (buf) * STAGE4 + tid;                     \
        if (NLD > 1 || last_ok) d_[0] = pre0;                             \
        if (NLD > 2 || (NLD == 2 && last_ok)) d_[512] = pre1;             \
        if (NLD > 3 || (NLD == 3 && last_ok)) d_[1024] = pre2;            \
        if (NLD == 4 && last_ok) d_[1536] = pre3;                         \
    }
    KS_GLOAD(0)
    KS_LSTORE(0)
    __syncthreads();
    uint4 acc[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) acc[k] = uint4{0u, 0u, 0u, 0u};
    const uint16_t *dig = sDig[wave * 4 + gl];
    for (int st = 0; st < NS; st++) {
        if (st + 1 < NS) {
            KS_GLOAD(st + 1)
        }
        const int p0 = st * SJ, ii = p0 / a.t, j0 = p0 % a.t;   // t is a multiple of SJ: the pairs of a stage belong to one coordinate
        const uint32_t hi = dig[ii];
        const uint4 *row[SJ];
#pragma unroll
        for (int pp = 0; pp < SJ; pp++) {
            const uint32_t d = (hi >> (14 - 2 * (j0 + pp))) & 3u;
            row[pp] = sL + (d ? ROW4 + (st & 1) * STAGE4 + (pp * 3 + (int)d - 1) * ROW4 : 0) + c;
        }
        // KS_CHUNK reads in flight behind the KS_CHUNK being subtracted -- not all of a stage: the memory fence stops the optimiser, the
        // scheduling barrier the instruction scheduler from clustering them
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
                    ks_sub(t.x, v.x), ks_sub(t.y, v.y), ks_sub(t.z, v.z), ks_sub(t.w, v.w);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (st + 1 < NS) {
            KS_LSTORE((st + 1) & 1)
        }
        __syncthreads();
    }
#undef KS_GLOAD
#undef KS_LSTORE
    const long g = g0 + wave * 4 + gl;
    if (g < a.gates) {
        const int32_t *u1 = a.u + (size_t)g * a.rot_per_gate * 1025;
        unsigned int *out = reinterpret_cast<unsigned int *>(a.out) + (size_t)g * (a.n + 1);
        uint32_t b = 0;
        if (blockIdx.y == 0) {
            b = (uint32_t)u1[1024];
            if (a.rot_per_gate == 2) b += (uint32_t)u1[1025 + 1024] + (1u << 29);
        }
#pragma unroll
        for (int k = 0; k < Q; k++) {
            const uint32_t v4[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int col = 4 * (c + 16 * k) + e;
                uint32_t v = v4[e];
                if (col == a.n) v += b;
                if (col <= a.n) atomicAdd(out + col, v);
            }
        }
    }
}

inline int ks_words_per_lane(int n) { return (((n + 1 + 63) / 64) + 1) & ~1; }

__global__ __launch_bounds__(256) void sk_linear_kernel(const int32_t *__restrict__ in0, int32_t *__restrict__ out, size_t words, int negate) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q < words) out[q] = negate ? (int32_t)(0u - (uint32_t)in0[q]) : in0[q];
}

}  // namespace

// ======================================================================================================
// host side
// ======================================================================================================
struct thfhe_ctx {
    thfhe_params p;
    int device = 0;
    hipStream_t stream = nullptr;
    cplx *d_bk = nullptr;     // spectral key
    int32_t *d_ksk = nullptr; // padded rows
    int ks_w = 0;             // words per lane of a padded KSK row
    long ks_multi_min_gates = 1024;  // batches of at least this many gates use sk_keyswitch_multi_kernel (rows shared by the gates of a workgroup)
    long ks_staged_min_gates = 192;  // (measured: 128 gates 0.146 ms one gate per workgroup / 0.184 staged, 256 gates 0.381 / 0.201)
    // ... and, where its shape allows, batches from this size on sk_keyswitch_staged_kernel (rows staged in LDS, the digit selects an address)
    int coop_max_jobs = 768;    // remainders (batch mod 2048) up to this many rotations use the cooperative (latency) kernel
    int ring4_max_jobs = 1024;  // ... above it and up to this many, the four-wave ring kernel (launch_br)
    cplx *d_tw = nullptr;
    // workspace
    size_t cap_jobs = 0;
    int n_pad = 0;
    int32_t *d_bara = nullptr, *d_barb = nullptr, *d_u = nullptr;
    // staging for the host-buffer API
    size_t cap_stage = 0;
    int32_t *d_in[3] = {nullptr, nullptr, nullptr};
    int32_t *d_out = nullptr;
    // gate-DAG executor: wire table and index tables (grow-only, reused by every thfhe_dag_run on this context)
    DagBuffers dag;
    size_t dag_slice = 28672;  // gates per launch of a DAG level: 14 x 2048, so that a MUX slice (2 rotations per gate) stays under the 65 535 limit of the prologue's grid
    // profiling
    bool profiling = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    std::mutex mu;
};

namespace {

int ensure_workspace(thfhe_ctx *c, size_t jobs) {
    if (jobs <= c->cap_jobs) return THFHE_OK;
    (void)hipFree(c->d_bara);
    (void)hipFree(c->d_barb);
    (void)hipFree(c->d_u);
    c->d_bara = c->d_barb = c->d_u = nullptr;
    c->cap_jobs = 0;
    THFHE_HIP(hipMalloc(&c->d_bara, jobs * c->n_pad * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_barb, jobs * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_u, jobs * 1025 * sizeof(int32_t)));
    c->cap_jobs = jobs;
    return THFHE_OK;
}

int ensure_stage(thfhe_ctx *c, size_t words) {
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

// One launch of `a.jobs` rotations on one kernel shape.
template <int L>
void launch_coop(const BRArgs &a, hipStream_t s) {
#ifdef THFHE_VARIANTS
    static const int pace = std::getenv("THFHE_COOP_PACE") ? std::atoi(std::getenv("THFHE_COOP_PACE")) : 1;
    if (pace == 0) { hipLaunchKernelGGL((sk_blind_rotate_coop_kernel<L, 0>), dim3((unsigned)a.jobs), dim3(512), 0, s, a); return; }
    if (pace == 2) { hipLaunchKernelGGL((sk_blind_rotate_coop_kernel<L, 2>), dim3((unsigned)a.jobs), dim3(512), 0, s, a); return; }
    if (pace == 4) { hipLaunchKernelGGL((sk_blind_rotate_coop_kernel<L, 4>), dim3((unsigned)a.jobs), dim3(512), 0, s, a); return; }
#endif
    hipLaunchKernelGGL((sk_blind_rotate_coop_kernel<L, 1>), dim3((unsigned)a.jobs), dim3(512), 0, s, a);
}
template <int L>
void launch_ring4(const BRArgs &a, hipStream_t s) {
    hipLaunchKernelGGL((sk_blind_rotate_ring_kernel<L, 1, 4>), dim3((unsigned)((a.jobs + 3) / 4)), dim3(256), 0, s, a);
}
template <int L>
void launch_ring8(const BRArgs &a, hipStream_t s) {
    const dim3 grid((unsigned)((a.jobs + 7) / 8)), block(512);
#ifdef THFHE_VARIANTS  // developer A/B builds only: 8 = first transpose through the LDS (variant "r")
    static const int variant = std::getenv("THFHE_RING_VARIANT") ? std::atoi(std::getenv("THFHE_RING_VARIANT")) : 0;
    if (variant == 8) { hipLaunchKernelGGL((sk_blind_rotate_ring_kernel<L, 0>), grid, block, 0, s, a); return; }
    if (variant == 3) { hipLaunchKernelGGL((sk_blind_rotate_ring_kernel<L, 3>), grid, block, 0, s, a); return; }   // forward in registers, inverse through the LDS
#endif
    hipLaunchKernelGGL((sk_blind_rotate_ring_kernel<L, 1>), grid, block, 0, s, a);
}

// Kernel choice for a batch of rotations.  Measured on one MI355X (256 CUs, SK-128; profiles/r04_time_batch.txt): a round of the
// eight-wave ring kernel takes 14.9 ms whether its workgroups hold 1 025 or 2 048 jobs between them, a round of the four-wave shape
// 9.8 ms for up to 1 024 jobs, the cooperative kernel 3.0 ms per 256 jobs.  So a batch is cut into whole rounds of 2 048 jobs on the
// eight-wave kernel plus a remainder r on the cheapest shape: cooperative up to coop_max (default 768: 8.6 ms), four-wave ring up to
// ring4_max (1 024), four-wave ring + one cooperative round up to ring4_max + 256 (12.9 ms), else one more eight-wave round.
// (3 072 rotations: 30.1 ms as one launch of 384 eight-wave workgroups, 24.8 ms as 2 048 + 1 024.)  The pieces are independent jobs
// on disjoint slices of the same arrays, launched back to back on the context's stream.
template <int L>
void launch_br(const BRArgs &a, hipStream_t s, int coop_max, int ring4_max) {
    auto piece = [&](long first, long count) {
        BRArgs b = a;
        b.bara += first * a.n_pad, b.barb += first, b.out += first * 1025, b.jobs = count;
        return b;
    };
    constexpr long kRound = 2048;   // 256 CUs x 8 jobs
    const long full = (coop_max > 0 || ring4_max > 0) ? a.jobs / kRound * kRound : a.jobs;   // both thresholds 0: everything on the eight-wave kernel
    if (full > 0) launch_ring8<L>(piece(0, full), s);
    const long r = a.jobs - full;
    if (r == 0) return;
    if (r <= coop_max) launch_coop<L>(piece(full, r), s);
    else if (r <= ring4_max) launch_ring4<L>(piece(full, r), s);
    else if (ring4_max > 0 && coop_max > 0 && r <= ring4_max + (coop_max < 256 ? coop_max : 256)) {
        launch_ring4<L>(piece(full, ring4_max), s);
        launch_coop<L>(piece(full + ring4_max, r - ring4_max), s);
    } else launch_ring8<L>(piece(full, r), s);
}

// rotations (prologue + blind rotate) of `jobs` = gates * rot_per_gate jobs into c->d_u
int enqueue_rotations(thfhe_ctx *c, int op, const int32_t *d0, const int32_t *d1, const int32_t *d2, size_t gates,
                      int rot_per_gate, int32_t mu, const int32_t *d_ops = nullptr) {
    const size_t jobs = gates * rot_per_gate;
    int rc = ensure_workspace(c, jobs);
    if (rc) return rc;
    const int n = c->p.n;
    if (c->profiling) THFHE_HIP(hipEventRecord(c->ev[0], c->stream));
    dim3 pg((unsigned)((n + 1 + 255) / 256), (unsigned)jobs);
    hipLaunchKernelGGL(sk_prologue_kernel, pg, dim3(256), 0, c->stream, d0, d1, d2, op, d_ops, rot_per_gate, n, c->n_pad,
                       ilog2(2 * c->p.N), (long)jobs, c->d_bara, c->d_barb);
    if (c->profiling) THFHE_HIP(hipEventRecord(c->ev[1], c->stream));
    BRArgs a{c->d_bk, c->d_tw, c->d_bara, c->d_barb, c->d_u, (long)jobs, n, c->n_pad, c->p.Bgbit, mu};
    switch (c->p.l) {
    case 1: launch_br<1>(a, c->stream, c->coop_max_jobs, c->ring4_max_jobs); break;
    case 2: launch_br<2>(a, c->stream, c->coop_max_jobs, c->ring4_max_jobs); break;
    case 3: launch_br<3>(a, c->stream, c->coop_max_jobs, c->ring4_max_jobs); break;
    case 4: launch_br<4>(a, c->stream, c->coop_max_jobs, c->ring4_max_jobs); break;
    default: return thfhe_fail(THFHE_E_UNSUPPORTED, "decomposition length l must be 1..4");
    }
    if (c->profiling) THFHE_HIP(hipEventRecord(c->ev[2], c->stream));
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}

int enqueue_keyswitch(thfhe_ctx *c, const int32_t *d_u, int32_t *d_out, size_t gates, int rot_per_gate, bool timed) {
    const bool staged_shape = c->p.ks_basebit == 2 && c->p.ks_t <= 8 &&
                              (((c->ks_w == 8 || c->ks_w == 10) && c->p.ks_t % 4 == 0) || (c->ks_w == 18 && c->p.ks_t % 2 == 0));
    if (staged_shape && (long)gates >= c->ks_staged_min_gates) {
        // throughput variant: rows staged in LDS for 32 gates, the digit selects an address (sk_keyswitch_staged_kernel).  The coordinates are cut
        // into 16 ranges (8 from 2 048 gates on: measured) whose partial sums meet in the zeroed output: 1 024 workgroups at 4 096 gates, 512 at 1 024.
        // Measured on MI355X, n = 630: 4 096 gates 1.01 ms (one gate per workgroup 2.57, branch-selected rows 1.69), 1 024 gates 0.34 (0.58),
        // 512 gates 0.24 (0.43), 256 gates 0.20 (0.38).
        KSArgs k{c->d_ksk, d_u, d_out, (long)gates, rot_per_gate, c->p.n, c->p.ks_t, 2, gates >= 2048 ? 8 : 16};
        THFHE_HIP(hipMemsetAsync(d_out, 0, gates * (size_t)(c->p.n + 1) * sizeof(int32_t), c->stream));
        const dim3 sgrid((unsigned)((gates + 31) / 32), (unsigned)k.nsplit), sblock(512);
        if (c->ks_w == 8) hipLaunchKernelGGL((sk_keyswitch_staged_kernel<8, 4>), sgrid, sblock, 0, c->stream, k);
        else if (c->ks_w == 10) hipLaunchKernelGGL((sk_keyswitch_staged_kernel<10, 4>), sgrid, sblock, 0, c->stream, k);
        else hipLaunchKernelGGL((sk_keyswitch_staged_kernel<18, 2>), sgrid, sblock, 0, c->stream, k);
        if (timed && c->profiling) {
            THFHE_HIP(hipEventRecord(c->ev[3], c->stream));
            c->ev_valid = true;
        }
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    if (c->p.ks_basebit == 2 && (long)gates >= c->ks_multi_min_gates && (c->ks_w == 8 || c->ks_w == 10 || c->ks_w == 18)) {
        // the rows of an (i, j) loaded once into registers for the gates of a workgroup, selected per gate by wave-uniform branches: the shapes the
        // staged kernel does not take (t not a multiple of its stage depth).  The coordinate range is cut in four so that 2048+ workgroups keep
        // ~12 waves per CU in flight.
        constexpr int kSplit = 4;
        KSArgs k{c->d_ksk, d_u, d_out, (long)gates, rot_per_gate, c->p.n, c->p.ks_t, 2, kSplit};
        THFHE_HIP(hipMemsetAsync(d_out, 0, gates * (size_t)(c->p.n + 1) * sizeof(int32_t), c->stream));
        const dim3 block(256);
        if (c->ks_w == 8) hipLaunchKernelGGL((sk_keyswitch_multi_kernel<2, 0, 8>), dim3((unsigned)((gates + 7) / 8), kSplit), block, 0, c->stream, k);
        else if (c->ks_w == 10) hipLaunchKernelGGL((sk_keyswitch_multi_kernel<2, 1, 8>), dim3((unsigned)((gates + 7) / 8), kSplit), block, 0, c->stream, k);
        else hipLaunchKernelGGL((sk_keyswitch_multi_kernel<4, 1, 4>), dim3((unsigned)((gates + 3) / 4), kSplit), block, 0, c->stream, k);
        if (timed && c->profiling) {
            THFHE_HIP(hipEventRecord(c->ev[3], c->stream));
            c->ev_valid = true;
        }
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    const int nsplit = gates <= 32 ? 16 : (gates <= 128 ? 8 : (gates <= 512 ? 2 : 1));  // fill the chip at small batch sizes
    KSArgs k{c->d_ksk, d_u, d_out, (long)gates, rot_per_gate, c->p.n, c->p.ks_t, c->p.ks_basebit, nsplit};
    if (nsplit > 1) THFHE_HIP(hipMemsetAsync(d_out, 0, gates * (size_t)(c->p.n + 1) * sizeof(int32_t), c->stream));
    const dim3 grid((unsigned)gates, (unsigned)nsplit), block(256);
    switch (c->ks_w) {
#define THFHE_KS_CASE(W, X4, X2) \
    case W: hipLaunchKernelGGL((sk_keyswitch_kernel<X4, X2>), grid, block, 0, c->stream, k); break;
        THFHE_KS_CASE(2, 0, 1) THFHE_KS_CASE(4, 1, 0) THFHE_KS_CASE(6, 1, 1) THFHE_KS_CASE(8, 2, 0) THFHE_KS_CASE(10, 2, 1)
        THFHE_KS_CASE(12, 3, 0) THFHE_KS_CASE(14, 3, 1) THFHE_KS_CASE(16, 4, 0) THFHE_KS_CASE(18, 4, 1) THFHE_KS_CASE(20, 5, 0)
        THFHE_KS_CASE(22, 5, 1)
#undef THFHE_KS_CASE
    default: return thfhe_fail(THFHE_E_UNSUPPORTED, "LWE dimension n too large for the key-switch kernel (n <= 1407)");
    }
    if (timed && c->profiling) {
        THFHE_HIP(hipEventRecord(c->ev[3], c->stream));
        c->ev_valid = true;
    }
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}

int gates_dev_locked(thfhe_ctx *c, int op, const int32_t *d0, const int32_t *d1, const int32_t *d2, int32_t *dout, size_t count) {
    if (count == 0) return THFHE_OK;
    if (count > (size_t)INT32_MAX / 4) return thfhe_fail(THFHE_E_INVALID, "count too large");
    THFHE_HIP(hipSetDevice(c->device));
    if (op == THFHE_NOT || op == THFHE_COPY) {
        const size_t words = count * (c->p.n + 1);
        hipLaunchKernelGGL(sk_linear_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream, d0, dout, words,
                           op == THFHE_NOT ? 1 : 0);
        THFHE_HIP(hipGetLastError());
        return THFHE_OK;
    }
    Lin L;
    if (!gate_lin(op, 0, L) || op == kOpIdentity) return thfhe_fail(THFHE_E_INVALID, "unknown gate opcode");
    if (!d0 || !d1 || (op == THFHE_MUX && !d2)) return thfhe_fail(THFHE_E_INVALID, "null operand");
    const int rot = op == THFHE_MUX ? 2 : 1;
    int rc = enqueue_rotations(c, op, d0, d1, d2, count, rot, 1 << 29);
    if (rc) return rc;
    return enqueue_keyswitch(c, c->d_u, dout, count, rot, true);
}

}  // namespace

extern "C" {

int thfhe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int thfhe_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 16) return thfhe_fail(THFHE_E_INVALID, "buffer of at least 16 bytes expected");
    THFHE_HIP(hipDeviceGetPCIBusId(buf, len, device));
    return THFHE_OK;
}

int thfhe_ctx_create(const thfhe_params *p, const int32_t *bk_coeff, const int32_t *ksk, int device, thfhe_ctx **out) {
    if (!p || !bk_coeff || !ksk || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    *out = nullptr;
    if (p->torus_bits != 32 || p->parties != 1) return thfhe_fail(THFHE_E_UNSUPPORTED, "thfhe_ctx_create is the single-key Torus32 path; use thfhe_mk_ctx_create");
    if (p->N != 1024 || p->k != 1) return thfhe_fail(THFHE_E_UNSUPPORTED, "only N = 1024, k = 1 is implemented");
    if (p->l < 1 || p->l > 4 || p->Bgbit < 1 || p->Bgbit > 10 || p->l * p->Bgbit > 32)
        return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= l <= 4, Bgbit <= 10 (FP64 exactness bound), l*Bgbit <= 32");
    if (p->n < 1 || p->n > 1407) return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= n <= 1407");
    if (p->ks_t < 1 || p->ks_basebit < 1 || p->ks_t * p->ks_basebit > 31) return thfhe_fail(THFHE_E_INVALID, "bad key-switch parameters");
    if (thfhe_device_count() <= device || device < 0) return thfhe_fail(THFHE_E_NO_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    THFHE_HIP(hipSetDevice(device));
    thfhe_ctx *c = new (std::nothrow) thfhe_ctx;
    if (!c) return thfhe_fail(THFHE_E_NOMEM, "out of host memory");
    c->p = *p;
    c->device = device;
    c->n_pad = (p->n + 3) & ~3;
    c->ks_w = ks_words_per_lane(p->n);
    const int row_words = 64 * c->ks_w;
    int32_t *d_coeff = nullptr, *d_raw = nullptr;  // upload staging, freed on every path
    auto fail = [&](int code) {
        (void)hipFree(d_coeff);
        (void)hipFree(d_raw);
        thfhe_ctx_destroy(c);
        return code;
    };
#define CK(expr)                                                   \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) return fail(thfhe_fail_hip(e_, #expr)); \
    } while (0)
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &e : c->ev) CK(hipEventCreate(&e));
    // twiddles
    std::vector<cplx> tw(576 + 128);  // T1[512] T2[64] lane roots[128]
    make_twiddles_1024(tw.data(), tw.data() + 512);
    make_lane_roots_1024(tw.data() + 576);
    CK(hipMalloc(&c->d_tw, tw.size() * sizeof(cplx)));
    CK(hipMemcpyAsync(c->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream));
    // bootstrapping key: upload coefficients, transform on device
    const long npolys = (long)p->n * 2 * p->l * 2;
    CK(hipMalloc(&d_coeff, (size_t)npolys * 1024 * sizeof(int32_t)));
    CK(hipMemcpyAsync(d_coeff, bk_coeff, (size_t)npolys * 1024 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CK(hipMalloc(&c->d_bk, (size_t)npolys * 1024 * sizeof(cplx)));
    hipLaunchKernelGGL(sk_key_transform_kernel, dim3((unsigned)((npolys + 3) / 4)), dim3(256), 0, c->stream, d_coeff, npolys, c->d_tw, c->d_bk);
    CK(hipGetLastError());
    // key-switching key: pad rows to 640 words
    const long rows = (long)p->N * p->ks_t * ((1 << p->ks_basebit) - 1);
    CK(hipMalloc(&d_raw, (size_t)rows * (p->n + 1) * sizeof(int32_t)));
    CK(hipMemcpyAsync(d_raw, ksk, (size_t)rows * (p->n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CK(hipMalloc(&c->d_ksk, (size_t)rows * row_words * sizeof(int32_t)));
    hipLaunchKernelGGL(sk_ksk_pad_kernel, dim3((unsigned)rows), dim3(256), 0, c->stream, d_raw, rows, p->n, row_words, c->d_ksk);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c->stream));
    (void)hipFree(d_coeff);
    (void)hipFree(d_raw);
#undef CK
    *out = c;
    return THFHE_OK;
}

void thfhe_ctx_destroy(thfhe_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    {
        std::lock_guard<std::mutex> g(c->mu);  // a call still running on another thread finishes first
        if (c->stream) (void)hipStreamSynchronize(c->stream);
    }
    (void)hipFree(c->d_bk);
    (void)hipFree(c->d_ksk);
    (void)hipFree(c->d_tw);
    (void)hipFree(c->d_bara);
    (void)hipFree(c->d_barb);
    (void)hipFree(c->d_u);
    for (auto &p : c->d_in) (void)hipFree(p);
    (void)hipFree(c->d_out);
    c->dag.release();
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int thfhe_ctx_params(const thfhe_ctx *c, thfhe_params *out) {
    if (!c || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    *out = c->p;
    return THFHE_OK;
}

void *thfhe_dev_alloc(thfhe_ctx *c, size_t bytes) {
    if (!c) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void thfhe_dev_free(thfhe_ctx *c, void *p) {
    if (c) (void)hipSetDevice(c->device);
    (void)hipFree(p);
}
int thfhe_copy_h2d(thfhe_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    THFHE_HIP(hipSetDevice(c->device));
    THFHE_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}
int thfhe_copy_d2h(thfhe_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    THFHE_HIP(hipSetDevice(c->device));
    THFHE_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}
int thfhe_reserve(thfhe_ctx *c, size_t max_count) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    return ensure_workspace(c, max_count * 2);
}
int thfhe_sync(thfhe_ctx *c) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}
#ifdef THFHE_STAMPS
int thfhe_debug_read_stamps(unsigned long long *dst, size_t count) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
int thfhe_set_coop_threshold(thfhe_ctx *c, int max_jobs) {
    if (!c || max_jobs < 0) return thfhe_fail(THFHE_E_INVALID, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    c->coop_max_jobs = max_jobs;
    return THFHE_OK;
}
int thfhe_set_ring4_threshold(thfhe_ctx *c, int max_jobs) {
    if (!c || max_jobs < 0) return thfhe_fail(THFHE_E_INVALID, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    c->ring4_max_jobs = max_jobs;
    return THFHE_OK;
}
int thfhe_set_profiling(thfhe_ctx *c, int enabled) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    c->profiling = enabled != 0;
    c->ev_valid = false;
    return THFHE_OK;
}
int thfhe_last_timings(thfhe_ctx *c, float ms[4]) {
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

int thfhe_gates_dev(thfhe_ctx *c, int op, const int32_t *d0, const int32_t *d1, const int32_t *d2, int32_t *dout, size_t count) {
    if (!c || !d0 || !dout) return thfhe_fail(THFHE_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    return gates_dev_locked(c, op, d0, d1, d2, dout, count);
}

int thfhe_gates(thfhe_ctx *c, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2, int32_t *out, size_t count) {
    if (!c || !in0 || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t words = count * (c->p.n + 1), bytes = words * sizeof(int32_t);
    int rc = ensure_stage(c, words);
    if (rc) return rc;
    const int32_t *src[3] = {in0, in1, in2};
    for (int q = 0; q < 3; q++)
        if (src[q]) THFHE_HIP(hipMemcpyAsync(c->d_in[q], src[q], bytes, hipMemcpyHostToDevice, c->stream));
    rc = gates_dev_locked(c, op, c->d_in[0], in1 ? c->d_in[1] : nullptr, in2 ? c->d_in[2] : nullptr, c->d_out, count);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));  // after all input copies: aliasing-safe
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

int thfhe_gates_mixed(thfhe_ctx *c, const int32_t *ops, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count) {
    if (!c || !ops || !in0 || !in1 || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    for (size_t g = 0; g < count; g++)
        if (ops[g] < THFHE_NAND || ops[g] > THFHE_ORYN) return thfhe_fail(THFHE_E_INVALID, "thfhe_gates_mixed takes two-input bootstrapped gates only");
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t words = count * (c->p.n + 1), bytes = words * sizeof(int32_t);
    int rc = ensure_stage(c, words);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_in[0], in0, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_in[1], in1, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_in[2], ops, count * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));  // staging buffer 2 holds the opcodes
    rc = enqueue_rotations(c, THFHE_NAND, c->d_in[0], c->d_in[1], nullptr, count, 1, 1 << 29, c->d_in[2]);
    if (rc) return rc;
    rc = enqueue_keyswitch(c, c->d_u, c->d_out, count, 1, true);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

// Gate-DAG evaluation: ASAP levelising scheduler (thfhe_dag.h) + device-resident executor.  The reference's
// applications issue these gates as sequential boots* calls (src/KNN_medical_data.cpp:127-489), once per test record (:676-691); here
// every level is one blind-rotate launch per gate class over ALL instances, the wire tables stay in HBM and nothing synchronises with
// the host between levels.
int thfhe_dag_run_batch(thfhe_ctx *c, const int32_t *inputs, size_t n_inputs, const int32_t *gates, size_t n_gates, size_t instances,
                        const int32_t *out_wires, size_t n_out, int32_t *outputs, int64_t *stats) {
    if (!c || (!inputs && n_inputs) || (!gates && n_gates) || (!outputs && n_gates) || (!out_wires && n_out)) return thfhe_fail(THFHE_E_INVALID, "null argument");
    DagPlan plan;
    int rc = dag_plan(gates, n_inputs, n_gates, [](int op) { return op == THFHE_NOT || op == THFHE_COPY ? 2 : (op == THFHE_MUX ? 1 : (op >= THFHE_NAND && op <= THFHE_ORYN ? 0 : -1)); },
                      plan);
    if (rc) return rc;
    if (stats) plan.fill_stats(stats);
    std::lock_guard<std::mutex> lk(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const int words = c->p.n + 1;
    return dag_execute(
        plan, c->dag, c->stream, words, n_inputs, n_gates, instances, inputs, out_wires, n_out, outputs, c->dag_slice,
        [&](size_t max_gates, int32_t **in, int32_t **out) {
            int r = ensure_workspace(c, 2 * max_gates);
            if (!r) r = ensure_stage(c, max_gates * words);
            in[0] = c->d_in[0], in[1] = c->d_in[1], in[2] = c->d_in[2], *out = c->d_out;
            return r;
        },
        [&](int cls, const int32_t *d_ops, size_t n) {
            const bool is_mux = cls == 1;
            int r = enqueue_rotations(c, is_mux ? THFHE_MUX : THFHE_NAND, c->d_in[0], c->d_in[1], is_mux ? c->d_in[2] : nullptr, n, is_mux ? 2 : 1, 1 << 29,
                                      is_mux ? nullptr : d_ops);
            if (!r) r = enqueue_keyswitch(c, c->d_u, c->d_out, n, is_mux ? 2 : 1, false);
            return r;
        });
}

int thfhe_set_dag_slice(thfhe_ctx *c, size_t max_gates) {
    if (!c || max_gates < 1 || max_gates > 32767) return thfhe_fail(THFHE_E_INVALID, "slice must be 1 .. 32767 gates");
    std::lock_guard<std::mutex> g(c->mu);
    c->dag_slice = max_gates;
    return THFHE_OK;
}

int thfhe_dag_run(thfhe_ctx *c, int32_t *wires, size_t n_inputs, const int32_t *gates, size_t n_gates, int64_t *stats) {
    if (!wires) return thfhe_fail(THFHE_E_INVALID, "null argument");
    return thfhe_dag_run_batch(c, wires, n_inputs, gates, n_gates, 1, nullptr, 0, wires + n_inputs * (size_t)(c ? c->p.n + 1 : 0), stats);
}

int thfhe_bootstrap_wo_keyswitch(thfhe_ctx *c, int32_t mu, const int32_t *x, int32_t *out_N1, size_t count) {
    if (!c || !x || !out_N1) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t words = count * (c->p.n + 1);
    int rc = ensure_stage(c, words);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_in[0], x, words * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    rc = enqueue_rotations(c, kOpIdentity, c->d_in[0], c->d_in[0], nullptr, count, 1, mu);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out_N1, c->d_u, count * 1025 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

int thfhe_bootstrap(thfhe_ctx *c, int32_t mu, const int32_t *x, int32_t *out, size_t count) {
    if (!c || !x || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t words = count * (c->p.n + 1);
    int rc = ensure_stage(c, words);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_in[0], x, words * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    rc = enqueue_rotations(c, kOpIdentity, c->d_in[0], c->d_in[0], nullptr, count, 1, mu);
    if (rc) return rc;
    rc = enqueue_keyswitch(c, c->d_u, c->d_out, count, 1, false);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, words * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

int thfhe_keyswitch(thfhe_ctx *c, const int32_t *in_N1, int32_t *out, size_t count) {
    if (!c || !in_N1 || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    int rc = ensure_workspace(c, count);
    if (rc) return rc;
    rc = ensure_stage(c, count * (c->p.n + 1));
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_u, in_N1, count * 1025 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    rc = enqueue_keyswitch(c, c->d_u, c->d_out, count, 1, false);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, count * (c->p.n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

}  // extern "C"
