// thfhe_ccs.hip -- the CCS multi-key scheme's gate bootstrapping (the reference's `mk_bootstrap` / `mk_gate_nand`) on gfx950.
//
// Reference path: mk_gate_nand (J/mk_gates.jl:7-13) -> mk_bootstrap (J/mk_internals.jl:855-858)
//   = mk_bootstrap_wo_keyswitch (:841-852): P*n CMuxes on an MKRLweSample accumulator (a_0 .. a_{P-1}, b) of Torus32 polynomials,
//     party-major (mk_blind_rotate :816-828); one CMux = acc + UniProduct_old(X^a acc - acc, bk[j, party], pk, crs, party)
//     (mk_mux_rotate :805-812, UniProduct_old :477-536); mk_rlwe_extract_sample (:141-148)
//   + mk_keyswitch (:714-728): party p key-switches ITS extracted mask with ks[p]; b = u.b + sum of the parts' b.
//
// UniProduct is two dependent external products:
//   stage 1   g^{-1} of the P+1 rotated-difference polynomials -> u_i = <g^{-1}(t_i), d>, v_i = <g^{-1}(t_i), pk_i.b> (i < P),
//             u_0 = <g^{-1}(t_b), d>, v_0 = -<g^{-1}(t_b), crs.a>;  acc.a_i += u_i, acc.b += u_0
//   stage 2   g^{-1}(v_i) -> acc.b += sum_i <g^{-1}(v_i), f0>,  acc.a_party += sum_i <g^{-1}(v_i), f1>
// with the engine's exact split-limb FP64 transform (two balanced 16-bit limbs per Torus32 key coefficient).  One 512-thread workgroup
// = one gate.  The (P+1) l digit rows are processed in batches of up to eight rows = eight forward transforms on eight waves, the
// batch's spectra in eight 8-KiB LDS slots that double as transpose scratch (a forward transform runs inside its own slot before
// publishing; a barrier frees the slots for the inverse transforms).  Stage-1 outputs are (group, u|v, limb) tasks; stage-2 outputs
// are eight (w0|w1, limb, row parity) roles that accumulate PARTIAL spectral sums over all batches (the inverse is linear, every
// partial sum is an exact integer polynomial) and add round(S) << 16h into the accumulator with LDS atomics (integer adds commute).
// LDS: T1 8 + acc 36 + v 36 + slots 64 = 144 KiB (sized for P <= 8).
#include <hip/hip_runtime.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"
#include "thfhe_lane.h"
#include "thfhe_mk_shared.h"

using namespace thfhe;

namespace {

constexpr int kCcsMaxParties = 8;

// coefficient-domain Torus32 polynomials -> two-limb spectra [poly][limb][slot m][lane], scaled by 1/512
__global__ __launch_bounds__(256) void ccs_key_transform_kernel(const int32_t *__restrict__ polys, long npolys, const cplx *__restrict__ tw,
                                                                 cplx *__restrict__ spec) {
    __shared__ cplx sT1[512];
    __shared__ cplx sX[4][512];
    for (int t = threadIdx.x; t < 512; t += 256) sT1[t] = tw[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const W64 w64{tw[512 + 1 * 8 + (lane & 7)]};
    const long q = (long)blockIdx.x * 4 + wave;
    if (q >= npolys) return;
    cplx zlo[8], zhi[8];
    key_limbs_to_z(lane, polys + q * 1024, zlo, zhi);
    wave_fft_fwd_s(lane, zlo, sX[wave], sT1, w64);
    wave_fft_fwd_s(lane, zhi, sX[wave], sT1, w64);
    cplx *dst = spec + q * 1024;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        dst[m * 64 + lane] = cplx{zlo[m].re * (1.0 / 512), zlo[m].im * (1.0 / 512)};
        dst[512 + m * 64 + lane] = cplx{zhi[m].re * (1.0 / 512), zhi[m].im * (1.0 / 512)};
    }
}

struct CCSArgs {
    const cplx *bk;   // [(party*n + j)][3: d, f0, f1][l][limb][512]
    const cplx *pk;   // [party][l][limb][512]
    const cplx *crs;  // [l][limb][512]
    const cplx *tw;
    const int32_t *bara;  // [jobs][w_pad]
    const int32_t *barb;
    int32_t *out;         // [jobs][P*N+1]
    long jobs;
    int parties, n, l, w_pad, Bgbit;
    int32_t mu;
};

// S += sum over the given rows of spectrum(slot) * key(level)
__device__ __forceinline__ void ccs_mac(int lane, cplx (&S)[8], const cplx *slot, const cplx *key_chunk) {
    cplx z[8], b[8];
    load8(lane, b, key_chunk);
#pragma unroll
    for (int m = 0; m < 8; m++) z[m] = slot[m * 64 + lane];
    mac8r(S, z, b);
}

__global__ __launch_bounds__(512, 2) void ccs_blind_rotate_kernel(CCSArgs a) {
    __shared__ cplx sT1[512];
    __shared__ int32_t sAcc[kCcsMaxParties + 1][1024];
    __shared__ int32_t sV[kCcsMaxParties + 1][1024];
    __shared__ cplx sSlot[8][512];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    sT1[threadIdx.x] = a.tw[threadIdx.x];
    const W64 w64{a.tw[512 + 1 * 8 + (lane & 7)]};
    const long job = blockIdx.x;
    const int P = a.parties, L = a.l, Bgbit = a.Bgbit;
    const int G = 8 / L < 4 ? 8 / L : 4;  // polynomial groups per batch: G*L <= 8 forward transforms, 4*G <= 16 stage-1 output tasks
    const uniform_i32_ptr bara = as_uniform(a.bara + job * a.w_pad);
    const uint32_t offset = decomp_offset32(L, Bgbit);
    for (int q = threadIdx.x; q < (P + 1) * 1024; q += 512) {
        int32_t v = 0;
        if (q >= P * 1024) {
            const int e = ((q - P * 1024) + a.barb[job]) & 2047;  // X^{-barb} * (mu, ..., mu)
            v = (e & 1024) ? (int32_t)(0u - (uint32_t)a.mu) : a.mu;
        }
        (&sAcc[0][0])[q] = v;
    }
    __syncthreads();

    for (int pj = 0; pj < P * a.n; pj++) {  // party-major, key index inner: J/mk_internals.jl:816-828
        const int ai = bara[pj];
        if (ai == 0) continue;
        const int a2n = ai & 2047, party = pj / a.n;
        const cplx *ue = a.bk + (size_t)pj * 3 * L * 1024;  // d | f0 | f1, each [l][limb][512]
        for (int q = threadIdx.x; q < (P + 1) * 1024; q += 512) (&sV[0][0])[q] = 0;
        // ---- stage 1: u and v of every polynomial group ------------------------------------------------------------------
        for (int g0 = 0; g0 <= P; g0 += G) {
            const int gb = (P + 1 - g0) < G ? (P + 1 - g0) : G;
            if (wave < gb * L) {
                uint32_t t[16];
                cplx z[8];
                load_rotated16(lane, sAcc[g0 + wave / L], a2n, offset, t);
                digits_to_z(t, (wave % L) + 1, Bgbit, z);
                wave_fft_fwd_s(lane, z, sSlot[wave], sT1, w64);
                wave_sync();
#pragma unroll
                for (int m = 0; m < 8; m++) sSlot[wave][m * 64 + lane] = z[m];
            }
            __syncthreads();  // spectra of the batch published; its rotated reads of the accumulator are done
            cplx S[2][8];
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int m = 0; m < 8; m++) S[s][m] = cplx{0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int task = wave + 8 * s;  // (group in batch, u|v, limb)
                if (task < gb * 4) {
                    const int gi = task >> 2, kind = (task >> 1) & 1, h = task & 1, grp = g0 + gi;
                    const cplx *key = kind == 0 ? ue : (grp < P ? a.pk + (size_t)grp * L * 1024 : a.crs);
                    for (int lv = 0; lv < L; lv++) ccs_mac(lane, S[s], sSlot[gi * L + lv], key + ((size_t)lv * 2 + h) * 512);
                }
            }
            __syncthreads();  // spectra consumed: the slots are transpose scratch from here on
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int task = wave + 8 * s;
                if (task < gb * 4) {
                    const int gi = task >> 2, kind = (task >> 1) & 1, h = task & 1, grp = g0 + gi;
                    wave_fft_inv_s(lane, S[s], sSlot[wave], sT1, w64);
                    unsigned int *dst = reinterpret_cast<unsigned int *>(kind == 0 ? sAcc[grp] : sV[grp]);
                    const bool neg = kind == 1 && grp == P;  // v_0 = -<g^{-1}(t_b), crs.a>
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        const int q = lane + 64 * m;
                        uint32_t vr = round_lo32(S[s][m].re) << (16 * h), vi = round_lo32(S[s][m].im) << (16 * h);
                        if (neg) vr = 0u - vr, vi = 0u - vi;
                        atomicAdd(dst + q, vr);
                        atomicAdd(dst + q + 512, vi);
                    }
                }
            }
            __syncthreads();  // u, v of the batch complete; scratch free
        }
        // ---- stage 2: acc.b += sum <g^{-1}(v_i), f0>, acc.a[party] += sum <g^{-1}(v_i), f1> -------------------------------------
        {
            const int o = wave >> 2, h = (wave >> 1) & 1, par = wave & 1;  // role: (w0|w1, limb, row parity)
            const cplx *key = ue + (size_t)(1 + o) * L * 1024;
            cplx S[8];
#pragma unroll
            for (int m = 0; m < 8; m++) S[m] = cplx{0.0, 0.0};
            for (int g0 = 0; g0 <= P; g0 += G) {
                const int gb = (P + 1 - g0) < G ? (P + 1 - g0) : G;
                if (wave < gb * L) {
                    uint32_t t[16];
                    cplx z[8];
                    const int32_t *v = sV[g0 + wave / L];
#pragma unroll
                    for (int m = 0; m < 16; m++) t[m] = (uint32_t)v[lane + 64 * m] + offset;
                    digits_to_z(t, (wave % L) + 1, Bgbit, z);
                    wave_fft_fwd_s(lane, z, sSlot[wave], sT1, w64);
                    wave_sync();
#pragma unroll
                    for (int m = 0; m < 8; m++) sSlot[wave][m * 64 + lane] = z[m];
                }
                __syncthreads();
                for (int slot = par; slot < gb * L; slot += 2) ccs_mac(lane, S, sSlot[slot], key + ((size_t)(slot % L) * 2 + h) * 512);
                __syncthreads();  // spectra consumed before the next batch (or the inverse transforms) reuse the slots
            }
            wave_fft_inv_s(lane, S, sSlot[wave], sT1, w64);
            unsigned int *dst = reinterpret_cast<unsigned int *>(o == 0 ? sAcc[P] : sAcc[party]);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int q = lane + 64 * m;
                atomicAdd(dst + q, round_lo32(S[m].re) << (16 * h));
                atomicAdd(dst + q + 512, round_lo32(S[m].im) << (16 * h));
            }
        }
        __syncthreads();  // accumulator updated before the next rotation reads it
    }
    // mk_rlwe_extract_sample (J/mk_internals.jl:141-148): a[:, p] = reverse_polynomial(acc.a_p), b = acc.b[0]
    int32_t *out = a.out + job * ((size_t)P * 1024 + 1);
    for (int q = threadIdx.x; q < P * 1024; q += 512) {
        const int p = q >> 10, j = q & 1023;
        out[q] = j == 0 ? sAcc[p][0] : (int32_t)(0u - (uint32_t)sAcc[p][1024 - j]);
    }
    if (threadIdx.x == 0) out[(size_t)P * 1024] = sAcc[P][0];
}


// ------------------------------------------------------------------------------------------------------
// The 16-party set (mktfhe_parameters_16party, J/mk_api.jl:185-191: l = 12, Bgbit = 2, 17 accumulator polynomials, 204 digit rows per stage):
// same arithmetic, other shape.  The accumulator (68 KiB) stays in LDS; the v polynomials of a step live in global memory (`vbuf`, 68 KiB per
// job: written with integer atomics in stage 1, read back once in stage 2 -- both through the L2, agent-scope accesses); the digit rows of ONE
// polynomial go through the eight slots in level batches of up to eight, the four (u|v, limb) output tasks of the group accumulating partial
// spectral sums over the level batches; stage 2 walks all (P+1) l rows in batches of eight with the eight (w0|w1, limb, parity) roles of the
// kernel above.  Correctness first: at 8 960 CMuxes of ~420 transforms per gate this set is seconds per gate on any hardware.
// LDS: T1 8 + acc 68 + slots 64 = 140 KiB.
// ------------------------------------------------------------------------------------------------------
constexpr int kCcsWideMaxParties = 16;

__global__ __launch_bounds__(512, 2) void ccs_blind_rotate_wide_kernel(CCSArgs a, int32_t *__restrict__ vbuf) {
    __shared__ cplx sT1[512];
    __shared__ int32_t sAcc[kCcsWideMaxParties + 1][1024];
    __shared__ cplx sSlot[8][512];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    sT1[threadIdx.x] = a.tw[threadIdx.x];
    const W64 w64{a.tw[512 + 1 * 8 + (lane & 7)]};
    const long job = blockIdx.x;
    const int P = a.parties, L = a.l, Bgbit = a.Bgbit;
    const uniform_i32_ptr bara = as_uniform(a.bara + job * a.w_pad);
    const uint32_t offset = decomp_offset32(L, Bgbit);
    int32_t *V = vbuf + (size_t)job * (P + 1) * 1024;
    for (int q = threadIdx.x; q < (P + 1) * 1024; q += 512) {
        int32_t v = 0;
        if (q >= P * 1024) {
            const int e = ((q - P * 1024) + a.barb[job]) & 2047;  // X^{-barb} * (mu, ..., mu)
            v = (e & 1024) ? (int32_t)(0u - (uint32_t)a.mu) : a.mu;
        }
        (&sAcc[0][0])[q] = v;
    }
    __syncthreads();

    for (int pj = 0; pj < P * a.n; pj++) {  // party-major, key index inner: J/mk_internals.jl:816-828
        const int ai = bara[pj];
        if (ai == 0) continue;
        const int a2n = ai & 2047, party = pj / a.n;
        const cplx *ue = a.bk + (size_t)pj * 3 * L * 1024;  // d | f0 | f1, each [l][limb][512]
        for (int q = threadIdx.x; q < (P + 1) * 1024; q += 512) __hip_atomic_store(V + q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __syncthreads();
        // ---- stage 1: u and v of every polynomial ---------------------------------------------------------------------------
        for (int grp = 0; grp <= P; grp++) {
            const int kind = (wave >> 1) & 1, h = wave & 1;   // output task of waves 0..3: (u|v, limb)
            const cplx *key = kind == 0 ? ue : (grp < P ? a.pk + (size_t)grp * L * 1024 : a.crs);
            cplx S[8];
#pragma unroll
            for (int m = 0; m < 8; m++) S[m] = cplx{0.0, 0.0};
            for (int lb = 0; lb < L; lb += 8) {
                const int nb = L - lb < 8 ? L - lb : 8;
                if (wave < nb) {
                    uint32_t t[16];
                    cplx z[8];
                    load_rotated16(lane, sAcc[grp], a2n, offset, t);
                    digits_to_z(t, lb + wave + 1, Bgbit, z);
                    wave_fft_fwd_s(lane, z, sSlot[wave], sT1, w64);
                    wave_sync();
#pragma unroll
                    for (int m = 0; m < 8; m++) sSlot[wave][m * 64 + lane] = z[m];
                }
                __syncthreads();  // spectra of the level batch published
                if (wave < 4)
                    for (int lv = 0; lv < nb; lv++) ccs_mac(lane, S, sSlot[lv], key + ((size_t)(lb + lv) * 2 + h) * 512);
                __syncthreads();  // spectra consumed: the slots are free for the next level batch / transpose scratch
            }
            if (wave < 4) {   // every rotated read of acc[grp] is done (barrier above): its u may be added now
                wave_fft_inv_s(lane, S, sSlot[wave], sT1, w64);
                const bool neg = kind == 1 && grp == P;  // v_0 = -<g^{-1}(t_b), crs.a>
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int q = lane + 64 * m;
                    uint32_t vr = round_lo32(S[m].re) << (16 * h), vi = round_lo32(S[m].im) << (16 * h);
                    if (neg) vr = 0u - vr, vi = 0u - vi;
                    if (kind == 0) {
                        atomicAdd(reinterpret_cast<unsigned int *>(sAcc[grp]) + q, vr);
                        atomicAdd(reinterpret_cast<unsigned int *>(sAcc[grp]) + q + 512, vi);
                    } else {
                        __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(V + (size_t)grp * 1024) + q, vr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(V + (size_t)grp * 1024) + q + 512, vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();  // u of the group complete; scratch free
        }
        __threadfence();
        __syncthreads();  // every v is in the L2
        // ---- stage 2: acc.b += sum <g^{-1}(v_i), f0>, acc.a[party] += sum <g^{-1}(v_i), f1> -------------------------------------
        {
            const int o = wave >> 2, h = (wave >> 1) & 1, par = wave & 1;  // role: (w0|w1, limb, row parity)
            const cplx *key = ue + (size_t)(1 + o) * L * 1024;
            cplx S[8];
#pragma unroll
            for (int m = 0; m < 8; m++) S[m] = cplx{0.0, 0.0};
            const int rows = (P + 1) * L;
            for (int r0 = 0; r0 < rows; r0 += 8) {
                const int nb = rows - r0 < 8 ? rows - r0 : 8;
                if (wave < nb) {
                    const int r = r0 + wave;
                    uint32_t t[16];
                    cplx z[8];
                    const int32_t *v = V + (size_t)(r / L) * 1024;
#pragma unroll
                    for (int m = 0; m < 16; m++) t[m] = (uint32_t)__hip_atomic_load(v + lane + 64 * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + offset;
                    digits_to_z(t, (r % L) + 1, Bgbit, z);
                    wave_fft_fwd_s(lane, z, sSlot[wave], sT1, w64);
                    wave_sync();
#pragma unroll
                    for (int m = 0; m < 8; m++) sSlot[wave][m * 64 + lane] = z[m];
                }
                __syncthreads();
                for (int slot = par; slot < nb; slot += 2) ccs_mac(lane, S, sSlot[slot], key + ((size_t)((r0 + slot) % L) * 2 + h) * 512);
                __syncthreads();  // spectra consumed before the next batch (or the inverse transforms) reuse the slots
            }
            wave_fft_inv_s(lane, S, sSlot[wave], sT1, w64);
            unsigned int *dst = reinterpret_cast<unsigned int *>(o == 0 ? sAcc[P] : sAcc[party]);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int q = lane + 64 * m;
                atomicAdd(dst + q, round_lo32(S[m].re) << (16 * h));
                atomicAdd(dst + q + 512, round_lo32(S[m].im) << (16 * h));
            }
        }
        __syncthreads();  // accumulator updated before the next rotation reads it
    }
    // mk_rlwe_extract_sample (J/mk_internals.jl:141-148): a[:, p] = reverse_polynomial(acc.a_p), b = acc.b[0]
    int32_t *out = a.out + job * ((size_t)P * 1024 + 1);
    for (int q = threadIdx.x; q < P * 1024; q += 512) {
        const int p = q >> 10, j = q & 1023;
        out[q] = j == 0 ? sAcc[p][0] : (int32_t)(0u - (uint32_t)sAcc[p][1024 - j]);
    }
    if (threadIdx.x == 0) out[(size_t)P * 1024] = sAcc[P][0];
}

}  // namespace

struct thfhe_ccs_ctx {
    thfhe_params p;
    int device = 0;
    hipStream_t stream = nullptr;
    cplx *d_bk = nullptr, *d_pk = nullptr, *d_crs = nullptr, *d_tw = nullptr;
    int32_t *d_ksk = nullptr;
    int row_words = 0, w_pad = 0, words = 0;
    size_t cap = 0;
    int32_t *d_bara = nullptr, *d_barb = nullptr, *d_u = nullptr, *d_in[2] = {nullptr, nullptr}, *d_out = nullptr;
    int32_t *d_v = nullptr;   // wide shape (more than 8 parties or 8 levels): the v polynomials of a step, int32[jobs][P+1][1024]
    bool wide = false;
    std::mutex mu;
};

namespace {
int ccs_ensure(thfhe_ccs_ctx *c, size_t jobs) {
    if (jobs <= c->cap) return THFHE_OK;
    for (int32_t **q : {&c->d_bara, &c->d_barb, &c->d_u, &c->d_in[0], &c->d_in[1], &c->d_out, &c->d_v}) {
        (void)hipFree(*q);
        *q = nullptr;
    }
    c->cap = 0;
    const size_t rec = (size_t)c->words + 1;
    THFHE_HIP(hipMalloc(&c->d_bara, jobs * c->w_pad * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_barb, jobs * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_u, jobs * ((size_t)c->p.parties * 1024 + 1) * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_in[0], jobs * rec * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_in[1], jobs * rec * sizeof(int32_t)));
    THFHE_HIP(hipMalloc(&c->d_out, jobs * rec * sizeof(int32_t)));
    if (c->wide) THFHE_HIP(hipMalloc(&c->d_v, jobs * ((size_t)c->p.parties + 1) * 1024 * sizeof(int32_t)));
    c->cap = jobs;
    return THFHE_OK;
}

int ccs_run(thfhe_ccs_ctx *c, MKLin L, const int32_t *in0, const int32_t *in1, int32_t mu, int32_t *out, size_t count) {
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    int rc = ccs_ensure(c, count);
    if (rc) return rc;
    const size_t rec = (size_t)c->words + 1, bytes = count * rec * sizeof(int32_t);
    THFHE_HIP(hipMemcpyAsync(c->d_in[0], in0, bytes, hipMemcpyHostToDevice, c->stream));
    if (in1) THFHE_HIP(hipMemcpyAsync(c->d_in[1], in1, bytes, hipMemcpyHostToDevice, c->stream));
    dim3 pg((unsigned)((c->words + 1 + 255) / 256), (unsigned)count);
    hipLaunchKernelGGL(mk_prologue_kernel, pg, dim3(256), 0, c->stream, c->d_in[0], in1 ? c->d_in[1] : c->d_in[0], c->d_in[0], L, L, (const int32_t *)nullptr, 1,
                       c->words, c->w_pad, 11, (long)count, c->d_bara, c->d_barb);
    CCSArgs a{c->d_bk, c->d_pk, c->d_crs, c->d_tw, c->d_bara, c->d_barb, c->d_u, (long)count, c->p.parties, c->p.n, c->p.l, c->w_pad, c->p.Bgbit, mu};
    if (c->wide) hipLaunchKernelGGL(ccs_blind_rotate_wide_kernel, dim3((unsigned)count), dim3(512), 0, c->stream, a, c->d_v);
    else hipLaunchKernelGGL(ccs_blind_rotate_kernel, dim3((unsigned)count), dim3(512), 0, c->stream, a);
    MKKSArgs k{c->d_ksk, c->d_u, c->d_out, (long)count, c->p.n, c->p.ks_t, c->p.ks_basebit, c->p.parties, c->row_words, 1024, c->p.parties * 1024 + 1, 1024};
    const int nsplit = count * c->p.parties <= 64 ? 16 : (count * c->p.parties <= 256 ? 4 : 1);
    THFHE_HIP(hipMemsetAsync(c->d_out, 0, bytes, c->stream));
    mk_launch_keyswitch(k, nsplit, c->stream);
    THFHE_HIP(hipGetLastError());
    THFHE_HIP(hipMemcpyAsync(out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}
}  // namespace

extern "C" {

int thfhe_ccs_ctx_create(const thfhe_params *p, const int32_t *bk, const int32_t *pk, const int32_t *crs, const int32_t *ksk, int device,
                         thfhe_ccs_ctx **out) {
    if (!p || !bk || !pk || !crs || !ksk || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    *out = nullptr;
    if (p->torus_bits != 32) return thfhe_fail(THFHE_E_UNSUPPORTED, "thfhe_ccs_ctx_create is the Torus32 CCS multi-key path");
    if (p->N != 1024 || p->k != 1) return thfhe_fail(THFHE_E_UNSUPPORTED, "only N = 1024, k = 1 is implemented");
    if (p->parties < 1 || p->parties > kCcsWideMaxParties) return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= parties <= 16");
    if (p->l < 1 || p->l > 16 || p->Bgbit < 1 || p->Bgbit > 10 || p->l * p->Bgbit > 32)
        return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= l <= 16, Bgbit <= 10, l*Bgbit <= 32");
    if ((long)(p->parties + 1) * p->l * (1L << (p->Bgbit - 1)) > 3072)
        return thfhe_fail(THFHE_E_UNSUPPORTED, "(parties+1) * l * 2^(Bgbit-1) exceeds the FP64 exactness bound of the stage-2 sums");
    if (p->n < 1 || p->n > 767) return thfhe_fail(THFHE_E_UNSUPPORTED, "need 1 <= n <= 767");
    if (p->ks_t < 1 || p->ks_basebit < 1 || p->ks_t * p->ks_basebit > 31) return thfhe_fail(THFHE_E_INVALID, "bad key-switch parameters");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return thfhe_fail(THFHE_E_NO_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    THFHE_HIP(hipSetDevice(device));
    thfhe_ccs_ctx *c = new (std::nothrow) thfhe_ccs_ctx;
    if (!c) return thfhe_fail(THFHE_E_NOMEM, "out of host memory");
    c->p = *p;
    c->device = device;
    c->wide = p->parties > kCcsMaxParties || p->l > 8;   // the 16-party shape: ccs_blind_rotate_wide_kernel
    c->words = p->parties * p->n;
    c->w_pad = (c->words + 3) & ~3;
    c->row_words = 128 * ((p->n + 1 + 127) / 128);
    int32_t *d_coeff = nullptr, *d_raw = nullptr;  // upload staging, freed on every path
    auto fail = [&](int code) {
        (void)hipFree(d_coeff);
        (void)hipFree(d_raw);
        thfhe_ccs_ctx_destroy(c);
        return code;
    };
#define CK(expr)                                                      \
    do {                                                              \
        hipError_t e_ = (expr);                                       \
        if (e_ != hipSuccess) return fail(thfhe_fail_hip(e_, #expr)); \
    } while (0)
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    std::vector<cplx> tw(576);
    make_twiddles_1024(tw.data(), tw.data() + 512);
    CK(hipMalloc(&c->d_tw, tw.size() * sizeof(cplx)));
    CK(hipMemcpyAsync(c->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream));
    struct Tab { const int32_t *src; long npolys; cplx **dst; };
    const Tab tabs[3] = {{bk, (long)p->parties * p->n * 3 * p->l, &c->d_bk}, {pk, (long)p->parties * p->l, &c->d_pk}, {crs, (long)p->l, &c->d_crs}};
    for (const Tab &t : tabs) {
        CK(hipMalloc(&d_coeff, (size_t)t.npolys * 1024 * sizeof(int32_t)));
        CK(hipMemcpyAsync(d_coeff, t.src, (size_t)t.npolys * 1024 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        CK(hipMalloc(t.dst, (size_t)t.npolys * 1024 * sizeof(cplx)));
        hipLaunchKernelGGL(ccs_key_transform_kernel, dim3((unsigned)((t.npolys + 3) / 4)), dim3(256), 0, c->stream, d_coeff, t.npolys, c->d_tw, *t.dst);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(c->stream));
        (void)hipFree(d_coeff);
        d_coeff = nullptr;
    }
    const long rows = (long)p->parties * 1024 * p->ks_t * ((1 << p->ks_basebit) - 1);
    CK(hipMalloc(&d_raw, (size_t)rows * (p->n + 1) * sizeof(int32_t)));
    CK(hipMemcpyAsync(d_raw, ksk, (size_t)rows * (p->n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CK(hipMalloc(&c->d_ksk, (size_t)rows * c->row_words * sizeof(int32_t)));
    hipLaunchKernelGGL(mk_ksk_pad_kernel, dim3((unsigned)rows), dim3(256), 0, c->stream, d_raw, rows, p->n, c->row_words, c->d_ksk);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c->stream));
    (void)hipFree(d_raw);
#undef CK
    *out = c;
    return THFHE_OK;
}

void thfhe_ccs_ctx_destroy(thfhe_ccs_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *q : {(void *)c->d_bk, (void *)c->d_pk, (void *)c->d_crs, (void *)c->d_tw, (void *)c->d_ksk, (void *)c->d_bara, (void *)c->d_barb, (void *)c->d_u,
                    (void *)c->d_in[0], (void *)c->d_in[1], (void *)c->d_out, (void *)c->d_v})
        (void)hipFree(q);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int thfhe_ccs_gates(thfhe_ccs_ctx *c, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count) {
    if (!c || !in0 || !in1 || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    MKLin L;
    if (!(op == THFHE_NAND || op == THFHE_OR || op == THFHE_AND || op == THFHE_XOR) || !mk_gate_lin(op, 0, L))
        return thfhe_fail(THFHE_E_INVALID, "thfhe_ccs_gates takes NAND (the reference's mk_gate_nand) / AND / OR / XOR");
    return ccs_run(c, L, in0, in1, 1 << 29, out, count);
}

int thfhe_ccs_bootstrap(thfhe_ccs_ctx *c, int32_t mu, const int32_t *x, int32_t *out, size_t count) {
    if (!c || !x || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    MKLin L;
    mk_gate_lin(kOpIdentity, 0, L);
    return ccs_run(c, L, x, nullptr, mu, out, count);
}

}  // extern "C"
