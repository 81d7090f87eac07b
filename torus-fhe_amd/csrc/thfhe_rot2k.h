// thfhe_rot2k.h -- blind rotation of ONE RLWE accumulator over the Torus64 ring of degree 2048 by a TGSW-shaped key with ANY number of
// digit rows (they pass through the LDS in batches of six) and one- or two-part digits.  Shared by
//   thfhe_kms.hip   the KMS scheme's TLev rotation (mk_ith_blind_rotate, J/new_mk_internals.jl:210-225)
//   thfhe_mk.hip    the 3-gen sets whose l x digit parts exceed what the one-pass N = 2048 kernel holds in LDS (the 256-party set: l = 2,
//                   Bgbit = 18 -> two 9-bit parts per level, eight row parts; J/mk_api.jl:304-310)
// Included inside each translation unit's anonymous namespace (kernels are per-TU copies).
#pragma once

__device__ __forceinline__ void kms_pin() { asm volatile("" ::: "memory"); }

// torus polynomials int64[npolys][2048] -> limb spectra [poly][limb h][half][512], scaled by 1/1024 (one wave per (poly, limb))
__global__ __launch_bounds__(256) void kms_key_transform_kernel(const int64_t *__restrict__ polys, long npolys, const cplx *__restrict__ tw,
                                                                 cplx *__restrict__ spec) {
    __shared__ cplx sT1[2][512];
    __shared__ cplx sX[4][512];
    for (int t = threadIdx.x; t < 1024; t += 256) (&sT1[0][0])[t] = tw[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const W64 w64{tw[1024 + 1 * 8 + (lane & 7)]};
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= npolys * 4) return;
    cplx z[16], y0[8], y1[8];
    key_limbs64_to_z16(lane, polys + (item >> 2) * 2048, (int)(item & 3), z);
    split2048(z, y0, y1);
    wave_fft_fwd_t<1>(lane, y0, sX[wave], sT1[0], w64);
    wave_fft_fwd_t<5>(lane, y1, sX[wave], sT1[1], w64);
    cplx *dst = spec + (size_t)item * 1024;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        dst[m * 64 + lane] = cplx{y0[m].re * (1.0 / 1024), y0[m].im * (1.0 / 1024)};
        dst[512 + m * 64 + lane] = cplx{y1[m].re * (1.0 / 1024), y1[m].im * (1.0 / 1024)};
    }
}

struct KmsBRArgs {
    const cplx *bk;       // the party's key spectra [j][row part][column o][limb h][half][512]
    const cplx *tw;
    const int32_t *bara;  // [gates][bara_stride]: mod-switched mask words, n of them used per gate
    int64_t *acc_out;     // [gates * l_lev][2][2048]: TLev sample s of gate g at job g * l_lev + s
    const int64_t *acc_in;  // null: TLev accumulators start at the trivial gadget samples; else one RLWE sample per gate (l_lev = 1) starts here
    long jobs;
    int n, lg, bg, parts, lo_bits, l_lev, bg_lev;
    int bara_stride;      // words between the mask rows of consecutive gates (>= n)
};

__global__ __launch_bounds__(512, 2) void kms_tlev_rotate_kernel(KmsBRArgs a) {
    constexpr int BATCH = 6;  // row parts whose spectra sit in the LDS at the same time (16 KiB each)
    __shared__ cplx sT1[2][512];
    __shared__ int64_t sAcc[4096];
    __shared__ cplx sSpec[BATCH * 1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int t = threadIdx.x; t < 1024; t += 512) (&sT1[0][0])[t] = a.tw[t];
    const W64 w64{a.tw[1024 + 1 * 8 + (lane & 7)]};
    const long job = blockIdx.x;
    const long gate = job / a.l_lev;
    const int sample = (int)(job % a.l_lev);
    const int32_t *bara = a.bara + gate * a.bara_stride;
    const int lg = a.lg, bg = a.bg, parts = a.parts, lo_bits = a.lo_bits;
    const int RP = 2 * lg * parts;
    uint64_t offset = 0;
    for (int p = 1; p <= lg; p++) offset += (1ull << (bg - 1)) << (64 - p * bg);
    // tlev_trivial_int(levpar, lwepar, 1): mask 0, body = gadget value of level `sample` on the constant coefficient     (J/tlev.jl:37-66)
    if (a.acc_in) {   // mk_single_blind_rotate (J/new_mk_internals.jl:226-238): the caller's RLWE sample
        for (int q = threadIdx.x; q < 4096; q += 512) sAcc[q] = a.acc_in[job * 4096 + q];
    } else {
        for (int q = threadIdx.x; q < 4096; q += 512) sAcc[q] = 0;
        __syncthreads();
        if (threadIdx.x == 0) sAcc[2048] = (int64_t)(1ull << (64 - (sample + 1) * a.bg_lev));
    }
    __syncthreads();
    const int o = wave >> 2, h = wave & 3;
    unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc) + o * 2048;
    auto chunk = [&](int step, int rp, int half) { return a.bk + (((((size_t)step * RP + rp) * 2 + o) * 4 + h) * 2 + half) * 512; };

    int i = 0;
    while (i < a.n && bara[i] == 0) i++;
    while (i < a.n) {
        const int a2n = __builtin_amdgcn_readfirstlane(bara[i]) & 4095;   // uniform over the workgroup
        int inext = i + 1;
        while (inext < a.n && bara[inext] == 0) inext++;
        cplx S0[8], S1[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S0[m] = S1[m] = cplx{0.0, 0.0};
        for (int b0 = 0; b0 < RP; b0 += BATCH) {
            const int nb = RP - b0 < BATCH ? RP - b0 : BATCH;
            if (wave < nb) {
                const int rp = b0 + wave, r = rp / parts, part = rp % parts;
                const int64_t *ap = sAcc + (r / lg) * 2048;
                const int shift = 64 - ((r % lg) + 1) * bg;
                const uint64_t mask = (1ull << bg) - 1ull;
                const int32_t half_bg = 1 << (bg - 1), half_lo = 1 << (lo_bits - 1), mask_lo = (1 << lo_bits) - 1;
                constexpr double R = 0.70710678118654752440;
                cplx y0[8], y1[8];
                int a2n_b = a2n;
                asm volatile("" : "+s"(a2n_b));   // opaque per batch: the 32 rotated indices and sign predicates are recomputed, not kept (spilled) across batches
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    double d[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint64_t v = rot_minus_self64_n<2048>(ap, lane + 64 * m + 512 * q, a2n_b) + offset;
                        int32_t dg = (int32_t)((v >> shift) & mask) - half_bg;                     // decompose, J/tgsw.jl:112-138 (64-bit words)
                        if (parts == 2) {
                            const int32_t lo = ((dg + half_lo) & mask_lo) - half_lo;               // balanced low part
                            dg = part ? (dg - lo) >> lo_bits : lo;
                        }
                        d[q] = (double)dg;
                    }
                    const cplx w{(d[1] - d[3]) * R, (d[1] + d[3]) * R};
                    y0[m] = cplx{d[0] + w.re, d[2] + w.im};
                    y1[m] = cplx{d[0] - w.re, d[2] - w.im};
                    if (m & 1) kms_pin();   // at most 8 of the 32 rotated 64-bit reads in flight: the partial spectra keep 64 registers busy here
                }
                cplx *xb = sSpec + wave * 1024;
                // the swizzled LDS slot maps are a few integer operations per address: recomputed here (opaque lane) instead of hoisted out
                // of the CMux loop by the compiler, which then spilled the 25 address registers to scratch
                int ln = lane;
                asm volatile("" : "+v"(ln));
                wave_fft_fwd_t<1>(ln, y0, xb, sT1[0], w64);
                wave_fft_fwd_t<5>(ln, y1, xb, sT1[1], w64);
                wave_sync();
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    xb[m * 64 + lane] = y0[m];
                    xb[512 + m * 64 + lane] = y1[m];
                }
            }
            // key chunks of this batch: requested after the transforms (the partial spectra S0 / S1 stay alive across the batches of a step,
            // so there are no registers for chunks in flight under the transforms), then one row part ahead inside the batch
            cplx bA[8], bB[8];
            load8(lane, bA, chunk(i, b0, 0));
            load8(lane, bB, chunk(i, b0, 1));
            __syncthreads();  // this batch's spectra published (and, for the first batch, every rotated read of the accumulator done)
            for (int q = 0; q < nb; q++) {
                const int rp = b0 + q;
                const int rn = q + 1 < nb ? rp + 1 : rp;   // the last row part of a batch re-requests itself (unconditional loads)
                cplx z[8];
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[q * 1024 + m * 64 + lane];
                mac8r(S0, z, bA);
                kms_pin();
                load8(lane, bA, chunk(i, rn, 0));
                kms_pin();
#pragma unroll
                for (int m = 0; m < 8; m++) z[m] = sSpec[q * 1024 + 512 + m * 64 + lane];
                mac8r(S1, z, bB);
                kms_pin();
                load8(lane, bB, chunk(i, rn, 1));
                kms_pin();
            }
            __syncthreads();  // spectra consumed: the area is free for the next batch / the inverse transforms
        }
        {
            cplx *xb = sSpec + wave * 512;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            wave_fft_inv_t<1>(ln, S0, xb, sT1[0], w64);
            wave_fft_inv_t<5>(ln, S1, xb, sT1[1], w64);
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
        }
        __syncthreads();  // accumulator updated and scratch free before the next rotation
        i = inext;
    }
    for (int q = threadIdx.x; q < 4096; q += 512) a.acc_out[job * 4096 + q] = sAcc[q];
}

// ------------------------------------------------------------------------------------------------------
// Two jobs per workgroup (batches above one job per CU).  The one-job kernel above is bound by the key bytes a CU pulls through its
// vector-memory path (2 l_gsw x parts x 2 x 4 chunks of 16 KiB per CMux: 1.5 MiB for the 2-party KMS set); here every key chunk multiplies
// the digit spectra of TWO jobs (the TLev samples of a gate share key and rotation; any two jobs of a launch share the key).  Two
// accumulators (64 KiB) leave 96 KiB for spectra = the HALF spectra (one twist) of six row parts of both jobs, so a batch of row parts
// runs as two half passes (even outputs, twist 1, into S0; odd outputs, twist 5, into S1) from digits that are extracted once: staged as
// 64-bit words in the (still free) spectrum area, cut and packed into 16-bit fields of 16 registers per forward task.  With more than six
// row parts the partial spectra S1 of both jobs (64 VGPRs) leave the registers between batches (a 128 KiB slice of `park` per workgroup,
// stored after a batch's second multiply, fetched back in front of the next one): the forward transforms of the later batches would not
// fit next to all four partial spectra.  Table-free twisted transforms ("qs" form); barriers wait for the LDS only.
// ------------------------------------------------------------------------------------------------------
template <int VM>
__device__ __forceinline__ void lds_barrier() {   // workgroup barrier that orders LDS traffic and leaves up to VM vector-memory loads in flight
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}
// forward half transform of task f (slot f) from its packed digits
template <int HALF>
__device__ __forceinline__ void r2k_transform(int f, int lane, cplx *sSpec, const uint32_t (&pk)[8][2], const LaneRoots &roots, const W64 &w64) {
    constexpr double R = 0.70710678118654752440;
    cplx y[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const double d0 = (double)((int32_t)(pk[m][0] << 16) >> 16), d1 = (double)((int32_t)pk[m][0] >> 16);
        const double d2 = (double)((int32_t)(pk[m][1] << 16) >> 16), d3 = (double)((int32_t)pk[m][1] >> 16);
        // split2048 of z[m] = (d0, d2), z[m + 8] = (d1, d3): y = z[m] +- e^{i pi/4} z[m + 8]
        const cplx w{(d1 - d3) * R, (d1 + d3) * R};
        y[m] = HALF == 0 ? cplx{d0 + w.re, d2 + w.im} : cplx{d0 - w.re, d2 - w.im};
    }
    cplx *slot = sSpec + f * 512;   // the transpose runs inside the task's own, not yet published, spectrum slot
    const int ln = opaque_lane(lane);   // slot addresses are formed here, not hoisted out of the step loop and spilled
    wave_fft_fwd_tq<HALF == 0 ? 1 : 5>(ln, y, slot, LaneRoots{opaque_cplx(roots.b), opaque_cplx(roots.s)}, w64);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) slot[m * 64 + ln] = y[m];
}

// two forward tasks of one wave, step by step next to each other: the second task's arithmetic fills the first one's LDS round trips
template <int HALF>
__device__ __forceinline__ void r2k_transform_two(int f0, int f1, int lane, cplx *sSpec, const uint32_t (&pk0)[8][2], const uint32_t (&pk1)[8][2],
                                                  const LaneRoots &roots, const W64 &w64) {
    constexpr double R = 0.70710678118654752440;
    constexpr int T = HALF == 0 ? 1 : 5;
    cplx y0[8], y1[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const double a0 = (double)((int32_t)(pk0[m][0] << 16) >> 16), a1 = (double)((int32_t)pk0[m][0] >> 16);
        const double a2 = (double)((int32_t)(pk0[m][1] << 16) >> 16), a3 = (double)((int32_t)pk0[m][1] >> 16);
        const double b0 = (double)((int32_t)(pk1[m][0] << 16) >> 16), b1 = (double)((int32_t)pk1[m][0] >> 16);
        const double b2 = (double)((int32_t)(pk1[m][1] << 16) >> 16), b3 = (double)((int32_t)pk1[m][1] >> 16);
        const cplx wa{(a1 - a3) * R, (a1 + a3) * R}, wb{(b1 - b3) * R, (b1 + b3) * R};
        y0[m] = HALF == 0 ? cplx{a0 + wa.re, a2 + wa.im} : cplx{a0 - wa.re, a2 - wa.im};
        y1[m] = HALF == 0 ? cplx{b0 + wb.re, b2 + wb.im} : cplx{b0 - wb.re, b2 - wb.im};
    }
    const int ln = opaque_lane(lane);
    cplx *slot0 = sSpec + f0 * 512, *slot1 = sSpec + f1 * 512;
    const LaneRoots r{opaque_cplx(roots.b), opaque_cplx(roots.s)};
    wave_fft_fwd_tq_two<T, T>(ln, y0, y1, slot0, slot1, r, r, w64);
    wave_sync();
#pragma unroll
    for (int m = 0; m < 8; m++) slot0[m * 64 + ln] = y0[m];
#pragma unroll
    for (int m = 0; m < 8; m++) slot1[m * 64 + ln] = y1[m];
}

struct R2KDigits {
    int lg, bg, parts, lo_bits;
    uint64_t offset;
};
// 64-bit words (X^a acc - acc) + offset of the accumulator polynomials p_lo .. p_lo + npoly - 1 of both jobs -> stage[(job * 2 + poly) * 2048 + c]
__device__ __forceinline__ void r2k_stage(int wave, int lane, const int64_t (*sAcc)[4096], uint64_t *stage, int ai0, int ai1, uint64_t offset,
                                          int p_lo, int npoly) {
    const int wpi = 4 / npoly;             // waves per (job, polynomial): 4 or 2
    const int item = wave / wpi, piece = wave % wpi;
    const int g = item / npoly, j = p_lo + item % npoly;
    const int ai = g ? ai1 : ai0;
    if (ai == 0) return;
    const int a2n = ai & 4095;
    const int64_t *ap = sAcc[g] + j * 2048;
    uint64_t *dst = stage + (g * 2 + j) * 2048;
    const int cnt = 32 / wpi;              // coefficients per lane: 8 or 16
    const int base = piece * (2048 / wpi) + lane;
    for (int k0 = 0; k0 < cnt; k0 += 8) {
        uint64_t v[8];
        rot_minus_self64_batch<2048, 8>(ap, base + 64 * k0, a2n, v);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[base + 64 * (k0 + k)] = v[k] + offset;
    }
}
// digits of row part rp of job g for the 32 coefficients this lane transforms, four 16-bit fields per radix-2 group     (decompose, J/tgsw.jl:112-138)
__device__ __forceinline__ void r2k_pack(int g, int rp, int lane, const uint64_t *stage, const R2KDigits &dg, uint32_t (&pk)[8][2]) {
    const int r = rp / dg.parts, part = rp % dg.parts;   // uniform per wave
    const uint64_t *src = stage + (g * 2 + r / dg.lg) * 2048;
    const int shift = 64 - ((r % dg.lg) + 1) * dg.bg;
    const uint64_t mask = (1ull << dg.bg) - 1ull;
    const int32_t half_bg = 1 << (dg.bg - 1);
    const int lo_bits = dg.parts == 2 ? dg.lo_bits : 1;
    const int32_t half_lo = 1 << (lo_bits - 1), mask_lo = (1 << lo_bits) - 1;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        int32_t d[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint64_t t = src[lane + 64 * m + 512 * q];
            int32_t v = (int32_t)((t >> shift) & mask) - half_bg;
            if (dg.parts == 2) {
                const int32_t lo = ((v + half_lo) & mask_lo) - half_lo;   // balanced low part
                v = part ? (v - lo) >> lo_bits : lo;
            }
            d[q] = v;
        }
        pk[m][0] = ((uint32_t)d[0] & 0xffffu) | ((uint32_t)d[1] << 16);
        pk[m][1] = ((uint32_t)d[2] & 0xffffu) | ((uint32_t)d[3] << 16);
    }
}
// inverse transforms of the two half spectra of one job, radix-2 merge, round(S) << 16h into accumulator polynomial o (64-bit LDS atomics)
__device__ __forceinline__ void r2k_finish(int ln, int lane, int h, cplx (&S0)[8], cplx (&S1)[8], cplx *xb, unsigned long long *accu,
                                           const LaneRoots &roots1, const LaneRoots &roots5, const W64 &w64) {
    wave_fft_inv_tq<1>(ln, S0, xb, LaneRoots{opaque_cplx(roots1.b), opaque_cplx(roots1.s)}, w64);
    wave_fft_inv_tq<5>(ln, S1, xb, LaneRoots{opaque_cplx(roots5.b), opaque_cplx(roots5.s)}, w64);
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
}

// S += spectrum slot * key chunk, the slot read in two halves (16 registers of digit spectrum alive instead of 32)
#ifndef R2K_KMS_HALVES
#define R2K_KMS_HALVES 1
#endif
template <int HALVES = 1>
__device__ __forceinline__ void r2k_mac_slot(cplx (&S)[8], const cplx *slot, int lane, const cplx (&b)[8]) {
    if (!HALVES) {
        cplx z[8];
#pragma unroll
        for (int m = 0; m < 8; m++) z[m] = slot[m * 64 + lane];
        mac8r(S, z, b);
        return;
    }
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += 4) {
        cplx z[4];
#pragma unroll
        for (int m = 0; m < 4; m++) z[m] = slot[(m0 + m) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 4; m++) cfma(S[m0 + m], z[m], b[m0 + m]);
        kms_pin();
    }
}

__global__ __launch_bounds__(512, 2) void kms_tlev_rotate_pair_kernel(KmsBRArgs a, cplx *__restrict__ park) {
    constexpr int BATCH = 6;   // row parts of both jobs whose half spectra sit in the LDS at the same time (8 KiB each)
    constexpr int PRE = 2;     // key chunks in flight per wave
    __shared__ int64_t sAcc[2][4096];
    __shared__ cplx sSpec[2 * BATCH * 512];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    auto tw_w64 = [&](int ln) { return W64{a.tw[1024 + 1 * 8 + (ln & 7)]}; };             // phase-local constants: see mk_blind_rotate_pair2k_kernel
    auto tw_roots1 = [&](int ln) { return LaneRoots{a.tw[ln], a.tw[1216 + ln]}; };
    auto tw_roots5 = [&](int ln) { return LaneRoots{a.tw[512 + ln], a.tw[1216 + ln]}; };
    const long job0 = 2 * (long)blockIdx.x;
    const bool has1 = job0 + 1 < a.jobs;
    const int32_t *bara0 = a.bara + (job0 / a.l_lev) * a.bara_stride;
    const int32_t *bara1 = has1 ? a.bara + ((job0 + 1) / a.l_lev) * a.bara_stride : bara0;
    R2KDigits dg;
    dg.lg = a.lg;
    dg.bg = a.bg;
    dg.parts = a.parts;
    dg.lo_bits = a.lo_bits;
    dg.offset = 0;
    for (int p = 1; p <= a.lg; p++) dg.offset += (1ull << (a.bg - 1)) << (64 - p * a.bg);
    const int RP = 2 * a.lg * a.parts;
    // tlev_trivial_int(levpar, lwepar, 1): mask 0, body = gadget value of level `sample` on the constant coefficient     (J/tlev.jl:37-66)
    for (int q = threadIdx.x; q < 8192; q += 512) {
        const int g = q >> 12;
        if (g == 0 || has1) sAcc[g][q & 4095] = a.acc_in ? a.acc_in[(job0 + g) * 4096 + (q & 4095)] : 0;
    }
    __syncthreads();
    if (!a.acc_in && threadIdx.x < 2 && (threadIdx.x == 0 || has1))
        sAcc[threadIdx.x][2048] = (int64_t)(1ull << (64 - ((int)((job0 + threadIdx.x) % a.l_lev) + 1) * a.bg_lev));
    __syncthreads();
    const int o = wave >> 2, h = wave & 3;
    auto chunk = [&](int step, int rp, int half) { return a.bk + (((((size_t)step * RP + rp) * 2 + o) * 4 + h) * 2 + half) * 512; };
    auto active = [&](int i) { return bara0[i] != 0 || (has1 && bara1[i] != 0); };   // uniform over the workgroup
    cplx *mypark = park + (((size_t)blockIdx.x * 8 + wave) * 2) * 512;
    uint64_t *stage = reinterpret_cast<uint64_t *>(sSpec);

    int i = 0;
    while (i < a.n && !active(i)) i++;
    while (i < a.n) {
        const int ai0 = bara0[i], ai1 = has1 ? bara1[i] : 0;
        int inext = i + 1;
        while (inext < a.n && !active(inext)) inext++;
        cplx S0a[8], S0b[8];
#pragma unroll
        for (int m = 0; m < 8; m++) S0a[m] = S0b[m] = cplx{0.0, 0.0};
        for (int b0 = 0; b0 < RP; b0 += BATCH) {
            const int nb = RP - b0 < BATCH ? RP - b0 : BATCH;
            const bool first = b0 == 0, last = b0 + BATCH >= RP;
            const int gl = opaque_lane(lane);   // lane index of the global (key / park) addresses: formed here, not hoisted out of the loops and spilled
            // ---- digits of the batch: forward task f < 2 nb = (job f / nb, row part b0 + f % nb) -> slot (job) * BATCH + f % nb
            const int f1 = wave + 8;
            const bool t0 = wave < 2 * nb && ((wave / nb) ? ai1 : ai0) != 0;
            const bool t1 = f1 < 2 * nb && ((f1 / nb) ? ai1 : ai0) != 0;
            const int s0 = (wave / nb) * BATCH + wave % nb, s1 = (f1 / nb) * BATCH + f1 % nb;
            {
                const int p_lo = (b0 / a.parts) / a.lg, p_hi = ((b0 + nb - 1) / a.parts) / a.lg;
                r2k_stage(wave, lane, sAcc, stage, ai0, ai1, dg.offset, p_lo, p_hi - p_lo + 1);
            }
            lds_barrier<0>();
            uint32_t pk0[8][2], pk1[8][2];
            if (t0) r2k_pack(wave / nb, b0 + wave % nb, lane, stage, dg, pk0);
            if (t1) r2k_pack(f1 / nb, b0 + f1 % nb, lane, stage, dg, pk1);
            lds_barrier<0>();   // staged words consumed: the slots are free for the spectra
            // ---- half pass 0: even outputs (twist 1)
            cplx B[PRE][8];
            kms_pin();
#pragma unroll
            for (int r = 0; r < PRE; r++) load8(gl, B[r], chunk(i, b0 + (r < nb ? r : 0), 0));
            kms_pin();
            if (t0 || t1) {
                const int ln = opaque_lane(lane);
                const W64 w64 = tw_w64(ln);
                const LaneRoots roots1 = tw_roots1(ln);
                if (t0) r2k_transform<0>(s0, lane, sSpec, pk0, roots1, w64);
                if (t1) r2k_transform<0>(s1, lane, sSpec, pk1, roots1, w64);
            }
            lds_barrier<8 * PRE>();   // spectra published
#pragma unroll
            for (int q = 0; q < BATCH; q++) {
                if (q < nb) {
                    if (ai0 != 0) r2k_mac_slot<R2K_KMS_HALVES>(S0a, sSpec + q * 512, lane, B[q % PRE]);
                    if (ai1 != 0) r2k_mac_slot<R2K_KMS_HALVES>(S0b, sSpec + (BATCH + q) * 512, lane, B[q % PRE]);
                    kms_pin();
                    if (q + PRE < nb) load8(gl, B[q % PRE], chunk(i, b0 + q + PRE, 0));
                    kms_pin();
                }
            }
            lds_barrier<0>();   // spectra consumed
            // ---- half pass 1: odd outputs (twist 5), same digits
            if (t0 || t1) {
                const int ln = opaque_lane(lane);
                const W64 w64 = tw_w64(ln);
                const LaneRoots roots5 = tw_roots5(ln);
                if (t0) r2k_transform<1>(s0, lane, sSpec, pk0, roots5, w64);
                if (t1) r2k_transform<1>(s1, lane, sSpec, pk1, roots5, w64);
            }
            cplx S1a[8], S1b[8];
            kms_pin();
            if (first) {
#pragma unroll
                for (int m = 0; m < 8; m++) S1a[m] = S1b[m] = cplx{0.0, 0.0};
            } else {
                load8(gl, S1a, mypark);
                load8(gl, S1b, mypark + 512);
            }
#pragma unroll
            for (int r = 0; r < PRE; r++) load8(gl, B[r], chunk(i, b0 + (r < nb ? r : 0), 1));
            kms_pin();
            lds_barrier<8 * PRE + 16>();
#pragma unroll
            for (int q = 0; q < BATCH; q++) {
                if (q < nb) {
                    if (ai0 != 0) r2k_mac_slot<R2K_KMS_HALVES>(S1a, sSpec + q * 512, lane, B[q % PRE]);
                    if (ai1 != 0) r2k_mac_slot<R2K_KMS_HALVES>(S1b, sSpec + (BATCH + q) * 512, lane, B[q % PRE]);
                    kms_pin();
                    if (q + PRE < nb) load8(gl, B[q % PRE], chunk(i, b0 + q + PRE, 1));
                    kms_pin();
                }
            }
            if (!last) {   // S1 leaves the registers until the next batch's second multiply
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    mypark[m * 64 + gl] = S1a[m];
                    mypark[512 + m * 64 + gl] = S1b[m];
                }
            }
            lds_barrier<0>();   // spectra consumed: the area is staging / transpose scratch from here on; every rotated read of the accumulators is done
            if (last) {
                // ---- inverse transforms, merge, accumulate
                cplx *xb = sSpec + wave * 512;
                const int ln = opaque_lane(lane);
                const W64 w64 = tw_w64(ln);
                const LaneRoots roots1 = tw_roots1(ln), roots5 = tw_roots5(ln);
                if (ai0 != 0) r2k_finish(ln, lane, h, S0a, S1a, xb, reinterpret_cast<unsigned long long *>(sAcc[0]) + o * 2048, roots1, roots5, w64);
                if (ai1 != 0) r2k_finish(ln, lane, h, S0b, S1b, xb, reinterpret_cast<unsigned long long *>(sAcc[1]) + o * 2048, roots1, roots5, w64);
                lds_barrier<0>();   // accumulators updated and scratch free before the next step
            }
        }
        i = inext;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 8192; q += 512) {
        const int g = q >> 12;
        if (g == 0 || has1) a.acc_out[(job0 + g) * 4096 + (q & 4095)] = sAcc[g][q & 4095];
    }
}

// launch: one job per workgroup up to `pair_threshold` jobs (one per CU), two above; the park buffer (128 KiB per workgroup) is only
// needed when the row parts do not fit one batch
struct Rot2kPark {
    cplx *buf = nullptr;
    size_t cap_wgs = 0;
};
inline int rot2k_launch(const KmsBRArgs &a, hipStream_t stream, long pair_threshold, Rot2kPark &park) {
    if (a.jobs > pair_threshold) {
        const size_t wgs = ((size_t)a.jobs + 1) / 2;
        const bool need_park = 2 * a.lg * a.parts > 6;
        if (need_park && wgs > park.cap_wgs) {
            THFHE_HIP(hipStreamSynchronize(stream));
            (void)hipFree(park.buf);
            park.buf = nullptr;
            park.cap_wgs = 0;
            THFHE_HIP(hipMalloc(&park.buf, wgs * 8 * 2 * 512 * sizeof(cplx)));
            park.cap_wgs = wgs;
        }
        hipLaunchKernelGGL(kms_tlev_rotate_pair_kernel, dim3((unsigned)wgs), dim3(512), 0, stream, a, need_park ? park.buf : (cplx *)nullptr);
    } else {
        hipLaunchKernelGGL(kms_tlev_rotate_kernel, dim3((unsigned)a.jobs), dim3(512), 0, stream, a);
    }
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}
