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
