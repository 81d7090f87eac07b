// thfhe_rot4k.h -- blind rotation on the Torus64 ring of degree 4096: the reference's wide-base 3-gen sets "64 parties, for fft" and
// "512 parties" (J/mk_api.jl:277-283, 316-322: N = 4096, l = 1, Bgbit = 27 -> three balanced 9-bit digit parts, six row parts).
// Structure of kms_tlev_rotate_pair_kernel (thfhe_rot2k.h) with the roles of "two jobs x two halves" taken by the FOUR quarter transforms
// of ONE job (radix-4 split, thfhe_lane.h): accumulator 64 KiB + twelve 8 KiB spectrum slots = 160 KiB of LDS; a step runs as two passes
// (quarters 0 / 1, then 2 / 3): rotated 64-bit words staged in the (free) spectrum area, every forward task = (row part, quarter) cuts its
// digits, combines the radix-4 group and transforms; every (output, limb) wave multiplies into its four partial quarter spectra; then four
// inverse transforms, the radix-4 merge and the integer atomics.  These sets run 46 k - 374 k sequential CMuxes per gate: the kernel is
// written for exactness and for fitting the CU, not tuned.  Included inside thfhe_mk.hip's anonymous namespace after thfhe_rot2k.h.
#pragma once
// nothing is scheduled across this point: without it the multiply-accumulates of a row sink below the requests for the next row's key chunk and the
// registers of both rows, the spectrum points and the partial sums are live at once
#define R4K_FENCE() do { kms_pin(); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef R4K_EARLY_KEY
#define R4K_EARLY_KEY 0
#endif

// torus polynomials int64[npolys][4096] -> limb spectra [poly][limb h][quarter][512], scaled by 1/2048 (one wave per (poly, limb))
__global__ __launch_bounds__(256) void r4k_key_transform_kernel(const int64_t *__restrict__ polys, long npolys, const cplx *__restrict__ tw,
                                                                 cplx *__restrict__ spec) {
    __shared__ cplx sX[4][512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= npolys * 4) return;
    const W64 w64{tw[1024 + 1 * 8 + (lane & 7)]};
    const cplx ratio = tw[1216 + lane];
    const int64_t *poly = polys + (item >> 2) * 4096;
    const int h = (int)(item & 3);
    cplx *dst = spec + (size_t)item * 2048;
#pragma unroll
    for (int qt = 0; qt < 4; qt++) {
        cplx y[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            cplx u[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                double a[4], b[4];
                split_limbs64(poly[lane + 64 * m + 512 * s], a);
                split_limbs64(poly[lane + 64 * m + 512 * s + 2048], b);
                u[s] = cplx{a[h], b[h]};
            }
            pre4096(u);
            y[m] = qt == 0 ? comb4096<0>(u) : qt == 1 ? comb4096<1>(u) : qt == 2 ? comb4096<2>(u) : comb4096<3>(u);
        }
        const LaneRoots roots{tw[1280 + qt * 64 + lane], ratio};
        if (qt == 0) wave_fft_fwd_tq<1, 64>(lane, y, sX[wave], roots, w64);
        if (qt == 1) wave_fft_fwd_tq<5, 64>(lane, y, sX[wave], roots, w64);
        if (qt == 2) wave_fft_fwd_tq<9, 64>(lane, y, sX[wave], roots, w64);
        if (qt == 3) wave_fft_fwd_tq<13, 64>(lane, y, sX[wave], roots, w64);
        wave_sync();
#pragma unroll
        for (int m = 0; m < 8; m++) dst[qt * 512 + m * 64 + lane] = cplx{y[m].re * (1.0 / 2048), y[m].im * (1.0 / 2048)};
    }
}

struct R4KArgs {
    const cplx *bk;       // key spectra [step][row part][output o][limb h][quarter][512]
    const cplx *tw;
    const int32_t *bara;  // [jobs][bara_stride]: mod-switched mask words (mod 8192), n of them used per job
    int64_t *acc;         // [jobs][2][4096], rotated in place
    long jobs;
    int n, l, Bgbit, parts, pw, bara_stride;
};

// digits (level, part) of the four complex points of one radix-4 group, taken from the staged 64-bit words       (decompose, J/tgsw.jl:112-138)
__device__ __forceinline__ void r4k_group_digits(const uint64_t *src, int c, int shift, uint32_t mask, int32_t half, int parts, int part, int pw,
                                                 cplx (&u)[4]) {
    const int32_t hp = pw ? 1 << (pw - 1) : 0, mp = (1 << pw) - 1;
    // the eight staged words of the group first (their LDS reads in flight together), then branch-free digit arithmetic: a loop over the parts
    // inside the unrolled body made every read wait on its own (281 of the kernel's 310 LDS waits covered a single read)
    uint32_t t[4][2];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int w = 0; w < 2; w++) t[s][w] = (uint32_t)(src[c + 512 * s + 2048 * w] >> 32);
#pragma unroll
    for (int s = 0; s < 4; s++) {
        double d[2];
#pragma unroll
        for (int w = 0; w < 2; w++) {
            int32_t v = (int32_t)((t[s][w] >> shift) & mask) - half;
            if (parts > 1) {   // balanced parts, least significant first; at most three; uniform selects
                const int32_t lo0 = ((v + hp) & mp) - hp, v1 = (v - lo0) >> pw;
                const int32_t lo1 = ((v1 + hp) & mp) - hp, v2 = (v1 - lo1) >> pw;
                const int32_t p1 = parts > 2 ? lo1 : v1;
                v = part == 0 ? lo0 : (part == 1 ? p1 : v2);
            }
            d[w] = (double)v;
        }
        u[s] = cplx{d[0], d[1]};
    }
}
// Register budget: a wave's four partial quarter spectra are 128 VGPRs; next to the digit work of a pass (two tasks of 32 VGPRs + key rows) they
// spilled 1 108 B per lane.  So a pass keeps only ITS two partial spectra (64 VGPRs): pass 0 parks them in global memory (`park`, 16 KiB per wave:
// cplx[workgroup][wave][2][512], coalesced 16-B accesses, L2-resident) and the inverse phase takes them back after pass 1's two are transformed.
__global__ __launch_bounds__(512, 2) void r4k_rotate_kernel(R4KArgs a, cplx *__restrict__ park) {
    constexpr int ROWP = 6;   // row parts (2 l parts): the reference's sets have exactly six; fewer are allowed
    constexpr int PRE = 2;
    __shared__ int64_t sAcc[2 * 4096];
    __shared__ cplx sSpec[2 * ROWP * 512];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long job = blockIdx.x;
    const int32_t *bara = a.bara + job * a.bara_stride;
    int64_t *gacc = a.acc + job * 8192;
    cplx *mypark = park + (((size_t)blockIdx.x * 8 + wave) * 2) * 512;
    const int RP = 2 * a.l * a.parts;
    const uint64_t offset = decomp_offset64(a.l, a.Bgbit);
    for (int q = threadIdx.x; q < 8192; q += 512) sAcc[q] = gacc[q];
    __syncthreads();
    const int o = wave >> 2, h = wave & 3;
    unsigned long long *accu = reinterpret_cast<unsigned long long *>(sAcc) + o * 4096;
    uint64_t *stage = reinterpret_cast<uint64_t *>(sSpec);
    auto chunk = [&](int step, int rp, int qt) { return a.bk + (((((size_t)step * RP + rp) * 2 + o) * 4 + h) * 4 + qt) * 512; };
    // forward tasks of a pass: f < 2 RP = (quarter-in-pass f / RP, row part f % RP) -> slot (f / RP) * ROWP + f % RP
    const int f0 = wave, f1 = wave + 8;
    const bool t0 = f0 < 2 * RP, t1 = f1 < 2 * RP;
    const int rp0 = f0 % RP, rp1 = f1 % RP, q0 = f0 / RP, q1 = f1 / RP;

    for (int i = 0; i < a.n; i++) {
        const int ai = bara[i];
        if (ai == 0) continue;   // uniform
        const int a2n = ai & 8191;
        cplx S[2][8];   // the partial spectra of the pass at hand (quarters 2P, 2P + 1)
        auto pass = [&](auto pass_index) {   // quarters 2P and 2P + 1; P is a compile-time constant
            constexpr int P = decltype(pass_index)::value;
            // ---- rotated words of both accumulator polynomials: 16 coefficients per lane
            {
                const int j = wave >> 2;
                const int64_t *ap = sAcc + j * 4096;
                uint64_t v[16];
                rot_minus_self64_batch<4096, 16>(ap, (wave & 3) * 1024 + lane, a2n, v);
#pragma unroll
                for (int k = 0; k < 16; k++) stage[j * 4096 + (wave & 3) * 1024 + lane + 64 * k] = v[k] + offset;
            }
            lds_barrier<0>();
            // ---- digits + radix-4 combination of this wave's tasks (in registers), then the staged words may be overwritten
            cplx y0[8], y1[8];
            auto digits = [&](int rp, int qt, cplx (&y)[8]) {
                const int r = rp / a.parts, part = rp % a.parts;   // uniform per wave; r = (accumulator polynomial j) * l + level
                const uint64_t *src = stage + (r / a.l) * 4096;
                const int shift = 32 - ((r % a.l) + 1) * a.Bgbit;
                const uint32_t mask = (1u << a.Bgbit) - 1u;
                const int32_t half = 1 << (a.Bgbit - 1);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    cplx u[4];
                    r4k_group_digits(src, lane + 64 * m, shift, mask, half, a.parts, part, a.pw, u);
                    pre4096(u);
                    y[m] = P == 0 ? (qt == 0 ? comb4096<0>(u) : comb4096<1>(u)) : (qt == 0 ? comb4096<2>(u) : comb4096<3>(u));
                }
            };
            if (t0) digits(rp0, q0, y0);
            if (t1) digits(rp1, q1, y1);
            lds_barrier<0>();
            // ---- forward quarter transforms into the slots; first key chunks requested
            cplx B[PRE][8];
            const int gl = opaque_lane(lane);
#if R4K_EARLY_KEY
            kms_pin();
            load8(gl, B[0], chunk(i, 0, 2 * P));
            load8(gl, B[1], chunk(i, 0, 2 * P + 1));
            kms_pin();
#endif
            {
                const int ln = opaque_lane(lane);
                const W64 w64{a.tw[1024 + 1 * 8 + (ln & 7)]};
                const cplx ratio = a.tw[1216 + ln];
                auto transform = [&](int qt, int slot, cplx (&y)[8]) {
                    cplx *xb = sSpec + slot * 512;
                    const LaneRoots roots{a.tw[1280 + (2 * P + qt) * 64 + ln], ratio};
                    if (P == 0 && qt == 0) wave_fft_fwd_tq<1, 64>(ln, y, xb, roots, w64);
                    if (P == 0 && qt == 1) wave_fft_fwd_tq<5, 64>(ln, y, xb, roots, w64);
                    if (P == 1 && qt == 0) wave_fft_fwd_tq<9, 64>(ln, y, xb, roots, w64);
                    if (P == 1 && qt == 1) wave_fft_fwd_tq<13, 64>(ln, y, xb, roots, w64);
                    wave_sync();
#pragma unroll
                    for (int m = 0; m < 8; m++) xb[m * 64 + ln] = y[m];
                };
                if (t0) transform(q0, q0 * ROWP + rp0, y0);
                if (t1) transform(q1, q1 * ROWP + rp1, y1);
            }
#if R4K_EARLY_KEY
            lds_barrier<8 * PRE>();   // spectra published
#else
            // the first key rows are requested AFTER the transforms: held across them (64 VGPRs next to two tasks' points and twiddles) they were
            // spilled as soon as they arrived
            kms_pin();
            load8(gl, B[0], chunk(i, 0, 2 * P));
            load8(gl, B[1], chunk(i, 0, 2 * P + 1));
            kms_pin();
            lds_barrier<8 * PRE>();   // spectra published
#endif
            // ---- multiply: row part q, quarters 2P (chunk in B[0]) and 2P + 1 (B[1])
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int m = 0; m < 8; m++) S[t][m] = cplx{0.0, 0.0};
            // a ROLLED loop over the row parts: unrolled (six conditional bodies, each with its requests for the next row) the allocator kept the
            // rows of several bodies live at once and spilled 900 B per lane; the unconditional request of the last body re-reads its own row
#pragma unroll 1
            for (int q = 0; q < RP; q++) {
                const int qn = q + 1 < RP ? q + 1 : q;
                r2k_mac_slot<1>(S[0], sSpec + q * 512, lane, B[0]);
                R4K_FENCE();
                load8(gl, B[0], chunk(i, qn, 2 * P));
                R4K_FENCE();
                r2k_mac_slot<1>(S[1], sSpec + (ROWP + q) * 512, lane, B[1]);
                R4K_FENCE();
                load8(gl, B[1], chunk(i, qn, 2 * P + 1));
                R4K_FENCE();
            }
            lds_barrier<0>();   // spectra consumed
        };
        pass(std::integral_constant<int, 0>{});
        {   // quarters 0 / 1 leave the registers for the length of pass 1
            const int gl = opaque_lane(lane);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                mypark[m * 64 + gl] = S[0][m];
                mypark[512 + m * 64 + gl] = S[1][m];
            }
        }
        pass(std::integral_constant<int, 1>{});
        // ---- four inverse quarter transforms, radix-4 merge, round(S) << 16h into accumulator polynomial o
        {
            cplx *xb = sSpec + wave * 512;
            const int ln = opaque_lane(lane);
            const W64 w64{a.tw[1024 + 1 * 8 + (ln & 7)]};
            const cplx ratio = a.tw[1216 + ln];
            cplx Sa[8], Sb[8];
            kms_pin();
            load8(ln, Sa, mypark);         // requested ahead of the two transforms that do not need them
            load8(ln, Sb, mypark + 512);
            kms_pin();
            wave_fft_inv_tq<9, 64>(ln, S[0], xb, LaneRoots{a.tw[1280 + 128 + ln], ratio}, w64);
            wave_fft_inv_tq<13, 64>(ln, S[1], xb, LaneRoots{a.tw[1280 + 192 + ln], ratio}, w64);
            wave_fft_inv_tq<1, 64>(ln, Sa, xb, LaneRoots{a.tw[1280 + ln], ratio}, w64);
            wave_fft_inv_tq<5, 64>(ln, Sb, xb, LaneRoots{a.tw[1280 + 64 + ln], ratio}, w64);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                cplx z[4];
                merge4096(Sa[m], Sb[m], S[0][m], S[1][m], z);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int c = lane + 64 * m + 512 * s;
                    atomicAdd(accu + c, (unsigned long long)round_i64(z[s].re) << (16 * h));
                    atomicAdd(accu + c + 2048, (unsigned long long)round_i64(z[s].im) << (16 * h));
                }
            }
        }
        lds_barrier<0>();   // accumulator updated and scratch free before the next step
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 8192; q += 512) gacc[q] = sAcc[q];
}
