// thfhe_pm_kernels.h -- the exact small x torus polynomial multiply-accumulate kernels (see thfhe_polymac.hip for the method).
// Included INSIDE the anonymous namespace of a translation unit (thfhe_polymac.hip: the thfhe_pm_* entry points; thfhe_kms.hip: the
// device-resident relinearisation of the KMS scheme), so that every unit has its own internal-linkage copies of the kernels.
#ifndef THFHE_PM_KERNELS_H
#define THFHE_PM_KERNELS_H

// limb h (balanced, 16 bit) of the two coefficients a lane folds into one complex point
template <int TB>
__device__ __forceinline__ cplx limb_pair(const void *poly, int q0, int q1, int h) {
    if (TB == 32) {
        const int32_t *p = static_cast<const int32_t *>(poly);
        double l0, h0, l1, h1;
        split_limbs32(p[q0], l0, h0);
        split_limbs32(p[q1], l1, h1);
        return h == 0 ? cplx{l0, l1} : cplx{h0, h1};
    } else {
        const int64_t *p = static_cast<const int64_t *>(poly);
        double a[4], b[4];
        split_limbs64(p[q0], a);
        split_limbs64(p[q1], b);
        return cplx{a[h], b[h]};
    }
}

// torus polynomials -> limb spectra [poly][limb][half][512], scaled by 1/(NN/2); one wave per (poly, limb)
template <int NN, int TB>
__global__ __launch_bounds__(256) void pm_torus_transform_kernel(const void *__restrict__ torus, long npolys, const cplx *__restrict__ tw,
                                                                  cplx *__restrict__ spec) {
    constexpr int LIMBS = TB / 16, HALVES = NN / 1024;
    __shared__ cplx sT1[HALVES][512];
    __shared__ cplx sX[4][512];
    for (int t = threadIdx.x; t < HALVES * 512; t += 256) (&sT1[0][0])[t] = tw[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const W64 w64{tw[HALVES * 512 + 1 * 8 + (lane & 7)]};
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= npolys * LIMBS) return;
    const int h = (int)(item % LIMBS);
    const char *poly = static_cast<const char *>(torus) + (size_t)(item / LIMBS) * NN * (TB / 8);
    cplx *dst = spec + (size_t)item * HALVES * 512;
    if (NN == 1024) {
        cplx z[8];
#pragma unroll
        for (int m = 0; m < 8; m++) z[m] = limb_pair<TB>(poly, lane + 64 * m, lane + 64 * m + 512, h);
        wave_fft_fwd_s(lane, z, sX[wave], sT1[0], w64);
#pragma unroll
        for (int m = 0; m < 8; m++) dst[m * 64 + lane] = cplx{z[m].re * (1.0 / 512), z[m].im * (1.0 / 512)};
    } else {
        cplx z[16], y0[8], y1[8];
#pragma unroll
        for (int m = 0; m < 16; m++) z[m] = limb_pair<TB>(poly, lane + 64 * m, lane + 64 * m + 1024, h);
        split2048(z, y0, y1);
        wave_fft_fwd_t<1>(lane, y0, sX[wave], sT1[0], w64);
        wave_fft_fwd_t<5>(lane, y1, sX[wave], sT1[HALVES - 1], w64);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            dst[m * 64 + lane] = cplx{y0[m].re * (1.0 / 1024), y0[m].im * (1.0 / 1024)};
            dst[512 + m * 64 + lane] = cplx{y1[m].re * (1.0 / 1024), y1[m].im * (1.0 / 1024)};
        }
    }
}

struct PMArgs {
    const int32_t *small;   // [n_small][NN]
    const cplx *spec;       // limb spectra of the torus polynomials
    const int32_t *terms;   // [n_terms][4] = (out, small, torus, sign), grouped by `out`
    const int32_t *first;   // [n_out + 1]: terms of output j are first[j] .. first[j+1]-1
    const void *addend;     // [n_out][NN] or null
    void *out;              // [n_out][NN]
    const cplx *tw;
    long n_out;
    int *too_big;           // set when a small coefficient leaves [-4096, 4096]
};

template <int NN, int TB>
__global__ __launch_bounds__(256) void pm_mac_kernel(PMArgs a) {
    constexpr int LIMBS = TB / 16, HALVES = NN / 1024, PER = NN / 64;  // coefficients per lane
    typedef typename std::conditional<TB == 32, uint32_t, uint64_t>::type word;
    __shared__ cplx sT1[HALVES][512];
    __shared__ cplx sX[4][512];
    for (int t = threadIdx.x; t < HALVES * 512; t += 256) (&sT1[0][0])[t] = a.tw[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const W64 w64{a.tw[HALVES * 512 + 1 * 8 + (lane & 7)]};
    const long j = (long)blockIdx.x * 4 + wave;
    if (j >= a.n_out) return;
    word r[PER];   // coefficient lane + 64 m
#pragma unroll
    for (int m = 0; m < PER; m++) r[m] = a.addend ? static_cast<const word *>(a.addend)[(size_t)j * NN + lane + 64 * m] : (word)0;
    int big = 0;
    for (int t = a.first[j]; t < a.first[j + 1]; t++) {
        const int32_t *sp = a.small + (size_t)a.terms[4 * t + 1] * NN;
        const cplx *K = a.spec + (size_t)a.terms[4 * t + 2] * LIMBS * HALVES * 512;
        const bool neg = a.terms[4 * t + 3] < 0;
        cplx y0[8], y1[8];   // spectrum of the small operand (two halves for NN = 2048)
        if (NN == 1024) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int32_t u = sp[lane + 64 * m], v = sp[lane + 64 * m + 512];
                big |= (u > 4096 || u < -4096 || v > 4096 || v < -4096);
                y0[m] = cplx{(double)u, (double)v};
            }
            wave_fft_fwd_s(lane, y0, sX[wave], sT1[0], w64);
        } else {
            cplx z[16];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const int32_t u = sp[lane + 64 * m], v = sp[lane + 64 * m + 1024];
                big |= (u > 4096 || u < -4096 || v > 4096 || v < -4096);
                z[m] = cplx{(double)u, (double)v};
            }
            split2048(z, y0, y1);
            wave_fft_fwd_t<1>(lane, y0, sX[wave], sT1[0], w64);
            wave_fft_fwd_t<5>(lane, y1, sX[wave], sT1[HALVES - 1], w64);
        }
#pragma unroll
        for (int h = 0; h < LIMBS; h++) {
            const cplx *Kh = K + (size_t)h * HALVES * 512;
            cplx p0[8], p1[8];
#pragma unroll
            for (int m = 0; m < 8; m++) p0[m] = cmul(y0[m], Kh[m * 64 + lane]);
            if (NN == 1024) {
                wave_fft_inv_s(lane, p0, sX[wave], sT1[0], w64);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const word vr = (word)round_i64(p0[m].re) << (16 * h), vi = (word)round_i64(p0[m].im) << (16 * h);
                    r[m] = neg ? r[m] - vr : r[m] + vr;
                    r[m + 8] = neg ? r[m + 8] - vi : r[m + 8] + vi;
                }
            } else {
#pragma unroll
                for (int m = 0; m < 8; m++) p1[m] = cmul(y1[m], Kh[512 + m * 64 + lane]);
                wave_fft_inv_t<1>(lane, p0, sX[wave], sT1[0], w64);
                wave_fft_inv_t<5>(lane, p1, sX[wave], sT1[HALVES - 1], w64);
                cplx lo[8], hi[8];
                merge2048(p0, p1, lo, hi);   // coefficients (j, j+512) in lo / hi real parts, (j+1024, j+1536) in the imaginary parts
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const word v0 = (word)round_i64(lo[m].re) << (16 * h), v1 = (word)round_i64(hi[m].re) << (16 * h);
                    const word v2 = (word)round_i64(lo[m].im) << (16 * h), v3 = (word)round_i64(hi[m].im) << (16 * h);
                    r[m % PER] = neg ? r[m % PER] - v0 : r[m % PER] + v0;
                    r[(m + 8) % PER] = neg ? r[(m + 8) % PER] - v1 : r[(m + 8) % PER] + v1;
                    r[(m + 16) % PER] = neg ? r[(m + 16) % PER] - v2 : r[(m + 16) % PER] + v2;
                    r[(m + 24) % PER] = neg ? r[(m + 24) % PER] - v3 : r[(m + 24) % PER] + v3;
                }
            }
        }
    }
    if (big) atomicOr(a.too_big, 1);
#pragma unroll
    for (int m = 0; m < PER; m++) static_cast<word *>(a.out)[(size_t)j * NN + lane + 64 * m] = r[m];
}


#endif
