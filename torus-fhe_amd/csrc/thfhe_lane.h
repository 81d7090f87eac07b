// thfhe_lane.h -- lane-level arithmetic of the MI355X blind-rotate engine.
//
// One 64-lane wavefront owns one bootstrapping job.  Every function here is the code ONE LANE runs
// between two wave-level LDS exchanges ("segments").  The HIP kernels (thfhe_kernels.hip) call the
// segments back to back with a wave-scope fence in between; tests/emu/lane_emu.cpp (test-only, never
// linked into libthfhe_hip.so) replays the same segments on the host, looping over the 64 lanes, so
// the index algebra below is unit-tested against the CPU oracle without a GPU.
//
// Transform.  A real polynomial p of degree N = 1024 (mod X^N+1) is folded into N/2 = 512 complex
// points z_j = p_j + i p_{j+512} and evaluated at the roots x_k = zeta^(4k+1), zeta = exp(i pi/N):
//     P_k = sum_j z_j zeta^j w^(jk),   w = exp(2 pi i / 512)
// (the reference does the same folding with the opposite sign convention, 3-gen-mk-tfhe/src/
// polynomials.jl:208-214; the convention is internal because the bootstrapping key is transformed by
// this very code).  512 = 8*8*8: three in-register radix-8 passes, two LDS transposes per transform.
// With j = j0 + 8 j1 + 64 j2 and k = k0 + 8 k1 + 64 k2:
//     pass 1: DFT8 over j2 of z*C[j2]          (C[m] = zeta^(64 m), wave-uniform constants)
//             twiddle T1[k0][lane] = zeta^(lane (4 k0 + 1)),  lane = j0 + 8 j1   (merges the twist)
//     pass 2: DFT8 over j1, twiddle T2[k1][j0] = exp(2 pi i j0 k1 / 64)
//     pass 3: DFT8 over j0          -> lane (k1 + 8 k0), register k2 holds P[k0 + 8 k1 + 64 k2]
// The inverse runs the three passes backwards with conjugated twiddles; its 1/512 is folded into the
// bootstrapping-key spectra.  Spectra never leave this register order, so no permutation pass exists.
//
// Exactness.  Torus32 key coefficients are split into two balanced 16-bit limbs; digits are
// |d| <= 2^(Bgbit-1).  Each limb product sum is an integer of magnitude < 2^37 computed in FP64 with
// worst-case rounding error < 2^-7, so rounding to nearest recovers it exactly
// and lo + (hi << 16) mod 2^32 equals the reference's exact product (tgsw_extern_mul_wo_FFT,
// 3-gen-mk-tfhe/src/tgsw.jl:152-156).
#ifndef THFHE_LANE_H
#define THFHE_LANE_H

#include <limits.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define THFHE_FN __host__ __device__ __forceinline__
#else
#define THFHE_FN inline __attribute__((always_inline))
#endif

namespace thfhe {

constexpr int kLanes = 64;
constexpr int kXbufSlots = 8 * 72;  // padded transpose buffer, in complex slots (9216 B)

struct alignas(16) cplx {
    double re, im;
};

THFHE_FN cplx cmul(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
THFHE_FN cplx cmul_conj(cplx a, cplx b) {  // a * conj(b)
    return cplx{a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im};
}
THFHE_FN cplx cadd(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
THFHE_FN cplx csub(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
// multiply by s*i (s = +1 / -1)
template <int S>
THFHE_FN cplx mul_si(cplx a) {
    return S > 0 ? cplx{-a.im, a.re} : cplx{a.im, -a.re};
}

// C[m] = exp(i pi m / 16), m = 0..7  (zeta^(64 m) for N = 1024)
#define THFHE_C_RE(m) ((m) == 0 ? 1.0 : (m) == 1 ? 0.98078528040323044913 : (m) == 2 ? 0.92387953251128675613 : (m) == 3 ? 0.83146961230254523708 : (m) == 4 ? 0.70710678118654752440 : (m) == 5 ? 0.55557023301960222474 : (m) == 6 ? 0.38268343236508977173 : 0.19509032201612826785)
#define THFHE_C_IM(m) ((m) == 0 ? 0.0 : (m) == 1 ? 0.19509032201612826785 : (m) == 2 ? 0.38268343236508977173 : (m) == 3 ? 0.55557023301960222474 : (m) == 4 ? 0.70710678118654752440 : (m) == 5 ? 0.83146961230254523708 : (m) == 6 ? 0.92387953251128675613 : 0.98078528040323044913)

// 8-point DFT, natural order in and out:  y[k] <- sum_m y[m] exp(S * 2 pi i m k / 8)
template <int S>
THFHE_FN void dft8(cplx (&y)[8]) {
    constexpr double R = 0.70710678118654752440;
    // even half: DFT4 of (y0, y2, y4, y6)
    cplx t0 = cadd(y[0], y[4]), t1 = csub(y[0], y[4]);
    cplx t2 = cadd(y[2], y[6]), t3 = mul_si<S>(csub(y[2], y[6]));
    cplx e0 = cadd(t0, t2), e1 = cadd(t1, t3), e2 = csub(t0, t2), e3 = csub(t1, t3);
    // odd half: DFT4 of (y1, y3, y5, y7)
    cplx u0 = cadd(y[1], y[5]), u1 = csub(y[1], y[5]);
    cplx u2 = cadd(y[3], y[7]), u3 = mul_si<S>(csub(y[3], y[7]));
    cplx o0 = cadd(u0, u2), o1 = cadd(u1, u3), o2 = csub(u0, u2), o3 = csub(u1, u3);
    // w^1 = (1 + S i)/sqrt2,  w^2 = S i,  w^3 = (-1 + S i)/sqrt2
    cplx p1 = S > 0 ? cplx{o1.re - o1.im, o1.re + o1.im} : cplx{o1.re + o1.im, o1.im - o1.re};
    cplx p3 = S > 0 ? cplx{-o3.re - o3.im, o3.re - o3.im} : cplx{o3.im - o3.re, -o3.re - o3.im};
    cplx q2 = mul_si<S>(o2);
    y[0] = cadd(e0, o0);
    y[4] = csub(e0, o0);
    y[1] = cplx{e1.re + R * p1.re, e1.im + R * p1.im};
    y[5] = cplx{e1.re - R * p1.re, e1.im - R * p1.im};
    y[2] = cadd(e2, q2);
    y[6] = csub(e2, q2);
    y[3] = cplx{e3.re + R * p3.re, e3.im + R * p3.im};
    y[7] = cplx{e3.re - R * p3.re, e3.im - R * p3.im};
}

// ---- transpose-buffer slot maps (complex slots; conflict-free for ds_read/write_b128) ----
THFHE_FN int xs_a(int k0, int lane) { return k0 * 72 + lane; }                              // (k0 ; j0 + 8 j1)
THFHE_FN int xs_b(int j1, int lane) { return (lane >> 3) * 72 + j1 * 8 + (lane & 7); }      // lane = j0 + 8 k0
THFHE_FN int xs_c(int k1, int lane) { return (lane >> 3) * 72 + k1 * 9 + (lane & 7); }      // lane = j0 + 8 k0
THFHE_FN int xs_d(int j0, int lane) { return (lane >> 3) * 72 + (lane & 7) * 9 + j0; }      // lane = k1 + 8 k0

// ---- forward transform, three segments -------------------------------------------------------------
// z[m] on entry: folded coefficients (p[lane + 64 m], p[lane + 64 m + 512])
THFHE_FN void fwd_seg1(int lane, cplx (&z)[8], cplx *xbuf, const cplx *T1) {
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
    dft8<+1>(z);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) xbuf[xs_a(k0, lane)] = cmul(z[k0], T1[k0 * 64 + lane]);
}
// segment 2 is split into its load half and its compute+store half: on the GPU all lanes of the wave
// finish the loads before any lane stores (the DFT needs all eight inputs), the emulator needs the split.
THFHE_FN void fwd_seg2_ld(int lane, cplx (&z)[8], const cplx *xbuf) {
#pragma unroll
    for (int j1 = 0; j1 < 8; j1++) z[j1] = xbuf[xs_b(j1, lane)];
}
THFHE_FN void fwd_seg2_st(int lane, cplx (&z)[8], cplx *xbuf, const cplx *T2) {
    dft8<+1>(z);
    xbuf[xs_c(0, lane)] = z[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) xbuf[xs_c(k1, lane)] = cmul(z[k1], T2[k1 * 8 + (lane & 7)]);
}
THFHE_FN void fwd_seg3_ld(int lane, cplx (&z)[8], const cplx *xbuf) {
#pragma unroll
    for (int j0 = 0; j0 < 8; j0++) z[j0] = xbuf[xs_d(j0, lane)];
}
THFHE_FN void fwd_seg3(int lane, cplx (&z)[8], const cplx *xbuf) {
    fwd_seg3_ld(lane, z, xbuf);
    dft8<+1>(z);
}

// ---- inverse transform (unnormalised: returns 512 * p), three segments ------------------------------
THFHE_FN void inv_seg1(int lane, cplx (&z)[8], cplx *xbuf, const cplx *T2) {
    dft8<-1>(z);
    xbuf[xs_d(0, lane)] = z[0];
#pragma unroll
    for (int j0 = 1; j0 < 8; j0++) xbuf[xs_d(j0, lane)] = cmul_conj(z[j0], T2[j0 * 8 + (lane & 7)]);
}
THFHE_FN void inv_seg2_ld(int lane, cplx (&z)[8], const cplx *xbuf) {
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) z[k1] = xbuf[xs_c(k1, lane)];
}
THFHE_FN void inv_seg2_st(int lane, cplx (&z)[8], cplx *xbuf) {
    dft8<-1>(z);
#pragma unroll
    for (int j1 = 0; j1 < 8; j1++) xbuf[xs_b(j1, lane)] = z[j1];
}
THFHE_FN void inv_seg3_ld(int lane, cplx (&z)[8], const cplx *xbuf) {
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) z[k0] = xbuf[xs_a(k0, lane)];
}
THFHE_FN void inv_seg3_fin(int lane, cplx (&z)[8], const cplx *T1) {
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) z[k0] = cmul_conj(z[k0], T1[k0 * 64 + lane]);
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
}
THFHE_FN void inv_seg3(int lane, cplx (&z)[8], const cplx *xbuf, const cplx *T1) {
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) z[k0] = cmul_conj(xbuf[xs_a(k0, lane)], T1[k0 * 64 + lane]);
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
}

// ---- variant with an unpadded, XOR-swizzled 512-slot transpose buffer and pass-2 twiddles as powers of one per-lane
// root w = exp(2 pi i (lane & 7) / 64) (w, w^2, w^4 held in registers; w^3, w^5, w^6, w^7 by one product each).
// Used by the LDS-ring kernel, whose 160 KiB LDS budget has no room for padding or for the T2 table.
//   transpose 1: column (j1*8 + j0) of row k0 is XORed with 8*(k0 & 1);
//   transpose 2: column (k1*8 + (j0 ^ k1)) of row k0 is XORed with 8*((k0 >> 1) & 1)          
THFHE_FN int ys_a(int k0, int lane) { return k0 * 64 + (lane ^ ((k0 & 1) << 3)); }
THFHE_FN int ys_b(int j1, int lane) { return (lane >> 3) * 64 + ((j1 * 8 + (lane & 7)) ^ (((lane >> 3) & 1) << 3)); }
THFHE_FN int ys_c(int k1, int lane) { return (lane >> 3) * 64 + ((k1 * 8 + ((lane & 7) ^ k1)) ^ (((lane >> 4) & 1) << 3)); }
THFHE_FN int ys_d(int j0, int lane) { return (lane >> 3) * 64 + (((lane & 7) * 8 + (j0 ^ (lane & 7))) ^ (((lane >> 4) & 1) << 3)); }

struct W64 {  // per-lane root w = omega_64^(lane & 7); its powers are rebuilt in every pass-2 twiddle step (registers are scarcer than FP64 ops)
    cplx w1;
};
THFHE_FN void w64_powers(const W64 &w, cplx (&p)[8]) {
    p[0] = cplx{1.0, 0.0};
    p[1] = w.w1;
    p[2] = cmul(w.w1, w.w1);
    p[3] = cmul(p[2], w.w1);
    p[4] = cmul(p[2], p[2]);
    p[5] = cmul(p[4], w.w1);
    p[6] = cmul(p[4], p[2]);
    p[7] = cmul(p[4], p[3]);
}

THFHE_FN void fwds_seg1(int lane, cplx (&z)[8], cplx *xbuf, const cplx *T1) {
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
    dft8<+1>(z);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) xbuf[ys_a(k0, lane)] = cmul(z[k0], T1[k0 * 64 + lane]);
}
THFHE_FN void fwds_seg2_ld(int lane, cplx (&z)[8], const cplx *xbuf) {
#pragma unroll
    for (int j1 = 0; j1 < 8; j1++) z[j1] = xbuf[ys_b(j1, lane)];
}
THFHE_FN void fwds_seg2_st(int lane, cplx (&z)[8], cplx *xbuf, const W64 &w) {
    dft8<+1>(z);
    cplx p[8];
    w64_powers(w, p);
    xbuf[ys_c(0, lane)] = z[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) xbuf[ys_c(k1, lane)] = cmul(z[k1], p[k1]);
}
THFHE_FN void fwds_seg3(int lane, cplx (&z)[8], const cplx *xbuf) {
#pragma unroll
    for (int j0 = 0; j0 < 8; j0++) z[j0] = xbuf[ys_d(j0, lane)];
    dft8<+1>(z);
}
THFHE_FN void invs_seg1(int lane, cplx (&z)[8], cplx *xbuf, const W64 &w) {
    dft8<-1>(z);
    cplx p[8];
    w64_powers(w, p);
    xbuf[ys_d(0, lane)] = z[0];
#pragma unroll
    for (int j0 = 1; j0 < 8; j0++) xbuf[ys_d(j0, lane)] = cmul_conj(z[j0], p[j0]);
}
THFHE_FN void invs_seg2_ld(int lane, cplx (&z)[8], const cplx *xbuf) {
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) z[k1] = xbuf[ys_c(k1, lane)];
}
THFHE_FN void invs_seg2_st(int lane, cplx (&z)[8], cplx *xbuf) {
    dft8<-1>(z);
#pragma unroll
    for (int j1 = 0; j1 < 8; j1++) xbuf[ys_b(j1, lane)] = z[j1];
}
THFHE_FN void invs_seg3(int lane, cplx (&z)[8], const cplx *xbuf, const cplx *T1) {
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) z[k0] = cmul_conj(xbuf[ys_a(k0, lane)], T1[k0 * 64 + lane]);
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
}

// ---- variant "r" (second-generation LDS-ring kernel): padded 576-slot buffer (three LDS address registers per wave instead of
// the 25 the XOR maps need) and pass-1 twiddles rebuilt from two per-lane roots instead of read from the 8 KiB T1 table:
//     T1[k0][lane] = zeta^(lane (4 k0 + 1)) = b * s^k0,    b = zeta^lane,  s = zeta^(4 lane)
// (one square and seven products per transform, two interleaved chains of depth 4; the table reads were a dependent LDS round trip
// in front of every transpose store, the products are FP64 work the SIMD has room for).  Same spectra order as the other variants.
struct LaneRoots {
    cplx b, s;
};
THFHE_FN void fwdr_seg1(int lane, cplx (&z)[8], cplx *xbuf, const LaneRoots &r) {
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
    dft8<+1>(z);
    const cplx s2 = cmul(r.s, r.s);
    cplx e = r.b, o = cmul(r.b, r.s);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 2) {
        xbuf[xs_a(k0, lane)] = cmul(z[k0], e);
        xbuf[xs_a(k0 + 1, lane)] = cmul(z[k0 + 1], o);
        if (k0 < 6) {
            e = cmul(e, s2);
            o = cmul(o, s2);
        }
    }
}
THFHE_FN void fwdr_seg2_st(int lane, cplx (&z)[8], cplx *xbuf, const W64 &w) {
    dft8<+1>(z);
    cplx p[8];
    w64_powers(w, p);
    xbuf[xs_c(0, lane)] = z[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) xbuf[xs_c(k1, lane)] = cmul(z[k1], p[k1]);
}
THFHE_FN void invr_seg1(int lane, cplx (&z)[8], cplx *xbuf, const W64 &w) {
    dft8<-1>(z);
    cplx p[8];
    w64_powers(w, p);
    xbuf[xs_d(0, lane)] = z[0];
#pragma unroll
    for (int j0 = 1; j0 < 8; j0++) xbuf[xs_d(j0, lane)] = cmul_conj(z[j0], p[j0]);
}
THFHE_FN void invr_seg3(int lane, cplx (&z)[8], const cplx *xbuf, const LaneRoots &r) {
    const cplx s2 = cmul(r.s, r.s);
    cplx e = r.b, o = cmul(r.b, r.s);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 2) {
        z[k0] = cmul_conj(xbuf[xs_a(k0, lane)], e);
        z[k0 + 1] = cmul_conj(xbuf[xs_a(k0 + 1, lane)], o);
        if (k0 < 6) {
            e = cmul(e, s2);
            o = cmul(o, s2);
        }
    }
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
}

// ---- variant "q": as "r", but the FIRST transpose (register index <-> lane bits 3..5) never touches the LDS: the wave exchanges
// registers between its lanes with v_permlane32_swap (lane bit 5), v_permlane16_swap (bit 4) and a row_ror:8 DPP move (bit 3) --
// 80 32-bit VALU instructions per transpose in place of 8 ds_write_b128 + 8 ds_read_b128 (136 cycles of the CU's one LDS pipe, the
// busiest unit of the first-generation kernel).  The lane functions below are the per-lane halves; the exchange itself is
// wave_transpose_hi3 (thfhe_common.h on the device, lanes_transpose_hi3 in the emulator).
THFHE_FN void fwdq_seg1(cplx (&z)[8], const LaneRoots &r) {
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
    dft8<+1>(z);
    const cplx s2 = cmul(r.s, r.s);
    cplx e = r.b, o = cmul(r.b, r.s);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 2) {
        z[k0] = cmul(z[k0], e);
        z[k0 + 1] = cmul(z[k0 + 1], o);
        if (k0 < 6) {
            e = cmul(e, s2);
            o = cmul(o, s2);
        }
    }
}
THFHE_FN void invq_seg3(cplx (&z)[8], const LaneRoots &r) {
    const cplx s2 = cmul(r.s, r.s);
    cplx e = r.b, o = cmul(r.b, r.s);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 2) {
        z[k0] = cmul_conj(z[k0], e);
        z[k0 + 1] = cmul_conj(z[k0 + 1], o);
        if (k0 < 6) {
            e = cmul(e, s2);
            o = cmul(o, s2);
        }
    }
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
}

// Register-lean form of the "q" pass-1 twiddles for kernels at the VGPR limit (LDS-ring kernel): the compiler hoists the eight products
// b s^k0 of fwdq_seg1 out of the CMux loop (32 VGPRs) and then spills some of them; here only the even ones e_j = b s^(2j) live across
// the loop (16 VGPRs + s), the odd ones e_j s are formed in place (four more complex products per transform).  `s` is passed through an
// empty asm so that these products are not hoisted as well.
struct LaneTw {
    cplx e[4], s;
};
THFHE_FN LaneTw make_lane_tw(const LaneRoots &r) {
    LaneTw t;
    const cplx s2 = cmul(r.s, r.s);
    t.e[0] = r.b;
#pragma unroll
    for (int j = 1; j < 4; j++) t.e[j] = cmul(t.e[j - 1], s2);
    t.s = r.s;
    return t;
}
THFHE_FN cplx opaque_cplx(cplx v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v.re), "+v"(v.im));
#endif
    return v;
}
THFHE_FN void fwdq_seg1(cplx (&z)[8], const LaneTw &t) {
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
    dft8<+1>(z);
    const cplx s = opaque_cplx(t.s);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        z[2 * j] = cmul(z[2 * j], t.e[j]);
        z[2 * j + 1] = cmul(z[2 * j + 1], cmul(t.e[j], s));
    }
}
THFHE_FN void invq_seg3(cplx (&z)[8], const LaneTw &t) {
    const cplx s = opaque_cplx(t.s);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        z[2 * j] = cmul_conj(z[2 * j], t.e[j]);
        z[2 * j + 1] = cmul_conj(z[2 * j + 1], cmul(t.e[j], s));
    }
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], cplx{THFHE_C_RE(m), THFHE_C_IM(m)});
}

// ---- integer helpers ---------------------------------------------------------------------------------
// coefficient q of X^a * p - p for p in LDS, a in [0, 2N)          (mul_by_monomial, J/rlwe.jl:130-131)
THFHE_FN uint32_t rot_minus_self32(const int32_t *p, int q, int a2n, int N) {
    int e = (q - a2n) & (2 * N - 1);
    uint32_t r = (uint32_t)p[e & (N - 1)];
    if (e & N) r = 0u - r;
    return r - (uint32_t)p[q];
}
// signed gadget digit of (v = coefficient + offset)                   (decompose, J/tgsw.jl:112-138)
THFHE_FN double digit32(uint32_t v, int shift, uint32_t mask, int32_t half) {
    return (double)((int32_t)((v >> shift) & mask) - half);
}
THFHE_FN uint32_t decomp_offset32(int l, int Bgbit) {
    uint32_t off = 0;
    for (int p = 1; p <= l; p++) off += (1u << (Bgbit - 1)) << (32 - p * Bgbit);
    return off;
}
// round-to-nearest of x (|x| < 2^51) as a wrapping 32-bit integer: add 1.5*2^52, keep the low mantissa word
THFHE_FN uint32_t round_lo32(double x) {
    double y = x + 6755399441055744.0;
    uint64_t b;
    __builtin_memcpy(&b, &y, 8);
    return (uint32_t)b;
}
// balanced 16-bit limb split of a Torus32 word: v = lo + 65536 * hi, lo in [-2^15, 2^15)
THFHE_FN void split_limbs32(int32_t v, double &lo, double &hi) {
    int32_t l = (int32_t)(int16_t)(uint16_t)v;
    int32_t h = (int32_t)(((int64_t)v - l) >> 16);
    lo = (double)l;
    hi = (double)h;
}

// mod-switch to Z_2N:  decode_message(x, 2N)                           (J/numeric-functions.jl:70-73)
THFHE_FN int32_t modswitch2n(int32_t x, int log2_2n) {
    int32_t y = (int32_t)((uint32_t)x + (1u << (32 - log2_2n - 1)));
    return y >> (32 - log2_2n);
}


// ---- CMux building blocks (single key, Torus32 ring, k = 1) ----------------------------------------
// Spectral bootstrapping key: [i][row r = j*l + p][column c][limb h][slot m][lane] complex, i.e. one
// coalesced 1 KiB line per (i, r, c, h, m); 8 KiB per limb polynomial ("8 B per coefficient" per limb).
THFHE_FN size_t bk_spec_index(int i, int r, int c, int h, int rows) {
    return ((((size_t)i * rows + r) * 2 + c) * 2 + h) * 512;  // + m*64 + lane
}

// t[m] = (X^a * acc_j - acc_j)[lane + 64 m] + offset, m = 0..15      (J/bootstrap.jl:21 + J/tgsw.jl:125-137)
THFHE_FN void load_rotated16(int lane, const int32_t *acc_poly, int a2n, uint32_t offset, uint32_t (&t)[16]) {
    // all 32 LDS reads first, the arithmetic after a scheduling fence: left to itself the compiler (in the CCS kernels) read one word, waited for it
    // (s_waitcnt lgkmcnt(0)), used it and only then read the next -- 32 exposed LDS round trips per digit row
    uint32_t r[16], s[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int e = (lane + 64 * m - a2n) & 2047;
        r[m] = (uint32_t)acc_poly[e & 1023];
        s[m] = (uint32_t)acc_poly[lane + 64 * m];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int e = (lane + 64 * m - a2n) & 2047;
        t[m] = ((e & 1024) ? 0u - r[m] : r[m]) - s[m] + offset;      // = rot_minus_self32(acc_poly, lane + 64 m, a2n, 1024) + offset
    }
}
// folded complex input of the level-p digit polynomial (p = 1..l)
THFHE_FN void digits_to_z(const uint32_t (&t)[16], int p, int Bgbit, cplx (&z)[8]) {
    const int shift = 32 - p * Bgbit;
    const uint32_t mask = (1u << Bgbit) - 1u;
    const int32_t half = 1 << (Bgbit - 1);
#pragma unroll
    for (int m = 0; m < 8; m++) z[m] = cplx{digit32(t[m], shift, mask, half), digit32(t[m + 8], shift, mask, half)};
}
// Fused form of load_rotated16 + digits_to_z for kernels that re-read the accumulator per digit level (second-generation ring kernel):
// z[m] = (digit_p of (X^a acc - acc)[lane + 64 m], digit_p of ...[lane + 64 m + 512]), level p = 1..l.  Fewer integer instructions:
//   * e = (lane - a) mod 2N is formed once; e + 64 m needs no second reduction (bit 10 of the sum is the sign, bits 0..9 the index);
//   * the sign is applied as (r ^ M) - M with M = -sign (a one-bit signed field extract);
//   * the balanced digit is ONE signed bit-field extract of v + offset + half_p, half_p = half a digit at level p: adding it turns the
//     unsigned field F of the reference's ((v + offset) >> shift) & mask - Bg/2 (J/tgsw.jl:125-137) into (F + Bg/2) mod Bg, whose
//     two's-complement reading is F - Bg/2 (carries of the addition only travel upwards, out of the field).
//   * device form: byte offsets (one add + one mask per rotated address, the polynomial being 4 KiB-aligned in LDS), sign applied as
//     (r + M) ^ M, the level-dependent constant folded with the unrotated word, v_bfe_i32 for sign and digit: 8 integer instructions +
//     1 conversion per coefficient (the portable form compiles to 10-11).
THFHE_FN int32_t sbfe32(uint32_t v, int shift, int width) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sbfe((int32_t)v, (uint32_t)shift, (uint32_t)width);
#else
    return (int32_t)(v << (32 - shift - width)) >> (32 - width);
#endif
}
THFHE_FN void rotated_digits_z(int lane, const int32_t *p, int a2n, int level, int l, int Bgbit, cplx (&z)[8]) {
    const int shift = 32 - level * Bgbit;
    const uint32_t off = decomp_offset32(l, Bgbit) + ((1u << (Bgbit - 1)) << shift);
    const uint32_t e4 = ((uint32_t)(lane - a2n) & 2047u) << 2;   // byte offset of coefficient (lane - a) mod 2N, bit 12 = sign
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) const void *)p;   // LDS, 4 KiB-aligned (the kernels' sAcc)
#endif
    // all 32 LDS reads first (rotated word and own word of the 16 coefficients), the arithmetic after a scheduling fence: the compiler otherwise
    // chains them -- read, s_waitcnt lgkmcnt(0), use, next read -- and the latency kernel's forward phase carried a dozen exposed LDS round trips
    uint32_t r[8][2], s[8][2];
#pragma unroll
    for (int m = 0; m < 8; m++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = lane + 64 * m + 512 * q;          // coefficient index
            const uint32_t f = e4 + (uint32_t)(256 * m + 2048 * q);
#if defined(__HIP_DEVICE_COMPILE__)
            r[m][q] = (uint32_t) * (__attribute__((address_space(3))) const int32_t *)(size_t)(base | (f & 0xFFCu));
#else
            r[m][q] = (uint32_t)p[(f & 0xFFCu) >> 2];
#endif
            s[m][q] = (uint32_t)p[c];
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int m = 0; m < 8; m++) {
        double d[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t f = e4 + (uint32_t)(256 * m + 2048 * q);
            const uint32_t M = (uint32_t)sbfe32(f, 12, 1);   // -1 where X^a wraps with a sign flip
            const uint32_t v = ((r[m][q] + M) ^ M) + (off - s[m][q]);
            d[q] = (double)sbfe32(v, shift, Bgbit);
        }
        z[m] = cplx{d[0], d[1]};
    }
}
// Two-step form: index, sign and subtraction are done once per coefficient and polynomial, a level then costs one signed bit-field
// extract and one conversion:   t = ((X^a acc - acc) + offset) ^ offset,  offset = sum_p (Bg/2) << (32 - p Bgbit).
// XOR with Bg/2 inside a field is "+ Bg/2 mod Bg" without carries into the neighbouring fields, so the two's-complement reading of
// field p of t is F_p - Bg/2, the reference's balanced digit (J/tgsw.jl:125-137).  The fields of the first NF coefficients this lane
// owns are kept across the levels of a polynomial, the other 16 - NF are re-read per level as in rotated_digits_z (NF = 16 in the
// LDS-ring kernel: the compiler parks some of the fields in scratch, which measured faster than re-reading them -- 29.9 ms per 4096
// gates against 30.1 for NF = 8 and 30.5 for the largest spill-free NF = 5).  Coefficient j = lane + 64 j (j < 16).
THFHE_FN uint32_t rotated_word(int lane, const int32_t *p, uint32_t e4, int j, uint32_t off) {
    const uint32_t f = e4 + (uint32_t)(256 * j);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) const void *)p;   // LDS, 4 KiB-aligned
    const uint32_t r = (uint32_t) * (__attribute__((address_space(3))) const int32_t *)(size_t)(base | (f & 0xFFCu));
#else
    const uint32_t r = (uint32_t)p[(f & 0xFFCu) >> 2];
#endif
    const uint32_t M = (uint32_t)sbfe32(f, 12, 1);
    return ((r + M) ^ M) + (off - (uint32_t)p[lane + 64 * j]);
}
// BATCH: all 2 NF LDS reads first, the arithmetic after a scheduling fence.  The eight-wave ring kernel's schedule already keeps the reads together;
// in its four-wave shape (one wave per SIMD, nobody to cover a round trip) the compiler chained them: read, s_waitcnt lgkmcnt(0), use, next read.
template <int NF, bool BATCH = false>
THFHE_FN void rotated_fields_keep(int lane, const int32_t *p, int a2n, int l, int Bgbit, uint32_t (&t)[NF]) {
    const uint32_t off = decomp_offset32(l, Bgbit);
    const uint32_t e4 = ((uint32_t)(lane - a2n) & 2047u) << 2;
    if (!BATCH) {
#pragma unroll
        for (int j = 0; j < NF; j++) t[j] = rotated_word(lane, p, e4, j, off) ^ off;
        return;
    }
    uint32_t r[NF], s[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) {
        const uint32_t f = e4 + (uint32_t)(256 * j);
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) const void *)p;   // LDS, 4 KiB-aligned
        r[j] = (uint32_t) * (__attribute__((address_space(3))) const int32_t *)(size_t)(base | (f & 0xFFCu));
#else
        r[j] = (uint32_t)p[(f & 0xFFCu) >> 2];
#endif
        s[j] = (uint32_t)p[lane + 64 * j];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int j = 0; j < NF; j++) {
        const uint32_t M = (uint32_t)sbfe32(e4 + (uint32_t)(256 * j), 12, 1);
        t[j] = (((r[j] + M) ^ M) + (off - s[j])) ^ off;
    }
}
template <int NF>
THFHE_FN void mixed_digits_z(int lane, const int32_t *p, int a2n, int level, int l, int Bgbit, const uint32_t (&t)[NF], cplx (&z)[8]) {
    const int shift = 32 - level * Bgbit;
    const uint32_t off = decomp_offset32(l, Bgbit) + ((1u << (Bgbit - 1)) << shift);
    const uint32_t e4 = ((uint32_t)(lane - a2n) & 2047u) << 2;
    double d[16];
#pragma unroll
    for (int j = 0; j < 16; j++) d[j] = (double)sbfe32(j < NF ? t[j < NF ? j : 0] : rotated_word(lane, p, e4, j, off), shift, Bgbit);
#pragma unroll
    for (int m = 0; m < 8; m++) z[m] = cplx{d[m], d[m + 8]};
}
// s += z * b as four fused multiply-adds (two dependent pairs)
THFHE_FN void cfma(cplx &s, cplx z, cplx b) {
    s.re = __builtin_fma(z.re, b.re, s.re);
    s.im = __builtin_fma(z.re, b.im, s.im);
    s.re = __builtin_fma(-z.im, b.im, s.re);
    s.im = __builtin_fma(z.im, b.re, s.im);
}
// S[m] += z[m] * B[m*64 + lane], one slice at a time with the smallest register footprint: the ring kernels run at the 256-VGPR
// limit, where this form (18 spilled registers) beats the batched-FMA form below (31+) by 4-20 % (measured)
THFHE_FN void mac8_lean(int lane, cplx (&S)[8], const cplx (&z)[8], const cplx *B) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
        cplx b = B[m * 64 + lane];
        S[m].re += z[m].re * b.re - z[m].im * b.im;
        S[m].im += z[m].re * b.im + z[m].im * b.re;
    }
}
// S[m] += z[m] * B[m*64 + lane]; DEPTH key slices are requested ahead of the multiplies (kernels with registers to spare)
template <int DEPTH = 8>
THFHE_FN void mac8(int lane, cplx (&S)[8], const cplx (&z)[8], const cplx *B) {
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += DEPTH) {
        cplx b[DEPTH];
#pragma unroll
        for (int q = 0; q < DEPTH; q++) b[q] = B[(m0 + q) * 64 + lane];
#pragma unroll
        for (int q = 0; q < DEPTH; q++) cfma(S[m0 + q], z[m0 + q], b[q]);
    }
}
// register-resident key chunk: b[m] = B[m*64 + lane]; S[m] += z[m] * b[m]
THFHE_FN void load8(int lane, cplx (&b)[8], const cplx *B) {
#pragma unroll
    for (int m = 0; m < 8; m++) b[m] = B[m * 64 + lane];
}
THFHE_FN void mac8r(cplx (&S)[8], const cplx (&z)[8], const cplx (&b)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) cfma(S[m], z[m], b[m]);
}
// acc_poly[q] += round(lo) + (round(hi) << 16)  for the 16 coefficients this lane owns
THFHE_FN void acc_update16(int lane, int32_t *acc_poly, const cplx (&zlo)[8], const cplx (&zhi)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
        int q = lane + 64 * m;
        uint32_t vr = round_lo32(zlo[m].re) + (round_lo32(zhi[m].re) << 16);
        uint32_t vi = round_lo32(zlo[m].im) + (round_lo32(zhi[m].im) << 16);
        acc_poly[q] = (int32_t)((uint32_t)acc_poly[q] + vr);
        acc_poly[q + 512] = (int32_t)((uint32_t)acc_poly[q + 512] + vi);
    }
}
// initial accumulator: acc = (0, X^{-barb} * (mu, ..., mu))            (J/bootstrap.jl:60-62,84)
THFHE_FN void acc_init16(int lane, int32_t *acc_mask, int32_t *acc_body, int barb, int32_t mu) {
#pragma unroll
    for (int m = 0; m < 16; m++) {
        int q = lane + 64 * m;
        int e = (q + barb) & 2047;
        acc_mask[q] = 0;
        acc_body[q] = (e & 1024) ? (int32_t)(0u - (uint32_t)mu) : mu;
    }
}
// sample extraction into an LWE(N) record: a'_0 = a_0, a'_j = -a_{N-j}, b = body_0   (J/rlwe.jl:64-68)
THFHE_FN void extract16(int lane, const int32_t *acc_mask, const int32_t *acc_body, int32_t *out) {
#pragma unroll
    for (int m = 0; m < 16; m++) {
        int q = lane + 64 * m;
        out[q] = q == 0 ? acc_mask[0] : (int32_t)(0u - (uint32_t)acc_mask[1024 - q]);
    }
    if (lane == 0) out[1024] = acc_body[0];
}
// folded limb inputs of a key polynomial for the key transform
THFHE_FN void key_limbs_to_z(int lane, const int32_t *poly, cplx (&zlo)[8], cplx (&zhi)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
        double l0, h0, l1, h1;
        split_limbs32(poly[lane + 64 * m], l0, h0);
        split_limbs32(poly[lane + 64 * m + 512], l1, h1);
        zlo[m] = cplx{l0, l1};
        zhi[m] = cplx{h0, h1};
    }
}


// ---- 3-gen multi-key building blocks (Torus64 ring, k = 1)        J/tgsw_3gen.jl:102-113, J/3gen_mk_internals.jl:59-95 ----
// Accumulator acc = [c1 (mask), c0 (body)] as int64[2][1024].  Digit rows r = j*l + lv: j = 0 digits of c1, j = 1 digits
// of c0.  Outputs o = 0: c1' = sum g(c0) P4 + g(c1) P3 ; o = 1: c0' = sum g(c0) P1 + g(c1) P2.  Every key polynomial is
// split into four balanced 16-bit limbs.  Spectral key stream order: [party][i][row r][limb h][output o][slot m][lane],
// one 8 KiB chunk per (party, i, r, h, o).
THFHE_FN int mk_part_index(int j, int o) {  // which of part_1..part_4 (0..3) multiplies digit set j for output o
    return o == 0 ? (j == 0 ? 2 : 3) : (j == 0 ? 1 : 0);
}
THFHE_FN size_t mk_chunk_index(long pi /* party*n + i */, int r, int h, int o, int rows) {
    return ((((size_t)pi * rows + r) * 4 + h) * 2 + o);  // * 512 complex
}
THFHE_FN uint64_t rot_minus_self64(const int64_t *p, int q, int a2n) {
    int e = (q - a2n) & 2047;
    uint64_t r = (uint64_t)p[e & 1023];
    if (e & 1024) r = 0ull - r;
    return r - (uint64_t)p[q];
}
THFHE_FN uint64_t decomp_offset64(int l, int Bgbit) {
    uint64_t off = 0;
    for (int p = 1; p <= l; p++) off += (1ull << (Bgbit - 1)) << (64 - p * Bgbit);
    return off;
}
// top 32 bits of (X^a acc_j - acc_j)[lane + 64 m] + offset: every digit lives there because l*Bgbit <= 32
THFHE_FN void load_rotated16_hi(int lane, const int64_t *acc_poly, int a2n, uint64_t offset, uint32_t (&t)[16]) {
#pragma unroll
    for (int m = 0; m < 16; m++) t[m] = (uint32_t)((rot_minus_self64(acc_poly, lane + 64 * m, a2n) + offset) >> 32);
}
// four balanced 16-bit limbs of a Torus64 word: v = l0 + l1 2^16 + l2 2^32 + l3 2^48
THFHE_FN void split_limbs64(int64_t v, double (&l)[4]) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
        int64_t lo = (int64_t)(int16_t)(uint16_t)v;
        l[q] = (double)lo;
        v = (v - lo) >> 16;
    }
    l[3] = (double)v;
}
THFHE_FN void key_limbs64_to_z(int lane, const int64_t *poly, cplx (&z)[4][8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
        double a[4], b[4];
        split_limbs64(poly[lane + 64 * m], a);
        split_limbs64(poly[lane + 64 * m + 512], b);
#pragma unroll
        for (int q = 0; q < 4; q++) z[q][m] = cplx{a[q], b[q]};
    }
}
// round-to-nearest of x (|x| < 2^51) as int64
THFHE_FN int64_t round_i64(double x) {
    double y = x + 6755399441055744.0;
    int64_t b;
    __builtin_memcpy(&b, &y, 8);
    return b - 0x4338000000000000ll;
}
// acc_poly[q] += sum_h round(limb_h) << 16h     for the 16 coefficients this lane owns
THFHE_FN void acc_update16_64(int lane, int64_t *acc_poly, const cplx (&S)[4][8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
        int q = lane + 64 * m;
        uint64_t vr = 0, vi = 0;
#pragma unroll
        for (int h = 0; h < 4; h++) {
            vr += (uint64_t)round_i64(S[h][m].re) << (16 * h);
            vi += (uint64_t)round_i64(S[h][m].im) << (16 * h);
        }
        acc_poly[q] = (int64_t)((uint64_t)acc_poly[q] + vr);
        acc_poly[q + 512] = (int64_t)((uint64_t)acc_poly[q + 512] + vi);
    }
}
THFHE_FN void acc_init16_64(int lane, int64_t *acc_mask, int64_t *acc_body, int barb, int64_t mu) {
#pragma unroll
    for (int m = 0; m < 16; m++) {
        int q = lane + 64 * m;
        int e = (q + barb) & 2047;
        acc_mask[q] = 0;
        acc_body[q] = (e & 1024) ? (int64_t)(0ull - (uint64_t)mu) : mu;
    }
}
// t64tot32 = trunc(Int32, Float64(d) / 2^32)                         J/numeric-functions.jl:109-111
THFHE_FN int32_t t64tot32(int64_t d) {
    double v = (double)d * (1.0 / 4294967296.0);
    v = v < 0 ? -__builtin_floor(-v) : __builtin_floor(v);
    return v >= 2147483648.0 ? INT32_MIN : (int32_t)v;
}
// rlwe_extract_sample_64                                             J/rlwe.jl:70-74
THFHE_FN void extract16_64(int lane, const int64_t *acc_mask, const int64_t *acc_body, int32_t *out) {
#pragma unroll
    for (int m = 0; m < 16; m++) {
        int q = lane + 64 * m;
        out[q] = t64tot32(q == 0 ? acc_mask[0] : (int64_t)(0ull - (uint64_t)acc_mask[1024 - q]));
    }
    if (lane == 0) out[1024] = t64tot32(acc_body[0]);
}


// ---- N = 2048 (BASELINE config 5): 1024 complex points = one radix-2 split + two twisted 512-point transforms -----------
// z_j = p_j + i p_{j+1024}, P_k = sum_{j<1024} z_j zeta^(j(4k+1)), zeta = exp(i pi / 2048).  With j = j' + 512 s:
//     zeta^(512 (4k+1)) = (-1)^k e^{i pi/4}, so
//     P_{2k''}   = sum_{j'<512} (z_j' + e^{i pi/4} z_{j'+512}) zeta^(8 j' k'') zeta^(1 j')        ("twist" T = 1)
//     P_{2k''+1} = sum_{j'<512} (z_j' - e^{i pi/4} z_{j'+512}) zeta^(8 j' k'') zeta^(5 j')        (T = 5)
// Each half is the 512-point transform above with the twist zeta2^j' (= zeta^(2 j'), T = 2) replaced by zeta^(T j'): only
// the pass-1 constants C_T[m] = zeta^(64 T m) = exp(i pi T m / 32) and the pass-1 table T1_T[k0][lane] = zeta^(lane (8 k0 + T))
// change; passes 2 and 3 and the swizzled buffer maps are shared.  Spectrum register order: [half][slot m][lane].
THFHE_FN cplx e32(int q) {  // exp(i pi q / 32)
    constexpr double C[17] = {1.0, 0.995184726672196886231, 0.980785280403230449119, 0.956940335732208864931, 0.923879532511286756101,
                              0.881921264348355029715, 0.831469612302545237081, 0.773010453362736960797, 0.707106781186547524382,
                              0.634393284163645498203, 0.555570233019602224757, 0.471396736825997648545, 0.382683432365089771723,
                              0.290284677254462367645, 0.195090322016128267857, 0.098017140329560601996, 0.0};
    const int quad = (q >> 4) & 3, r = q & 15;
    const double c = C[r], s = C[16 - r];
    return quad == 0 ? cplx{c, s} : quad == 1 ? cplx{-s, c} : quad == 2 ? cplx{-c, -s} : cplx{s, -c};
}
template <int T>
THFHE_FN void fwdt_seg1(int lane, cplx (&z)[8], cplx *xbuf, const cplx *T1t) {
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul(z[m], e32(T * m));
    dft8<+1>(z);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) xbuf[ys_a(k0, lane)] = cmul(z[k0], T1t[k0 * 64 + lane]);
}
template <int T>
THFHE_FN void invt_seg3(int lane, cplx (&z)[8], const cplx *xbuf, const cplx *T1t) {
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) z[k0] = cmul_conj(xbuf[ys_a(k0, lane)], T1t[k0 * 64 + lane]);
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], e32(T * m));
}
// "qs" form of the twisted halves (the two-gate N = 2048 kernel, whose LDS has no room for the two T1 tables): the pass-1 twiddles are
// rebuilt from per-lane roots, T1_T[k0][lane] = b_T s^k0 with b_T = zeta^(T lane), s = zeta^(8 lane), and the first transpose happens in
// registers (wave_transpose_hi3) -- same spectra order, so key spectra made by the table variant multiply with these.
THFHE_FN cplx e64(int q) {  // exp(i pi q / 64)
    constexpr double C[33] = {1.0, 0.998795456205172392715, 0.995184726672196886245, 0.989176509964780973452, 0.980785280403230449126,
                              0.970031253194543992604, 0.956940335732208864936, 0.941544065183020778413, 0.923879532511286756128,
                              0.903989293123443331586, 0.881921264348355029713, 0.857728610000272069902, 0.831469612302545237079,
                              0.803207531480644909807, 0.773010453362736960811, 0.740951125354959091176, 0.707106781186547524401,
                              0.671558954847018400625, 0.634393284163645498215, 0.595699304492433343467, 0.555570233019602224743,
                              0.514102744193221726594, 0.471396736825997648556, 0.427555093430282094321, 0.382683432365089771728,
                              0.336889853392220050689, 0.290284677254462367636, 0.242980179903263889948, 0.195090322016128267848,
                              0.146730474455361751659, 0.0980171403295606019942, 0.049067674327418014255, 0.0};
    const int quad = (q >> 5) & 3, r = q & 31;
    const double c = C[r], s = C[32 - r];
    return quad == 0 ? cplx{c, s} : quad == 1 ? cplx{-s, c} : quad == 2 ? cplx{-c, -s} : cplx{s, -c};
}
// pass-1 constants zeta^(64 T m) of the twisted halves (DEN = 32: ring of degree 2048) / quarters (DEN = 64: degree 4096)
template <int DEN>
THFHE_FN cplx e_den(int q) {
    return DEN == 32 ? e32(q) : e64(q);
}
template <int T, int DEN = 32>
THFHE_FN void fwdtq_seg1(cplx (&z)[8], const LaneRoots &r) {
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul(z[m], e_den<DEN>(T * m));
    dft8<+1>(z);
    const cplx s2 = cmul(r.s, r.s);
    cplx e = r.b, o = cmul(r.b, r.s);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 2) {
        z[k0] = cmul(z[k0], e);
        z[k0 + 1] = cmul(z[k0 + 1], o);
        if (k0 < 6) {
            e = cmul(e, s2);
            o = cmul(o, s2);
        }
    }
}
template <int T, int DEN = 32>
THFHE_FN void invtq_seg3(cplx (&z)[8], const LaneRoots &r) {
    const cplx s2 = cmul(r.s, r.s);
    cplx e = r.b, o = cmul(r.b, r.s);
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 2) {
        z[k0] = cmul_conj(z[k0], e);
        z[k0 + 1] = cmul_conj(z[k0 + 1], o);
        if (k0 < 6) {
            e = cmul(e, s2);
            o = cmul(o, s2);
        }
    }
    dft8<-1>(z);
#pragma unroll
    for (int m = 1; m < 8; m++) z[m] = cmul_conj(z[m], e_den<DEN>(T * m));
}
// radix-2 split of the 16 folded points a lane holds (z[m] <-> j = lane + 64 m) and its inverse (unnormalised: x2)
THFHE_FN void split2048(const cplx (&z)[16], cplx (&y0)[8], cplx (&y1)[8]) {
    constexpr double R = 0.70710678118654752440;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const cplx w{(z[m + 8].re - z[m + 8].im) * R, (z[m + 8].re + z[m + 8].im) * R};  // e^{i pi/4} z
        y0[m] = cadd(z[m], w);
        y1[m] = csub(z[m], w);
    }
}
THFHE_FN void merge2048(const cplx (&a)[8], const cplx (&b)[8], cplx (&lo)[8], cplx (&hi)[8]) {
    constexpr double R = 0.70710678118654752440;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const cplx d = csub(a[m], b[m]);
        lo[m] = cadd(a[m], b[m]);
        hi[m] = cplx{(d.re + d.im) * R, (d.im - d.re) * R};  // e^{-i pi/4} d
    }
}
// ---- N = 4096 (the 64-party "for fft" and 512-party 3-gen sets, J/mk_api.jl:277-283, 316-322): 2048 complex points = one radix-4 split +
// four twisted 512-point transforms.  z_j = p_j + i p_{j+2048}, P_k = sum_{j<2048} z_j zeta^(j(4k+1)), zeta = exp(i pi / 4096).  With
// j = j' + 512 s and k = 4 k'' + t:   zeta^(j(4k+1)) = omega_512^(j' k'') zeta^(j'(4t+1)) (e^{i pi/8} i^t)^s, so
//     P_{4k''+t} = sum_{j'<512} y^t_j' zeta^(T j') omega_512^(j' k''),   y^t_j' = sum_s (e^{i pi/8} i^t)^s z_{j'+512 s},   T = 4t + 1
// i.e. quarter t is the twisted 512-point transform with twist T in {1, 5, 9, 13}: constants exp(i pi T m / 64), per-lane root
// b_T = zeta^(T lane), the same ratio exp(i pi lane / 256) as on the ring of degree 2048.  Spectrum order [quarter][slot m][lane].
// u_s = e^{i pi s/8} z_s for the four points of a radix-4 group
THFHE_FN void pre4096(cplx (&z)[4]) {
    constexpr double C8 = 0.923879532511286756128, S8 = 0.382683432365089771728, R = 0.70710678118654752440;
    z[1] = cplx{z[1].re * C8 - z[1].im * S8, z[1].re * S8 + z[1].im * C8};
    z[2] = cplx{(z[2].re - z[2].im) * R, (z[2].re + z[2].im) * R};
    z[3] = cplx{z[3].re * S8 - z[3].im * C8, z[3].re * C8 + z[3].im * S8};
}
// y^t = u0 + i^t u1 + (-1)^t u2 + (-i)^t u3
template <int QT>
THFHE_FN cplx comb4096(const cplx (&u)[4]) {
    if (QT == 0) return cplx{u[0].re + u[1].re + u[2].re + u[3].re, u[0].im + u[1].im + u[2].im + u[3].im};
    if (QT == 1) return cplx{u[0].re - u[1].im - u[2].re + u[3].im, u[0].im + u[1].re - u[2].im - u[3].re};
    if (QT == 2) return cplx{u[0].re - u[1].re + u[2].re - u[3].re, u[0].im - u[1].im + u[2].im - u[3].im};
    return cplx{u[0].re + u[1].im - u[2].re - u[3].im, u[0].im - u[1].re - u[2].im + u[3].re};
}
// inverse (unnormalised: x4): z_s = e^{-i pi s/8} (a0 + (-i)^s a1 + (-1)^s a2 + i^s a3)
THFHE_FN void merge4096(const cplx &a0, const cplx &a1, const cplx &a2, const cplx &a3, cplx (&z)[4]) {
    constexpr double C8 = 0.923879532511286756128, S8 = 0.382683432365089771728, R = 0.70710678118654752440;
    z[0] = cplx{a0.re + a1.re + a2.re + a3.re, a0.im + a1.im + a2.im + a3.im};
    const cplx v1{a0.re + a1.im - a2.re - a3.im, a0.im - a1.re - a2.im + a3.re};
    const cplx v2{a0.re - a1.re + a2.re - a3.re, a0.im - a1.im + a2.im - a3.im};
    const cplx v3{a0.re - a1.im - a2.re + a3.im, a0.im + a1.re - a2.im - a3.re};
    z[1] = cplx{v1.re * C8 + v1.im * S8, v1.im * C8 - v1.re * S8};
    z[2] = cplx{(v2.re + v2.im) * R, (v2.im - v2.re) * R};
    z[3] = cplx{v3.re * S8 + v3.im * C8, v3.im * S8 - v3.re * C8};
}
// generic-degree integer helpers (NN = ring degree, a power of two)
template <int NN>
THFHE_FN uint64_t rot_minus_self64_n(const int64_t *p, int q, int a2n) {
    int e = (q - a2n) & (2 * NN - 1);
    uint64_t r = (uint64_t)p[e & (NN - 1)];
    if (e & NN) r = 0ull - r;
    return r - (uint64_t)p[q];
}
// out[k] = (X^a p - p)[q0 + 64 k], k < K: all 2 K LDS reads first, the sign / subtraction after a scheduling fence (left to itself the compiler reads,
// waits for and uses one word at a time in the kernels that stage rotated words: K exposed LDS round trips instead of one)
template <int NN, int K>
THFHE_FN void rot_minus_self64_batch(const int64_t *p, int q0, int a2n, uint64_t (&out)[K]) {
    uint64_t r[K], s[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int e = (q0 + 64 * k - a2n) & (2 * NN - 1);
        r[k] = (uint64_t)p[e & (NN - 1)];
        s[k] = (uint64_t)p[q0 + 64 * k];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int e = (q0 + 64 * k - a2n) & (2 * NN - 1);
        out[k] = ((e & NN) ? 0ull - r[k] : r[k]) - s[k];
    }
}
// t[m] = top 32 bits of (X^a acc_j - acc_j)[lane + 64 m] + offset, m = 0..31   (N = 2048)
THFHE_FN void load_rotated32_hi(int lane, const int64_t *acc_poly, int a2n, uint64_t offset, uint32_t (&t)[32]) {
#pragma unroll
    for (int m = 0; m < 32; m++) t[m] = (uint32_t)((rot_minus_self64_n<2048>(acc_poly, lane + 64 * m, a2n) + offset) >> 32);
}
THFHE_FN void digits_to_z16(const uint32_t (&t)[32], int p, int Bgbit, cplx (&z)[16]) {
    const int shift = 32 - p * Bgbit;
    const uint32_t mask = (1u << Bgbit) - 1u;
    const int32_t half = 1 << (Bgbit - 1);
#pragma unroll
    for (int m = 0; m < 16; m++) z[m] = cplx{digit32(t[m], shift, mask, half), digit32(t[m + 16], shift, mask, half)};
}
THFHE_FN void key_limbs64_to_z16(int lane, const int64_t *poly, int h, cplx (&z)[16]) {  // limb h of a degree-2048 key polynomial
#pragma unroll
    for (int m = 0; m < 16; m++) {
        double a[4], b[4];
        split_limbs64(poly[lane + 64 * m], a);
        split_limbs64(poly[lane + 64 * m + 1024], b);
        z[m] = cplx{a[h], b[h]};
    }
}
template <int NN>
THFHE_FN void acc_init_64_n(int lane, int64_t *acc_mask, int64_t *acc_body, int barb, int64_t mu) {
#pragma unroll
    for (int m = 0; m < NN / 64; m++) {
        int q = lane + 64 * m;
        int e = (q + barb) & (2 * NN - 1);
        acc_mask[q] = 0;
        acc_body[q] = (e & NN) ? (int64_t)(0ull - (uint64_t)mu) : mu;
    }
}
template <int NN>
THFHE_FN void extract_64_n(int lane, const int64_t *acc_mask, const int64_t *acc_body, int32_t *out) {
#pragma unroll
    for (int m = 0; m < NN / 64; m++) {
        int q = lane + 64 * m;
        out[q] = t64tot32(q == 0 ? acc_mask[0] : (int64_t)(0ull - (uint64_t)acc_mask[NN - q]));
    }
    if (lane == 0) out[NN] = t64tot32(acc_body[0]);
}
// spectral key stream for N = 2048: [party*n + i][row r][limb h][output o][half][slot m][lane], 16 KiB per (pi, r, h, o)
THFHE_FN size_t mk_chunk_index_2k(long pi, int r, int h, int o, int rows) { return mk_chunk_index(pi, r, h, o, rows) * 2; }  // * 512 complex

}  // namespace thfhe

// ---- host-side twiddle table (double precision from long double; identical bytes on device) ---------
#include <cmath>
namespace thfhe {
// T1[k0*64 + lane] = exp(i pi lane (4 k0 + 1) / 1024), T2[k1*8 + j0] = exp(2 pi i j0 k1 / 64)
inline void make_twiddles_1024(cplx *T1 /*512*/, cplx *T2 /*64*/) {
    const long double PI = 3.14159265358979323846264338327950288L;
    for (int k0 = 0; k0 < 8; k0++)
        for (int lane = 0; lane < 64; lane++) {
            long double ang = PI * (long double)(lane * (4 * k0 + 1)) / 1024.0L;
            T1[k0 * 64 + lane] = cplx{(double)cosl(ang), (double)sinl(ang)};
        }
    for (int k1 = 0; k1 < 8; k1++)
        for (int j0 = 0; j0 < 8; j0++) {
            long double ang = 2.0L * PI * (long double)(j0 * k1) / 64.0L;
            T2[k1 * 8 + j0] = cplx{(double)cosl(ang), (double)sinl(ang)};
        }
}
// per-lane roots of variant "r": roots[2*lane] = exp(i pi lane / 1024), roots[2*lane + 1] = exp(i pi 4 lane / 1024)
inline void make_lane_roots_1024(cplx *roots /*128*/) {
    const long double PI = 3.14159265358979323846264338327950288L;
    for (int lane = 0; lane < 64; lane++) {
        long double a = PI * (long double)lane / 1024.0L, b = PI * (long double)(4 * lane) / 1024.0L;
        roots[2 * lane] = cplx{(double)cosl(a), (double)sinl(a)};
        roots[2 * lane + 1] = cplx{(double)cosl(b), (double)sinl(b)};
    }
}
// N = 2048, "qs" form: the common ratio s[lane] = exp(i pi 8 lane / 2048) of the pass-1 twiddles (b_T = T1_T[0][lane] comes from the tables)
inline void make_lane_ratio_2048(cplx *s /*64*/) {
    const long double PI = 3.14159265358979323846264338327950288L;
    for (int lane = 0; lane < 64; lane++) {
        long double a = PI * (long double)(8 * lane) / 2048.0L;
        s[lane] = cplx{(double)cosl(a), (double)sinl(a)};
    }
}
// N = 4096: b_T[lane] = exp(i pi T lane / 4096) for the four twists T = 1, 5, 9, 13 (roots[q * 64 + lane], q = quarter)
inline void make_lane_roots_4096(cplx *roots /*256*/) {
    const long double PI = 3.14159265358979323846264338327950288L;
    for (int q = 0; q < 4; q++)
        for (int lane = 0; lane < 64; lane++) {
            long double a = PI * (long double)((4 * q + 1) * lane) / 4096.0L;
            roots[q * 64 + lane] = cplx{(double)cosl(a), (double)sinl(a)};
        }
}
// N = 2048: T1_T[k0*64 + lane] = exp(i pi lane (8 k0 + T) / 2048) for the two twists T = 1 (even outputs) and T = 5 (odd outputs)
inline void make_twiddles_2048(cplx *T1a /*512, T = 1*/, cplx *T1b /*512, T = 5*/) {
    const long double PI = 3.14159265358979323846264338327950288L;
    for (int k0 = 0; k0 < 8; k0++)
        for (int lane = 0; lane < 64; lane++) {
            long double a1 = PI * (long double)(lane * (8 * k0 + 1)) / 2048.0L, a5 = PI * (long double)(lane * (8 * k0 + 5)) / 2048.0L;
            T1a[k0 * 64 + lane] = cplx{(double)cosl(a1), (double)sinl(a1)};
            T1b[k0 * 64 + lane] = cplx{(double)cosl(a5), (double)sinl(a5)};
        }
}
}  // namespace thfhe

#endif  // THFHE_LANE_H
