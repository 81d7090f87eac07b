// lane_emu.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libthfhe_hip.so, never reachable from the
// product API).  Replays the per-lane segments of torus-fhe_amd/csrc/thfhe_lane.h on the host, looping
// over the 64 lanes of a wavefront between the wave-level LDS exchanges, so the kernel's index algebra
// and FP64 exactness margin are checked against the CPU oracle in the `-m "not gpu"` suite.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../torus-fhe_amd/csrc/thfhe_lane.h"

using namespace thfhe;

namespace {
struct Wave {
    cplx T1[512], T2[64];
    cplx xbuf[kXbufSlots];
    Wave() { make_twiddles_1024(T1, T2); }
    void fwd(cplx (*z)[8]) {  // z[lane][8]
        for (int l = 0; l < 64; l++) fwd_seg1(l, z[l], xbuf, T1);
        for (int l = 0; l < 64; l++) fwd_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) fwd_seg2_st(l, z[l], xbuf, T2);
        for (int l = 0; l < 64; l++) fwd_seg3(l, z[l], xbuf);
    }
    void inv(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) inv_seg1(l, z[l], xbuf, T2);
        for (int l = 0; l < 64; l++) inv_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) inv_seg2_st(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) inv_seg3(l, z[l], xbuf, T1);
    }
};
struct WaveS {  // the LDS-ring kernel's variant: swizzled 512-slot buffer, pass-2 twiddles as powers
    cplx T1[512], T2[64];
    cplx xbuf[512];
    W64 w[64];
    WaveS() {
        make_twiddles_1024(T1, T2);
        for (int l = 0; l < 64; l++) w[l] = W64{T2[1 * 8 + (l & 7)]};
    }
    void fwd(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) fwds_seg1(l, z[l], xbuf, T1);
        for (int l = 0; l < 64; l++) fwds_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) fwds_seg2_st(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) fwds_seg3(l, z[l], xbuf);
    }
    void inv(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) invs_seg1(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) invs_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) invs_seg2_st(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) invs_seg3(l, z[l], xbuf, T1);
    }
};
struct WaveR {  // second-generation ring kernel: padded buffer, pass-1 twiddles from per-lane roots, pass-2 twiddles as powers
    cplx T1[512], T2[64], roots[128];
    cplx xbuf[kXbufSlots];
    W64 w[64];
    LaneRoots r[64];
    WaveR() {
        make_twiddles_1024(T1, T2);
        make_lane_roots_1024(roots);
        for (int l = 0; l < 64; l++) {
            w[l] = W64{T2[1 * 8 + (l & 7)]};
            r[l] = LaneRoots{roots[2 * l], roots[2 * l + 1]};
        }
    }
    void fwd(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) fwdr_seg1(l, z[l], xbuf, r[l]);
        for (int l = 0; l < 64; l++) fwd_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) fwdr_seg2_st(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) fwd_seg3(l, z[l], xbuf);
    }
    void inv(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) invr_seg1(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) inv_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) inv_seg2_st(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) invr_seg3(l, z[l], xbuf, r[l]);
    }
};
// host model of wave_transpose_hi3: register index (bits 2,1,0) <-> lane bits (5,4,3)
void lanes_transpose_hi3(cplx (*z)[8]) {
    static cplx t[64][8];
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 8; r++) t[(l & 7) | (r << 3)][l >> 3] = z[l][r];
    memcpy(z, t, sizeof(t));
}
struct WaveQ {  // third-generation ring kernel: as WaveR, first transpose in registers
    WaveR base;
    void fwd(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) fwdq_seg1(z[l], base.r[l]);
        lanes_transpose_hi3(z);
        for (int l = 0; l < 64; l++) fwdr_seg2_st(l, z[l], base.xbuf, base.w[l]);
        for (int l = 0; l < 64; l++) fwd_seg3(l, z[l], base.xbuf);
    }
    void inv(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) invr_seg1(l, z[l], base.xbuf, base.w[l]);
        for (int l = 0; l < 64; l++) {
            inv_seg2_ld(l, z[l], base.xbuf);
            dft8<-1>(z[l]);
        }
        lanes_transpose_hi3(z);
        for (int l = 0; l < 64; l++) invq_seg3(z[l], base.r[l]);
    }
};
struct WaveQS {  // multi-key kernels: first transpose in registers, second through the XOR-swizzled 512-slot buffer (no room for padding)
    WaveR base;
    void fwd(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) fwdq_seg1(z[l], base.r[l]);
        lanes_transpose_hi3(z);
        for (int l = 0; l < 64; l++) fwds_seg2_st(l, z[l], base.xbuf, base.w[l]);
        for (int l = 0; l < 64; l++) fwds_seg3(l, z[l], base.xbuf);
    }
    void inv(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) invs_seg1(l, z[l], base.xbuf, base.w[l]);
        for (int l = 0; l < 64; l++) {
            invs_seg2_ld(l, z[l], base.xbuf);
            dft8<-1>(z[l]);
        }
        lanes_transpose_hi3(z);
        for (int l = 0; l < 64; l++) invq_seg3(z[l], base.r[l]);
    }
};
}  // namespace

extern "C" {

// coefficient-domain key polynomials -> spectral layout [poly][limb][m][lane], scaled by 1/512
void emu_transform_key_polys(const int32_t *polys, int64_t npolys, double *spec /* npolys*2*512*2 doubles */) {
    Wave w;
    static cplx zlo[64][8], zhi[64][8];
    for (int64_t q = 0; q < npolys; q++) {
        for (int l = 0; l < 64; l++) key_limbs_to_z(l, polys + q * 1024, zlo[l], zhi[l]);
        w.fwd(zlo);
        w.fwd(zhi);
        cplx *out = reinterpret_cast<cplx *>(spec) + q * 2 * 512;
        for (int l = 0; l < 64; l++)
            for (int m = 0; m < 8; m++) {
                out[m * 64 + l] = cplx{zlo[l][m].re * (1.0 / 512), zlo[l][m].im * (1.0 / 512)};
                out[512 + m * 64 + l] = cplx{zhi[l][m].re * (1.0 / 512), zhi[l][m].im * (1.0 / 512)};
            }
    }
}

// exact negacyclic product of a small-coefficient polynomial with a Torus32 polynomial, via the lane code.
// Also returns the worst distance of any inverse-transform output from an integer (exactness margin).
double emu_polymul(const int32_t *small, const int32_t *b, int32_t *out) {
    Wave w;
    std::vector<double> spec(2 * 512 * 2);
    emu_transform_key_polys(b, 1, spec.data());
    const cplx *B = reinterpret_cast<const cplx *>(spec.data());
    static cplx z[64][8], slo[64][8], shi[64][8];
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) z[l][m] = cplx{(double)small[l + 64 * m], (double)small[l + 64 * m + 512]};
    w.fwd(z);
    memset(slo, 0, sizeof(slo));
    memset(shi, 0, sizeof(shi));
    for (int l = 0; l < 64; l++) {
        mac8(l, slo[l], z[l], B);
        mac8(l, shi[l], z[l], B + 512);
    }
    w.inv(slo);
    w.inv(shi);
    double worst = 0;
    std::vector<int32_t> acc(1024, 0);
    for (int l = 0; l < 64; l++) {
        for (int m = 0; m < 8; m++)
            for (double v : {slo[l][m].re, slo[l][m].im, shi[l][m].re, shi[l][m].im}) {
                double d = v - __builtin_rint(v);
                if (d < 0) d = -d;
                if (d > worst) worst = d;
            }
        acc_update16(l, acc.data(), slo[l], shi[l]);
    }
    memcpy(out, acc.data(), sizeof(int32_t) * 1024);
    return worst;
}

// one CMux on acc[2][1024] with the spectral key of index i (bk_spec laid out by emu_transform_key_polys over
// the coefficient table [n][2l][2][1024], i.e. poly index ((i*2l + r)*2 + c)).   Mirrors blind_rotate_kernel.
void emu_mux_rotate(const double *bk_spec, int l_levels, int Bgbit, int i, int barai, int32_t *acc) {
    Wave w;
    const cplx *BK = reinterpret_cast<const cplx *>(bk_spec);
    const int rows = 2 * l_levels;
    const uint32_t offset = decomp_offset32(l_levels, Bgbit);
    const int a2n = barai & 2047;
    static cplx S[64][2][2][8], z[64][8];
    static uint32_t t[64][16];
    memset(S, 0, sizeof(S));
    for (int j = 0; j < 2; j++) {
        for (int l = 0; l < 64; l++) load_rotated16(l, acc + j * 1024, a2n, offset, t[l]);
        for (int p = 1; p <= l_levels; p++) {
            for (int l = 0; l < 64; l++) digits_to_z(t[l], p, Bgbit, z[l]);
            w.fwd(z);
            int r = j * l_levels + (p - 1);
            for (int l = 0; l < 64; l++)
                for (int c = 0; c < 2; c++)
                    for (int h = 0; h < 2; h++) mac8(l, S[l][c][h], z[l], BK + bk_spec_index(i, r, c, h, rows));
        }
    }
    for (int c = 0; c < 2; c++) {
        static cplx lo[64][8], hi[64][8];
        for (int l = 0; l < 64; l++) {
            memcpy(lo[l], S[l][c][0], sizeof(lo[l]));
            memcpy(hi[l], S[l][c][1], sizeof(hi[l]));
        }
        w.inv(lo);
        w.inv(hi);
        for (int l = 0; l < 64; l++) acc_update16(l, acc + c * 1024, lo[l], hi[l]);
    }
}

// whole blind rotation + extraction of one job (bara[n], barb given), mirrors the kernel's control flow
void emu_blind_rotate(const double *bk_spec, int n, int l_levels, int Bgbit, const int32_t *bara, int barb, int32_t mu,
                      int32_t *out /* 1025 */) {
    std::vector<int32_t> acc(2048);
    for (int l = 0; l < 64; l++) acc_init16(l, acc.data(), acc.data() + 1024, barb, mu);
    for (int i = 0; i < n; i++)
        if (bara[i] != 0) emu_mux_rotate(bk_spec, l_levels, Bgbit, i, bara[i], acc.data());
    for (int l = 0; l < 64; l++) extract16(l, acc.data(), acc.data() + 1024, out);
}
}

extern "C" {
// debug/unit-test entry points: raw transforms.  zin: 512 complex in natural order j (z_j = p_j + i p_{j+512});
// out[lane][m] complex in register order.
void emu_fwd_raw(const double *zin, double *out) {
    Wave w;
    static cplx z[64][8];
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) z[l][m] = cplx{zin[2 * (l + 64 * m)], zin[2 * (l + 64 * m) + 1]};
    w.fwd(z);
    memcpy(out, z, sizeof(z));
}
void emu_inv_raw(const double *in, double *zout) {
    Wave w;
    static cplx z[64][8];
    memcpy(z, in, sizeof(z));
    w.inv(z);
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) {
            zout[2 * (l + 64 * m)] = z[l][m].re;
            zout[2 * (l + 64 * m) + 1] = z[l][m].im;
        }
}
}

extern "C" {
// the fused rotate + decompose of the second-generation ring kernel must give the digits of load_rotated16 + digits_to_z; returns mismatches
int emu_rotated_digits_crosscheck(const int32_t *acc /*[1024]*/, int a2n, int l, int Bgbit) {
    int bad = 0;
    const uint32_t offset = decomp_offset32(l, Bgbit);
    for (int lane = 0; lane < 64; lane++) {
        uint32_t t[16];
        load_rotated16(lane, acc, a2n, offset, t);
        for (int p = 1; p <= l; p++) {
            cplx z0[8], z1[8];
            digits_to_z(t, p, Bgbit, z0);
            rotated_digits_z(lane, acc, a2n, p, l, Bgbit, z1);
            for (int m = 0; m < 8; m++) bad += (z0[m].re != z1[m].re) + (z0[m].im != z1[m].im);
            uint32_t f7[7], f12[16];   // two-step form (fields once per polynomial, one extract per level), some / all fields kept
            rotated_fields_keep<7>(lane, acc, a2n, l, Bgbit, f7);
            mixed_digits_z<7>(lane, acc, a2n, p, l, Bgbit, f7, z1);
            for (int m = 0; m < 8; m++) bad += (z0[m].re != z1[m].re) + (z0[m].im != z1[m].im);
            rotated_fields_keep<16>(lane, acc, a2n, l, Bgbit, f12);
            mixed_digits_z<16>(lane, acc, a2n, p, l, Bgbit, f12, z1);
            for (int m = 0; m < 8; m++) bad += (z0[m].re != z1[m].re) + (z0[m].im != z1[m].im);
        }
    }
    return bad;
}
}

extern "C" {
// swizzled-variant transforms must give the same spectra (same register order) as the padded variant, and the same
// exact products: forward with one variant, inverse with the other.
double emu_variant_crosscheck(const int32_t *small, const int32_t *b, int32_t *out) {
    Wave w;
    WaveS ws;
    std::vector<double> spec(2 * 512 * 2);
    emu_transform_key_polys(b, 1, spec.data());   // key transformed by the padded variant
    const cplx *B = reinterpret_cast<const cplx *>(spec.data());
    static cplx z[64][8], z2[64][8], slo[64][8], shi[64][8];
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) z2[l][m] = z[l][m] = cplx{(double)small[l + 64 * m], (double)small[l + 64 * m + 512]};
    ws.fwd(z);   // digits transformed by the swizzled variant
    w.fwd(z2);
    double dmax = 0;
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) {
            double d = __builtin_fabs(z[l][m].re - z2[l][m].re) + __builtin_fabs(z[l][m].im - z2[l][m].im);
            if (d > dmax) dmax = d;
        }
    memset(slo, 0, sizeof(slo));
    memset(shi, 0, sizeof(shi));
    for (int l = 0; l < 64; l++) {
        mac8(l, slo[l], z[l], B);
        mac8(l, shi[l], z[l], B + 512);
    }
    ws.inv(slo);
    ws.inv(shi);
    std::vector<int32_t> acc(1024, 0);
    for (int l = 0; l < 64; l++) acc_update16(l, acc.data(), slo[l], shi[l]);
    memcpy(out, acc.data(), sizeof(int32_t) * 1024);
    return dmax;
}
}

// variant "r" (computed pass-1 twiddles): same spectra as the table variant up to rounding, and exact products with margin.
// Returns the worst distance of an inverse-transform output from an integer; *dmax = largest spectrum difference to the table variant.
template <class WV>
static double variant_crosscheck(const int32_t *small, const int32_t *b, int32_t *out, double *dmax_out) {
    Wave w;
    WV wr;
    std::vector<double> spec(2 * 512 * 2);
    emu_transform_key_polys(b, 1, spec.data());   // key transformed by the table variant (as sk_key_transform_kernel does)
    const cplx *B = reinterpret_cast<const cplx *>(spec.data());
    static cplx z[64][8], z2[64][8], slo[64][8], shi[64][8];
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) z2[l][m] = z[l][m] = cplx{(double)small[l + 64 * m], (double)small[l + 64 * m + 512]};
    wr.fwd(z);
    w.fwd(z2);
    double dmax = 0;
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) {
            double d = __builtin_fabs(z[l][m].re - z2[l][m].re) + __builtin_fabs(z[l][m].im - z2[l][m].im);
            if (d > dmax) dmax = d;
        }
    *dmax_out = dmax;
    memset(slo, 0, sizeof(slo));
    memset(shi, 0, sizeof(shi));
    for (int l = 0; l < 64; l++) {
        mac8(l, slo[l], z[l], B);
        mac8(l, shi[l], z[l], B + 512);
    }
    wr.inv(slo);
    wr.inv(shi);
    double worst = 0;
    std::vector<int32_t> acc(1024, 0);
    for (int l = 0; l < 64; l++) {
        for (int m = 0; m < 8; m++)
            for (double v : {slo[l][m].re, slo[l][m].im, shi[l][m].re, shi[l][m].im}) {
                double d = v - __builtin_rint(v);
                if (d < 0) d = -d;
                if (d > worst) worst = d;
            }
        acc_update16(l, acc.data(), slo[l], shi[l]);
    }
    memcpy(out, acc.data(), sizeof(int32_t) * 1024);
    return worst;
}
extern "C" {
double emu_roots_variant_crosscheck(const int32_t *small, const int32_t *b, int32_t *out, double *dmax_out) {
    return variant_crosscheck<WaveR>(small, b, out, dmax_out);
}
double emu_regtranspose_variant_crosscheck(const int32_t *small, const int32_t *b, int32_t *out, double *dmax_out) {
    return variant_crosscheck<WaveQ>(small, b, out, dmax_out);
}
double emu_regtranspose_swizzled_variant_crosscheck(const int32_t *small, const int32_t *b, int32_t *out, double *dmax_out) {
    return variant_crosscheck<WaveQS>(small, b, out, dmax_out);
}
}

extern "C" {
// ---- 3-gen multi-key: key transform + one CMux, mirroring mk_blind_rotate_ring_kernel's data flow --------------------
// bk: int64[P][n][4][l][1024] (coefficient domain) -> spectral stream [pi][r][h][o][m][lane], scaled by 1/512
void emu_mk_transform_key(const int64_t *bk, long PN /* parties*n */, int l, double *spec) {
    WaveS w;
    const int rows = 2 * l;
    static cplx z[4][64][8];
    cplx *out = reinterpret_cast<cplx *>(spec);
    for (long pi = 0; pi < PN; pi++)
        for (int r = 0; r < rows; r++)
            for (int o = 0; o < 2; o++) {
                const int j = r / l, lv = r % l;
                const int64_t *poly = bk + (((size_t)pi * 4 + mk_part_index(j, o)) * l + lv) * 1024;
                for (int ln = 0; ln < 64; ln++) {
                    cplx zz[4][8];
                    key_limbs64_to_z(ln, poly, zz);
                    for (int h = 0; h < 4; h++) memcpy(z[h][ln], zz[h], sizeof(zz[h]));
                }
                for (int h = 0; h < 4; h++) {
                    w.fwd(z[h]);
                    cplx *dst = out + mk_chunk_index(pi, r, h, o, rows) * 512;
                    for (int ln = 0; ln < 64; ln++)
                        for (int m = 0; m < 8; m++) dst[m * 64 + ln] = cplx{z[h][ln][m].re * (1.0 / 512), z[h][ln][m].im * (1.0 / 512)};
                }
            }
}
void emu_mk_mux_rotate(const double *spec, int l, int Bgbit, long pi, int barai, int64_t *acc /* [2][1024] */) {
    WaveS w;
    const cplx *BK = reinterpret_cast<const cplx *>(spec);
    const int rows = 2 * l, a2n = barai & 2047;
    const uint64_t offset = decomp_offset64(l, Bgbit);
    static cplx S[2][64][4][8], z[64][8];
    static uint32_t t[64][16];
    memset(S, 0, sizeof(S));
    for (int r = 0; r < rows; r++) {
        if (r % l == 0)
            for (int ln = 0; ln < 64; ln++) load_rotated16_hi(ln, acc + (r / l) * 1024, a2n, offset, t[ln]);
        // digits of the 64-bit value with Bgbit bits per level are digits of its top 32 bits
        for (int ln = 0; ln < 64; ln++) digits_to_z(t[ln], (r % l) + 1, Bgbit, z[ln]);
        w.fwd(z);
        for (int o = 0; o < 2; o++)
            for (int h = 0; h < 4; h++)
                for (int ln = 0; ln < 64; ln++) mac8(ln, S[o][ln][h], z[ln], BK + mk_chunk_index(pi, r, h, o, rows) * 512);
    }
    for (int o = 0; o < 2; o++) {
        static cplx lim[4][64][8];
        for (int h = 0; h < 4; h++) {
            for (int ln = 0; ln < 64; ln++) memcpy(lim[h][ln], S[o][ln][h], sizeof(lim[h][ln]));
            w.inv(lim[h]);
        }
        for (int ln = 0; ln < 64; ln++) {
            cplx SS[4][8];
            for (int h = 0; h < 4; h++) memcpy(SS[h], lim[h][ln], sizeof(SS[h]));
            acc_update16_64(ln, acc + o * 1024, SS);
        }
    }
}
void emu_mk_extract(const int64_t *acc, int32_t *out) {
    for (int ln = 0; ln < 64; ln++) extract16_64(ln, acc, acc + 1024, out);
}
}

// ---- N = 2048: radix-2 split + two twisted 512-point transforms (thfhe_lane.h, "N = 2048" section) ---------------------------
namespace {
struct Wave2K {
    cplx T1a[512], T1b[512], T2[64], scratch[512];
    cplx xbuf[512];
    W64 w[64];
    Wave2K() {
        make_twiddles_2048(T1a, T1b);
        make_twiddles_1024(scratch, T2);
        for (int l = 0; l < 64; l++) w[l] = W64{T2[1 * 8 + (l & 7)]};
    }
    template <int T>
    void fwd(cplx (*z)[8]) {
        const cplx *T1 = T == 1 ? T1a : T1b;
        for (int l = 0; l < 64; l++) fwdt_seg1<T>(l, z[l], xbuf, T1);
        for (int l = 0; l < 64; l++) fwds_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) fwds_seg2_st(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) fwds_seg3(l, z[l], xbuf);
    }
    template <int T>
    void inv(cplx (*z)[8]) {
        const cplx *T1 = T == 1 ? T1a : T1b;
        for (int l = 0; l < 64; l++) invs_seg1(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) invs_seg2_ld(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) invs_seg2_st(l, z[l], xbuf);
        for (int l = 0; l < 64; l++) invt_seg3<T>(l, z[l], xbuf, T1);
    }
    // z16[lane][16] -> spectra y0[lane][8] (even outputs), y1[lane][8] (odd outputs)
    void fwd2k(cplx (*z16)[16], cplx (*y0)[8], cplx (*y1)[8]) {
        for (int l = 0; l < 64; l++) split2048(z16[l], y0[l], y1[l]);
        fwd<1>(y0);
        fwd<5>(y1);
    }
    void inv2k(cplx (*s0)[8], cplx (*s1)[8], cplx (*lo)[8], cplx (*hi)[8]) {  // returns 1024 * z
        inv<1>(s0);
        inv<5>(s1);
        for (int l = 0; l < 64; l++) merge2048(s0[l], s1[l], lo[l], hi[l]);
    }
};
}  // namespace

extern "C" {
// zin: 1024 complex in natural order j; out: [half][lane][m] complex; back: inverse of out, natural order, scaled by 1024
void emu_fwd_raw_2k(const double *zin, double *out, double *back) {
    Wave2K w;
    static cplx z[64][16], y0[64][8], y1[64][8], lo[64][8], hi[64][8];
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 16; m++) z[l][m] = cplx{zin[2 * (l + 64 * m)], zin[2 * (l + 64 * m) + 1]};
    w.fwd2k(z, y0, y1);
    memcpy(out, y0, sizeof(y0));
    memcpy(out + 64 * 8 * 2, y1, sizeof(y1));
    w.inv2k(y0, y1, lo, hi);
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) {
            back[2 * (l + 64 * m)] = lo[l][m].re;
            back[2 * (l + 64 * m) + 1] = lo[l][m].im;
            back[2 * (l + 64 * (m + 8))] = hi[l][m].re;
            back[2 * (l + 64 * (m + 8)) + 1] = hi[l][m].im;
        }
}
// bk: int64[PN][4][l][2048] -> spectral stream [pi][r][h][o][half][m][lane], scaled by 1/1024
void emu_mk_transform_key_2k(const int64_t *bk, long PN, int l, double *spec) {
    Wave2K w;
    const int rows = 2 * l;
    static cplx z[64][16], y0[64][8], y1[64][8];
    cplx *out = reinterpret_cast<cplx *>(spec);
    for (long pi = 0; pi < PN; pi++)
        for (int r = 0; r < rows; r++)
            for (int o = 0; o < 2; o++)
                for (int h = 0; h < 4; h++) {
                    const int j = r / l, lv = r % l;
                    const int64_t *poly = bk + (((size_t)pi * 4 + mk_part_index(j, o)) * l + lv) * 2048;
                    for (int ln = 0; ln < 64; ln++) key_limbs64_to_z16(ln, poly, h, z[ln]);
                    w.fwd2k(z, y0, y1);
                    cplx *dst = out + mk_chunk_index_2k(pi, r, h, o, rows) * 512;
                    for (int ln = 0; ln < 64; ln++)
                        for (int m = 0; m < 8; m++) {
                            dst[m * 64 + ln] = cplx{y0[ln][m].re * (1.0 / 1024), y0[ln][m].im * (1.0 / 1024)};
                            dst[512 + m * 64 + ln] = cplx{y1[ln][m].re * (1.0 / 1024), y1[ln][m].im * (1.0 / 1024)};
                        }
                }
}
// one CMux on acc int64[2][2048]; returns the worst distance of an inverse-transform output from an integer
double emu_mk_mux_rotate_2k(const double *spec, int l, int Bgbit, long pi, int barai, int64_t *acc) {
    Wave2K w;
    const cplx *BK = reinterpret_cast<const cplx *>(spec);
    const int rows = 2 * l, a2n = barai & 4095;
    const uint64_t offset = decomp_offset64(l, Bgbit);
    static cplx S0[2][4][64][8], S1[2][4][64][8], z[64][16], y0[64][8], y1[64][8], lo[64][8], hi[64][8];
    static uint32_t t[64][32];
    memset(S0, 0, sizeof(S0));
    memset(S1, 0, sizeof(S1));
    for (int r = 0; r < rows; r++) {
        if (r % l == 0)
            for (int ln = 0; ln < 64; ln++) load_rotated32_hi(ln, acc + (r / l) * 2048, a2n, offset, t[ln]);
        for (int ln = 0; ln < 64; ln++) digits_to_z16(t[ln], (r % l) + 1, Bgbit, z[ln]);
        w.fwd2k(z, y0, y1);
        for (int o = 0; o < 2; o++)
            for (int h = 0; h < 4; h++)
                for (int ln = 0; ln < 64; ln++) {
                    const cplx *B = BK + mk_chunk_index_2k(pi, r, h, o, rows) * 512;
                    mac8(ln, S0[o][h][ln], y0[ln], B);
                    mac8(ln, S1[o][h][ln], y1[ln], B + 512);
                }
    }
    double worst = 0;
    for (int o = 0; o < 2; o++)
        for (int h = 0; h < 4; h++) {
            w.inv2k(S0[o][h], S1[o][h], lo, hi);
            for (int ln = 0; ln < 64; ln++)
                for (int m = 0; m < 8; m++) {
                    const int q = ln + 64 * m;
                    const double v[4] = {lo[ln][m].re, hi[ln][m].re, lo[ln][m].im, hi[ln][m].im};
                    const int idx[4] = {q, q + 512, q + 1024, q + 1536};
                    for (int e = 0; e < 4; e++) {
                        double d = __builtin_fabs(v[e] - __builtin_rint(v[e]));
                        if (d > worst) worst = d;
                        acc[o * 2048 + idx[e]] = (int64_t)((uint64_t)acc[o * 2048 + idx[e]] + ((uint64_t)round_i64(v[e]) << (16 * h)));
                    }
                }
        }
    return worst;
}
void emu_mk_extract_2k(const int64_t *acc, int32_t *out) {
    for (int ln = 0; ln < 64; ln++) extract_64_n<2048>(ln, acc, acc + 2048, out);
}
}

// ---- table-free twisted transforms ("tq" form: N = 2048 two-gate kernels) and the ring of degree 4096 (four twisted quarters) -----------
namespace {
struct WaveTQ {
    cplx T1a[512], T1b[512], T2[64], scratch[512], ratio[64], roots4k[256];
    cplx xbuf[512];
    W64 w[64];
    WaveTQ() {
        make_twiddles_2048(T1a, T1b);
        make_twiddles_1024(scratch, T2);
        make_lane_ratio_2048(ratio);
        make_lane_roots_4096(roots4k);
        for (int l = 0; l < 64; l++) w[l] = W64{T2[1 * 8 + (l & 7)]};
    }
    template <int T, int DEN>
    LaneRoots roots(int l) const {
        if (DEN == 32) return LaneRoots{T == 1 ? T1a[l] : T1b[l], ratio[l]};
        return LaneRoots{roots4k[((T - 1) / 4) * 64 + l], ratio[l]};
    }
    template <int T, int DEN>
    void fwd(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) fwdtq_seg1<T, DEN>(z[l], roots<T, DEN>(l));
        lanes_transpose_hi3(z);
        for (int l = 0; l < 64; l++) fwds_seg2_st(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) fwds_seg3(l, z[l], xbuf);
    }
    template <int T, int DEN>
    void inv(cplx (*z)[8]) {
        for (int l = 0; l < 64; l++) invs_seg1(l, z[l], xbuf, w[l]);
        for (int l = 0; l < 64; l++) {
            invs_seg2_ld(l, z[l], xbuf);
            dft8<-1>(z[l]);
        }
        lanes_transpose_hi3(z);
        for (int l = 0; l < 64; l++) invtq_seg3<T, DEN>(z[l], roots<T, DEN>(l));
    }
};
}  // namespace

extern "C" {
// N = 2048: the table-free form gives the spectra of the table form (same order); returns the largest difference
double emu_tq_vs_table_2k(const double *zin) {
    Wave2K wt;
    WaveTQ wq;
    static cplx z[64][16], y0[64][8], y1[64][8], q0[64][8], q1[64][8];
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 16; m++) z[l][m] = cplx{zin[2 * (l + 64 * m)], zin[2 * (l + 64 * m) + 1]};
    for (int l = 0; l < 64; l++) split2048(z[l], y0[l], y1[l]);
    memcpy(q0, y0, sizeof(y0));
    memcpy(q1, y1, sizeof(y1));
    wt.fwd<1>(y0);
    wt.fwd<5>(y1);
    wq.fwd<1, 32>(q0);
    wq.fwd<5, 32>(q1);
    double worst = 0;
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) {
            worst = __builtin_fmax(worst, __builtin_fabs(y0[l][m].re - q0[l][m].re) + __builtin_fabs(y0[l][m].im - q0[l][m].im));
            worst = __builtin_fmax(worst, __builtin_fabs(y1[l][m].re - q1[l][m].re) + __builtin_fabs(y1[l][m].im - q1[l][m].im));
        }
    // and the inverse pair returns 1024 z
    wq.inv<1, 32>(q0);
    wq.inv<5, 32>(q1);
    static cplx lo[64][8], hi[64][8];
    for (int l = 0; l < 64; l++) merge2048(q0[l], q1[l], lo[l], hi[l]);
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) {
            worst = __builtin_fmax(worst, __builtin_fabs(lo[l][m].re / 1024 - z[l][m].re) + __builtin_fabs(lo[l][m].im / 1024 - z[l][m].im));
            worst = __builtin_fmax(worst, __builtin_fabs(hi[l][m].re / 1024 - z[l][m + 8].re) + __builtin_fabs(hi[l][m].im / 1024 - z[l][m + 8].im));
        }
    return worst;
}
// N = 4096: exact negacyclic product of a small-digit polynomial d (|d| <= 2^8) with a Torus64 polynomial k, through the radix-4 split, the
// four twisted quarter transforms, limb spectra scaled by 1/2048, inverse, merge and rounding -- the arithmetic of r4k_rotate_kernel.
// Returns the worst distance of an inverse output from an integer.
double emu_polymul_4k(const int32_t *d, const int64_t *k, int64_t *out) {
    WaveTQ w;
    static cplx yd[4][64][8], yk[4][4][64][8], S[4][4][64][8];
    auto fwd4 = [&](cplx (*y)[64][8]) {
        w.fwd<1, 64>(y[0]);
        w.fwd<5, 64>(y[1]);
        w.fwd<9, 64>(y[2]);
        w.fwd<13, 64>(y[3]);
    };
    for (int l = 0; l < 64; l++)
        for (int m = 0; m < 8; m++) {
            cplx u[4];
            for (int s = 0; s < 4; s++) u[s] = cplx{(double)d[l + 64 * m + 512 * s], (double)d[l + 64 * m + 512 * s + 2048]};
            pre4096(u);
            yd[0][l][m] = comb4096<0>(u);
            yd[1][l][m] = comb4096<1>(u);
            yd[2][l][m] = comb4096<2>(u);
            yd[3][l][m] = comb4096<3>(u);
        }
    fwd4(yd);
    for (int h = 0; h < 4; h++) {
        for (int l = 0; l < 64; l++)
            for (int m = 0; m < 8; m++) {
                cplx u[4];
                for (int s = 0; s < 4; s++) {
                    double a[4], b[4];
                    split_limbs64(k[l + 64 * m + 512 * s], a);
                    split_limbs64(k[l + 64 * m + 512 * s + 2048], b);
                    u[s] = cplx{a[h], b[h]};
                }
                pre4096(u);
                yk[h][0][l][m] = comb4096<0>(u);
                yk[h][1][l][m] = comb4096<1>(u);
                yk[h][2][l][m] = comb4096<2>(u);
                yk[h][3][l][m] = comb4096<3>(u);
            }
        fwd4(yk[h]);
    }
    double worst = 0;
    for (int q = 0; q < 4096; q++) out[q] = 0;
    for (int h = 0; h < 4; h++) {
        for (int t = 0; t < 4; t++)
            for (int l = 0; l < 64; l++)
                for (int m = 0; m < 8; m++) {
                    const cplx a = yd[t][l][m], b = cplx{yk[h][t][l][m].re * (1.0 / 2048), yk[h][t][l][m].im * (1.0 / 2048)};
                    S[h][t][l][m] = cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
                }
        w.inv<1, 64>(S[h][0]);
        w.inv<5, 64>(S[h][1]);
        w.inv<9, 64>(S[h][2]);
        w.inv<13, 64>(S[h][3]);
        for (int l = 0; l < 64; l++)
            for (int m = 0; m < 8; m++) {
                cplx z[4];
                merge4096(S[h][0][l][m], S[h][1][l][m], S[h][2][l][m], S[h][3][l][m], z);
                for (int s = 0; s < 4; s++) {
                    const int c = l + 64 * m + 512 * s;
                    worst = __builtin_fmax(worst, __builtin_fabs(z[s].re - __builtin_rint(z[s].re)));
                    worst = __builtin_fmax(worst, __builtin_fabs(z[s].im - __builtin_rint(z[s].im)));
                    out[c] = (int64_t)((uint64_t)out[c] + ((uint64_t)round_i64(z[s].re) << (16 * h)));
                    out[c + 2048] = (int64_t)((uint64_t)out[c + 2048] + ((uint64_t)round_i64(z[s].im) << (16 * h)));
                }
            }
    }
    return worst;
}
}

