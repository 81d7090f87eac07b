/*
 * thfhe_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Exact-integer restatement of the gate-bootstrapping hot path of
 * Animesh005/Torus-FHE (reference mounted read-only at /root/reference; paths below are
 * relative to it, J/ = 3-gen-mk-tfhe/src/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (torus-fhe_amd/) never does.
 *
 * Parity pinning: the restatement follows the reference's *exact* multiply twins
 * (tgsw_extern_mul_wo_FFT, J/tgsw.jl:152-156; mux_rotate_wo_FFT, J/bootstrap.jl:25-29), i.e.
 * integer negacyclic convolution with wrap-around mod 2^32 / 2^64, and is pinned by the
 * reference's committed ciphertext fixtures test/bootstrap_modules/{cloud1..4,sum,carry,diff,...}.data (record format,
 * bit order, +-1/8 encoding, adder/subtractor wiring, noise envelope) -- see
 * tests/test_oracle_fixtures.py.  Ciphertext-level equality with the reference's own
 * bootstrapped outputs (sum.data...) is NOT achievable: they were produced with the authors'
 * (unrecoverable) bootstrapping key and libtfhe's approximate double FFT.  For the multi-key
 * (3-gen) path the reference holds no ciphertext fixtures and Julia is not installed here:
 * MK ciphertext parity vs the reference is UNPINNED (decrypt-equality + noise only).
 *
 * Layouts (all little-endian, row-major, innermost index last):
 *   LWE record (single key)      int32[n+1]          = a[0..n), b
 *   extracted LWE record         int32[kN+1]         = a[0..kN), b
 *   MK LWE record (P parties)    int32[P*n+1]        = a[p*n+i] (column p of J/mk_internals.jl:23-37), b
 *   BK  (single key, coeff dom.) int32[n][(k+1)l][k+1][N]   row r = j*l + p  (block j, level p)
 *                                 = TGSW_z(s_i) rows of J/tgsw.jl:65-101 (samples[p,j]), poly c of row
 *   KSK (single key)             int32[kN][t][base-1][n+1]  entry (i,j,h-1) = ks[h,j,i] of J/keyswitch.jl:35-38
 *   BK  (3-gen MK, coeff dom.)   int64[P][n][4][l][N]       part q (1..4 -> 0..3), level p   (J/tgsw_3gen.jl:3-20)
 *   KSK (3-gen MK)               int32[P][N][t][base-1][n+1]
 */
#ifndef THFHE_ORACLE_H
#define THFHE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t n;          /* LWE dimension                         (J/api.jl:4-21 lwe_size)            */
    int32_t N;          /* ring degree                           (rlwe_polynomial_degree)            */
    int32_t k;          /* RLWE mask size (1 everywhere in the reference)                            */
    int32_t l;          /* gadget decomposition length           (bs_decomp_length)                  */
    int32_t Bgbit;      /* log2 of gadget base                   (bs_log2_base)                      */
    int32_t ks_t;       /* key-switch decomposition length       (ks_decomp_length)                  */
    int32_t ks_basebit; /* key-switch log2 base                  (ks_log2_base)                      */
    int32_t torus_bits; /* 32: Torus32 ring (single key); 64: Torus64 ring (3-gen MK, rlwe_is32=false) */
    int32_t parties;    /* 1 for single key                                                          */
} oracle_params;

/* gate opcodes (shared numbering with include/thfhe_hip.h) */
enum {
    OR_GATE_NAND = 0, OR_GATE_OR = 1, OR_GATE_AND = 2, OR_GATE_XOR = 3, OR_GATE_XNOR = 4,
    OR_GATE_NOR = 5, OR_GATE_ANDNY = 6, OR_GATE_ANDYN = 7, OR_GATE_ORNY = 8, OR_GATE_ORYN = 9,
    OR_GATE_MUX = 10, OR_GATE_NOT = 11, OR_GATE_COPY = 12, OR_GATE_AND3 = 13
};

/* ---- scalar / polynomial primitives (unit-tested one by one) -------------------------------- */
int32_t oracle_modswitch(int32_t x, int32_t N);                                   /* J/numeric-functions.jl:70-73 */
void oracle_mul_by_monomial32(const int32_t *p, int32_t shift, int32_t N, int32_t *out);   /* J/rlwe.jl:130-131 */
void oracle_mul_by_monomial64(const int64_t *p, int32_t shift, int32_t N, int64_t *out);
void oracle_decompose32(const int32_t *p, int32_t N, int32_t l, int32_t Bgbit, int32_t *digits /*[l][N]*/); /* J/tgsw.jl:112-138 */
void oracle_decompose64(const int64_t *p, int32_t N, int32_t l, int32_t Bgbit, int64_t *digits /*[l][N]*/);
void oracle_polymul_schoolbook32(const int32_t *a, const int32_t *b, int32_t N, int32_t *out);  /* exact, mod X^N+1, mod 2^32 */
void oracle_polymul_schoolbook64(const int64_t *a, const int64_t *b, int32_t N, int64_t *out);
void oracle_polymul_ntt32(const int32_t *small, const int32_t *b, int32_t N, int32_t *out);     /* exact via Goldilocks NTT; |small| < 2^15 */
void oracle_polymul_ntt64(const int64_t *small, const int64_t *b, int32_t N, int64_t *out);
int32_t oracle_t64tot32(int64_t d);                                               /* J/numeric-functions.jl:109-111 */

/* ---- single-key context ------------------------------------------------------------------- */
typedef struct oracle_ctx oracle_ctx;
oracle_ctx *oracle_ctx_create(const oracle_params *p, const int32_t *bk, const int32_t *ksk);
void oracle_ctx_destroy(oracle_ctx *c);

/* acc (k+1 polys of N) += BK_i (.) (X^barai * acc - acc)           J/bootstrap.jl:19-23
 * use_schoolbook = 1 -> O(N^2) reference multiply, 0 -> NTT multiply (bit-identical); single-key functions also take 2 -> the reference's
 * own APPROXIMATE Complex{Float64} transform (J/polynomials.jl:208-247): same decryptions, low bits differ by FFT rounding noise. */
void oracle_fft_polymul32(const int32_t *x, const int32_t *y, int32_t N, int32_t *out); /* transformed_mul, J/polynomials.jl:245-247: the reference's approximate Complex{Float64} product */
void oracle_mux_rotate(const oracle_ctx *c, int32_t i, int32_t barai, int32_t *acc, int use_schoolbook);
/* x: LWE(n) record -> out: LWE(kN) record, mu = output message     J/bootstrap.jl:75-88 */
void oracle_bootstrap_wo_keyswitch(const oracle_ctx *c, int32_t mu, const int32_t *x, int32_t *out, int use_schoolbook);
/* in: LWE(kN) record -> out: LWE(n) record                          J/keyswitch.jl:45-80 */
void oracle_keyswitch(const oracle_ctx *c, const int32_t *in, int32_t *out);
/* full gates on `count` independent records (OpenMP over gates)     J/gates.jl:15-177
 * in2 only for MUX; NOT/COPY ignore bk.  returns 0 or -1 on bad op. */
int oracle_gates(const oracle_ctx *c, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                 int32_t *out, size_t count, int use_schoolbook);
/* the linear prologue only (temp = const +- x +- y), for unit tests   J/gates.jl */
int oracle_gate_prologue(const oracle_params *p, int op, int which, const int32_t *in0, const int32_t *in1,
                         const int32_t *in2, int32_t *tmp);

/* ---- 3-gen multi-key context -------------------------------------------------------------- */
typedef struct oracle_mk_ctx oracle_mk_ctx;
oracle_mk_ctx *oracle_mk_ctx_create(const oracle_params *p, const int64_t *bk, const int32_t *ksk);
void oracle_mk_ctx_destroy(oracle_mk_ctx *c);
void oracle_mk_mux_rotate(const oracle_mk_ctx *c, int32_t party, int32_t i, int32_t barai, int64_t *acc, int use_schoolbook); /* J/3gen_mk_internals.jl:59-62 */
void oracle_mk_bootstrap_wo_keyswitch(const oracle_mk_ctx *c, int64_t mu, const int32_t *x, int32_t *out, int use_schoolbook); /* :99-109 */
void oracle_mk_keyswitch(const oracle_mk_ctx *c, const int32_t *in, int32_t *out);                                               /* J/mk_internals.jl:730-744 */
int oracle_mk_gates(const oracle_mk_ctx *c, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                    int32_t *out, size_t count, int use_schoolbook);                                                              /* J/3gen_mk_gates.jl:8-150 */

/* ---- key generation / encryption / decryption (host-side, for tests only) ----------------- */
/* deterministic xoshiro256** streams; sigma in torus units (fraction of 1) */
void oracle_keygen_sk(const oracle_params *p, uint64_t seed, double sigma_bk, double sigma_ks,
                      const int32_t *lwe_key_in /* NULL -> sample uniform binary */,
                      int32_t *lwe_key /*[n]*/, int32_t *rlwe_key /*[k][N]*/, int32_t *bk, int32_t *ksk);
void oracle_keygen_mk(const oracle_params *p, uint64_t seed, double sigma_bk, double sigma_ks,
                      int32_t *lwe_keys /*[P][n]*/, int64_t *rlwe_keys /*[P][N]*/, int64_t *bk, int32_t *ksk);
void oracle_lwe_encrypt(const int32_t *key, int32_t n, int32_t mu, double sigma, uint64_t seed, uint64_t idx, int32_t *rec);
int32_t oracle_lwe_phase(const int32_t *key, int32_t n, const int32_t *rec);
void oracle_mk_lwe_encrypt(const int32_t *keys /*[P][n]*/, int32_t n, int32_t P, int32_t mu, double sigma,
                           uint64_t seed, uint64_t idx, int32_t *rec);
int32_t oracle_mk_lwe_phase(const int32_t *keys, int32_t n, int32_t P, const int32_t *rec);

/* ---- CCS multi-key scheme: mk_bootstrap / mk_gate_nand (J/mk_internals.jl:477-536,714-728,805-858; J/mk_gates.jl:7-13) ----
 * bk int32[P][n][3][l][N] = d1, f0, f1 of every MKTGswUESample; pk int32[P][l][N]; crs int32[l][N]; ksk int32[P][N][t][base-1][n+1];
 * accumulator int32[P+1][N] = (a_0 .. a_{P-1}, b); extracted sample int32[P*N+1] */
typedef struct oracle_ccs_ctx oracle_ccs_ctx;
oracle_ccs_ctx *oracle_ccs_ctx_create(const oracle_params *p, const int32_t *bk, const int32_t *pk, const int32_t *crs, const int32_t *ksk);
void oracle_ccs_ctx_destroy(oracle_ccs_ctx *c);
void oracle_ccs_uniproduct(const oracle_ccs_ctx *c, int32_t party, int32_t j, const int32_t *acc, int32_t *out, int use_schoolbook);
void oracle_ccs_mux_rotate(const oracle_ccs_ctx *c, int32_t party, int32_t j, int32_t barai, int32_t *acc, int use_schoolbook);
void oracle_ccs_bootstrap_wo_keyswitch(const oracle_ccs_ctx *c, int32_t mu, const int32_t *x, int32_t *out, int use_schoolbook);
void oracle_ccs_keyswitch(const oracle_ccs_ctx *c, const int32_t *in, int32_t *out);
int oracle_ccs_gates(const oracle_ccs_ctx *c, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count, int use_schoolbook);
void oracle_keygen_ccs(const oracle_params *p, uint64_t seed, double sigma_bk, double sigma_ks, int32_t *lwe_keys, int32_t *rlwe_keys,
                       int32_t *bk, int32_t *pk, int32_t *crs, int32_t *ksk);

/* ---- LWE -> TLWE conversion, threshold partial / final decryption (k = 1)   src/libthfhe.cpp:270-348 ---- */
void oracle_tlwe_from_lwe(const int32_t *lwe /*[N+1]*/, int32_t N, int32_t *tlwe_a /*[N]*/, int32_t *tlwe_b /*[N]*/);
void oracle_partial_decrypt(const int32_t *key_share, const int32_t *tlwe_a, const int32_t *noise, int32_t N, int32_t *partial);
int32_t oracle_final_decrypt(const int32_t *tlwe_b, const int32_t *partials /*[t][N]*/, int32_t t, int32_t N, int32_t *result);

/* ---- KMS multi-key scheme (mk_bootstrap_new / mk_gate_nand_new)   3-gen-mk-tfhe/src/new_mk_internals.jl, tlev.jl ---- */
typedef struct {
    int32_t n, N, parties;
    int32_t l_gsw, bg_gsw; /* per-party TGSW blind rotation of the TLev accumulator */
    int32_t l_lev, bg_lev; /* the TLev accumulator */
    int32_t l_uni, bg_uni; /* uni-encryption / public keys / shared key */
    int32_t ks_t, ks_basebit;
} oracle_kms_params;
typedef struct oracle_kms_ctx oracle_kms_ctx;
oracle_kms_ctx *oracle_kms_ctx_create(const oracle_kms_params *p, const int64_t *gsw, const int64_t *uni, const int64_t *pk, const int64_t *crs,
                                      const int32_t *ksk);
void oracle_kms_ctx_destroy(oracle_kms_ctx *c);
void oracle_kms_rlwe_rotate(const oracle_kms_ctx *c, int32_t party, const int32_t *bara, int64_t *acc /* [2][N] in/out */, int use_schoolbook);
void oracle_kms_tlev_rotate(const oracle_kms_ctx *c, int32_t party, const int32_t *bara, int64_t *lev, int use_schoolbook);
void oracle_kms_uniproduct(const oracle_kms_ctx *c, int32_t party, const int64_t *e, int64_t *out, int use_schoolbook);
void oracle_kms_lev_rlwe_mul(const oracle_kms_ctx *c, int32_t party, int64_t *accum, const int64_t *lev, int use_schoolbook);
void oracle_kms_bootstrap_wo_keyswitch(const oracle_kms_ctx *c, int64_t mu, const int32_t *x, int32_t *out, int use_schoolbook);
void oracle_kms_bootstrap_wo_keyswitch_ex(const oracle_kms_ctx *c, int64_t mu, const int32_t *x, int32_t *out, int use_schoolbook, int fast_boot);
void oracle_kms_keyswitch(const oracle_kms_ctx *c, const int32_t *in, int32_t *out);
int oracle_kms_gates(const oracle_kms_ctx *c, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count, int use_schoolbook);
int oracle_kms_gates_ex(const oracle_kms_ctx *c, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count, int use_schoolbook,
                        int fast_boot);   /* fast_boot: mk_blind_rotate_new_v2, J/new_mk_internals.jl:255-269 */

int oracle_max_threads(void);
void oracle_set_threads(int n); /* OpenMP team size for the batch entry points */

#ifdef __cplusplus
}
#endif
#endif
