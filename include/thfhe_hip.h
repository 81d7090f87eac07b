/*
 * thfhe_hip.h -- C ABI of libthfhe_hip.so, the MI355X-native gate-bootstrapping engine.
 *
 * Drop-in boundary for ONE hot path of Animesh005/Torus-FHE: bootstrapped gate evaluation
 * (blind rotate = n x CMux, sample extraction, key switching).  Reference interfaces replaced
 * (paths relative to the reference tree, J/ = 3-gen-mk-tfhe/src/):
 *
 *   thfhe_gates                      <- gate_nand/or/and/xor/xnor/nor/andny/andyn/orny/oryn/mux/not  J/gates.jl:15-177
 *                                       == libtfhe's extern "C" bootsNAND/AND/OR/XOR/.../MUX/NOT that the C++ side
 *                                       calls (src/KNN_medical_data.cpp:130,142-151,227,388-396; src/Convert.cpp:31)
 *   thfhe_bootstrap                  <- bootstrap(bk, ks, mu, x)                    J/bootstrap.jl:98-101
 *   thfhe_bootstrap_wo_keyswitch     <- bootstrap_wo_keyswitch(bk, mu, x)           J/bootstrap.jl:75-88
 *   thfhe_keyswitch                  <- keyswitch(ks, sample)                       J/keyswitch.jl:45-80
 *   thfhe_ctx_create                 <- BootstrapKey(...) forward_transform step    J/bootstrap.jl:6-15 (key -> transformed key)
 *                                       + KeyswitchKey table upload                 J/keyswitch.jl:7-42
 *   thfhe_mk_*                       <- mk_bootstrap_3gen / mk_gate_*_3gen          J/3gen_mk_internals.jl:99-116, J/3gen_mk_gates.jl:8-150
 *   bootsNAND ... (tfhe_shim.h)      <- the libtfhe symbols themselves (struct-compatible shims)
 *
 * All entry points are plain C: pointers + sizes, no C++/torch types.  Return value: 0 on success,
 * negative THFHE_E_* on failure (thfhe_last_error() gives a message).  There is NO CPU fallback: if
 * no HIP device is usable every compute call fails with THFHE_E_NO_DEVICE.
 *
 * Data layouts (little-endian, row-major, innermost last):
 *   LWE record                  int32[n+1]   = a[0..n), b                       (LweSample, J/lwe.jl:21-29)
 *   extracted LWE record        int32[N+1]
 *   bk_coeff  (single key)      int32[n][(k+1)l][k+1][N], row r = j*l + p (block j, level p) -- libtfhe's
 *                               TGswSample.all_sample order; coefficient domain (Torus32)
 *   ksk       (single key)      int32[N][t][base-1][n+1], entry (i, j, h-1) = KS[h, j, i]    (J/keyswitch.jl:35-38)
 *   MK record (P parties)       int32[P*n+1] = a[p*n + i], b                     (MKLweSample, J/mk_internals.jl:23-37)
 *   mk bk_coeff                 int64[P][n][4][l][N]  (part_1..part_4, level)    (TGswSample_3gen, J/tgsw_3gen.jl:3-20)
 *   mk ksk                      int32[P][N][t][base-1][n+1]
 */
#ifndef THFHE_HIP_H
#define THFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct thfhe_params {
    int32_t n;          /* LWE dimension                (lwe_size, J/api.jl:4-21)                     */
    int32_t N;          /* ring degree                  (rlwe_polynomial_degree); 1024 supported      */
    int32_t k;          /* RLWE mask size; 1 supported                                                 */
    int32_t l;          /* gadget decomposition length  (bs_decomp_length); 1..4                       */
    int32_t Bgbit;      /* log2 gadget base             (bs_log2_base); l*Bgbit <= 32, Bgbit <= 10     */
    int32_t ks_t;       /* key-switch length            (ks_decomp_length)                             */
    int32_t ks_basebit; /* key-switch log2 base         (ks_log2_base)                                 */
    int32_t torus_bits; /* 32 = Torus32 ring (single key), 64 = Torus64 ring (3-gen multi-key)         */
    int32_t parties;    /* 1 = single key                                                              */
} thfhe_params;

/* gate opcodes (J/gates.jl; J/3gen_mk_gates.jl for AND3) */
enum thfhe_gate {
    THFHE_NAND = 0, THFHE_OR = 1, THFHE_AND = 2, THFHE_XOR = 3, THFHE_XNOR = 4, THFHE_NOR = 5,
    THFHE_ANDNY = 6, THFHE_ANDYN = 7, THFHE_ORNY = 8, THFHE_ORYN = 9, THFHE_MUX = 10,
    THFHE_NOT = 11, THFHE_COPY = 12, THFHE_AND3 = 13
};

enum thfhe_error {
    THFHE_OK = 0, THFHE_E_INVALID = -1, THFHE_E_UNSUPPORTED = -2, THFHE_E_NO_DEVICE = -3,
    THFHE_E_HIP = -4, THFHE_E_NOMEM = -5
};

typedef struct thfhe_ctx thfhe_ctx;

const char *thfhe_last_error(void);
int thfhe_device_count(void);
/* PCI bus id ("0000:c1:00.0") of HIP device `device` as this process sees it: bench.py prints it per rank so that a multi-GPU record
 * shows N ranks on N different devices. */
int thfhe_device_pci_bus_id(int device, char *buf, int len);

/* Upload + transform the keys to device `device`.  bk_coeff / ksk are HOST pointers, borrowed only
 * for the duration of the call. */
int thfhe_ctx_create(const thfhe_params *params, const int32_t *bk_coeff, const int32_t *ksk, int device,
                     thfhe_ctx **out);
void thfhe_ctx_destroy(thfhe_ctx *ctx);
int thfhe_ctx_params(const thfhe_ctx *ctx, thfhe_params *out);

/* ---- host-buffer API: the drop-in level.  in0/in1/in2/out are HOST arrays of `count` records.  `out` may alias
 * an input (the reference's callers do, src/KNN_medical_data.cpp:256,395).  Thread-safe per ctx. */
int thfhe_gates(thfhe_ctx *ctx, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                int32_t *out, size_t count);
/* One launch for a level of a gate DAG: gate g applies ops[g] (any two-input bootstrapped gate NAND..ORYN) to
 * (in0[g], in1[g]).  ops is a HOST array of `count` opcodes. */
int thfhe_gates_mixed(thfhe_ctx *ctx, const int32_t *ops, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count);
/* Gate-DAG evaluation: the levelising scheduler + device-resident executor for the reference's circuits (FullAdder / difference /
 * distance / sort_with_distance ..., src/KNN_medical_data.cpp:127-489, issued there as sequential boots* calls).
 *   wires  HOST table int32[n_inputs + n_gates][n+1]: rows [0, n_inputs) hold the input ciphertexts, row n_inputs + g receives gate g
 *   gates  HOST int32[n_gates][4] = (opcode, in0, in1, in2), topological order, operands are earlier wire ids (unused = -1);
 *          opcodes: the two-input bootstrapped gates, THFHE_MUX, THFHE_NOT, THFHE_COPY
 * Gates are scheduled ASAP into levels; each level is ONE blind-rotate launch per gate class (two-input with per-gate opcodes, MUX);
 * the wire table stays in HBM and the host is not synchronised between levels.
 * stats (optional) int64[4] = {levels, bootstrap launches, blind rotations, widest level}. */
int thfhe_dag_run(thfhe_ctx *ctx, int32_t *wires, size_t n_inputs, const int32_t *gates, size_t n_gates, int64_t *stats);
/* `instances` independent evaluations of ONE gate list, level by level: a level's launch holds instances x its gates.  This is the
 * reference's loop over test records around one circuit (`for i < test_row_size`, src/KNN_medical_data.cpp:676-691): the deep, narrow
 * part of a decision (a ripple carry holds 1-3 gates per level) fills the chip only when many records walk it side by side.
 *   inputs     HOST int32[instances][n_inputs][n+1]
 *   out_wires  HOST wire ids to return (n_out of them); NULL: every gate wire, i.e. n_inputs .. n_inputs + n_gates - 1
 *   outputs    HOST int32[instances][n_out (or n_gates)][n+1]
 * Gate outputs are deterministic functions of their operands, so instance q's wires equal thfhe_dag_run on inputs[q] bit for bit. */
int thfhe_dag_run_batch(thfhe_ctx *ctx, const int32_t *inputs, size_t n_inputs, const int32_t *gates, size_t n_gates, size_t instances,
                        const int32_t *out_wires, size_t n_out, int32_t *outputs, int64_t *stats);
/* A level whose instances x gates exceed `max_gates` runs as several launches of at most that many gates (default 28 672 = 14 rounds of the
 * throughput kernel; bounds the staging memory and the launch grid).  1 .. 32 767. */
int thfhe_set_dag_slice(thfhe_ctx *ctx, size_t max_gates);
int thfhe_bootstrap(thfhe_ctx *ctx, int32_t mu, const int32_t *x, int32_t *out, size_t count);
int thfhe_bootstrap_wo_keyswitch(thfhe_ctx *ctx, int32_t mu, const int32_t *x, int32_t *out_N1, size_t count);
int thfhe_keyswitch(thfhe_ctx *ctx, const int32_t *in_N1, int32_t *out, size_t count);

/* ---- device-buffer API: records already resident in HBM (what bench.py times).  Pointers come from
 * thfhe_dev_alloc (or any hipMalloc in this process).  Calls enqueue on the context's stream and
 * return; thfhe_sync waits. */
void *thfhe_dev_alloc(thfhe_ctx *ctx, size_t bytes);
void thfhe_dev_free(thfhe_ctx *ctx, void *p);
int thfhe_copy_h2d(thfhe_ctx *ctx, void *dst, const void *src, size_t bytes);
int thfhe_copy_d2h(thfhe_ctx *ctx, void *dst, const void *src, size_t bytes);
int thfhe_reserve(thfhe_ctx *ctx, size_t max_count); /* pre-size the workspace (no allocation afterwards) */
int thfhe_gates_dev(thfhe_ctx *ctx, int op, const int32_t *d_in0, const int32_t *d_in1, const int32_t *d_in2,
                    int32_t *d_out, size_t count);
int thfhe_sync(thfhe_ctx *ctx);

/* Kernel choice for a batch of rotations (gates; a MUX is two).  Whole rounds of 2 048 rotations (eight per CU of an MI355X) run on the
 * LDS-ring throughput kernel, eight gates per workgroup.  The remainder r runs on the cooperative latency kernel (one workgroup per gate) if
 * r <= the cooperative threshold (default 768), on the four-wave shape of the ring kernel (four gates per workgroup, one wave per SIMD) if
 * r <= the ring4 threshold (default 1 024), on both if r <= ring4 + 256, else on one more eight-wave round.  Both thresholds 0: everything
 * on the eight-wave kernel.  Every shape computes the same words. */
int thfhe_set_coop_threshold(thfhe_ctx *ctx, int max_jobs);
int thfhe_set_ring4_threshold(thfhe_ctx *ctx, int max_jobs);

/* Per-kernel device timing: when enabled, every *_dev call brackets each kernel with HIP events on the
 * context's stream.  After thfhe_sync, thfhe_last_timings returns milliseconds of the most recent call:
 * ms[0] = prologue (linear part + mod-switch), ms[1] = blind rotate, ms[2] = key switch, ms[3] = total. */
int thfhe_set_profiling(thfhe_ctx *ctx, int enabled);
int thfhe_last_timings(thfhe_ctx *ctx, float ms[4]);

/* ---- 3-gen multi-key (Torus64 ring) --------------------------------------------------------------- */
typedef struct thfhe_mk_ctx thfhe_mk_ctx;
/* Ring degrees: N = 1024 (l <= 4, Bgbit <= 10), N = 2048 (l <= 3; bases of 11 .. 27 bit are cut into balanced 9-bit digit parts: the 16 .. 256-party
 * sets of J/mk_api.jl:214-310), N = 4096 (at most six digit rows: the 64-party "for fft" and the 512-party set, J/mk_api.jl:277-283, 316-322).
 * bk_coeff int64[P][n][4][l][N] (part_1 .. part_4 of MKBootstrapKeyPart_3gen), ksk int32[P][N][t][base-1][n+1]. */
int thfhe_mk_ctx_create(const thfhe_params *params, const int64_t *bk_coeff, const int32_t *ksk, int device,
                        thfhe_mk_ctx **out);
void thfhe_mk_ctx_destroy(thfhe_mk_ctx *ctx);
int thfhe_mk_gates(thfhe_mk_ctx *ctx, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                   int32_t *out, size_t count);
/* one launch for a DAG level of two-input 3-gen gates (NAND / OR / AND / XOR), per-gate opcodes in the HOST array ops */
int thfhe_mk_gates_mixed(thfhe_mk_ctx *ctx, const int32_t *ops, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count);
/* Batches of at most `max_single_jobs` rotations run one gate per workgroup (latency), larger ones two gates per workgroup sharing
 * every key chunk (throughput; l <= 3 on the ring of degree 1024, every set on the ring of degree 2048).  Default 256 = one workgroup per CU of an MI355X. */
int thfhe_mk_set_pair_threshold(thfhe_mk_ctx *ctx, long max_single_jobs);
/* Gate-DAG evaluation for the 3-gen scheme (same contract as thfhe_dag_run; records of P*n+1 words): the reference's multi-key integer
 * circuits mk_add_3gen ... mk_int_mul_3gen (J/3gen_mk_gates.jl:183-362).  Opcodes: NAND / OR / AND / XOR, AND3, MUX, NOT, COPY. */
int thfhe_mk_dag_run(thfhe_mk_ctx *ctx, int32_t *wires, size_t n_inputs, const int32_t *gates, size_t n_gates, int64_t *stats);
/* `instances` evaluations of one 3-gen gate list side by side (same contract as thfhe_dag_run_batch; records of P*n+1 words) */
int thfhe_mk_dag_run_batch(thfhe_mk_ctx *ctx, const int32_t *inputs, size_t n_inputs, const int32_t *gates, size_t n_gates, size_t instances,
                           const int32_t *out_wires, size_t n_out, int32_t *outputs, int64_t *stats);
int thfhe_mk_set_dag_slice(thfhe_mk_ctx *ctx, size_t max_gates); /* default 8 192 */
int thfhe_mk_bootstrap(thfhe_mk_ctx *ctx, int64_t mu, const int32_t *x, int32_t *out, size_t count);
/* Party-sharded building blocks (optional mode: a rank holds only the keys of a contiguous block of m
 * parties, i.e. a context created with parties = m from those parties' key parts; m = 1 is one rank per party).  Below
 * nb = m * n (the block's mask words) and P = the key set's total party count.  All pointers are DEVICE pointers; calls enqueue
 * on the context's stream.  The accumulator travels between ranks as int64[count][2][N] (mask polynomial, body polynomial).
 *   prologue       : the gate's linear part (J/3gen_mk_gates.jl; op = -1: identity, i.e. plain mk_bootstrap_3gen of in0; which = 0 / 1
 *                    selects the first / second AND of the 3-gen MUX) + mod-switch (J/numeric-functions.jl:70-73) of this block's nb
 *                    mask words [first_word, first_word + nb) of records with rec_words = P*n + 1 words, and of b.
 *   rotate_partial : run this context's nb CMuxes (J/3gen_mk_internals.jl:66-84, party-major) on every accumulator.  d_bara =
 *                    int32[count][nb] mod-switched mask words of this block; d_acc_in == NULL starts from X^{-barb} * mu (first block).
 *   extract        : rlwe_extract_sample_64 (J/rlwe.jl:70-74) -> int32[count][N+1]
 *   keyswitch      : keyswitch of the extracted samples with this context's key(s) -> int32[count][nb+1]      */
int thfhe_mk_prologue_dev(thfhe_mk_ctx *ctx, int op, int which, const int32_t *d_in0, const int32_t *d_in1, const int32_t *d_in2,
                          int rec_words, int first_word, int32_t *d_bara, int32_t *d_barb, size_t count);
int thfhe_mk_rotate_partial_dev(thfhe_mk_ctx *ctx, const int32_t *d_bara, const int32_t *d_barb, int64_t mu,
                                const int64_t *d_acc_in, int64_t *d_acc_out, size_t count);
int thfhe_mk_extract_dev(thfhe_mk_ctx *ctx, const int64_t *d_acc, int32_t *d_u, size_t count);
int thfhe_mk_keyswitch_dev(thfhe_mk_ctx *ctx, const int32_t *d_u, int32_t *d_out, size_t count);
/* Enqueue every later call on the caller's HIP stream (e.g. the stream the caller's RCCL communicator synchronises with);
 * NULL returns to the context's own stream.  Waits for work already enqueued. */
int thfhe_mk_set_stream(thfhe_mk_ctx *ctx, void *hip_stream);
void *thfhe_mk_dev_alloc(thfhe_mk_ctx *ctx, size_t bytes);
void thfhe_mk_dev_free(thfhe_mk_ctx *ctx, void *p);
int thfhe_mk_copy_h2d(thfhe_mk_ctx *ctx, void *dst, const void *src, size_t bytes);
int thfhe_mk_copy_d2h(thfhe_mk_ctx *ctx, void *dst, const void *src, size_t bytes);
int thfhe_mk_reserve(thfhe_mk_ctx *ctx, size_t max_count);
int thfhe_mk_gates_dev(thfhe_mk_ctx *ctx, int op, const int32_t *d_in0, const int32_t *d_in1,
                       const int32_t *d_in2, int32_t *d_out, size_t count);
int thfhe_mk_sync(thfhe_mk_ctx *ctx);
int thfhe_mk_set_profiling(thfhe_mk_ctx *ctx, int enabled);
int thfhe_mk_last_timings(thfhe_mk_ctx *ctx, float ms[4]);

/* ---- CCS multi-key scheme: the reference's `mk_bootstrap` / `mk_gate_nand` ---------------------------------
 *   thfhe_ccs_gates      <- mk_gate_nand(ck, x, y)            J/mk_gates.jl:7-13 (AND / OR / XOR share the bootstrap with their own linear part)
 *   thfhe_ccs_bootstrap  <- mk_bootstrap(bk, ks, mu, x)       J/mk_internals.jl:855-858 (UniProduct_old :477-536, mk_keyswitch :714-728)
 *   thfhe_ccs_ctx_create <- MKBootstrapKey(parts, shared_key) J/mk_internals.jl:778-802 (forward_transform of every key polynomial)
 * Torus32, N = 1024, k = 1.  HOST tables, coefficient domain:
 *   bk  int32[P][n][3][l][N]   d1, f0, f1 of every MKTGswUESample (J/mk_internals.jl:338-448)
 *   pk  int32[P][l][N]         PublicKey.b;    crs int32[l][N]  SharedKey.a;    ksk int32[P][N][t][base-1][n+1]
 * Records: int32[P*n+1] = a[p*n + i], b (MKLweSample). */
typedef struct thfhe_ccs_ctx thfhe_ccs_ctx;
int thfhe_ccs_ctx_create(const thfhe_params *params, const int32_t *bk, const int32_t *pk, const int32_t *crs, const int32_t *ksk, int device,
                         thfhe_ccs_ctx **out);
void thfhe_ccs_ctx_destroy(thfhe_ccs_ctx *ctx);
int thfhe_ccs_gates(thfhe_ccs_ctx *ctx, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count);
int thfhe_ccs_bootstrap(thfhe_ccs_ctx *ctx, int32_t mu, const int32_t *x, int32_t *out, size_t count);

/* ---- LWE -> TLWE conversion and threshold partial / final decryption: the step after the gate path in the reference's C++
 * applications.  k = 1, N = 1024; all pointers are HOST arrays.
 *   thfhe_tlwe_from_lwe     <- TLweFromLwe(ring_cipher, cipher, tlwe_params)       src/libthfhe.cpp:340-348, src/KNN_medical_data.cpp:492-500
 *                              lwe int32[count][N+1] -> tlwe_a int32[count][N] (a'[0] = a[0], a'[i] = -a[N-i]), tlwe_b int32[count][N] (b'[0] = b)
 *   thfhe_partial_decrypt   <- ThFHEKeyShare::PartialDecrypt / partialDecrypt       src/libthfhe.cpp:270-293, src/threshold_decryption_functions.cpp:441-480
 *                              partial[c] = key_share (*) tlwe_a[c] + noise[c]; (*) exact negacyclic product mod 2^32 (torusPolynomialAddMulR);
 *                              key_share int32[N] with |s| <= 512; noise (the caller's smudging Gaussian) may be NULL
 *   thfhe_final_decrypt     <- finalDecrypt                                          src/libthfhe.cpp:296-315
 *                              result[c] = tlwe_b[c] - partials[0][c] + sum_{i>=1} partials[i][c]; bits[c] = result[c][0] > 0; result may be NULL */
typedef struct thfhe_poly_ctx thfhe_poly_ctx;
int thfhe_poly_ctx_create(int device, int N, thfhe_poly_ctx **out);
void thfhe_poly_ctx_destroy(thfhe_poly_ctx *ctx);
int thfhe_tlwe_from_lwe(thfhe_poly_ctx *ctx, const int32_t *lwe, int32_t *tlwe_a, int32_t *tlwe_b, size_t count);
int thfhe_partial_decrypt(thfhe_poly_ctx *ctx, const int32_t *key_share, const int32_t *tlwe_a, const int32_t *noise, int32_t *partial, size_t count);
int thfhe_final_decrypt(thfhe_poly_ctx *ctx, const int32_t *tlwe_b, const int32_t *partials /*[t][count][N]*/, int t, int32_t *result, int32_t *bits,
                        size_t count);

/* ---- multi-key KEY GENERATION arithmetic on the device --------------------------------------------------------------
 * Exact multiply-accumulate of small-coefficient polynomials with torus polynomials, the only non-trivial arithmetic of
 *   tgsw_encrypt_3gen                3-gen-mk-tfhe/src/tgsw_3gen.jl:41-95     part_1..4 = r1 (*) B, r2 (*) B, r2 (*) A, r1 (*) A  (+ m g + e)
 *   PublicKey / CommonPubKey_3gen    3-gen-mk-tfhe/src/mk_internals.jl:266-345 b = z (*) a + e ; B = sum of the parties' b
 *   mk_tgsw_encrypt (CCS)            3-gen-mk-tfhe/src/mk_internals.jl:390-446 d1 = r (*) a + m g + e ; f0 = s (*) f1 + r g + e
 * The randomness (keys, masks, Gaussian noise) is an INPUT, so the device result equals the host key generation bit for bit.
 *   out[j] = addend[j] + sum over the terms (j, s, t, sign) of sign * small[s] (*) torus[t]     mod X^N + 1, mod 2^torus_bits
 *   small  int32[n_small][N], |coefficient| <= 4096;  torus / addend / out  int32 or int64 [.][N] by torus_bits;
 *   terms  int32[n_terms][4] = (out, small, torus, +1 | -1), outputs in ascending order; addend may be NULL.
 * N = 1024 (torus_bits 32 or 64) and N = 2048 (torus_bits 64). */
typedef struct thfhe_pm_ctx thfhe_pm_ctx;
int thfhe_pm_ctx_create(int device, int N, int torus_bits, thfhe_pm_ctx **out);
void thfhe_pm_ctx_destroy(thfhe_pm_ctx *ctx);
int thfhe_pm_mac(thfhe_pm_ctx *ctx, const int32_t *small, size_t n_small, const void *torus, size_t n_torus, const int32_t *terms, size_t n_terms,
                 const void *addend, void *out, size_t n_out);

/* ---- KMS multi-key scheme: mk_bootstrap_new / mk_gate_nand_new ----------------------------------------------
 * reference: 3-gen-mk-tfhe/src/new_mk_internals.jl (mk_ith_blind_rotate :210-225, mk_lev_rlwe_mul :185-207, UniProduct_new :85-127,
 * mk_bootstrap_new :321-325), tlev.jl, new_mk_gates.jl:1-7, parameter sets mk_api.jl:12-30,64-82,120-138 (ring degree 2048, Torus64).
 * The context holds the parties' TGSW bootstrapping keys (spectral, device) and key-switch keys:
 *   gsw  int64[P][n][2 l_gsw][2][N]  row = block * l_gsw + level, column 0 = mask, 1 = body (coefficient domain)
 *   ksk  int32[P][N][t][base-1][n+1]
 * thfhe_kms_tlev_rotate   <- mk_ith_blind_rotate: for `count` gates, party `party`: bara int32[count][n] (the party's mod-switched mask
 *                            words) -> lev int64[count][l_lev][2][N], the rotated TLev accumulator (mask, body per level)
 * thfhe_kms_rlwe_rotate   <- mk_single_blind_rotate (new_mk_internals.jl:226-238, the fast_boot route :255-269): acc int64[count][2][N]
 *                            (mask, body), rotated in place by the party's n TGSW-encrypted key bits
 * thfhe_kms_keyswitch     <- mk_keyswitch (mk_internals.jl:714-728): u int32[count][P N + 1] -> out int32[count][P n + 1]
 * thfhe_kms_set_relin_keys <- the rest of MKBootstrapKey_new (mk_api.jl:440-455): uni int64[P][3][l_uni][N] (d, f0, f1 of every party's
 *                            uni-encryption), pk int64[P][l_uni][N] (public keys), crs int64[l_uni][N] (shared key); transformed once
 * thfhe_kms_lev_rlwe_mul  <- mk_lev_rlwe_mul (new_mk_internals.jl:185-207 = tlev_extern_mul + UniProduct_new :85-127): accum
 *                            int64[count][P+1][N] (a_0 .. a_{P-1}, b) in place, lev int64[count][l_lev][2][N]
 * thfhe_kms_bootstrap     <- mk_bootstrap_new (new_mk_internals.jl:315-325) / mk_bootstrap_wo_keyswitch_new (:302-313): x int32[count][P n+1]
 *                            -> u int32[count][P N + 1] (before the key switch; may be null) and / or out int32[count][P n + 1] (may be null);
 *                            fast_boot != 0 selects mk_blind_rotate_new_v2 (:255-269)
 * thfhe_kms_gates         <- mk_gate_nand_new (new_mk_gates.jl:1-7) and the other two-input gates of gates.jl on the same bootstrap
 * Between the gate's linear part and the key switch everything stays in HBM: prologue, per party the TLev rotation, gadget
 * decompositions and exact polynomial multiply-accumulates of the relinearisation, extraction. */
typedef struct {
    int32_t n, N, parties;
    int32_t l_gsw, bg_gsw;
    int32_t l_lev, bg_lev;
    int32_t l_uni, bg_uni;
    int32_t ks_t, ks_basebit;
} thfhe_kms_params;
typedef struct thfhe_kms_ctx thfhe_kms_ctx;
int thfhe_kms_ctx_create(const thfhe_kms_params *p, const int64_t *gsw, const int32_t *ksk, int device, thfhe_kms_ctx **out);
void thfhe_kms_ctx_destroy(thfhe_kms_ctx *ctx);
int thfhe_kms_tlev_rotate(thfhe_kms_ctx *ctx, int party, const int32_t *bara, int64_t *lev, size_t count);
int thfhe_kms_rlwe_rotate(thfhe_kms_ctx *ctx, int party, const int32_t *bara, int64_t *acc, size_t count);
int thfhe_kms_keyswitch(thfhe_kms_ctx *ctx, const int32_t *u, int32_t *out, size_t count);
int thfhe_kms_set_relin_keys(thfhe_kms_ctx *ctx, const int64_t *uni, const int64_t *pk, const int64_t *crs);
int thfhe_kms_lev_rlwe_mul(thfhe_kms_ctx *ctx, int party, int64_t *accum, const int64_t *lev, size_t count);
int thfhe_kms_bootstrap(thfhe_kms_ctx *ctx, int64_t mu, const int32_t *x, int32_t *u, int32_t *out, size_t count, int fast_boot);
int thfhe_kms_gates(thfhe_kms_ctx *ctx, int op, const int32_t *x, const int32_t *y, int32_t *out, size_t count, int fast_boot);
/* Party-sharded KMS evaluation, device-resident (thfhe/kms_sharded.py): in mk_blind_rotate_new the per-party TLev rotations read nothing the
 * relinearisation writes (new_mk_internals.jl:241-252), so ranks rotate disjoint blocks of parties, exchange the TLev accumulators once
 * (RCCL all-gather on device tensors) and finish replicated.  All pointers are DEVICE pointers; op = opcode NAND .. ORYN or -1 (plain
 * mk_bootstrap_new of x, mu = 1/8).
 *   rotate_parties_dev : gate linear part + mod-switch, then mk_ith_blind_rotate for parties [first_party, first_party + n_parties)
 *                        -> d_lev int64[n_parties][count][l_lev][2][N]                                   (returns after enqueueing)
 *   finish_dev         : trivial accumulator, mk_lev_rlwe_mul for p = 0 .. P-1 with d_lev_all int64[P][count][l_lev][2][N], extraction,
 *                        key switch -> d_out int32[count][P n + 1]                                        (returns after the digit-range check)
 *   set_stream         : enqueue on the caller's HIP stream (NULL: the context's own)                                                    */
int thfhe_kms_rotate_parties_dev(thfhe_kms_ctx *ctx, int op, const int32_t *d_x, const int32_t *d_y, int first_party, int n_parties, int64_t *d_lev,
                                 size_t count);
int thfhe_kms_finish_dev(thfhe_kms_ctx *ctx, int op, const int32_t *d_x, const int32_t *d_y, const int64_t *d_lev_all, int32_t *d_out, size_t count);
int thfhe_kms_set_stream(thfhe_kms_ctx *ctx, void *hip_stream);
/* Launches of at most `max_single_jobs` TLev / RLWE rotations run one job per workgroup, larger ones two jobs per workgroup that share
 * every key chunk (kms_tlev_rotate_pair_kernel).  Default 256 = one workgroup per CU of an MI355X. */
int thfhe_kms_set_pair_threshold(thfhe_kms_ctx *ctx, long max_single_jobs);

#ifdef __cplusplus
}
#endif
#endif /* THFHE_HIP_H */
