/*
 * tfhe_shim.h -- struct-compatible replacements for the libtfhe gate entry points the reference's C++ side links
 * against (README.md:102, src/KNN_medical_data.cpp:130,142-151,227,388-396, src/Convert.cpp:31,
 * src/TN_bootstrap.cpp:13).  libtfhe itself is NOT part of the reference tree (un-vendored submodule); the struct
 * layouts below are the x86-64 layouts recovered from the DWARF info of the reference's prebuilt bin/KNN_medical_data
 * (SURVEY.md section 8b).  A program built against <tfhe/tfhe.h> can link libthfhe_hip.so in place of
 * libtfhe-spqlios-fma for these symbols: the cloud key's coefficient-domain TGSW samples (bk->bk) and key-switching
 * key (bk->ks) are read once per key set, transformed and cached on the GPU; bkFFT (SPQLIOS-private) is never read.
 *
 * All functions: void f(LweSample* result, const LweSample* ca, const LweSample* cb, const CloudKeySet* bk);
 * result may alias an input.  Thread-safe (OpenMP callers, src/KNN_medical_data.cpp:681).
 */
#ifndef THFHE_TFHE_SHIM_H
#define THFHE_TFHE_SHIM_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t Torus32;

typedef struct LweParams {
    int32_t n;
    double alpha_min;
    double alpha_max;
} LweParams;

typedef struct LweSample {
    Torus32 *a;
    Torus32 b;
    double current_variance;
} LweSample;

typedef struct LweKeySwitchKey {
    int32_t n;        /* input dimension (k*N) */
    int32_t t;
    int32_t basebit;
    int32_t base;
    const LweParams *out_params;
    LweSample *ks0_raw;
    LweSample **ks1_raw;
    LweSample ***ks;  /* ks[i][j][h], h in [0, base) */
} LweKeySwitchKey;

typedef struct TLweParams {
    int32_t N;
    int32_t k;
    double alpha_min;
    double alpha_max;
    LweParams extracted_lweparams;
} TLweParams;

typedef struct TorusPolynomial {
    int32_t N;
    Torus32 *coefsT;
} TorusPolynomial;

typedef struct TLweSample {
    TorusPolynomial *a; /* k+1 polynomials, a[k] is the body */
    TorusPolynomial *b;
    double current_variance;
    int32_t k;
} TLweSample;

typedef struct TGswParams {
    int32_t l;
    int32_t Bgbit;
    int32_t Bg;
    int32_t halfBg;
    uint32_t maskMod;
    const TLweParams *tlwe_params;
    int32_t kpl;
    Torus32 *h;
    uint32_t offset;
} TGswParams;

typedef struct TGswSample {
    TLweSample *all_sample; /* (k+1)*l rows, row j*l + p */
    TLweSample **bloc_sample;
    int32_t k;
    int32_t l;
} TGswSample;

typedef struct LweBootstrappingKey {
    const LweParams *in_out_params;
    const TGswParams *bk_params;
    const TLweParams *accum_params;
    const LweParams *extract_params;
    TGswSample *bk;
    LweKeySwitchKey *ks;
} LweBootstrappingKey;

typedef struct TFheGateBootstrappingParameterSet {
    int32_t ks_t;
    int32_t ks_basebit;
    const LweParams *in_out_params;
    const TGswParams *tgsw_params;
} TFheGateBootstrappingParameterSet;

typedef struct TFheGateBootstrappingCloudKeySet {
    const TFheGateBootstrappingParameterSet *params;
    const LweBootstrappingKey *bk;
    const void *bkFFT; /* LweBootstrappingKeyFFT*: never dereferenced */
} TFheGateBootstrappingCloudKeySet;

/* ---- secret-key side (layouts from the same DWARF: LweKey 16 B, TLweKey 16 B, TGswKey 40 B, SecretKeySet 48 B) ---- */
typedef struct LweKey {
    const LweParams *params;
    int32_t *key;
} LweKey;

typedef struct IntPolynomial {
    int32_t N;
    int32_t *coefs;
} IntPolynomial;

typedef struct TLweKey {
    const TLweParams *params;
    IntPolynomial *key; /* k polynomials */
} TLweKey;

typedef struct TGswKey {
    const TGswParams *params;
    const TLweParams *tlwe_params;
    IntPolynomial *key; /* = tlwe_key.key */
    TLweKey tlwe_key;
} TGswKey;

typedef struct TFheGateBootstrappingSecretKeySet {
    const TFheGateBootstrappingParameterSet *params;
    const LweKey *lwe_key;
    const TGswKey *tgsw_key;
    TFheGateBootstrappingCloudKeySet cloud;
} TFheGateBootstrappingSecretKeySet;

void bootsNAND(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsAND(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsXOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsXNOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsNOR(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsANDNY(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsANDYN(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsORNY(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsORYN(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk);
void bootsMUX(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c, const TFheGateBootstrappingCloudKeySet *bk);
void bootsNOT(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk);
void bootsCOPY(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk);
void bootsCONSTANT(LweSample *result, int32_t value, const TFheGateBootstrappingCloudKeySet *bk);

/* ---- the host half of the libtfhe surface (csrc/tfhe_host.cpp): what the reference's programs call around the gates.
 * Reference call sites: src/KeyGen.cpp:31-57 (parameters, seed, key generation, key files), src/Convert.cpp:35-70 (bootsSymEncrypt /
 * bootsSymDecrypt, key files, ciphertext arrays), src/KNN_medical_data.cpp:23-121,163-180 (default parameters, ciphertext files),
 * src/libthfhe.cpp:316-338 (new_LweParams / new_TLweParams / new_TGswParams), src/bootstrap_modules.cpp:52-55,95 (the seed and bit order of
 * the committed fixtures).  libtfhe is not in the reference tree; these follow its published behaviour.  Pinned by the reference's fixtures:
 * the seeded LWE key, the LweSample file record, encoding / decryption.  Key-set files are this library's own containers (unpinned). */
LweParams *new_LweParams(int32_t n, double alpha_min, double alpha_max);
TLweParams *new_TLweParams(int32_t N, int32_t k, double alpha_min, double alpha_max);
TGswParams *new_TGswParams(int32_t l, int32_t Bgbit, const TLweParams *tlwe_params);
void delete_LweParams(LweParams *p);
void delete_TLweParams(TLweParams *p);
void delete_TGswParams(TGswParams *p);
TFheGateBootstrappingParameterSet *new_default_gate_bootstrapping_parameters(int32_t minimum_lambda);
void delete_gate_bootstrapping_parameters(TFheGateBootstrappingParameterSet *params);
void tfhe_random_generator_setSeed(uint32_t *values, int32_t size);
Torus32 gaussian32(Torus32 message, double sigma);
Torus32 modSwitchToTorus32(int32_t mu, int32_t Msize);
int32_t modSwitchFromTorus32(Torus32 phase, int32_t Msize);
TFheGateBootstrappingSecretKeySet *new_random_gate_bootstrapping_secret_keyset(const TFheGateBootstrappingParameterSet *params);
void delete_gate_bootstrapping_secret_keyset(TFheGateBootstrappingSecretKeySet *keyset);
void delete_gate_bootstrapping_cloud_keyset(TFheGateBootstrappingCloudKeySet *keyset);
LweSample *new_gate_bootstrapping_ciphertext(const TFheGateBootstrappingParameterSet *params);
LweSample *new_gate_bootstrapping_ciphertext_array(int32_t nbelems, const TFheGateBootstrappingParameterSet *params);
void delete_gate_bootstrapping_ciphertext(LweSample *sample);
void delete_gate_bootstrapping_ciphertext_array(int32_t nbelems, LweSample *samples);
void bootsSymEncrypt(LweSample *result, int32_t message, const TFheGateBootstrappingSecretKeySet *key);
int32_t bootsSymDecrypt(const LweSample *sample, const TFheGateBootstrappingSecretKeySet *key);
void export_gate_bootstrapping_ciphertext_toFile(FILE *F, const LweSample *sample, const TFheGateBootstrappingParameterSet *params);
void import_gate_bootstrapping_ciphertext_fromFile(FILE *F, LweSample *sample, const TFheGateBootstrappingParameterSet *params);
void export_tfheGateBootstrappingParameterSet_toFile(FILE *F, const TFheGateBootstrappingParameterSet *params);
TFheGateBootstrappingParameterSet *new_tfheGateBootstrappingParameterSet_fromFile(FILE *F);
void export_tfheGateBootstrappingCloudKeySet_toFile(FILE *F, const TFheGateBootstrappingCloudKeySet *cloud);
TFheGateBootstrappingCloudKeySet *new_tfheGateBootstrappingCloudKeySet_fromFile(FILE *F);
void export_tfheGateBootstrappingSecretKeySet_toFile(FILE *F, const TFheGateBootstrappingSecretKeySet *key);
TFheGateBootstrappingSecretKeySet *new_tfheGateBootstrappingSecretKeySet_fromFile(FILE *F);

/* torus polynomials of the threshold decryption that follows the gates (ThFHEKeyShare::PartialDecrypt / finalDecrypt, src/libthfhe.cpp:270-314).
 * torusPolynomialAddMulR[FFT]: result += poly1 (*) poly2 mod X^N + 1, exact, ON THE GPU (N = 1024, |poly1 coefficients| <= 512). */
TorusPolynomial *new_TorusPolynomial(int32_t N);
void delete_TorusPolynomial(TorusPolynomial *p);
void torusPolynomialCopy(TorusPolynomial *result, const TorusPolynomial *sample);
void torusPolynomialAddTo(TorusPolynomial *result, const TorusPolynomial *poly2);
void torusPolynomialSubTo(TorusPolynomial *result, const TorusPolynomial *poly2);
void torusPolynomialAddMulR(TorusPolynomial *result, const IntPolynomial *poly1, const TorusPolynomial *poly2);
void torusPolynomialAddMulRFFT(TorusPolynomial *result, const IntPolynomial *poly1, const TorusPolynomial *poly2);

/* Batched form for callers that hold arrays of samples (e.g. one adder level): count gates of one kind. */
int thfhe_tfhe_gate_batch(int op, LweSample *result, const LweSample *ca, const LweSample *cb, const LweSample *cc,
                          int32_t count, const TFheGateBootstrappingCloudKeySet *bk);
/* Drop the cached device context of a key set (call before freeing the key). */
void thfhe_tfhe_forget_key(const TFheGateBootstrappingCloudKeySet *bk);

#ifdef __cplusplus
}
#endif
#endif
