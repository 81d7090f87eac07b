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

/* Batched form for callers that hold arrays of samples (e.g. one adder level): count gates of one kind. */
int thfhe_tfhe_gate_batch(int op, LweSample *result, const LweSample *ca, const LweSample *cb, const LweSample *cc,
                          int32_t count, const TFheGateBootstrappingCloudKeySet *bk);
/* Drop the cached device context of a key set (call before freeing the key). */
void thfhe_tfhe_forget_key(const TFheGateBootstrappingCloudKeySet *bk);

#ifdef __cplusplus
}
#endif
#endif
