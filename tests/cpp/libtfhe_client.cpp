// libtfhe_client.cpp -- a program written like the reference's own C++ programs, against include/tfhe_shim.h ONLY (TEST CODE): it
// includes no thfhe_* header, calls nothing but libtfhe's names, and links libthfhe_hip.so where the reference links
// libtfhe-spqlios-fma.  The flow is the reference's:
//   keygen     src/KeyGen.cpp:31-57               parameters, seed {100, 20032, 21341}, new_random_gate_bootstrapping_secret_keyset, key files
//   host       src/bootstrap_modules.cpp:76-95    import the committed ciphertext files, decrypt, re-export      (no GPU needed)
//   evaluate   src/Convert.cpp:29-33, src/bootstrap_modules.cpp:20-44,412-432   32 x bootsAND and the FullAdder on the GPU, export, decrypt
//
//   libtfhe_client keygen   <work_dir>
//   libtfhe_client host     <work_dir> <golden_dir>
//   libtfhe_client evaluate <work_dir> <golden_dir>
// Every mode prints "key: value" lines and exits non-zero on the first mismatch with the reference's numbers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tfhe_shim.h"

static std::string path(const char *dir, const char *name) { return std::string(dir) + "/" + name; }
static FILE *open_or_die(const std::string &p, const char *mode) {
    FILE *f = fopen(p.c_str(), mode);
    if (!f) { perror(p.c_str()); exit(2); }
    return f;
}
static void fail(const char *what) {
    fprintf(stderr, "libtfhe_client: MISMATCH: %s\n", what);
    exit(1);
}
static std::vector<unsigned char> slurp(const std::string &p) {
    FILE *f = open_or_die(p, "rb");
    std::vector<unsigned char> v;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + got);
    fclose(f);
    return v;
}

// MSB first: index 0 holds bit 31 (bootsSymEncrypt(&ct[31-i], (v>>i)&1, key), src/bootstrap_modules.cpp:95)
static uint32_t decrypt_word(const LweSample *c, const TFheGateBootstrappingSecretKeySet *key) {
    uint32_t v = 0;
    for (int i = 0; i < 32; i++) v = (v << 1) | (uint32_t)bootsSymDecrypt(&c[i], key);
    return v;
}
static LweSample *import_word(const std::string &file, const TFheGateBootstrappingParameterSet *params) {
    LweSample *c = new_gate_bootstrapping_ciphertext_array(32, params);
    FILE *f = open_or_die(file, "rb");
    for (int i = 0; i < 32; i++) import_gate_bootstrapping_ciphertext_fromFile(f, &c[i], params);
    fclose(f);
    return c;
}
static void export_word(const std::string &file, const LweSample *c, const TFheGateBootstrappingParameterSet *params) {
    FILE *f = open_or_die(file, "wb");
    for (int i = 0; i < 32; i++) export_gate_bootstrapping_ciphertext_toFile(f, &c[i], params);
    fclose(f);
}

static TFheGateBootstrappingSecretKeySet *make_keys() {   // src/KeyGen.cpp:33-38 with the default 128-bit set of src/KNN_medical_data.cpp:25-31
    TFheGateBootstrappingParameterSet *params = new_default_gate_bootstrapping_parameters(110);
    uint32_t seed[] = {100, 20032, 21341};
    tfhe_random_generator_setSeed(seed, 3);
    return new_random_gate_bootstrapping_secret_keyset(params);
}

// the reference's ripple-carry adder, src/bootstrap_modules.cpp:20-44 (MSB first; carrybit[31] is the carry in)
static void FullAdder(LweSample *sum2, LweSample *carrybit, const LweSample *input1, const LweSample *input2, const TFheGateBootstrappingCloudKeySet *bk) {
    LweSample *sum1 = new_gate_bootstrapping_ciphertext_array(32, bk->params);
    LweSample *carry1 = new_gate_bootstrapping_ciphertext_array(32, bk->params);
    LweSample *carry2 = new_gate_bootstrapping_ciphertext_array(32, bk->params);
    for (int i = 31; i >= 0; i--) {
        bootsXOR(&sum1[i], &input1[i], &input2[i], bk);
        bootsAND(&carry1[i], &input1[i], &input2[i], bk);
        bootsXOR(&sum2[i], &sum1[i], &carrybit[i], bk);
        bootsAND(&carry2[i], &sum1[i], &carrybit[i], bk);
        if (i != 0) bootsOR(&carrybit[i - 1], &carry1[i], &carry2[i], bk);
    }
    delete_gate_bootstrapping_ciphertext_array(32, sum1);
    delete_gate_bootstrapping_ciphertext_array(32, carry1);
    delete_gate_bootstrapping_ciphertext_array(32, carry2);
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s keygen|host|evaluate <work_dir> [<golden_dir>]\n", argv[0]); return 2; }
    const std::string mode = argv[1];
    const char *work = argv[2], *golden = argc > 3 ? argv[3] : "";

    if (mode == "keygen") {   // src/KeyGen.cpp:31-57
        TFheGateBootstrappingSecretKeySet *key = make_keys();
        const TFheGateBootstrappingParameterSet *params = key->params;
        printf("n: %d\nN: %d\nl: %d\nBgbit: %d\nks_t: %d\nks_basebit: %d\n", params->in_out_params->n, params->tgsw_params->tlwe_params->N, params->tgsw_params->l,
               params->tgsw_params->Bgbit, params->ks_t, params->ks_basebit);
        printf("lwe_key: ");
        for (int i = 0; i < params->in_out_params->n; i++) putchar('0' + key->lwe_key->key[i]);
        putchar('\n');
        FILE *f = open_or_die(path(work, "secret.key"), "wb");
        export_tfheGateBootstrappingSecretKeySet_toFile(f, key);
        fclose(f);
        f = open_or_die(path(work, "cloud.key"), "wb");
        export_tfheGateBootstrappingCloudKeySet_toFile(f, &key->cloud);
        fclose(f);
        f = open_or_die(path(work, "secret.params"), "wb");
        export_tfheGateBootstrappingParameterSet_toFile(f, params);
        fclose(f);
        delete_gate_bootstrapping_secret_keyset(key);
        delete_gate_bootstrapping_parameters(const_cast<TFheGateBootstrappingParameterSet *>(params));
        return 0;
    }

    // host / evaluate: the key set comes back from the files of the keygen step (src/Convert.cpp:50-70)
    FILE *f = open_or_die(path(work, "secret.key"), "rb");
    TFheGateBootstrappingSecretKeySet *key = new_tfheGateBootstrappingSecretKeySet_fromFile(f);
    fclose(f);
    f = open_or_die(path(work, "secret.params"), "rb");
    TFheGateBootstrappingParameterSet *params = new_tfheGateBootstrappingParameterSet_fromFile(f);
    fclose(f);
    if (params->in_out_params->n != 630 || params->tgsw_params->l != 3 || params->tgsw_params->Bgbit != 7 || params->ks_t != 8 || params->ks_basebit != 2)
        fail("parameter file does not hold the 128-bit default set");

    if (mode == "host") {
        // the reference's committed ciphertexts decrypt to its committed plaintexts under the seeded key
        const struct { const char *file; uint32_t value; } kat[] = {
            {"cloud1.data", 9876u}, {"cloud2.data", 686u}, {"cloud3.data", 1287u}, {"cloud4.data", 2000u}, {"allOne.data", 0xFFFFFFFFu}, {"allZero.data", 0u},
            {"lsbOne.data", 1u}, {"lsbZero.data", 0xFFFFFFFEu}, {"sum.data", 10562u}, {"diff.data", 9190u}, {"carry.data", 3448u}};
        for (const auto &k : kat) {
            LweSample *c = import_word(path(golden, k.file), params);
            const uint32_t v = decrypt_word(c, key);
            printf("%s: %u\n", k.file, v);
            if (v != k.value) fail(k.file);
            // export(import(x)) is byte-identical to the reference's file
            export_word(path(work, k.file), c, params);
            if (slurp(path(work, k.file)) != slurp(path(golden, k.file))) fail("re-exported ciphertext file differs from the reference's bytes");
            delete_gate_bootstrapping_ciphertext_array(32, c);
        }
        // fresh encryptions: variance field and noise level of the reference's fresh ciphertexts (alpha = 2^-15)
        LweSample *c = new_gate_bootstrapping_ciphertext_array(32, params);
        for (int i = 0; i < 32; i++) bootsSymEncrypt(&c[31 - i], (13452 >> i) & 1, key);   // test/plain22.txt
        if (decrypt_word(c, key) != 13452u) fail("bootsSymEncrypt / bootsSymDecrypt round trip");
        printf("fresh_variance: %.6e\n", c[0].current_variance);
        if (c[0].current_variance < 9.3e-10 || c[0].current_variance > 9.33e-10) fail("variance of a fresh ciphertext");
        delete_gate_bootstrapping_ciphertext_array(32, c);
        delete_gate_bootstrapping_secret_keyset(key);
        delete_gate_bootstrapping_parameters(params);
        printf("host: ok\n");
        return 0;
    }

    if (mode == "evaluate") {   // the GPU part: gates on the reference's input ciphertexts under the generated cloud key
        f = open_or_die(path(work, "cloud.key"), "rb");
        TFheGateBootstrappingCloudKeySet *cloud = new_tfheGateBootstrappingCloudKeySet_fromFile(f);
        fclose(f);
        LweSample *c1 = import_word(path(golden, "cloud1.data"), cloud->params);
        LweSample *c2 = import_word(path(golden, "cloud2.data"), cloud->params);
        LweSample *res = new_gate_bootstrapping_ciphertext_array(32, cloud->params);
        for (int i = 0; i < 32; i++) bootsAND(&res[i], &c1[i], &c2[i], cloud);   // Evaluate, src/Convert.cpp:29-33
        export_word(path(work, "and.data"), res, cloud->params);
        LweSample *back = import_word(path(work, "and.data"), params);
        printf("and: %u\n", decrypt_word(back, key));
        if (decrypt_word(back, key) != (9876u & 686u)) fail("32 x bootsAND");
        // adder of src/bootstrap_modules.cpp:412-432: carry in = bit 31 of allZero.data
        LweSample *zero = import_word(path(golden, "allZero.data"), cloud->params);
        LweSample *sum = new_gate_bootstrapping_ciphertext_array(32, cloud->params);
        LweSample *carry = new_gate_bootstrapping_ciphertext_array(32, cloud->params);
        bootsCOPY(&carry[31], &zero[31], cloud);
        FullAdder(sum, carry, c1, c2, cloud);
        export_word(path(work, "sum.data"), sum, cloud->params);
        export_word(path(work, "carry.data"), carry, cloud->params);
        LweSample *s2 = import_word(path(work, "sum.data"), params), *k2 = import_word(path(work, "carry.data"), params);
        printf("sum: %u\ncarry: %u\n", decrypt_word(s2, key), decrypt_word(k2, key));
        if (decrypt_word(s2, key) != 10562u) fail("FullAdder sum (reference: test/bootstrap_modules/sum.txt)");
        if (decrypt_word(k2, key) != 3448u) fail("FullAdder carries (reference: test/bootstrap_modules/carry.txt)");
        // the multiply of ThFHEKeyShare::PartialDecrypt (src/libthfhe.cpp:279-287) under the reference's own name: partial = share (*) a + noise
        {
            const int N = 1024;
            TorusPolynomial *acc = new_TorusPolynomial(N), *a = new_TorusPolynomial(N), *want = new_TorusPolynomial(N);
            std::vector<int32_t> share(N);
            uint32_t x = 12345u;
            for (int i = 0; i < N; i++) {
                x = x * 1664525u + 1013904223u;
                a->coefsT[i] = (Torus32)x;
                x = x * 1664525u + 1013904223u;
                share[i] = (int32_t)(x >> 22) - 512;            // key shares are small integers (|s| <= 512)
                x = x * 1664525u + 1013904223u;
                acc->coefsT[i] = (Torus32)(x >> 8);              // the smudging noise already in the accumulator
            }
            torusPolynomialCopy(want, acc);
            for (int i = 0; i < N; i++)                          // schoolbook negacyclic product, wrapping mod 2^32
                for (int j = 0; j < N; j++) {
                    const uint32_t t = (uint32_t)share[i] * (uint32_t)a->coefsT[j];
                    const int q = i + j;
                    if (q < N) want->coefsT[q] = (Torus32)((uint32_t)want->coefsT[q] + t);
                    else want->coefsT[q - N] = (Torus32)((uint32_t)want->coefsT[q - N] - t);
                }
            IntPolynomial sp{N, share.data()};
            torusPolynomialAddMulR(acc, &sp, a);
            if (memcmp(acc->coefsT, want->coefsT, sizeof(Torus32) * N) != 0) fail("torusPolynomialAddMulR differs from the schoolbook product");
            printf("addmulr: exact\n");
            delete_TorusPolynomial(acc), delete_TorusPolynomial(a), delete_TorusPolynomial(want);
        }
        for (LweSample *p : {c1, c2, res, back, zero, sum, carry, s2, k2}) delete_gate_bootstrapping_ciphertext_array(32, p);
        delete_gate_bootstrapping_cloud_keyset(cloud);
        delete_gate_bootstrapping_secret_keyset(key);
        printf("evaluate: ok\n");
        return 0;
    }
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    return 2;
}
