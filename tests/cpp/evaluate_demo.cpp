// evaluate_demo.cpp -- a C++ CLIENT of the drop-in boundary (TEST CODE; not part of the product library).
//
// Written the way the reference's C++ programs use libtfhe (src/Convert.cpp:28-33 `Evaluate`: bootsAND over the 32 bits of two words;
// src/KNN_medical_data.cpp:134-157 `FullAdder`: XOR / AND / OR with in-place carries; :681 `#pragma omp parallel for` callers), but
// compiled with plain g++ against include/tfhe_shim.h and linked with libthfhe_hip.so instead of libtfhe.  Key material and ciphertexts
// come from a blob written by the test harness (libtfhe's key files cannot be produced here); this program lays them out in libtfhe's
// struct graph, evaluates, and writes the result records back.
//
//   evaluate_demo <blob_in> <blob_out>
// blob_in : int32 header {n, N, l, Bgbit, ks_t, ks_basebit, nbits} | bk int32[n][2l][2][N] | ksk int32[N][t][base-1][n+1]
//           | word1 int32[nbits][n+1] | word2 int32[nbits][n+1] | carry_in int32[n+1]
// blob_out: and int32[nbits][n+1] | sum int32[nbits][n+1] | carry int32[nbits][n+1]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/tfhe_shim.h"

static std::vector<int32_t> read_blob(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<int32_t> v(bytes / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

struct Samples {  // new_gate_bootstrapping_ciphertext_array: caller-owned LweSample array
    std::vector<LweSample> s;
    std::vector<Torus32> store;
    Samples(int count, int n) : s(count), store((size_t)count * n, 0) {
        for (int i = 0; i < count; i++) s[i] = LweSample{store.data() + (size_t)i * n, 0, 0.0};
    }
    void load(const int32_t *rec, int n) {
        for (size_t i = 0; i < s.size(); i++) {
            memcpy(s[i].a, rec + i * (n + 1), sizeof(int32_t) * n);
            s[i].b = rec[i * (n + 1) + n];
        }
    }
    void dump(FILE *f, int n) const {
        for (const LweSample &x : s) {
            fwrite(x.a, 4, n, f);
            fwrite(&x.b, 4, 1, f);
        }
    }
};

// the reference's ripple-carry adder, src/KNN_medical_data.cpp:134-157 (MSB first; carrybit[nbits-1] is the carry in)
static void FullAdder(LweSample *sum2, LweSample *carrybit, LweSample *input1, LweSample *input2, const int nbits,
                      const TFheGateBootstrappingCloudKeySet *bk, int n) {
    Samples sum1(nbits, n), carry1(nbits, n), carry2(nbits, n);
    for (int i = nbits - 1; i >= 0; i--) {
        bootsXOR(&sum1.s[i], &input1[i], &input2[i], bk);
        bootsAND(&carry1.s[i], &input1[i], &input2[i], bk);
        bootsXOR(&sum2[i], &sum1.s[i], &carrybit[i], bk);
        bootsAND(&carry2.s[i], &sum1.s[i], &carrybit[i], bk);
        if (i != 0) bootsOR(&carrybit[i - 1], &carry1.s[i], &carry2.s[i], bk);
    }
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s blob_in blob_out\n", argv[0]); return 2; }
    std::vector<int32_t> blob = read_blob(argv[1]);
    const int32_t *h = blob.data();
    const int n = h[0], N = h[1], l = h[2], Bgbit = h[3], t = h[4], bb = h[5], nbits = h[6], base = 1 << bb;
    const int32_t *bk = h + 7, *ksk = bk + (size_t)n * 2 * l * 2 * N, *w1 = ksk + (size_t)N * t * (base - 1) * (n + 1);
    const int32_t *w2 = w1 + (size_t)nbits * (n + 1), *cin = w2 + (size_t)nbits * (n + 1);

    // ---- libtfhe's struct graph (layouts: include/tfhe_shim.h) -----------------------------------------------------------------
    LweParams in_out{n, 0.0, 0.0};
    TLweParams accum{N, 1, 0.0, 0.0, LweParams{N, 0.0, 0.0}};
    TGswParams tgsw{};
    tgsw.l = l, tgsw.Bgbit = Bgbit, tgsw.Bg = 1 << Bgbit, tgsw.halfBg = 1 << (Bgbit - 1), tgsw.maskMod = (1u << Bgbit) - 1;
    tgsw.tlwe_params = &accum, tgsw.kpl = 2 * l;
    std::vector<TorusPolynomial> polys((size_t)n * 2 * l * 2);
    std::vector<TLweSample> rows((size_t)n * 2 * l);
    std::vector<TGswSample> gsw(n);
    for (int i = 0; i < n; i++) {
        for (int r = 0; r < 2 * l; r++) {
            TorusPolynomial *p = &polys[((size_t)i * 2 * l + r) * 2];
            for (int c = 0; c < 2; c++) p[c] = TorusPolynomial{N, const_cast<int32_t *>(bk) + (((size_t)i * 2 * l + r) * 2 + c) * N};
            rows[(size_t)i * 2 * l + r] = TLweSample{p, p + 1, 0.0, 1};
        }
        gsw[i] = TGswSample{&rows[(size_t)i * 2 * l], nullptr, 1, l};
    }
    std::vector<LweSample> ks_samples((size_t)N * t * base);
    std::vector<LweSample *> ks_j((size_t)N * t);
    std::vector<LweSample **> ks_i(N);
    std::vector<int32_t> zero_row(n, 0);
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < t; j++) {
            LweSample *e = &ks_samples[((size_t)i * t + j) * base];
            e[0] = LweSample{zero_row.data(), 0, 0.0};  // h = 0 is the noiseless zero sample in libtfhe; never read by a key switch
            for (int hh = 1; hh < base; hh++) {
                const int32_t *row = ksk + (((size_t)i * t + j) * (base - 1) + (hh - 1)) * (n + 1);
                e[hh] = LweSample{const_cast<int32_t *>(row), row[n], 0.0};
            }
            ks_j[(size_t)i * t + j] = e;
        }
        ks_i[i] = &ks_j[(size_t)i * t];
    }
    LweKeySwitchKey ks{N, t, bb, base, &in_out, ks_samples.data(), ks_j.data(), ks_i.data()};
    LweBootstrappingKey lbk{&in_out, &tgsw, &accum, &accum.extracted_lweparams, gsw.data(), &ks};
    TFheGateBootstrappingParameterSet params{t, bb, &in_out, &tgsw};
    TFheGateBootstrappingCloudKeySet cloud{&params, &lbk, nullptr};

    Samples c1(nbits, n), c2(nbits, n), res_and(nbits, n), sum(nbits, n), carry(nbits, n);
    c1.load(w1, n);
    c2.load(w2, n);
    // Evaluate, src/Convert.cpp:28-33 -- here from OpenMP threads on a shared key, as src/KNN_medical_data.cpp:681 does
#pragma omp parallel for
    for (int i = 0; i < nbits; i++) bootsAND(&res_and.s[i], &c1.s[i], &c2.s[i], &cloud);
    // FullAdder with the carry chain written in place
    carry.load(w1, n);  // any initial content; FullAdder overwrites positions 0 .. nbits-2
    memcpy(carry.s[nbits - 1].a, cin, sizeof(int32_t) * n);
    carry.s[nbits - 1].b = cin[n];
    FullAdder(sum.s.data(), carry.s.data(), c1.s.data(), c2.s.data(), nbits, &cloud, n);

    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }
    res_and.dump(out, n);
    sum.dump(out, n);
    carry.dump(out, n);
    fclose(out);
    thfhe_tfhe_forget_key(&cloud);
    printf("evaluate_demo: %d-bit AND and FullAdder done\n", nbits);
    return 0;
}
