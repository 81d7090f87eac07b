// tfhe_shim.cpp -- the libtfhe gate symbols (include/tfhe_shim.h) on top of the batch C ABI.
#include "../../include/tfhe_shim.h"

#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_keyslot.h"

static_assert(sizeof(LweParams) == 24 && offsetof(LweParams, alpha_min) == 8, "LweParams layout");
static_assert(sizeof(LweSample) == 24 && offsetof(LweSample, b) == 8 && offsetof(LweSample, current_variance) == 16, "LweSample layout");
static_assert(offsetof(LweKeySwitchKey, out_params) == 16 && offsetof(LweKeySwitchKey, ks) == 40, "LweKeySwitchKey layout");
static_assert(sizeof(TLweParams) == 48 && offsetof(TLweParams, extracted_lweparams) == 24, "TLweParams layout");
static_assert(offsetof(TorusPolynomial, coefsT) == 8, "TorusPolynomial layout");
static_assert(offsetof(TLweSample, b) == 8 && offsetof(TLweSample, k) == 24, "TLweSample layout");
static_assert(offsetof(TGswParams, maskMod) == 16 && offsetof(TGswParams, tlwe_params) == 24 && offsetof(TGswParams, kpl) == 32 &&
                  offsetof(TGswParams, h) == 40 && offsetof(TGswParams, offset) == 48, "TGswParams layout");
static_assert(offsetof(TGswSample, k) == 16 && offsetof(TGswSample, l) == 20, "TGswSample layout");
static_assert(offsetof(LweBootstrappingKey, bk) == 32 && offsetof(LweBootstrappingKey, ks) == 40, "LweBootstrappingKey layout");
static_assert(offsetof(TFheGateBootstrappingParameterSet, in_out_params) == 8 && offsetof(TFheGateBootstrappingParameterSet, tgsw_params) == 16, "ParameterSet layout");
static_assert(offsetof(TFheGateBootstrappingCloudKeySet, bk) == 8 && offsetof(TFheGateBootstrappingCloudKeySet, bkFFT) == 16, "CloudKeySet layout");

namespace {

// The device context is cached per key-set ADDRESS.  A libtfhe client may delete a key set and load another one at the same
// address, so every entry carries a cheap fingerprint of the key it was built from (table pointers, shape, a few key words);
// a mismatch builds new device tables instead of evaluating under a stale key (the old context dies with its last caller).
struct Fingerprint {
    const void *bk_rows, *ks_rows;
    int32_t n, N, l, Bgbit, ks_t, ks_basebit;
    uint32_t words[8];
    bool operator==(const Fingerprint &o) const {
        if (bk_rows != o.bk_rows || ks_rows != o.ks_rows || n != o.n || N != o.N || l != o.l || Bgbit != o.Bgbit || ks_t != o.ks_t ||
            ks_basebit != o.ks_basebit)
            return false;
        for (int q = 0; q < 8; q++)
            if (words[q] != o.words[q]) return false;
        return true;
    }
};
Fingerprint fingerprint(const TFheGateBootstrappingCloudKeySet *bk) {
    const LweBootstrappingKey *b = bk->bk;
    Fingerprint f{};
    f.bk_rows = b->bk;
    f.ks_rows = b->ks->ks;
    f.n = b->in_out_params->n, f.N = b->accum_params->N, f.l = b->bk_params->l, f.Bgbit = b->bk_params->Bgbit;
    f.ks_t = b->ks->t, f.ks_basebit = b->ks->basebit;
    const int rows = (b->accum_params->k + 1) * f.l;
    for (int q = 0; q < 4; q++) {  // first / last TGSW sample, first / last row, a body and a mask coefficient each
        const TGswSample &s = b->bk[q < 2 ? 0 : f.n - 1];
        const TLweSample &r = s.all_sample[(q & 1) ? rows - 1 : 0];
        f.words[2 * q] = (uint32_t)r.a[0].coefsT[q];
        f.words[2 * q + 1] = (uint32_t)r.a[b->accum_params->k].coefsT[f.N - 1 - q];
    }
    return f;
}
struct Request {
    int op;
    LweSample *result;
    const LweSample *a, *b, *c;
    bool done = false;
};
// One Slot = device context + combining queue of one key set, held by shared_ptr for the whole of every call (thfhe_keyslot.h):
// a key swap at the same address or thfhe_tfhe_forget_key can never destroy a context, mutex or condition variable under a caller.
using Cache = thfhe_slot::Cache<thfhe_ctx, Fingerprint, Request>;
using Entry = Cache::SlotT;
Cache g_cache;

[[noreturn]] void die(const char *what) {
    // the libtfhe gate functions return void and abort on internal errors ("Errors")
    std::fprintf(stderr, "libthfhe_hip (tfhe shim): %s: %s\n", what, thfhe_last_error());
    std::abort();
}

bool build_ctx(const TFheGateBootstrappingCloudKeySet *bk, Entry &e) {
    const LweBootstrappingKey *b = bk->bk;
    thfhe_params p{};
    p.n = b->in_out_params->n;
    p.N = b->accum_params->N;
    p.k = b->accum_params->k;
    p.l = b->bk_params->l;
    p.Bgbit = b->bk_params->Bgbit;
    p.ks_t = b->ks->t;
    p.ks_basebit = b->ks->basebit;
    p.torus_bits = 32;
    p.parties = 1;
    const int rows = (p.k + 1) * p.l, N = p.N;
    std::vector<int32_t> bkc((size_t)p.n * rows * (p.k + 1) * N);
    for (int i = 0; i < p.n; i++)
        for (int r = 0; r < rows; r++)
            for (int c = 0; c <= p.k; c++) {
                const Torus32 *src = b->bk[i].all_sample[r].a[c].coefsT;
                int32_t *dst = bkc.data() + (((size_t)i * rows + r) * (p.k + 1) + c) * N;
                for (int q = 0; q < N; q++) dst[q] = src[q];
            }
    const int base = 1 << p.ks_basebit, Nin = b->ks->n;
    std::vector<int32_t> ksk((size_t)Nin * p.ks_t * (base - 1) * (p.n + 1));
    for (int i = 0; i < Nin; i++)
        for (int j = 0; j < p.ks_t; j++)
            for (int h = 1; h < base; h++) {
                const LweSample *s = &b->ks->ks[i][j][h];
                int32_t *dst = ksk.data() + ((((size_t)i * p.ks_t + j) * (base - 1)) + (h - 1)) * (p.n + 1);
                for (int q = 0; q < p.n; q++) dst[q] = s->a[q];
                dst[p.n] = s->b;
            }
    e.n = p.n;
    e.destroy = thfhe_ctx_destroy;
    const char *dev = std::getenv("THFHE_DEVICE");
    return thfhe_ctx_create(&p, bkc.data(), ksk.data(), dev ? std::atoi(dev) : 0, &e.ctx) == THFHE_OK;
}

Cache::Ptr get_ctx(const TFheGateBootstrappingCloudKeySet *bk) {
    Cache::Ptr s = g_cache.acquire(bk, fingerprint(bk), [&](Entry &e) { return build_ctx(bk, e); });
    if (!s) die("cannot create device context");
    return s;
}

// One evaluation of `count` gates of one kind on contiguous LweSample arrays.
void run_batch(const Entry &e, int op, LweSample *result, const LweSample *ca, const LweSample *cb, const LweSample *cc, int count) {
    const size_t rec = (size_t)e.n + 1;
    std::vector<int32_t> buf(4 * rec * count);
    int32_t *x = buf.data(), *y = x + rec * count, *z = y + rec * count, *o = z + rec * count;
    auto pack = [&](int32_t *dst, const LweSample *s) {
        for (int g = 0; g < count; g++) {
            for (int q = 0; q < e.n; q++) dst[g * rec + q] = s[g].a[q];
            dst[g * rec + e.n] = s[g].b;
        }
    };
    pack(x, ca);
    if (cb) pack(y, cb);
    if (cc) pack(z, cc);
    if (thfhe_gates(e.ctx, op, x, cb ? y : nullptr, cc ? z : nullptr, o, (size_t)count) != THFHE_OK) die("gate evaluation failed");
    for (int g = 0; g < count; g++) {  // inputs were fully read above: result may alias them
        for (int q = 0; q < e.n; q++) result[g].a[q] = o[g * rec + q];
        result[g].b = o[g * rec + e.n];
        result[g].current_variance = 0.0;
    }
}

// ---- combining the single-gate entry points ------------------------------------------------------------------------
// The reference calls boots* from an OpenMP loop on a shared key set (src/KNN_medical_data.cpp:681-691).  Calls that arrive
// while a launch is in flight are queued on the key's Slot and evaluated TOGETHER by the next leader thread in one mixed-opcode
// launch (thfhe_gates_mixed; MUX requests in one thfhe_gates call), so T concurrent callers cost one gate latency, not T.
void execute(const Entry &e, const std::vector<Request *> &batch) {
    const size_t rec = (size_t)e.n + 1;
    std::vector<Request *> two, mux;
    for (Request *r : batch) (r->op == THFHE_MUX ? mux : two).push_back(r);
    auto pack1 = [&](int32_t *dst, const LweSample *s) {
        for (int q = 0; q < e.n; q++) dst[q] = s->a[q];
        dst[e.n] = s->b;
    };
    auto unpack1 = [&](LweSample *dst, const int32_t *src) {
        for (int q = 0; q < e.n; q++) dst->a[q] = src[q];
        dst->b = src[e.n];
        dst->current_variance = 0.0;
    };
    if (!two.empty()) {
        const size_t cnt = two.size();
        std::vector<int32_t> buf(3 * rec * cnt), ops(cnt);
        int32_t *x = buf.data(), *y = x + rec * cnt, *o = y + rec * cnt;
        for (size_t g = 0; g < cnt; g++) {
            ops[g] = two[g]->op;
            pack1(x + g * rec, two[g]->a);
            pack1(y + g * rec, two[g]->b);
        }
        if (thfhe_gates_mixed(e.ctx, ops.data(), x, y, o, cnt) != THFHE_OK) die("gate evaluation failed");
        for (size_t g = 0; g < cnt; g++) unpack1(two[g]->result, o + g * rec);
    }
    if (!mux.empty()) {
        const size_t cnt = mux.size();
        std::vector<int32_t> buf(4 * rec * cnt);
        int32_t *x = buf.data(), *y = x + rec * cnt, *z = y + rec * cnt, *o = z + rec * cnt;
        for (size_t g = 0; g < cnt; g++) {
            pack1(x + g * rec, mux[g]->a);
            pack1(y + g * rec, mux[g]->b);
            pack1(z + g * rec, mux[g]->c);
        }
        if (thfhe_gates(e.ctx, THFHE_MUX, x, y, z, o, cnt) != THFHE_OK) die("gate evaluation failed");
        for (size_t g = 0; g < cnt; g++) unpack1(mux[g]->result, o + g * rec);
    }
}

void run_one(int op, LweSample *result, const LweSample *ca, const LweSample *cb, const LweSample *cc, const TFheGateBootstrappingCloudKeySet *bk) {
    const Cache::Ptr e = get_ctx(bk);  // keeps context, queue, mutex and condition variable alive until this call returns
    Request r{op, result, ca, cb, cc};
    thfhe_slot::combine(*e, r, [](Entry &slot, const std::vector<Request *> &batch) { execute(slot, batch); });
}

void run(int op, LweSample *result, const LweSample *ca, const LweSample *cb, const LweSample *cc, int count,
         const TFheGateBootstrappingCloudKeySet *bk) {
    const Cache::Ptr e = get_ctx(bk);
    run_batch(*e, op, result, ca, cb, cc, count);
}

}  // namespace

extern "C" {

#define THFHE_GATE2(NAME, OP)                                                                                                   \
    void NAME(LweSample *result, const LweSample *ca, const LweSample *cb, const TFheGateBootstrappingCloudKeySet *bk) {        \
        run_one(OP, result, ca, cb, nullptr, bk);                                                                                \
    }
THFHE_GATE2(bootsNAND, THFHE_NAND)
THFHE_GATE2(bootsOR, THFHE_OR)
THFHE_GATE2(bootsAND, THFHE_AND)
THFHE_GATE2(bootsXOR, THFHE_XOR)
THFHE_GATE2(bootsXNOR, THFHE_XNOR)
THFHE_GATE2(bootsNOR, THFHE_NOR)
THFHE_GATE2(bootsANDNY, THFHE_ANDNY)
THFHE_GATE2(bootsANDYN, THFHE_ANDYN)
THFHE_GATE2(bootsORNY, THFHE_ORNY)
THFHE_GATE2(bootsORYN, THFHE_ORYN)
#undef THFHE_GATE2

void bootsMUX(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c, const TFheGateBootstrappingCloudKeySet *bk) {
    run_one(THFHE_MUX, result, a, b, c, bk);
}
void bootsNOT(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk) {
    const int n = bk->params->in_out_params->n;  // not bootstrapped: plain negation (J/gates.jl:76-79)
    for (int q = 0; q < n; q++) result->a[q] = (Torus32)(0u - (uint32_t)ca->a[q]);
    result->b = (Torus32)(0u - (uint32_t)ca->b);
    result->current_variance = ca->current_variance;
}
void bootsCOPY(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk) {
    const int n = bk->params->in_out_params->n;
    for (int q = 0; q < n; q++) result->a[q] = ca->a[q];
    result->b = ca->b;
    result->current_variance = ca->current_variance;
}
void bootsCONSTANT(LweSample *result, int32_t value, const TFheGateBootstrappingCloudKeySet *bk) {
    const int n = bk->params->in_out_params->n;  // noiseless trivial +-1/8 (J/gates.jl:91-93)
    for (int q = 0; q < n; q++) result->a[q] = 0;
    result->b = value ? (1 << 29) : -(1 << 29);
    result->current_variance = 0.0;
}
int thfhe_tfhe_gate_batch(int op, LweSample *result, const LweSample *ca, const LweSample *cb, const LweSample *cc, int32_t count,
                          const TFheGateBootstrappingCloudKeySet *bk) {
    if (!result || !ca || !bk || count < 0) return THFHE_E_INVALID;
    if (count == 0) return THFHE_OK;
    run(op, result, ca, cb, cc, count, bk);
    return THFHE_OK;
}
void thfhe_tfhe_forget_key(const TFheGateBootstrappingCloudKeySet *bk) {
    // drops the cache's reference: the device context is destroyed now, or -- if calls on this key set are still running -- by the
    // last of them when it returns (thfhe_keyslot.h); later calls with this address build a fresh context
    g_cache.forget(bk);
}
}
