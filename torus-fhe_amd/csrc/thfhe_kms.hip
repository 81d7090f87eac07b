// thfhe_kms.hip -- the KMS multi-key scheme (mk_bootstrap_new / mk_gate_nand_new) on gfx950: device side.
//   reference: 3-gen-mk-tfhe/src/new_mk_internals.jl (mk_ith_blind_rotate :210-225, mk_mux_rotate_new :177-182, mk_lev_rlwe_mul :185-207,
//   UniProduct_new :85-127, mk_bootstrap_new :303-325), tlev.jl (TLev accumulators), mk_internals.jl:714-728 (mk_keyswitch).
//
// Where the time goes: per party, the TLev accumulator (l_lev RLWE samples over a Torus64 ring of degree 2048) is blind-rotated by the
// party's n TGSW-encrypted key bits -- n x l_lev CMuxes with 2 l_gsw digit rows each; everything after it (tlev_extern_mul, UniProduct_new:
// a few dozen polynomial products per party and gate) is less than 1 % of the arithmetic and goes through thfhe_pm_mac (thfhe_polymac.hip)
// under the host layer thfhe/kms.py.  This file holds
//   kms_tlev_rotate_kernel   one 512-thread workgroup per (gate, TLev sample): the N = 2048 cooperative structure of thfhe_mk.hip
//                            (digit rows transformed by up to six waves, (column, limb) multiply-accumulate + inverse on all eight,
//                            key chunks requested two ahead), generalised to what the reference's KMS sets need:
//                            * digits taken from the full 64-bit word (l_gsw Bgbit = 39 .. 48 bits);
//                            * digits wider than 10 bit are cut in two balanced parts d = d_lo + 2^w d_hi; the second part multiplies
//                              the key row shifted left by w bits (a second table, built once), so both parts accumulate into ONE set of
//                              spectra and every limb sum stays inside the FP64 exactness bound (12 x 2048 x 2^6 x 2^15 = 2^35.6);
//                            * any number of row parts: they pass through the LDS in batches of six.
//   thfhe_kms_keyswitch      mk_keyswitch: party p key-switches its own extracted mask (thfhe_mk_shared.h).
#include <hip/hip_runtime.h>

#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"
#include "thfhe_lane.h"
#include "thfhe_mk_shared.h"

using namespace thfhe;

namespace {

THFHE_STAMP_STORAGE

#include "thfhe_rot2k.h"

#include "thfhe_pm_kernels.h"

// ---- device-resident mk_bootstrap_new: everything between the gate's linear part and the key switch stays in HBM -----------------
// gate linear part (J/gates.jl, constants by opcode) + decode_message(., 2N) of every word + the accumulator X^{-barb} (mu, .., mu)
// as a trivial multi-key RLWE sample (J/new_mk_internals.jl:271-276).  bara is party-major: [P][G][n].
__global__ __launch_bounds__(256) void kms_prologue_kernel(const int32_t *__restrict__ x, const int32_t *__restrict__ y, int32_t cb, int32_t cx, int32_t cy,
                                                            int n, int P, long G, int64_t mu, int32_t *__restrict__ bara, int64_t *__restrict__ accum) {
    const long g = blockIdx.x;
    const int words = P * n + 1;
    __shared__ int s_barb;
    auto word = [&](int q) {
        uint32_t t = (uint32_t)cx * (uint32_t)x[g * words + q] + (y ? (uint32_t)cy * (uint32_t)y[g * words + q] : 0u);
        if (q == words - 1) t += (uint32_t)cb;
        return (int32_t)(t + (1u << 19)) >> 20;   // decode_message(t, 4096), J/numeric-functions.jl:70-73
    };
    for (int q = threadIdx.x; q < words; q += 256) {
        const int32_t v = word(q);
        if (q == words - 1) s_barb = v;
        else bara[((size_t)(q / n) * G + g) * n + (q % n)] = v;
    }
    __syncthreads();
    const int barb = s_barb;
    int64_t *acc = accum + (size_t)g * (P + 1) * 2048;
    for (int q = threadIdx.x; q < (P + 1) * 2048; q += 256) {
        const int i = q >> 11, t = q & 2047;
        acc[q] = i < P ? 0 : ((((t + barb) & 4095) >= 2048) ? -mu : mu);
    }
}
// decompose (J/tgsw.jl:112-138, 64-bit words): polys[index[j]] -> digits out[j][level][2048], level 1 (most significant) first
__global__ __launch_bounds__(256) void kms_decompose_kernel(const int64_t *__restrict__ polys, const int32_t *__restrict__ index, long n, int l, int bg,
                                                             int32_t *__restrict__ out) {
    const long j = blockIdx.x;
    const int q = blockIdx.y * 256 + threadIdx.x;
    uint64_t offset = 0;
    for (int p = 1; p <= l; p++) offset += (1ull << (bg - 1)) << (64 - p * bg);
    const uint64_t v = (uint64_t)polys[(size_t)(index ? index[j] : j) * 2048 + q] + offset;
    for (int p = 1; p <= l; p++) out[((size_t)j * l + (p - 1)) * 2048 + q] = (int32_t)((v >> (64 - p * bg)) & ((1ull << bg) - 1ull)) - (1 << (bg - 1));
}
// accum'[g][i] = (f - u)[g][pos[i]] (0 where polynomial i took no part), - w0 on the body, - w1 on the party's mask (J/new_mk_internals.jl:119-126,204-206)
__global__ __launch_bounds__(256) void kms_assemble_kernel(const int64_t *__restrict__ r, const int64_t *__restrict__ w01, const int32_t *__restrict__ pos, int ns,
                                                            int party, int P, int64_t *__restrict__ accum) {
    const long g = blockIdx.x;
    const int i = blockIdx.y;
    const int ps = pos[i];
    for (int q = threadIdx.x; q < 2048; q += 256) {
        uint64_t v = ps >= 0 ? (uint64_t)r[((size_t)g * ns + ps) * 2048 + q] : 0ull;
        if (i == P) v -= (uint64_t)w01[((size_t)g * 2 + 0) * 2048 + q];
        if (i == party) v -= (uint64_t)w01[((size_t)g * 2 + 1) * 2048 + q];
        accum[((size_t)g * (P + 1) + i) * 2048 + q] = (int64_t)v;
    }
}
// mk_rlwe_extract_sample_64 + t64tot32 (J/new_mk_internals.jl:294-299): u[g] = (a'_0 .. a'_{P-1}, b), a'_p[0] = a_p[0], a'_p[j] = -a_p[N - j]
__global__ __launch_bounds__(256) void kms_extract_kernel(const int64_t *__restrict__ accum, int P, int32_t *__restrict__ u) {
    const long g = blockIdx.x;
    const int p = blockIdx.y;
    const int64_t *a = accum + ((size_t)g * (P + 1) + p) * 2048;
    int32_t *dst = u + (size_t)g * (P * 2048 + 1) + (size_t)p * 2048;
    if (p == P) {
        if (threadIdx.x == 0) dst[0] = t64tot32(a[0]);
        return;
    }
    for (int q = threadIdx.x; q < 2048; q += 256) dst[q] = t64tot32(q == 0 ? a[0] : (int64_t)(0ull - (uint64_t)a[2048 - q]));
}
// fast_boot: the first party's RLWE sample (0, testvect) in, its rotated (mask, body) out as e = mask, f = body
__global__ __launch_bounds__(256) void kms_rlwe_init_kernel(const int64_t *__restrict__ accum, int P, int64_t *__restrict__ acc1) {
    const long g = blockIdx.x;
    for (int q = threadIdx.x; q < 2048; q += 256) {
        acc1[(size_t)g * 4096 + q] = 0;
        acc1[(size_t)g * 4096 + 2048 + q] = accum[((size_t)g * (P + 1) + P) * 2048 + q];
    }
}
__global__ __launch_bounds__(256) void kms_rlwe_split_kernel(const int64_t *__restrict__ acc1, long G, int64_t *__restrict__ ef) {
    const long g = blockIdx.x;
    for (int q = threadIdx.x; q < 2048; q += 256) {
        ef[(size_t)g * 2048 + q] = acc1[(size_t)g * 4096 + q];                  // e block [G][1][N]
        ef[((size_t)G + g) * 2048 + q] = acc1[(size_t)g * 4096 + 2048 + q];   // f block
    }
}
}  // namespace

struct thfhe_kms_ctx {
    thfhe_kms_params p;
    int device = 0;
    hipStream_t stream = nullptr;      // the stream every call enqueues on
    hipStream_t own_stream = nullptr;  // created with the context; `stream` differs only after thfhe_kms_set_stream
    cplx *d_tw = nullptr;
    Rot2kPark park;              // two jobs per workgroup: partial spectra between row-part batches (thfhe_rot2k.h)
    long pair_threshold = 256;   // launches of more TLev / RLWE rotations than this (one per CU) run two jobs per workgroup
    cplx *d_bk = nullptr;       // [party][j][row part][o][h][half][512]
    int32_t *d_ksk = nullptr;
    int parts = 1, lo_bits = 1, row_words = 0;
    size_t party_stride = 0;    // complex elements per party in d_bk
    void *d_buf[3] = {nullptr, nullptr, nullptr};
    size_t cap[3] = {0, 0, 0};
    // relinearisation keys as limb spectra: [party][d | f0 | f1][l_uni], then pk [P][l_uni], then crs [l_uni]   (thfhe_kms_set_relin_keys)
    cplx *d_relin = nullptr;
    int *d_flag = nullptr;
    enum { W_X, W_Y, W_BARA, W_ACCUM, W_LEV, W_LEVSPEC, W_SMALL, W_EF, W_R, W_V, W_W01, W_TERMS, W_FIRST, W_INDEX, W_U, W_OUT, W_ACC1, W_COUNT };
    void *d_w[W_COUNT] = {};
    size_t cap_w[W_COUNT] = {};
    // device-resident copies of the relinearisation index tables (terms / first of thfhe_pm_mac, decompose index lists, assemble positions):
    // they depend only on (table kind, party, gates per call, route), so a steady stream of equally sized calls uploads them once
    struct DevTab {
        int32_t *d = nullptr;
        size_t words = 0;
    };
    std::map<uint64_t, DevTab> tabs;
    size_t tab_bytes = 0;
    std::mutex mu;
};

namespace {
int kms_ensure(thfhe_kms_ctx *c, int slot, size_t bytes) {
    if (bytes <= c->cap[slot]) return THFHE_OK;
    (void)hipFree(c->d_buf[slot]);
    c->d_buf[slot] = nullptr;
    c->cap[slot] = 0;
    THFHE_HIP(hipMalloc(&c->d_buf[slot], bytes));
    c->cap[slot] = bytes;
    return THFHE_OK;
}
}  // namespace

extern "C" {

int thfhe_kms_ctx_create(const thfhe_kms_params *p, const int64_t *gsw, const int32_t *ksk, int device, thfhe_kms_ctx **out) {
    if (!p || !gsw || !ksk || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    *out = nullptr;
    if (p->N != 2048) return thfhe_fail(THFHE_E_UNSUPPORTED, "the KMS scheme is implemented for its reference ring degree N = 2048 (Torus64)");
    if (p->parties < 1 || p->n < 1 || p->n > 767) return thfhe_fail(THFHE_E_UNSUPPORTED, "need parties >= 1, 1 <= n <= 767");
    if (p->l_gsw < 1 || p->l_gsw > 8 || p->bg_gsw < 2 || p->bg_gsw > 14 || p->l_gsw * p->bg_gsw > 64)
        return thfhe_fail(THFHE_E_UNSUPPORTED, "gsw gadget: need 1 <= l <= 8, 2 <= Bgbit <= 14, l * Bgbit <= 64");
    if (p->l_lev < 1 || p->l_lev > 8 || p->bg_lev < 1 || p->l_lev * p->bg_lev > 64) return thfhe_fail(THFHE_E_UNSUPPORTED, "bad lev gadget");
    if (p->ks_t < 1 || p->ks_basebit < 1 || p->ks_t * p->ks_basebit > 31) return thfhe_fail(THFHE_E_INVALID, "bad key-switch parameters");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return thfhe_fail(THFHE_E_NO_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    THFHE_HIP(hipSetDevice(device));
    thfhe_kms_ctx *c = new (std::nothrow) thfhe_kms_ctx;
    if (!c) return thfhe_fail(THFHE_E_NOMEM, "out of host memory");
    c->p = *p;
    c->device = device;
    // exactness: RP * N * 2^(part width - 1) * 2^15 must stay below 2^37 (N = 2048 bound of DESIGN.md section 4.3).  Digits are cut in two
    // balanced parts when they are wider than 10 bit OR when the whole-digit sum would leave the bound (many rows: the 16-party set's
    // l = 5, Bgbit 9 gives 2^37.3)
    const int N = 2048;
    auto sum_bound = [&](int parts, int bits) { return (double)(2 * p->l_gsw * parts) * N * (double)(1 << (bits - 1)) * 32768.0; };
    c->lo_bits = (p->bg_gsw + 1) / 2;
    c->parts = (p->bg_gsw > 10 || sum_bound(1, p->bg_gsw) > 137438953472.0) ? 2 : 1;
    c->row_words = 128 * ((p->n + 1 + 127) / 128);
    const int RP = 2 * p->l_gsw * c->parts;
    const int part_bits = c->parts == 2 ? c->lo_bits : p->bg_gsw;
    if (sum_bound(c->parts, part_bits) > 137438953472.0 /* 2^37 */) {
        delete c;
        return thfhe_fail(THFHE_E_UNSUPPORTED, "gsw gadget outside the FP64 exactness bound of the N = 2048 transform");
    }
    int64_t *d_coeff = nullptr;
    int32_t *d_raw = nullptr;
    auto fail = [&](int code) {
        (void)hipFree(d_coeff);
        (void)hipFree(d_raw);
        thfhe_kms_ctx_destroy(c);
        return code;
    };
#define CK(expr)                                                      \
    do {                                                              \
        hipError_t e_ = (expr);                                       \
        if (e_ != hipSuccess) return fail(thfhe_fail_hip(e_, #expr)); \
    } while (0)
    CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    std::vector<cplx> tw(1216 + 64), unused(512);   // T1 (twist 1), T1 (twist 5), T2; [1216..): pass-1 ratio of the table-free transforms
    make_lane_ratio_2048(tw.data() + 1216);
    make_twiddles_2048(tw.data(), tw.data() + 512);
    make_twiddles_1024(unused.data(), tw.data() + 1024);
    CK(hipMalloc(&c->d_tw, tw.size() * sizeof(cplx)));
    CK(hipMemcpyAsync(c->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream));
    // key table: per party and key bit the 2 l rows x 2 columns of the TGSW sample; with two-part digits every row is followed by its copy
    // shifted left by lo_bits (wrapping): d (*) K = d_lo (*) K + d_hi (*) (K << lo_bits)
    const size_t polys_per_party = (size_t)p->n * RP * 2;
    c->party_stride = polys_per_party * 4 * 1024;
    CK(hipMalloc(&c->d_bk, (size_t)p->parties * c->party_stride * sizeof(cplx)));
    CK(hipMalloc(&d_coeff, polys_per_party * N * sizeof(int64_t)));
    std::vector<int64_t> host(polys_per_party * N);
    for (int q = 0; q < p->parties; q++) {
        for (int j = 0; j < p->n; j++)
            for (int r = 0; r < 2 * p->l_gsw; r++)
                for (int part = 0; part < c->parts; part++)
                    for (int col = 0; col < 2; col++) {
                        const int64_t *src = gsw + ((((size_t)q * p->n + j) * 2 * p->l_gsw + r) * 2 + col) * N;
                        int64_t *dst = host.data() + ((((size_t)j * RP + (size_t)r * c->parts + part) * 2) + col) * N;
                        const int sh = part * c->lo_bits;
                        for (int t = 0; t < N; t++) dst[t] = (int64_t)((uint64_t)src[t] << sh);
                    }
        CK(hipMemcpyAsync(d_coeff, host.data(), host.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(kms_key_transform_kernel, dim3((unsigned)((polys_per_party * 4 + 3) / 4)), dim3(256), 0, c->stream, d_coeff, (long)polys_per_party,
                           c->d_tw, c->d_bk + (size_t)q * c->party_stride);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(c->stream));   // `host` is reused for the next party
    }
    const long rows = (long)p->parties * N * p->ks_t * ((1 << p->ks_basebit) - 1);
    CK(hipMalloc(&d_raw, (size_t)rows * (p->n + 1) * sizeof(int32_t)));
    CK(hipMemcpyAsync(d_raw, ksk, (size_t)rows * (p->n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CK(hipMalloc(&c->d_ksk, (size_t)rows * c->row_words * sizeof(int32_t)));
    hipLaunchKernelGGL(mk_ksk_pad_kernel, dim3((unsigned)rows), dim3(256), 0, c->stream, d_raw, rows, p->n, c->row_words, c->d_ksk);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c->stream));
    (void)hipFree(d_coeff);
    (void)hipFree(d_raw);
#undef CK
    *out = c;
    return THFHE_OK;
}

void thfhe_kms_ctx_destroy(thfhe_kms_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    (void)hipFree(c->d_tw);
    (void)hipFree(c->park.buf);
    (void)hipFree(c->d_bk);
    (void)hipFree(c->d_ksk);
    for (auto &q : c->d_buf) (void)hipFree(q);
    for (auto &q : c->d_w) (void)hipFree(q);
    for (auto &kv : c->tabs) (void)hipFree(kv.second.d);
    (void)hipFree(c->d_relin);
    (void)hipFree(c->d_flag);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int thfhe_kms_tlev_rotate(thfhe_kms_ctx *c, int party, const int32_t *bara, int64_t *lev, size_t count) {
    if (!c || !bara || !lev) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (party < 0 || party >= c->p.parties) return thfhe_fail(THFHE_E_INVALID, "party out of range");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t jobs = count * c->p.l_lev;
    int rc = kms_ensure(c, 0, count * c->p.n * sizeof(int32_t));
    if (!rc) rc = kms_ensure(c, 1, jobs * 4096 * sizeof(int64_t));
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_buf[0], bara, count * c->p.n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    KmsBRArgs a{c->d_bk + (size_t)party * c->party_stride, c->d_tw, (const int32_t *)c->d_buf[0], (int64_t *)c->d_buf[1], nullptr, (long)jobs,
                c->p.n, c->p.l_gsw, c->p.bg_gsw, c->parts, c->lo_bits, c->p.l_lev, c->p.bg_lev, c->p.n};
    {
        int rc2 = rot2k_launch(a, c->stream, c->pair_threshold, c->park);
        if (rc2) return rc2;
    }
    THFHE_HIP(hipMemcpyAsync(lev, c->d_buf[1], jobs * 4096 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

int thfhe_kms_rlwe_rotate(thfhe_kms_ctx *c, int party, const int32_t *bara, int64_t *acc, size_t count) {
    if (!c || !bara || !acc) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (party < 0 || party >= c->p.parties) return thfhe_fail(THFHE_E_INVALID, "party out of range");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    int rc = kms_ensure(c, 0, count * c->p.n * sizeof(int32_t));
    if (!rc) rc = kms_ensure(c, 1, count * 4096 * sizeof(int64_t));
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_buf[0], bara, count * c->p.n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_buf[1], acc, count * 4096 * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    KmsBRArgs a{c->d_bk + (size_t)party * c->party_stride, c->d_tw, (const int32_t *)c->d_buf[0], (int64_t *)c->d_buf[1], (const int64_t *)c->d_buf[1],
                (long)count, c->p.n, c->p.l_gsw, c->p.bg_gsw, c->parts, c->lo_bits, 1, c->p.bg_lev, c->p.n};   // in place: a workgroup reads its sample before it writes it
    {
        int rc2 = rot2k_launch(a, c->stream, c->pair_threshold, c->park);
        if (rc2) return rc2;
    }
    THFHE_HIP(hipMemcpyAsync(acc, c->d_buf[1], count * 4096 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

int thfhe_kms_keyswitch(thfhe_kms_ctx *c, const int32_t *u, int32_t *out, size_t count) {
    if (!c || !u || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const int P = c->p.parties, N = c->p.N, n = c->p.n;
    const size_t in_words = count * ((size_t)P * N + 1), out_words = count * ((size_t)P * n + 1);
    int rc = kms_ensure(c, 1, in_words * sizeof(int32_t));
    if (!rc) rc = kms_ensure(c, 2, out_words * sizeof(int32_t));
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_buf[1], u, in_words * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemsetAsync(c->d_buf[2], 0, out_words * sizeof(int32_t), c->stream));
    MKKSArgs k{c->d_ksk, (const int32_t *)c->d_buf[1], (int32_t *)c->d_buf[2], (long)count, n, c->p.ks_t, c->p.ks_basebit, P, c->row_words, N, P * N + 1, N};
    const int nsplit = count <= 64 ? 8 : 2;
    mk_launch_keyswitch(k, nsplit, c->stream);
    THFHE_HIP(hipGetLastError());
    THFHE_HIP(hipMemcpyAsync(out, c->d_buf[2], out_words * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

}  // extern "C"

namespace {
int kms_w(thfhe_kms_ctx *c, int slot, size_t bytes) {
    if (bytes <= c->cap_w[slot]) return THFHE_OK;
    (void)hipFree(c->d_w[slot]);
    c->d_w[slot] = nullptr;
    c->cap_w[slot] = 0;
    THFHE_HIP(hipMalloc(&c->d_w[slot], bytes));
    c->cap_w[slot] = bytes;
    return THFHE_OK;
}
// host-built tables of one call stay alive until the stream has been synchronised (asynchronous copies read them)
struct KmsTables {
    std::deque<std::vector<int32_t>> keep;   // a deque: references to earlier tables stay valid when one is added
    std::vector<int32_t> &add() {
        keep.emplace_back();
        return keep.back();
    }
};
// A table that depends only on its key (kind, party, gates, variant): built on the host by `build` and uploaded the first time, then served
// from the device copy.  The cache is dropped wholesale when it passes 256 MB (the stream is drained first: queued kernels read it).
constexpr uint64_t kms_key(int kind, int party, size_t G, int extra) {
    return ((uint64_t)kind << 56) ^ ((uint64_t)(party & 0xFF) << 48) ^ ((uint64_t)(extra & 0xFF) << 40) ^ (uint64_t)(G & 0xFFFFFFFFFFull);
}
int kms_tab(thfhe_kms_ctx *c, KmsTables &tabs, uint64_t key, const std::function<void(std::vector<int32_t> &)> &build, const int32_t **d_out, size_t *words_out) {
    auto it = c->tabs.find(key);
    if (it == c->tabs.end()) {
        std::vector<int32_t> &host = tabs.add();   // stays alive until the stream has been synchronised (asynchronous copy)
        build(host);
        if (c->tab_bytes + host.size() * 4 > (size_t)256 << 20) {
            THFHE_HIP(hipStreamSynchronize(c->stream));
            for (auto &kv : c->tabs) (void)hipFree(kv.second.d);
            c->tabs.clear();
            c->tab_bytes = 0;
        }
        thfhe_kms_ctx::DevTab t;
        t.words = host.size();
        THFHE_HIP(hipMalloc(&t.d, (t.words ? t.words : 1) * 4));
        if (t.words) THFHE_HIP(hipMemcpyAsync(t.d, host.data(), t.words * 4, hipMemcpyHostToDevice, c->stream));
        c->tab_bytes += t.words * 4;
        it = c->tabs.emplace(key, t).first;
    }
    *d_out = it->second.d;
    if (words_out) *words_out = it->second.words;
    return THFHE_OK;
}
// out[j] = addend[j] + sum_terms sign * small[s] (*) spec[t], all operands device-resident; terms = (out, small, torus, sign), ascending in out.
// The table [terms (4 words each) | first (n_out + 1 words)] comes from the cache under `key`; `build_terms` fills the term list on a miss.
int kms_mac(thfhe_kms_ctx *c, KmsTables &tabs, uint64_t key, const int32_t *d_small, const cplx *d_spec, const std::function<void(std::vector<int32_t> &)> &build_terms,
            size_t n_out, const void *d_addend, void *d_out) {
    const int32_t *d_tab = nullptr;
    size_t words = 0;
    int rc = kms_tab(c, tabs, key, [&](std::vector<int32_t> &t) {
        build_terms(t);
        const size_t n_terms = t.size() / 4;
        std::vector<int32_t> first(n_out + 1, 0);
        for (size_t q = 0; q < n_terms; q++) first[t[4 * q] + 1]++;
        for (size_t j = 0; j < n_out; j++) first[j + 1] += first[j];
        t.insert(t.end(), first.begin(), first.end());
    }, &d_tab, &words);
    if (rc) return rc;
    const size_t n_terms = (words - (n_out + 1)) / 4;
    PMArgs a{d_small, d_spec, d_tab, d_tab + 4 * n_terms, d_addend, d_out, c->d_tw, (long)n_out, c->d_flag};
    hipLaunchKernelGGL((pm_mac_kernel<2048, 64>), dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, c->stream, a);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}
int kms_decompose(thfhe_kms_ctx *c, const int64_t *d_polys, const int32_t *d_index, size_t n, int l, int bg) {
    int rc = kms_w(c, thfhe_kms_ctx::W_SMALL, n * l * 2048 * sizeof(int32_t));
    if (rc) return rc;
    hipLaunchKernelGGL(kms_decompose_kernel, dim3((unsigned)n, 8), dim3(256), 0, c->stream, d_polys, d_index, (long)n, l, bg, (int32_t *)c->d_w[thfhe_kms_ctx::W_SMALL]);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}
// UniProduct_new on e, accum' = f - (u, u0 + w0, a_party += w1)   (J/new_mk_internals.jl:85-127, 204-206).  d_ef = e block [G][ns][N] followed
// by the f block; src[q] = which polynomial of the multi-key sample (0 .. P-1 masks, P body) row q is.  `route` (0: after a TLev product,
// 1: the fast_boot start) keys the cached index tables: the same (party, G) has a different src in the two routes.
int kms_relin_core(thfhe_kms_ctx *c, KmsTables &tabs, int party, size_t G, const std::vector<int> &src, int64_t *d_accum, int route) {
    typedef thfhe_kms_ctx K;
    const int P = c->p.parties, lu = c->p.l_uni, ns = (int)src.size();
    const size_t N = 2048;
    const int64_t *d_e = (const int64_t *)c->d_w[K::W_EF], *d_f = d_e + G * ns * N;
    auto T_D = [&](int l) { return (party * 3 + 0) * lu + l; };
    auto T_F = [&](int w, int l) { return (party * 3 + 1 + w) * lu + l; };
    auto T_PK = [&](int i, int l) { return P * 3 * lu + i * lu + l; };
    auto T_A = [&](int l) { return P * 3 * lu + P * lu + l; };
    int rc = kms_decompose(c, d_e, nullptr, G * ns, lu, c->p.bg_uni);
    if (!rc) rc = kms_w(c, K::W_R, G * ns * N * 8);
    if (!rc) rc = kms_w(c, K::W_V, G * N * 8);
    if (!rc) rc = kms_w(c, K::W_W01, G * 2 * N * 8);
    if (rc) return rc;
    const int32_t *d_small = (const int32_t *)c->d_w[K::W_SMALL];
    rc = kms_mac(c, tabs, kms_key(1, party, G, route), d_small, c->d_relin, [&](std::vector<int32_t> &t) {
        for (size_t g = 0; g < G; g++)
            for (int q = 0; q < ns; q++)
                for (int l = 0; l < lu; l++) {
                    const int32_t j = (int32_t)(g * ns + q);
                    t.insert(t.end(), {j, j * lu + l, T_D(l), -1});          // (f - u)_q = f_q - sum_l dec(e_q)[l] (*) d[l]
                }
    }, G * ns, d_f, c->d_w[K::W_R]);
    if (!rc) rc = kms_mac(c, tabs, kms_key(2, party, G, route), d_small, c->d_relin, [&](std::vector<int32_t> &t) {
        for (size_t g = 0; g < G; g++)
            for (int q = 0; q < ns; q++)
                for (int l = 0; l < lu; l++) {
                    const int32_t sm = (int32_t)((g * ns + q) * lu + l);
                    if (src[q] < P) t.insert(t.end(), {(int32_t)g, sm, T_PK(src[q], l), 1});   // v = sum_i <dec(e_i), pk_i> - <dec(e_b), crs>
                    else t.insert(t.end(), {(int32_t)g, sm, T_A(l), -1});
                }
    }, G, nullptr, c->d_w[K::W_V]);
    if (!rc) rc = kms_decompose(c, (const int64_t *)c->d_w[K::W_V], nullptr, G, lu, c->p.bg_uni);   // stream order: after the two products read W_SMALL
    if (rc) return rc;
    rc = kms_mac(c, tabs, kms_key(3, party, G, 0), (const int32_t *)c->d_w[K::W_SMALL], c->d_relin, [&](std::vector<int32_t> &t) {
        for (size_t g = 0; g < G; g++)
            for (int w = 0; w < 2; w++)
                for (int l = 0; l < lu; l++) t.insert(t.end(), {(int32_t)(g * 2 + w), (int32_t)(g * lu + l), T_F(w, l), 1});   // w0 = <dec(v), f0>, w1 = <dec(v), f1>
    }, G * 2, nullptr, c->d_w[K::W_W01]);
    if (rc) return rc;
    const int32_t *d_pos = nullptr;
    rc = kms_tab(c, tabs, kms_key(4, party, 0, route), [&](std::vector<int32_t> &pos) {
        pos.assign(64, -1);
        for (int q = 0; q < ns; q++) pos[src[q]] = q;
    }, &d_pos, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(kms_assemble_kernel, dim3((unsigned)G, (unsigned)(P + 1)), dim3(256), 0, c->stream, (const int64_t *)c->d_w[K::W_R],
                       (const int64_t *)c->d_w[K::W_W01], d_pos, ns, party, P, d_accum);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}
// mk_lev_rlwe_mul (J/new_mk_internals.jl:185-207): (e, f) = tlev_extern_mul of a_0 .. a_{party-1} and b with the TLev sample, then the core
int kms_lev_rlwe_mul_dev(thfhe_kms_ctx *c, KmsTables &tabs, int party, size_t G, int64_t *d_accum, const int64_t *d_lev) {
    typedef thfhe_kms_ctx K;
    const int P = c->p.parties, lv = c->p.l_lev;
    const size_t N = 2048;
    std::vector<int> src;
    for (int i = 0; i < party; i++) src.push_back(i);
    src.push_back(P);
    const int ns = (int)src.size();
    const int32_t *d_index = nullptr;
    int rc = kms_tab(c, tabs, kms_key(5, party, G, 0), [&](std::vector<int32_t> &index) {
        for (size_t g = 0; g < G; g++)
            for (int q = 0; q < ns; q++) index.push_back((int32_t)(g * (P + 1) + src[q]));
    }, &d_index, nullptr);
    if (!rc) rc = kms_decompose(c, d_accum, d_index, G * ns, lv, c->p.bg_lev);
    if (!rc) rc = kms_w(c, K::W_LEVSPEC, G * lv * 2 * 4 * 1024 * sizeof(cplx));
    if (!rc) rc = kms_w(c, K::W_EF, 2 * G * ns * N * 8);
    if (rc) return rc;
    hipLaunchKernelGGL((pm_torus_transform_kernel<2048, 64>), dim3((unsigned)((G * lv * 2 * 4 + 3) / 4)), dim3(256), 0, c->stream, (const void *)d_lev,
                       (long)(G * lv * 2), c->d_tw, (cplx *)c->d_w[K::W_LEVSPEC]);
    THFHE_HIP(hipGetLastError());
    rc = kms_mac(c, tabs, kms_key(6, party, G, 0), (const int32_t *)c->d_w[K::W_SMALL], (const cplx *)c->d_w[K::W_LEVSPEC], [&](std::vector<int32_t> &t1) {
        for (int w = 0; w < 2; w++)   // e block (w = 0: masks of the TLev samples), then f block (w = 1: bodies)
            for (size_t g = 0; g < G; g++)
                for (int q = 0; q < ns; q++)
                    for (int s = 0; s < lv; s++)
                        t1.insert(t1.end(), {(int32_t)((w * G + g) * ns + q), (int32_t)((g * ns + q) * lv + s), (int32_t)((g * lv + s) * 2 + w), 1});
    }, 2 * G * ns, nullptr, c->d_w[K::W_EF]);
    if (rc) return rc;
    return kms_relin_core(c, tabs, party, G, src, d_accum, 0);
}
int kms_launch_rotation(thfhe_kms_ctx *c, int party, const int32_t *d_bara, int64_t *d_out, const int64_t *d_in, size_t gates, int l_lev) {
    KmsBRArgs a{c->d_bk + (size_t)party * c->party_stride, c->d_tw, d_bara, d_out, d_in, (long)(gates * l_lev),
                c->p.n, c->p.l_gsw, c->p.bg_gsw, c->parts, c->lo_bits, l_lev, c->p.bg_lev, c->p.n};
    return rot2k_launch(a, c->stream, c->pair_threshold, c->park);
}
int kms_finish(thfhe_kms_ctx *c) {
    int flag = 0;
    THFHE_HIP(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    if (flag) return thfhe_fail(THFHE_E_UNSUPPORTED, "a gadget digit left [-4096, 4096] (outside the FP64 exactness bound of the relinearisation products)");
    return THFHE_OK;
}
// the whole gate / bootstrap on device buffers; x (and y) int32[G][P n + 1] on the host
int kms_bootstrap_body(thfhe_kms_ctx *c, KmsTables &tabs, int32_t cb, int32_t cx, int32_t cy, int64_t mu, const int32_t *x, const int32_t *y, int32_t *u_out,
                       int32_t *out, size_t G, int fast_boot) {
    typedef thfhe_kms_ctx K;
    if (!c->d_relin) return thfhe_fail(THFHE_E_INVALID, "thfhe_kms_set_relin_keys has not been called on this context");
    const int P = c->p.parties, n = c->p.n, lv = c->p.l_lev;
    const size_t N = 2048, words = (size_t)P * n + 1, uw = (size_t)P * N + 1;
    THFHE_HIP(hipSetDevice(c->device));
    int rc = kms_w(c, K::W_X, G * words * 4);
    if (!rc && y) rc = kms_w(c, K::W_Y, G * words * 4);
    if (!rc) rc = kms_w(c, K::W_BARA, (size_t)P * G * n * 4);
    if (!rc) rc = kms_w(c, K::W_ACCUM, G * (P + 1) * N * 8);
    if (!rc) rc = kms_w(c, K::W_LEV, G * lv * 2 * N * 8);
    if (!rc) rc = kms_w(c, K::W_U, G * uw * 4);
    if (!rc) rc = kms_w(c, K::W_OUT, G * words * 4);
    if (rc) return rc;
    THFHE_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_w[K::W_X], x, G * words * 4, hipMemcpyHostToDevice, c->stream));
    if (y) THFHE_HIP(hipMemcpyAsync(c->d_w[K::W_Y], y, G * words * 4, hipMemcpyHostToDevice, c->stream));
    int64_t *d_accum = (int64_t *)c->d_w[K::W_ACCUM];
    const int32_t *d_bara = (const int32_t *)c->d_w[K::W_BARA];
    hipLaunchKernelGGL(kms_prologue_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, (const int32_t *)c->d_w[K::W_X], y ? (const int32_t *)c->d_w[K::W_Y] : nullptr,
                       cb, cx, cy, n, P, (long)G, mu, (int32_t *)c->d_w[K::W_BARA], d_accum);
    THFHE_HIP(hipGetLastError());
    int first = 0;
    if (fast_boot) {   // mk_blind_rotate_new_v2 (J/new_mk_internals.jl:255-269)
        rc = kms_w(c, K::W_ACC1, G * 2 * N * 8);
        if (!rc) rc = kms_w(c, K::W_EF, 2 * G * N * 8);
        if (rc) return rc;
        hipLaunchKernelGGL(kms_rlwe_init_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, (const int64_t *)d_accum, P, (int64_t *)c->d_w[K::W_ACC1]);
        rc = kms_launch_rotation(c, 0, d_bara, (int64_t *)c->d_w[K::W_ACC1], (const int64_t *)c->d_w[K::W_ACC1], G, 1);
        if (rc) return rc;
        hipLaunchKernelGGL(kms_rlwe_split_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, (const int64_t *)c->d_w[K::W_ACC1], (long)G, (int64_t *)c->d_w[K::W_EF]);
        THFHE_HIP(hipGetLastError());
        rc = kms_relin_core(c, tabs, 0, G, std::vector<int>{P}, d_accum, 1);
        if (rc) return rc;
        first = 1;
    }
    for (int party = first; party < P; party++) {
        rc = kms_launch_rotation(c, party, d_bara + (size_t)party * G * n, (int64_t *)c->d_w[K::W_LEV], nullptr, G, lv);
        if (!rc) rc = kms_lev_rlwe_mul_dev(c, tabs, party, G, d_accum, (const int64_t *)c->d_w[K::W_LEV]);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kms_extract_kernel, dim3((unsigned)G, (unsigned)(P + 1)), dim3(256), 0, c->stream, (const int64_t *)d_accum, P, (int32_t *)c->d_w[K::W_U]);
    THFHE_HIP(hipGetLastError());
    if (u_out) THFHE_HIP(hipMemcpyAsync(u_out, c->d_w[K::W_U], G * uw * 4, hipMemcpyDeviceToHost, c->stream));
    if (out) {
        THFHE_HIP(hipMemsetAsync(c->d_w[K::W_OUT], 0, G * words * 4, c->stream));
        MKKSArgs k{c->d_ksk, (const int32_t *)c->d_w[K::W_U], (int32_t *)c->d_w[K::W_OUT], (long)G, n, c->p.ks_t, c->p.ks_basebit, P, c->row_words, (int)N,
                   (int)uw, (int)N};
        const int nsplit = G <= 64 ? 8 : 2;
        mk_launch_keyswitch(k, nsplit, c->stream);
        THFHE_HIP(hipGetLastError());
        THFHE_HIP(hipMemcpyAsync(out, c->d_w[K::W_OUT], G * words * 4, hipMemcpyDeviceToHost, c->stream));
    }
    return kms_finish(c);
}
// Every failure leaves through here: asynchronous copies still queued read the host tables in `tabs` and write the caller's buffers,
// so the stream is drained before `tabs` dies and before the call reports failure.
int kms_bootstrap_impl(thfhe_kms_ctx *c, int32_t cb, int32_t cx, int32_t cy, int64_t mu, const int32_t *x, const int32_t *y, int32_t *u_out, int32_t *out,
                       size_t G, int fast_boot) {
    std::lock_guard<std::mutex> lock(c->mu);
    KmsTables tabs;
    const int rc = kms_bootstrap_body(c, tabs, cb, cx, cy, mu, x, y, u_out, out, G, fast_boot);
    if (rc) (void)hipStreamSynchronize(c->stream);
    return rc;
}
int kms_lev_rlwe_mul_body(thfhe_kms_ctx *c, KmsTables &tabs, int party, int64_t *accum, const int64_t *lev, size_t count) {
    typedef thfhe_kms_ctx K;
    if (!c->d_relin) return thfhe_fail(THFHE_E_INVALID, "thfhe_kms_set_relin_keys has not been called on this context");
    THFHE_HIP(hipSetDevice(c->device));
    const size_t N = 2048, ab = count * (c->p.parties + 1) * N * 8, lb = count * c->p.l_lev * 2 * N * 8;
    int rc = kms_w(c, K::W_ACCUM, ab);
    if (!rc) rc = kms_w(c, K::W_LEV, lb);
    if (rc) return rc;
    THFHE_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_w[K::W_ACCUM], accum, ab, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_w[K::W_LEV], lev, lb, hipMemcpyHostToDevice, c->stream));
    rc = kms_lev_rlwe_mul_dev(c, tabs, party, count, (int64_t *)c->d_w[K::W_ACCUM], (const int64_t *)c->d_w[K::W_LEV]);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(accum, c->d_w[K::W_ACCUM], ab, hipMemcpyDeviceToHost, c->stream));
    return kms_finish(c);
}
}  // namespace

extern "C" {

int thfhe_kms_set_relin_keys(thfhe_kms_ctx *c, const int64_t *uni, const int64_t *pk, const int64_t *crs) {
    if (!c || !uni || !pk || !crs) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (c->p.l_uni < 1 || c->p.l_uni > 16 || c->p.bg_uni < 1 || c->p.bg_uni > 13 || c->p.l_uni * c->p.bg_uni > 64 || c->p.bg_lev > 13 || c->p.parties > 62)
        return thfhe_fail(THFHE_E_UNSUPPORTED, "relinearisation gadgets: need Bgbit <= 13 (digits inside [-4096, 4096]), l_uni <= 16, l * Bgbit <= 64");
    std::lock_guard<std::mutex> lock(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const int P = c->p.parties, lu = c->p.l_uni;
    const size_t N = 2048, n_polys = (size_t)P * 3 * lu + (size_t)P * lu + lu;
    void *d_raw = nullptr;
    THFHE_HIP(hipMalloc(&d_raw, n_polys * N * 8));
    hipError_t e = hipMemcpyAsync(d_raw, uni, (size_t)P * 3 * lu * N * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((char *)d_raw + (size_t)P * 3 * lu * N * 8, pk, (size_t)P * lu * N * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((char *)d_raw + (size_t)P * 4 * lu * N * 8, crs, (size_t)lu * N * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && !c->d_relin) e = hipMalloc(&c->d_relin, n_polys * 4 * 1024 * sizeof(cplx));
    if (e == hipSuccess && !c->d_flag) e = hipMalloc(&c->d_flag, sizeof(int));
    if (e == hipSuccess) {
        hipLaunchKernelGGL((pm_torus_transform_kernel<2048, 64>), dim3((unsigned)((n_polys * 4 + 3) / 4)), dim3(256), 0, c->stream, (const void *)d_raw, (long)n_polys,
                           c->d_tw, c->d_relin);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_raw);
    if (e != hipSuccess) return thfhe_fail_hip(e, "thfhe_kms_set_relin_keys");
    return THFHE_OK;
}

int thfhe_kms_lev_rlwe_mul(thfhe_kms_ctx *c, int party, int64_t *accum, const int64_t *lev, size_t count) {
    if (!c || !accum || !lev) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (party < 0 || party >= c->p.parties) return thfhe_fail(THFHE_E_INVALID, "party out of range");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> lock(c->mu);
    KmsTables tabs;
    const int rc = kms_lev_rlwe_mul_body(c, tabs, party, accum, lev, count);
    if (rc) (void)hipStreamSynchronize(c->stream);   // queued copies still read `tabs` / write the caller's buffer
    return rc;
}

int thfhe_kms_bootstrap(thfhe_kms_ctx *c, int64_t mu, const int32_t *x, int32_t *u, int32_t *out, size_t count, int fast_boot) {
    if (!c || !x || (!u && !out)) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    return kms_bootstrap_impl(c, 0, 1, 0, mu, x, nullptr, u, out, count, fast_boot);
}

// J/gates.jl:15-161, (constant, coefficient of x, coefficient of y) by opcode THFHE_NAND .. THFHE_ORYN; mu = 1/8 on Torus64
static const int32_t kKmsLin[10][3] = {{1 << 29, -1, -1}, {1 << 29, 1, 1}, {-(1 << 29), 1, 1}, {1 << 30, 2, 2}, {-(1 << 30), -2, -2},
                                       {-(1 << 29), -1, -1}, {-(1 << 29), -1, 1}, {-(1 << 29), 1, -1}, {1 << 29, -1, 1}, {1 << 29, 1, -1}};

int thfhe_kms_gates(thfhe_kms_ctx *c, int op, const int32_t *x, const int32_t *y, int32_t *out, size_t count, int fast_boot) {
    if (!c || !x || !y || !out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (op < 0 || op > 9) return thfhe_fail(THFHE_E_UNSUPPORTED, "the KMS scheme evaluates two-input bootstrapped gates (opcodes NAND .. ORYN)");
    if (count == 0) return THFHE_OK;
    return kms_bootstrap_impl(c, kKmsLin[op][0], kKmsLin[op][1], kKmsLin[op][2], (int64_t)1 << 61, x, y, nullptr, out, count, fast_boot);
}

// ---- party-sharded mode, device-resident (thfhe/kms_sharded.py; SURVEY.md section 8e side note) -----------------------------------------------
// The per-party TLev rotations read nothing the relinearisation writes (J/new_mk_internals.jl:241-252), so ranks rotate disjoint blocks of parties
// and exchange the TLev accumulators once.  All pointers are DEVICE pointers; both calls enqueue on the context's stream and return at once
// (rotate) / after the digit-range check (finish).  op = opcode NAND .. ORYN, or -1: plain mk_bootstrap_new of x with mu = 1/8.
static int kms_lin_of(int op, int32_t &cb, int32_t &cx, int32_t &cy) {
    if (op == -1) {
        cb = 0, cx = 1, cy = 0;
        return THFHE_OK;
    }
    if (op < 0 || op > 9) return thfhe_fail(THFHE_E_UNSUPPORTED, "the KMS scheme evaluates two-input bootstrapped gates (opcodes NAND .. ORYN)");
    cb = kKmsLin[op][0], cx = kKmsLin[op][1], cy = kKmsLin[op][2];
    return THFHE_OK;
}
// phase 1: gate linear part + mod-switch (J/numeric-functions.jl:70-73), then mk_ith_blind_rotate (J/new_mk_internals.jl:210-225) for parties
// [first_party, first_party + n_parties) -> d_lev int64[n_parties][count][l_lev][2][N]
int thfhe_kms_rotate_parties_dev(thfhe_kms_ctx *c, int op, const int32_t *d_x, const int32_t *d_y, int first_party, int n_parties, int64_t *d_lev, size_t count) {
    typedef thfhe_kms_ctx K;
    if (!c || !d_x || !d_lev) return thfhe_fail(THFHE_E_INVALID, "null argument");
    int32_t cb, cx, cy;
    int rc = kms_lin_of(op, cb, cx, cy);
    if (rc) return rc;
    if (cy != 0 && !d_y) return thfhe_fail(THFHE_E_INVALID, "null operand");
    if (first_party < 0 || n_parties < 0 || first_party + n_parties > c->p.parties) return thfhe_fail(THFHE_E_INVALID, "party block out of range");
    if (count == 0 || n_parties == 0) return THFHE_OK;
    std::lock_guard<std::mutex> lock(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const int P = c->p.parties, n = c->p.n, lv = c->p.l_lev;
    const size_t G = count, N = 2048;
    rc = kms_w(c, K::W_BARA, (size_t)P * G * n * 4);
    if (!rc) rc = kms_w(c, K::W_ACCUM, G * (P + 1) * N * 8);
    if (rc) return rc;
    hipLaunchKernelGGL(kms_prologue_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, d_x, cy != 0 ? d_y : nullptr, cb, cx, cy, n, P, (long)G, (int64_t)1 << 61,
                       (int32_t *)c->d_w[K::W_BARA], (int64_t *)c->d_w[K::W_ACCUM]);
    THFHE_HIP(hipGetLastError());
    for (int q = 0; q < n_parties && !rc; q++)
        rc = kms_launch_rotation(c, first_party + q, (const int32_t *)c->d_w[K::W_BARA] + (size_t)(first_party + q) * G * n, d_lev + (size_t)q * G * lv * 2 * N, nullptr, G, lv);
    return rc;
}
// phase 2: accum = X^{-barb} mu (trivial), then mk_lev_rlwe_mul for p = 0 .. P-1 with d_lev_all int64[P][count][l_lev][2][N] (sequential by construction,
// J/new_mk_internals.jl:276-283), mk_rlwe_extract_sample_64 + t64tot32, mk_keyswitch -> d_out int32[count][P n + 1]
int thfhe_kms_finish_dev(thfhe_kms_ctx *c, int op, const int32_t *d_x, const int32_t *d_y, const int64_t *d_lev_all, int32_t *d_out, size_t count) {
    typedef thfhe_kms_ctx K;
    if (!c || !d_x || !d_lev_all || !d_out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    int32_t cb, cx, cy;
    int rc = kms_lin_of(op, cb, cx, cy);
    if (rc) return rc;
    if (cy != 0 && !d_y) return thfhe_fail(THFHE_E_INVALID, "null operand");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->d_relin) return thfhe_fail(THFHE_E_INVALID, "thfhe_kms_set_relin_keys has not been called on this context");
    THFHE_HIP(hipSetDevice(c->device));
    KmsTables tabs;
    const int P = c->p.parties, n = c->p.n, lv = c->p.l_lev;
    const size_t G = count, N = 2048, words = (size_t)P * n + 1, uw = (size_t)P * N + 1;
    auto body = [&]() -> int {
        int r = kms_w(c, K::W_BARA, (size_t)P * G * n * 4);
        if (!r) r = kms_w(c, K::W_ACCUM, G * (P + 1) * N * 8);
        if (!r) r = kms_w(c, K::W_U, G * uw * 4);
        if (r) return r;
        THFHE_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
        int64_t *d_accum = (int64_t *)c->d_w[K::W_ACCUM];
        hipLaunchKernelGGL(kms_prologue_kernel, dim3((unsigned)G), dim3(256), 0, c->stream, d_x, cy != 0 ? d_y : nullptr, cb, cx, cy, n, P, (long)G, (int64_t)1 << 61,
                           (int32_t *)c->d_w[K::W_BARA], d_accum);
        THFHE_HIP(hipGetLastError());
        for (int party = 0; party < P; party++) {
            r = kms_lev_rlwe_mul_dev(c, tabs, party, G, d_accum, d_lev_all + (size_t)party * G * lv * 2 * N);
            if (r) return r;
        }
        hipLaunchKernelGGL(kms_extract_kernel, dim3((unsigned)G, (unsigned)(P + 1)), dim3(256), 0, c->stream, (const int64_t *)d_accum, P, (int32_t *)c->d_w[K::W_U]);
        THFHE_HIP(hipGetLastError());
        THFHE_HIP(hipMemsetAsync(d_out, 0, G * words * 4, c->stream));
        MKKSArgs k{c->d_ksk, (const int32_t *)c->d_w[K::W_U], d_out, (long)G, n, c->p.ks_t, c->p.ks_basebit, P, c->row_words, (int)N, (int)uw, (int)N};
        const int nsplit = G <= 64 ? 8 : 2;
        mk_launch_keyswitch(k, nsplit, c->stream);
        THFHE_HIP(hipGetLastError());
        return kms_finish(c);
    };
    rc = body();
    if (rc) (void)hipStreamSynchronize(c->stream);
    return rc;
}
// Enqueue every later call on the caller's HIP stream (the stream the caller's RCCL communicator orders with); NULL returns to the context's own stream.
int thfhe_kms_set_stream(thfhe_kms_ctx *c, void *hip_stream) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return THFHE_OK;
}

// Launches of at most `max_single_jobs` rotations run one job per workgroup, larger ones two jobs per workgroup sharing every key chunk.
int thfhe_kms_set_pair_threshold(thfhe_kms_ctx *c, long max_single_jobs) {
    if (!c) return thfhe_fail(THFHE_E_INVALID, "null ctx");
    if (max_single_jobs < 0) return thfhe_fail(THFHE_E_INVALID, "threshold must be >= 0");
    std::lock_guard<std::mutex> g(c->mu);
    c->pair_threshold = max_single_jobs;
    return THFHE_OK;
}

}  // extern "C"
