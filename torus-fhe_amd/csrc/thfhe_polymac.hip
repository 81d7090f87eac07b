// thfhe_polymac.hip -- exact multiply-accumulate of small-coefficient polynomials with torus polynomials on gfx950:
//     out[j] = addend[j] + sum over the terms (j, s, t, sign) of  sign * small[s] (*) torus[t]      mod X^N + 1, mod 2^32 / 2^64
// This is the arithmetic of multi-key KEY GENERATION -- tgsw_encrypt_3gen (J/tgsw_3gen.jl:41-95: r1 (*) B, r2 (*) B, r2 (*) A, r1 (*) A),
// PublicKey / CommonPubKey_3gen (J/mk_internals.jl:266-345: z (*) a), the CCS mk_tgsw_encrypt (J/mk_internals.jl:390-446:
// r (*) a_i, s (*) f1_i) -- with the randomness handed in as arrays, so that the GPU result equals the host (numpy) key generation
// bit for bit (thfhe/keygen.py, tests/test_gpu_keygen.py).
// Method: the engine's split-limb FP64 transform (thfhe_lane.h).  Every torus polynomial is split into balanced 16-bit limbs and
// transformed once (pm_torus_transform_kernel); one wave per output transforms each small operand once, multiplies it into the
// limb spectra of its partner and inverse-transforms every limb product separately: |product| <= N 2^(sb-1) 2^15 with |small| < 2^(sb-1),
// sb <= 13 -- at most 2^38 for N = 2048, inside the exactness bound.  Ring degrees 1024 and 2048 (radix-2 split
// + two twisted 512-point transforms), Torus32 (N = 1024) and Torus64.
#include <hip/hip_runtime.h>

#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"
#include "thfhe_lane.h"

using namespace thfhe;

namespace {

#include "thfhe_pm_kernels.h"

}  // namespace

struct thfhe_pm_ctx {
    int device = 0, N = 1024, torus_bits = 32;
    hipStream_t stream = nullptr;
    cplx *d_tw = nullptr;
    int *d_flag = nullptr;
    void *d_buf[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // small, torus, spec, terms, first, addend, out
    size_t cap[7] = {0, 0, 0, 0, 0, 0, 0};
    std::mutex mu;
};

namespace {
int pm_ensure(thfhe_pm_ctx *c, int slot, size_t bytes) {
    if (bytes <= c->cap[slot]) return THFHE_OK;
    (void)hipFree(c->d_buf[slot]);
    c->d_buf[slot] = nullptr;
    c->cap[slot] = 0;
    THFHE_HIP(hipMalloc(&c->d_buf[slot], bytes));
    c->cap[slot] = bytes;
    return THFHE_OK;
}

template <int NN, int TB>
int pm_run(thfhe_pm_ctx *c, size_t n_torus, const PMArgs &a) {
    hipLaunchKernelGGL((pm_torus_transform_kernel<NN, TB>), dim3((unsigned)((n_torus * (TB / 16) + 3) / 4)), dim3(256), 0, c->stream, c->d_buf[1], (long)n_torus,
                       c->d_tw, (cplx *)c->d_buf[2]);
    hipLaunchKernelGGL((pm_mac_kernel<NN, TB>), dim3((unsigned)((a.n_out + 3) / 4)), dim3(256), 0, c->stream, a);
    THFHE_HIP(hipGetLastError());
    return THFHE_OK;
}
}  // namespace

extern "C" {

int thfhe_pm_ctx_create(int device, int N, int torus_bits, thfhe_pm_ctx **out) {
    if (!out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    *out = nullptr;
    if (!((N == 1024 && (torus_bits == 32 || torus_bits == 64)) || (N == 2048 && torus_bits == 64)))
        return thfhe_fail(THFHE_E_UNSUPPORTED, "polynomial products: N = 1024 with Torus32 / Torus64, N = 2048 with Torus64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return thfhe_fail(THFHE_E_NO_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    THFHE_HIP(hipSetDevice(device));
    thfhe_pm_ctx *c = new (std::nothrow) thfhe_pm_ctx;
    if (!c) return thfhe_fail(THFHE_E_NOMEM, "out of host memory");
    c->device = device, c->N = N, c->torus_bits = torus_bits;
    std::vector<cplx> tw(1088);  // N = 1024: T1[512] T2[64]; N = 2048: T1(twist 1)[512] T1(twist 5)[512] T2[64]
    if (N == 2048) {
        std::vector<cplx> unused(512);
        make_twiddles_2048(tw.data(), tw.data() + 512);
        make_twiddles_1024(unused.data(), tw.data() + 1024);
    } else {
        make_twiddles_1024(tw.data(), tw.data() + 512);
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&c->d_tw, tw.size() * sizeof(cplx));
    if (e == hipSuccess) e = hipMalloc(&c->d_flag, sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        thfhe_pm_ctx_destroy(c);
        return thfhe_fail_hip(e, "thfhe_pm_ctx_create");
    }
    *out = c;
    return THFHE_OK;
}

void thfhe_pm_ctx_destroy(thfhe_pm_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_tw);
    (void)hipFree(c->d_flag);
    for (auto &p : c->d_buf) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int thfhe_pm_mac(thfhe_pm_ctx *c, const int32_t *small, size_t n_small, const void *torus, size_t n_torus, const int32_t *terms, size_t n_terms,
                 const void *addend, void *out, size_t n_out) {
    if (!c || !small || !torus || !out || (!terms && n_terms)) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (n_out == 0) return THFHE_OK;
    // validate the term list on the host: grouped by output, indices in range
    std::vector<int32_t> first(n_out + 1, 0);
    long prev = -1;
    for (size_t t = 0; t < n_terms; t++) {
        const int32_t j = terms[4 * t], s = terms[4 * t + 1], q = terms[4 * t + 2], sg = terms[4 * t + 3];
        if (j < 0 || (size_t)j >= n_out || s < 0 || (size_t)s >= n_small || q < 0 || (size_t)q >= n_torus || (sg != 1 && sg != -1) || j < prev)
            return thfhe_fail(THFHE_E_INVALID, "term list: need (out, small, torus, +-1) with indices in range and outputs in ascending order");
        prev = j;
        first[j + 1]++;
    }
    for (size_t j = 0; j < n_out; j++) first[j + 1] += first[j];
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t N = c->N, wb = c->torus_bits / 8, limbs = c->torus_bits / 16;
    const size_t bytes[7] = {n_small * N * 4, n_torus * N * wb, n_torus * limbs * (N / 2) * sizeof(cplx), (n_terms ? n_terms : 1) * 16, (n_out + 1) * 4,
                             n_out * N * wb, n_out * N * wb};
    for (int q = 0; q < 7; q++) {
        int rc = pm_ensure(c, q, bytes[q]);
        if (rc) return rc;
    }
    THFHE_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_buf[0], small, bytes[0], hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_buf[1], torus, bytes[1], hipMemcpyHostToDevice, c->stream));
    if (n_terms) THFHE_HIP(hipMemcpyAsync(c->d_buf[3], terms, n_terms * 16, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_buf[4], first.data(), bytes[4], hipMemcpyHostToDevice, c->stream));
    if (addend) THFHE_HIP(hipMemcpyAsync(c->d_buf[5], addend, bytes[5], hipMemcpyHostToDevice, c->stream));
    PMArgs a{(const int32_t *)c->d_buf[0], (const cplx *)c->d_buf[2], (const int32_t *)c->d_buf[3], (const int32_t *)c->d_buf[4],
             addend ? c->d_buf[5] : nullptr, c->d_buf[6], c->d_tw, (long)n_out, c->d_flag};
    int rc = c->N == 2048 ? pm_run<2048, 64>(c, n_torus, a) : (c->torus_bits == 64 ? pm_run<1024, 64>(c, n_torus, a) : pm_run<1024, 32>(c, n_torus, a));
    if (rc) return rc;
    int flag = 0;
    THFHE_HIP(hipMemcpyAsync(out, c->d_buf[6], bytes[6], hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    if (flag) return thfhe_fail(THFHE_E_UNSUPPORTED, "a small-operand coefficient exceeds 2^12 (outside the FP64 exactness bound of the product)");
    return THFHE_OK;
}

}  // extern "C"
