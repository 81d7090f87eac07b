// thfhe_threshold.hip -- the step AFTER the gate path in the reference's C++ applications, on gfx950:
//   TLweFromLwe           src/libthfhe.cpp:340-348 (= src/KNN_medical_data.cpp:492-500): LWE(N) -> ring sample (a', b')
//   PartialDecrypt        src/libthfhe.cpp:270-293, partialDecrypt src/threshold_decryption_functions.cpp:441-480:
//                         partial = key_share (*) a' + smudging noise, (*) = exact negacyclic product mod 2^32
//                         (libtfhe's torusPolynomialAddMulR; the reference's own nonFFTmul, :357-375, is the exact twin)
//   finalDecrypt          src/libthfhe.cpp:296-315: result = b' - partial_0 + sum_{i>=1} partial_i, bit = result[0] > 0
// The product is the blind-rotate engine's split-limb FP64 transform with the roles swapped: the small integer polynomial
// (the key share, |s| <= 2^9) is transformed once per call, every ciphertext mask is split into two balanced 16-bit limbs
// (two forward, two inverse transforms per ciphertext, one wave each).  |sum| <= N 2^9 2^15 = 2^34: inside the exactness bound.
#include <hip/hip_runtime.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/thfhe_hip.h"
#include "thfhe_common.h"
#include "thfhe_lane.h"

using namespace thfhe;

namespace {

__global__ __launch_bounds__(64) void share_transform_kernel(const int32_t *__restrict__ share, const cplx *__restrict__ tw, cplx *__restrict__ spec,
                                                             int *__restrict__ too_big) {
    __shared__ cplx sT1[512];
    __shared__ cplx sX[512];
    const int lane = threadIdx.x;
    for (int t = lane; t < 512; t += 64) sT1[t] = tw[t];
    __syncthreads();
    const W64 w64{tw[512 + 1 * 8 + (lane & 7)]};
    cplx z[8];
    int big = 0;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int32_t a = share[lane + 64 * m], b = share[lane + 64 * m + 512];
        big |= (a > 512 || a < -512 || b > 512 || b < -512);
        z[m] = cplx{(double)a, (double)b};
    }
    if (big) atomicOr(too_big, 1);
    wave_fft_fwd_s(lane, z, sX, sT1, w64);
#pragma unroll
    for (int m = 0; m < 8; m++) spec[m * 64 + lane] = cplx{z[m].re * (1.0 / 512), z[m].im * (1.0 / 512)};
}

// one wave per ciphertext: partial[c] = share (*) a[c] (+ noise[c])
__global__ __launch_bounds__(256) void partial_decrypt_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ noise,
                                                              const cplx *__restrict__ spec, const cplx *__restrict__ tw,
                                                              int32_t *__restrict__ out, long count) {
    __shared__ cplx sT1[512];
    __shared__ cplx sX[4][512];
    for (int t = threadIdx.x; t < 512; t += 256) sT1[t] = tw[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const W64 w64{tw[512 + 1 * 8 + (lane & 7)]};
    const long c = (long)blockIdx.x * 4 + wave;
    if (c >= count) return;
    cplx zlo[8], zhi[8], S[8];
    key_limbs_to_z(lane, a + c * 1024, zlo, zhi);
    load8(lane, S, spec);
    wave_fft_fwd_s(lane, zlo, sX[wave], sT1, w64);
    wave_fft_fwd_s(lane, zhi, sX[wave], sT1, w64);
#pragma unroll
    for (int m = 0; m < 8; m++) {
        zlo[m] = cmul(zlo[m], S[m]);
        zhi[m] = cmul(zhi[m], S[m]);
    }
    wave_fft_inv_s(lane, zlo, sX[wave], sT1, w64);
    wave_fft_inv_s(lane, zhi, sX[wave], sT1, w64);
    int32_t *o = out + c * 1024;
    const int32_t *e = noise ? noise + c * 1024 : nullptr;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int q = lane + 64 * m;
        uint32_t vr = round_lo32(zlo[m].re) + (round_lo32(zhi[m].re) << 16);
        uint32_t vi = round_lo32(zlo[m].im) + (round_lo32(zhi[m].im) << 16);
        if (e) {
            vr += (uint32_t)e[q];
            vi += (uint32_t)e[q + 512];
        }
        o[q] = (int32_t)vr;
        o[q + 512] = (int32_t)vi;
    }
}

__global__ __launch_bounds__(256) void tlwe_from_lwe_kernel(const int32_t *__restrict__ lwe, int32_t *__restrict__ ta, int32_t *__restrict__ tb, long count) {
    const long c = blockIdx.x;
    if (c >= count) return;
    const int32_t *x = lwe + c * 1025;
    for (int q = threadIdx.x; q < 1024; q += 256) {
        ta[c * 1024 + q] = q == 0 ? x[0] : (int32_t)(0u - (uint32_t)x[1024 - q]);
        tb[c * 1024 + q] = q == 0 ? x[1024] : 0;
    }
}

__global__ __launch_bounds__(256) void final_decrypt_kernel(const int32_t *__restrict__ tb, const int32_t *__restrict__ partials, int t, long count,
                                                            int32_t *__restrict__ result, int32_t *__restrict__ bits) {
    const long c = blockIdx.x;
    if (c >= count) return;
    for (int q = threadIdx.x; q < 1024; q += 256) {
        uint32_t v = (uint32_t)tb[c * 1024 + q];
        for (int i = 0; i < t; i++) {
            const uint32_t pv = (uint32_t)partials[((size_t)i * count + c) * 1024 + q];
            v = i == 0 ? v - pv : v + pv;
        }
        if (result) result[c * 1024 + q] = (int32_t)v;
        if (q == 0) bits[c] = (int32_t)v > 0 ? 1 : 0;
    }
}

}  // namespace

struct thfhe_poly_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    cplx *d_tw = nullptr, *d_spec = nullptr;
    int *d_flag = nullptr;
    void *d_buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    std::mutex mu;
};

namespace {
int ensure(thfhe_poly_ctx *c, int slot, size_t bytes) {
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

int thfhe_poly_ctx_create(int device, int N, thfhe_poly_ctx **out) {
    if (!out) return thfhe_fail(THFHE_E_INVALID, "null argument");
    *out = nullptr;
    if (N != 1024) return thfhe_fail(THFHE_E_UNSUPPORTED, "only N = 1024 (k = 1) is implemented");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return thfhe_fail(THFHE_E_NO_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    THFHE_HIP(hipSetDevice(device));
    thfhe_poly_ctx *c = new (std::nothrow) thfhe_poly_ctx;
    if (!c) return thfhe_fail(THFHE_E_NOMEM, "out of host memory");
    c->device = device;
    std::vector<cplx> tw(576);
    make_twiddles_1024(tw.data(), tw.data() + 512);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&c->d_tw, tw.size() * sizeof(cplx));
    if (e == hipSuccess) e = hipMalloc(&c->d_spec, 512 * sizeof(cplx));
    if (e == hipSuccess) e = hipMalloc(&c->d_flag, sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        thfhe_poly_ctx_destroy(c);
        return thfhe_fail_hip(e, "thfhe_poly_ctx_create");
    }
    *out = c;
    return THFHE_OK;
}

void thfhe_poly_ctx_destroy(thfhe_poly_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_tw);
    (void)hipFree(c->d_spec);
    (void)hipFree(c->d_flag);
    for (auto &p : c->d_buf) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int thfhe_tlwe_from_lwe(thfhe_poly_ctx *c, const int32_t *lwe, int32_t *tlwe_a, int32_t *tlwe_b, size_t count) {
    if (!c || !lwe || !tlwe_a || !tlwe_b) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    int rc = ensure(c, 0, count * 1025 * 4);
    if (!rc) rc = ensure(c, 1, count * 1024 * 4);
    if (!rc) rc = ensure(c, 2, count * 1024 * 4);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_buf[0], lwe, count * 1025 * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(tlwe_from_lwe_kernel, dim3((unsigned)count), dim3(256), 0, c->stream, (const int32_t *)c->d_buf[0], (int32_t *)c->d_buf[1],
                       (int32_t *)c->d_buf[2], (long)count);
    THFHE_HIP(hipGetLastError());
    THFHE_HIP(hipMemcpyAsync(tlwe_a, c->d_buf[1], count * 1024 * 4, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipMemcpyAsync(tlwe_b, c->d_buf[2], count * 1024 * 4, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

int thfhe_partial_decrypt(thfhe_poly_ctx *c, const int32_t *key_share, const int32_t *tlwe_a, const int32_t *noise, int32_t *partial, size_t count) {
    if (!c || !key_share || !tlwe_a || !partial) return thfhe_fail(THFHE_E_INVALID, "null argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t bytes = count * 1024 * 4;
    int rc = ensure(c, 0, bytes);
    if (!rc) rc = ensure(c, 1, bytes);
    if (!rc) rc = ensure(c, 2, bytes);
    if (!rc) rc = ensure(c, 3, 1024 * 4);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_buf[3], key_share, 1024 * 4, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_buf[0], tlwe_a, bytes, hipMemcpyHostToDevice, c->stream));
    if (noise) THFHE_HIP(hipMemcpyAsync(c->d_buf[1], noise, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(share_transform_kernel, dim3(1), dim3(64), 0, c->stream, (const int32_t *)c->d_buf[3], c->d_tw, c->d_spec, c->d_flag);
    hipLaunchKernelGGL(partial_decrypt_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, c->stream, (const int32_t *)c->d_buf[0],
                       noise ? (const int32_t *)c->d_buf[1] : nullptr, c->d_spec, c->d_tw, (int32_t *)c->d_buf[2], (long)count);
    THFHE_HIP(hipGetLastError());
    int flag = 0;
    THFHE_HIP(hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipMemcpyAsync(partial, c->d_buf[2], bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    if (flag) return thfhe_fail(THFHE_E_UNSUPPORTED, "key-share coefficients must satisfy |s| <= 512 (FP64 exactness bound)");
    return THFHE_OK;
}

int thfhe_final_decrypt(thfhe_poly_ctx *c, const int32_t *tlwe_b, const int32_t *partials, int t, int32_t *result, int32_t *bits, size_t count) {
    if (!c || !tlwe_b || !partials || !bits || t < 1) return thfhe_fail(THFHE_E_INVALID, "bad argument");
    if (count == 0) return THFHE_OK;
    std::lock_guard<std::mutex> g(c->mu);
    THFHE_HIP(hipSetDevice(c->device));
    const size_t bytes = count * 1024 * 4;
    int rc = ensure(c, 0, bytes);
    if (!rc) rc = ensure(c, 1, bytes * t);
    if (!rc) rc = ensure(c, 2, bytes);
    if (!rc) rc = ensure(c, 3, count * 4 > 4096 ? count * 4 : 4096);
    if (rc) return rc;
    THFHE_HIP(hipMemcpyAsync(c->d_buf[0], tlwe_b, bytes, hipMemcpyHostToDevice, c->stream));
    THFHE_HIP(hipMemcpyAsync(c->d_buf[1], partials, bytes * t, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(final_decrypt_kernel, dim3((unsigned)count), dim3(256), 0, c->stream, (const int32_t *)c->d_buf[0], (const int32_t *)c->d_buf[1], t,
                       (long)count, result ? (int32_t *)c->d_buf[2] : nullptr, (int32_t *)c->d_buf[3]);
    THFHE_HIP(hipGetLastError());
    if (result) THFHE_HIP(hipMemcpyAsync(result, c->d_buf[2], bytes, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipMemcpyAsync(bits, c->d_buf[3], count * 4, hipMemcpyDeviceToHost, c->stream));
    THFHE_HIP(hipStreamSynchronize(c->stream));
    return THFHE_OK;
}

}  // extern "C"
