// thfhe_common.h -- shared host/device helpers of libthfhe_hip.so (error plumbing, wave-level transform drivers).
#ifndef THFHE_COMMON_H
#define THFHE_COMMON_H

#include <hip/hip_runtime.h>

#include "../../include/thfhe_hip.h"
#include "thfhe_lane.h"

namespace thfhe {

constexpr int kOpIdentity = 100;  // internal opcode: tmp = x (plain bootstrap of a sample)

int thfhe_fail(int code, const char *msg);
int thfhe_fail_hip(hipError_t e, const char *what);

inline int ilog2(int x) {
    int l = 0;
    while ((1 << l) < x) l++;
    return l;
}

#define THFHE_HIP(expr)                                                    \
    do {                                                                   \
        hipError_t thfhe_e_ = (expr);                                      \
        if (thfhe_e_ != hipSuccess) return ::thfhe::thfhe_fail_hip(thfhe_e_, #expr); \
    } while (0)

#if defined(__HIPCC__)
#ifdef THFHE_STAMPS
// Diagnostic build only (make stamps -> torus-fhe_amd/lib/libthfhe_hip_stamps.so, never shipped): per-wave cycle totals of the phases of
// the ring kernel's CMux loop, s_memtime deltas summed over all CMuxes; read back with thfhe_debug_read_stamps.
#define THFHE_STAMP_STORAGE __device__ unsigned long long g_stamps[8 * 2048 * 8];
#define STAMP_DECL unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime()
#define STAMP(slot)                                              \
    do {                                                         \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        st_acc[slot] += now_ - st_t;                             \
        st_t = now_;                                             \
    } while (0)
#define STAMP_FLUSH(wg, wave)                                                                                  \
    do {                                                                                                       \
        if (lane == 0 && (wg) < 2048)                                                                          \
            for (int q_ = 0; q_ < 6; q_++) g_stamps[(((size_t)(wg)) * 8 + (wave)) * 8 + q_] = st_acc[q_];       \
    } while (0)
#else
#define THFHE_STAMP_STORAGE
#define STAMP_DECL
#define STAMP(slot)
#define STAMP_FLUSH(wg, wave)
#endif


// Wave-level ordering point between two LDS segments.  One wavefront's DS instructions execute in issue order,
// so no hardware barrier is needed -- this only stops the compiler from moving LDS accesses across the exchange.
// The fences are scoped to the LDS address space so that global (bootstrapping-key) loads may be scheduled across them.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// The lane index behind an empty asm: everything derived from it (the XOR-swizzled LDS slot maps: ~25 addresses of 1-3 integer
// instructions each) is recomputed where it is used.  Without it the compiler hoists those addresses out of the CMux loop as
// loop invariants and, being at the VGPR limit, spills them to scratch.
__device__ __forceinline__ int opaque_lane(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}
// forward / inverse folded negacyclic transform of the 8 points each lane holds (see thfhe_lane.h)
__device__ __forceinline__ void wave_fft_fwd(int lane, cplx (&z)[8], cplx *xb, const cplx *T1, const cplx *T2) {
    wave_sync();
    fwd_seg1(lane, z, xb, T1);
    wave_sync();
    fwd_seg2_ld(lane, z, xb);
    fwd_seg2_st(lane, z, xb, T2);
    wave_sync();
    fwd_seg3(lane, z, xb);
}
__device__ __forceinline__ void wave_fft_inv(int lane, cplx (&z)[8], cplx *xb, const cplx *T1, const cplx *T2) {
    wave_sync();
    inv_seg1(lane, z, xb, T2);
    wave_sync();
    inv_seg2_ld(lane, z, xb);
    inv_seg2_st(lane, z, xb);
    wave_sync();
    inv_seg3(lane, z, xb, T1);
}
// Wave-uniform words that no kernel of the launch writes (the mod-switched mask words a rotation walks: written by the prologue kernel before it):
// read through the constant address space, i.e. with s_load_dword into an SGPR.  As plain global loads they were global_load_dword + s_waitcnt vmcnt(0)
// at the top of every CMux -- a vector-memory round trip on the sequential chain, and a vmcnt(0) that also waits for whatever the wave has in flight.
typedef const int32_t __attribute__((address_space(4))) *uniform_i32_ptr;
__device__ __forceinline__ uniform_i32_ptr as_uniform(const int32_t *p) { return (uniform_i32_ptr)(size_t)p; }

// ---- LDS key ring (shared by the single-key and multi-key ring kernels) --------------------------------------------
// One LDS-DMA of this wave's 1 KiB slice of a key chunk: lane l fetches 16 B at gptr_lane into LDS at lds_byte_off + 16 l.
// Inline asm on purpose: the compiler then neither waits vmcnt(0) before every ring read nor reorders the hand-off.  The LDS base
// travels in m0 as a register-constrained INPUT ("{m0}"): the compiler itself materialises the value in m0 and knows it is live there,
// so nothing is clobbered behind its back (round 3 wrote m0 inside the asm and listed it as a clobber, which LLVM calls undefined
// for a reserved register).  s_nop 0 = the one wait state gfx9 wants between an SALU write of m0 and an LDS-DMA.
__device__ __forceinline__ void ring_dma(const cplx *gptr_lane, uint32_t lds_byte_off) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr_lane), "{m0}"(lds_byte_off) : "memory");
}
// Hand-off barrier of the ring: this wave's own slice of the chunk has landed (at most VM younger DMAs in flight), every LDS read it issued
// has returned, then the workgroup barrier.  Two forms: the wait inside an asm (default), or the s_waitcnt BUILTIN -- with the builtin the compiler's
// wait-count pass knows that nothing is pending on lgkmcnt after the barrier and counts the waits of reads issued right after it exactly (with the asm
// it puts a conservative lgkmcnt(0) in front of the first multiply after the barrier; harmless in the default order, where no read is pending there).
// gfx9 encoding of s_waitcnt: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14.
#ifndef THFHE_RING_ASM_BARRIER
#define THFHE_RING_ASM_BARRIER 1   // the builtin form times the same (profiles/r04_ring_multiply_phase.md); the asm form is the one every round measured
#endif
template <int VM>
__device__ __forceinline__ void ring_barrier() {
#if THFHE_RING_ASM_BARRIER
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
#else
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt((VM & 15) | (7 << 4) | (0 << 8) | ((VM >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}
__device__ __forceinline__ void wave_fft_fwd_s(int lane, cplx (&z)[8], cplx *xb, const cplx *T1, const W64 &w) {
    wave_sync();
    fwds_seg1(lane, z, xb, T1);
    wave_sync();
    fwds_seg2_ld(lane, z, xb);
    fwds_seg2_st(lane, z, xb, w);
    wave_sync();
    fwds_seg3(lane, z, xb);
}
__device__ __forceinline__ void wave_fft_inv_s(int lane, cplx (&z)[8], cplx *xb, const cplx *T1, const W64 &w) {
    wave_sync();
    invs_seg1(lane, z, xb, w);
    wave_sync();
    invs_seg2_ld(lane, z, xb);
    invs_seg2_st(lane, z, xb);
    wave_sync();
    invs_seg3(lane, z, xb, T1);
}
// variant "r": padded buffer, pass-1 twiddles from per-lane roots (thfhe_lane.h)
__device__ __forceinline__ void wave_fft_fwd_r(int lane, cplx (&z)[8], cplx *xb, const LaneRoots &r, const W64 &w) {
    wave_sync();
    fwdr_seg1(lane, z, xb, r);
    wave_sync();
    fwd_seg2_ld(lane, z, xb);
    fwdr_seg2_st(lane, z, xb, w);
    wave_sync();
    fwd_seg3(lane, z, xb);
}
__device__ __forceinline__ void wave_fft_inv_r(int lane, cplx (&z)[8], cplx *xb, const LaneRoots &r, const W64 &w) {
    wave_sync();
    invr_seg1(lane, z, xb, w);
    wave_sync();
    inv_seg2_ld(lane, z, xb);
    inv_seg2_st(lane, z, xb);
    wave_sync();
    invr_seg3(lane, z, xb, r);
}
// variant "q": first transpose in registers.  Exchange of the register index (bits 2,1,0) with lane bits (5,4,3):
//   stage A  z[r] (r < 4) of the upper half-wave <-> z[r + 4] of the lower half-wave      v_permlane32_swap
//   stage B  z[r] (bit 1 clear) of the odd 16-lane rows <-> z[r + 2] of the even rows      v_permlane16_swap
//   stage C  lanes with bit 3 clear take z[r] (r even) of lane ^ 8 into z[r + 1], lanes with bit 3 set take z[r + 1] into z[r]
//            (row_ror:8 DPP moves, bank masks 0x3 / 0xC)
// The exchange is its own inverse.
__device__ __forceinline__ void wave_transpose_hi3(cplx (&z)[8]) {
    uint32_t w[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) __builtin_memcpy(w[r], &z[r], 16);
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            auto q = __builtin_amdgcn_permlane32_swap(w[r][d], w[r + 4][d], false, false);
            w[r][d] = q[0];
            w[r + 4][d] = q[1];
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r & 2) continue;
            auto q = __builtin_amdgcn_permlane16_swap(w[r][d], w[r + 2][d], false, false);
            w[r][d] = q[0];
            w[r + 2][d] = q[1];
        }
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            const uint32_t a = w[r][d], b = w[r + 1][d];
            w[r + 1][d] = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xF, 0x3, false);
            w[r][d] = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xF, 0xC, false);
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) __builtin_memcpy(&z[r], w[r], 16);
}
template <class Roots>
__device__ __forceinline__ void wave_fft_fwd_q(int lane, cplx (&z)[8], cplx *xb, const Roots &r, const W64 &w) {
    fwdq_seg1(z, r);
    wave_transpose_hi3(z);
    wave_sync();
    fwdr_seg2_st(lane, z, xb, w);
    wave_sync();
    fwd_seg3(lane, z, xb);
}
template <class Roots>
__device__ __forceinline__ void wave_fft_inv_q(int lane, cplx (&z)[8], cplx *xb, const Roots &r, const W64 &w) {
    wave_sync();
    invr_seg1(lane, z, xb, w);
    wave_sync();
    inv_seg2_ld(lane, z, xb);
    dft8<-1>(z);
    wave_transpose_hi3(z);
    invq_seg3(z, r);
}
// variant "qs" (multi-key kernels, whose LDS has no room for padded buffers): first transpose in registers, pass-1 twiddles from the
// per-lane roots, second transpose through the XOR-swizzled 512-slot buffer -- one LDS crossing and no T1 table reads per transform
template <class Roots>
__device__ __forceinline__ void wave_fft_fwd_qs(int lane, cplx (&z)[8], cplx *xb, const Roots &r, const W64 &w) {
    fwdq_seg1(z, r);
    wave_transpose_hi3(z);
    wave_sync();
    fwds_seg2_st(lane, z, xb, w);
    wave_sync();
    fwds_seg3(lane, z, xb);
}
template <class Roots>
__device__ __forceinline__ void wave_fft_inv_qs(int lane, cplx (&z)[8], cplx *xb, const Roots &r, const W64 &w) {
    wave_sync();
    invs_seg1(lane, z, xb, w);
    wave_sync();
    invs_seg2_ld(lane, z, xb);
    dft8<-1>(z);
    wave_transpose_hi3(z);
    invq_seg3(z, r);
}
// N = 2048 halves: twisted 512-point transforms (thfhe_lane.h, "N = 2048" section); T1t = the table of twist T
template <int T>
__device__ __forceinline__ void wave_fft_fwd_t(int lane, cplx (&z)[8], cplx *xb, const cplx *T1t, const W64 &w) {
    wave_sync();
    fwdt_seg1<T>(lane, z, xb, T1t);
    wave_sync();
    fwds_seg2_ld(lane, z, xb);
    fwds_seg2_st(lane, z, xb, w);
    wave_sync();
    fwds_seg3(lane, z, xb);
}
template <int T>
__device__ __forceinline__ void wave_fft_inv_t(int lane, cplx (&z)[8], cplx *xb, const cplx *T1t, const W64 &w) {
    wave_sync();
    invs_seg1(lane, z, xb, w);
    wave_sync();
    invs_seg2_ld(lane, z, xb);
    invs_seg2_st(lane, z, xb);
    wave_sync();
    invt_seg3<T>(lane, z, xb, T1t);
}
// "qs" form of the twisted halves: first transpose in registers, pass-1 twiddles from per-lane roots (no T1 table), one LDS crossing
template <int T, int DEN = 32>
__device__ __forceinline__ void wave_fft_fwd_tq(int lane, cplx (&z)[8], cplx *xb, const LaneRoots &r, const W64 &w) {
    fwdtq_seg1<T, DEN>(z, r);
    wave_transpose_hi3(z);
    wave_sync();
    fwds_seg2_st(lane, z, xb, w);
    wave_sync();
    fwds_seg3(lane, z, xb);
}
template <int T, int DEN = 32>
__device__ __forceinline__ void wave_fft_inv_tq(int lane, cplx (&z)[8], cplx *xb, const LaneRoots &r, const W64 &w) {
    wave_sync();
    invs_seg1(lane, z, xb, w);
    wave_sync();
    invs_seg2_ld(lane, z, xb);
    dft8<-1>(z);
    wave_transpose_hi3(z);
    invtq_seg3<T, DEN>(z, r);
}

// Two transforms of one wave side by side.  A wavefront's DS instructions execute in issue order, so the second transform's arithmetic covers
// the first one's LDS round trip (and the other way round); the inverse pair shares ONE transpose buffer: the second one's stores may follow
// the first one's loads without a wait.
template <int TA, int TB, int DEN = 32>
__device__ __forceinline__ void wave_fft_fwd_tq_two(int lane, cplx (&za)[8], cplx (&zb)[8], cplx *xa, cplx *xb, const LaneRoots &ra, const LaneRoots &rb,
                                                    const W64 &w) {
    fwdtq_seg1<TA, DEN>(za, ra);
    fwdtq_seg1<TB, DEN>(zb, rb);
    wave_transpose_hi3(za);
    wave_transpose_hi3(zb);
    wave_sync();
    fwds_seg2_st(lane, za, xa, w);
    fwds_seg2_st(lane, zb, xb, w);
    wave_sync();
    fwds_seg3(lane, za, xa);
    fwds_seg3(lane, zb, xb);
}
template <int TA, int TB, int DEN = 32>
__device__ __forceinline__ void wave_fft_inv_tq_two(int lane, cplx (&za)[8], cplx (&zb)[8], cplx *xb, const LaneRoots &ra, const LaneRoots &rb, const W64 &w) {
    wave_sync();
    invs_seg1(lane, za, xb, w);
    wave_sync();
    invs_seg2_ld(lane, za, xb);
    wave_sync();
    invs_seg1(lane, zb, xb, w);
    wave_sync();
    dft8<-1>(za);
    wave_transpose_hi3(za);
    invs_seg2_ld(lane, zb, xb);
    invtq_seg3<TA, DEN>(za, ra);
    dft8<-1>(zb);
    wave_transpose_hi3(zb);
    invtq_seg3<TB, DEN>(zb, rb);
}

#endif

}  // namespace thfhe
#endif
