/*
 * thfhe_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See thfhe_oracle.h.
 *
 * Plain C restatement of the reference's gate-bootstrapping path with EXACT integer ring
 * arithmetic.  Every function cites the reference lines it follows (paths relative to
 * /root/reference, J/ = 3-gen-mk-tfhe/src/).  Two multiply engines give bit-identical results:
 *   - schoolbook negacyclic convolution (the literal meaning of the reference's *_wo_FFT path)
 *   - a 64-bit NTT over p = 2^64 - 2^32 + 1 with centred lifting (fast; used for batches and for
 *     the cpu_baseline timing).  tests/test_oracle_units.py checks NTT == schoolbook.
 * The GPU product uses a third, independent method (split-limb FP64 FFT), so GPU == oracle is a
 * genuine cross-check.
 *
 * Engine 2 (single-key path only) is NOT exact: it restates the arithmetic the reference really runs -- the folded N/2-point
 * Complex{Float64} transform of J/polynomials.jl:81-247 under tgsw_extern_mul (J/tgsw.jl:146-150), products of full 32-bit torus
 * words, rounded back with to_int32 -- so its low bits carry FFT rounding noise (far below the ciphertext noise), as the
 * reference's own outputs do.  It exists (a) to show that the exact engines and the reference's approximate path decrypt alike and
 * differ by that noise only (tests/test_oracle_units.py), (b) as the cpu_baseline of bench.py: the same algorithm class and cost as
 * the reference's CPU path, not the ~3x more expensive exact NTT.
 */
#include "thfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* per-thread scratch arena: the hot loop (n CMuxes per gate) allocates nothing -- one growing buffer per call site and
 * thread, kept for the life of the thread (OpenMP workers persist), so `oracle_gates` scales with the host's cores
 * instead of serialising on the allocator */
enum { SCR_MUX = 0, SCR_EXT_DIG, SCR_EXT_NTT, SCR_EXT_PROD, SCR_BOOT, SCR_KS, SCR_GATE, SCR_MKMUX, SCR_MKEXT_DIG, SCR_MKEXT_NTT,
       SCR_MKEXT_PROD, SCR_MKBOOT, SCR_MKKS, SCR_MKGATE, SCR_EXT_FFT, SCR_SLOTS };
static __thread void *scr_ptr[SCR_SLOTS];
static __thread size_t scr_cap[SCR_SLOTS];
static void *scr(int slot, size_t bytes) {
    if (scr_cap[slot] < bytes) {
        free(scr_ptr[slot]);
        scr_ptr[slot] = malloc(bytes);
        if (!scr_ptr[slot]) abort();
        scr_cap[slot] = bytes;
    }
    return scr_ptr[slot];
}

/* ============================================================================================
 * Goldilocks field  p = 2^64 - 2^32 + 1
 * ========================================================================================== */
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^64 mod p */

/* branch-free: data-dependent branches mispredict half the time on random residues */
static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    s += (0 - (uint64_t)(s < a)) & GL_EPS; /* wrapped: +2^64 == +EPS (a,b < p so no second wrap) */
    s -= (0 - (uint64_t)(s >= GL_P)) & GL_P;
    return s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    return d + ((0 - (uint64_t)(a < b)) & GL_P); /* borrow: + p (mod 2^64) */
}
static inline uint64_t gl_reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    /* x = lo + hi_lo*2^64 + hi_hi*2^96 == lo + hi_lo*(2^32-1) - hi_hi  (mod p) */
    uint64_t t0 = lo - hi_hi;
    t0 -= (0 - (uint64_t)(lo < hi_hi)) & GL_EPS;
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t r = t0 + t1;
    r += (0 - (uint64_t)(r < t1)) & GL_EPS;
    r -= (0 - (uint64_t)(r >= GL_P)) & GL_P;
    return r;
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((u128)a * b); }
static uint64_t gl_pow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline uint64_t gl_from_i64(int64_t v) { return v >= 0 ? (uint64_t)v : GL_P - (uint64_t)(-v); }
static inline int64_t gl_to_centered(uint64_t r) { return (r > GL_P / 2) ? -(int64_t)(GL_P - r) : (int64_t)r; }

typedef struct {
    int N;
    uint64_t *psi_rev;     /* psi^{bitrev(i)}  */
    uint64_t *psi_inv_rev; /* psi^{-bitrev(i)} */
    uint64_t n_inv;
} gl_tables;

#define GL_MAX_LOG 13
static gl_tables g_tables[GL_MAX_LOG + 1];

static unsigned bitrev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

static const gl_tables *gl_get_tables(int N) {
    int lg = 0;
    while ((1 << lg) < N) lg++;
    if (lg > GL_MAX_LOG || (1 << lg) != N) abort();
    gl_tables *t = &g_tables[lg];
    if (t->N == N) return t;
#pragma omp critical(gl_tables_init)
    {
        if (t->N != N) {
            uint64_t psi = gl_pow(7, (GL_P - 1) / (2 * (uint64_t)N)); /* 7 generates F_p^* */
            uint64_t psi_inv = gl_pow(psi, GL_P - 2);
            uint64_t *pr = (uint64_t *)malloc(sizeof(uint64_t) * N);
            uint64_t *pir = (uint64_t *)malloc(sizeof(uint64_t) * N);
            uint64_t a = 1, b = 1;
            for (int i = 0; i < N; i++) {
                unsigned j = bitrev((unsigned)i, lg);
                pr[j] = a;
                pir[j] = b;
                a = gl_mul(a, psi);
                b = gl_mul(b, psi_inv);
            }
            t->psi_rev = pr;
            t->psi_inv_rev = pir;
            t->n_inv = gl_pow((uint64_t)N, GL_P - 2);
#pragma omp flush
            t->N = N;
        }
    }
    return t;
}

/* negacyclic forward NTT, natural order in -> bit-reversed out (Cooley-Tukey, merged psi twist) */
static void gl_ntt_fwd(uint64_t *a, const gl_tables *T) {
    int N = T->N, t = N;
    for (int m = 1; m < N; m <<= 1) {
        t >>= 1;
        for (int i = 0; i < m; i++) {
            int j1 = 2 * i * t;
            uint64_t S = T->psi_rev[m + i];
            for (int j = j1; j < j1 + t; j++) {
                uint64_t U = a[j], V = gl_mul(a[j + t], S);
                a[j] = gl_add(U, V);
                a[j + t] = gl_sub(U, V);
            }
        }
    }
}
/* inverse: bit-reversed in -> natural out (Gentleman-Sande), includes 1/N */
static void gl_ntt_inv(uint64_t *a, const gl_tables *T) {
    int N = T->N, t = 1;
    for (int m = N; m > 1; m >>= 1) {
        int j1 = 0, h = m >> 1;
        for (int i = 0; i < h; i++) {
            uint64_t S = T->psi_inv_rev[h + i];
            for (int j = j1; j < j1 + t; j++) {
                uint64_t U = a[j], V = a[j + t];
                a[j] = gl_add(U, V);
                a[j + t] = gl_mul(gl_sub(U, V), S);
            }
            j1 += 2 * t;
        }
        t <<= 1;
    }
    for (int j = 0; j < N; j++) a[j] = gl_mul(a[j], T->n_inv);
}

/* ============================================================================================
 * scalar helpers
 * ========================================================================================== */
static int ilog2(int x) {
    int l = 0;
    while ((1 << l) < x) l++;
    return l;
}

/* decode_message(phase, 2N): (x + 2^(32-log2(2N)-1)) >> (32-log2(2N)), arithmetic
 * J/numeric-functions.jl:70-73 ; result in [-N, N) */
int32_t oracle_modswitch(int32_t x, int32_t N) {
    int lg = ilog2(2 * N);
    int32_t y = (int32_t)((uint32_t)x + (1u << (32 - lg - 1)));
    return y >> (32 - lg); /* arithmetic shift of a negative int32: gcc/clang implement it as such */
}

/* X^shift * p mod X^N+1, any integer shift (taken mod 2N)    J/rlwe.jl:130-131 (DarkIntegers mul_by_monomial) */
void oracle_mul_by_monomial32(const int32_t *p, int32_t shift, int32_t N, int32_t *out) {
    int32_t s = ((shift % (2 * N)) + 2 * N) % (2 * N);
    for (int j = 0; j < N; j++) {
        int d = j + s; /* destination exponent in [0, 3N) */
        int neg = 0;
        while (d >= N) {
            d -= N;
            neg ^= 1;
        }
        out[d] = neg ? (int32_t)(0u - (uint32_t)p[j]) : p[j];
    }
}
void oracle_mul_by_monomial64(const int64_t *p, int32_t shift, int32_t N, int64_t *out) {
    int32_t s = ((shift % (2 * N)) + 2 * N) % (2 * N);
    for (int j = 0; j < N; j++) {
        int d = j + s;
        int neg = 0;
        while (d >= N) {
            d -= N;
            neg ^= 1;
        }
        out[d] = neg ? (int64_t)(0ull - (uint64_t)p[j]) : p[j];
    }
}

/* signed gadget decomposition    J/tgsw.jl:112-138, offset from J/tgsw.jl:26-30 */
void oracle_decompose32(const int32_t *p, int32_t N, int32_t l, int32_t Bgbit, int32_t *digits) {
    uint32_t mask = (1u << Bgbit) - 1u, half = 1u << (Bgbit - 1), offset = 0;
    for (int q = 1; q <= l; q++) offset += half << (32 - q * Bgbit);
    for (int q = 1; q <= l; q++)
        for (int j = 0; j < N; j++) {
            uint32_t v = (uint32_t)p[j] + offset;
            digits[(q - 1) * N + j] = (int32_t)((v >> (32 - q * Bgbit)) & mask) - (int32_t)half;
        }
}
void oracle_decompose64(const int64_t *p, int32_t N, int32_t l, int32_t Bgbit, int64_t *digits) {
    uint64_t mask = (1ull << Bgbit) - 1ull, half = 1ull << (Bgbit - 1), offset = 0;
    for (int q = 1; q <= l; q++) offset += half << (64 - q * Bgbit);
    for (int q = 1; q <= l; q++)
        for (int j = 0; j < N; j++) {
            uint64_t v = (uint64_t)p[j] + offset;
            digits[(size_t)(q - 1) * N + j] = (int64_t)((v >> (64 - q * Bgbit)) & mask) - (int64_t)half;
        }
}

/* exact negacyclic products (the reference's IntPolynomial * TorusPolynomial of the _wo_FFT path) */
void oracle_polymul_schoolbook32(const int32_t *a, const int32_t *b, int32_t N, int32_t *out) {
    uint32_t *acc = (uint32_t *)calloc((size_t)N, sizeof(uint32_t));
    for (int i = 0; i < N; i++) {
        uint32_t ai = (uint32_t)a[i];
        if (!ai) continue;
        for (int j = 0; j < N - i; j++) acc[i + j] += ai * (uint32_t)b[j];
        for (int j = N - i; j < N; j++) acc[i + j - N] -= ai * (uint32_t)b[j];
    }
    memcpy(out, acc, sizeof(uint32_t) * (size_t)N);
    free(acc);
}
void oracle_polymul_schoolbook64(const int64_t *a, const int64_t *b, int32_t N, int64_t *out) {
    uint64_t *acc = (uint64_t *)calloc((size_t)N, sizeof(uint64_t));
    for (int i = 0; i < N; i++) {
        uint64_t ai = (uint64_t)a[i];
        if (!ai) continue;
        for (int j = 0; j < N - i; j++) acc[i + j] += ai * (uint64_t)b[j];
        for (int j = N - i; j < N; j++) acc[i + j - N] -= ai * (uint64_t)b[j];
    }
    memcpy(out, acc, sizeof(uint64_t) * (size_t)N);
    free(acc);
}

/* exact product via NTT: requires sum |small_i * b_j| < p/2, i.e. |small| < 2^15 with int32 b and N <= 2^13 */
void oracle_polymul_ntt32(const int32_t *small, const int32_t *b, int32_t N, int32_t *out) {
    const gl_tables *T = gl_get_tables(N);
    uint64_t *x = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)N), *y = x + N;
    for (int j = 0; j < N; j++) {
        x[j] = gl_from_i64(small[j]);
        y[j] = gl_from_i64(b[j]);
    }
    gl_ntt_fwd(x, T);
    gl_ntt_fwd(y, T);
    for (int j = 0; j < N; j++) x[j] = gl_mul(x[j], y[j]);
    gl_ntt_inv(x, T);
    for (int j = 0; j < N; j++) out[j] = (int32_t)(uint32_t)(uint64_t)gl_to_centered(x[j]);
    free(x);
}
/* Torus64: b = b_hi*2^32 + b_lo (b_lo unsigned 32, b_hi signed 32); both partial products exact */
void oracle_polymul_ntt64(const int64_t *small, const int64_t *b, int32_t N, int64_t *out) {
    const gl_tables *T = gl_get_tables(N);
    uint64_t *x = (uint64_t *)malloc(sizeof(uint64_t) * 3 * (size_t)N), *lo = x + N, *hi = x + 2 * N;
    for (int j = 0; j < N; j++) {
        x[j] = gl_from_i64(small[j]);
        lo[j] = (uint64_t)b[j] & 0xFFFFFFFFull;
        hi[j] = gl_from_i64(b[j] >> 32);
    }
    gl_ntt_fwd(x, T);
    gl_ntt_fwd(lo, T);
    gl_ntt_fwd(hi, T);
    for (int j = 0; j < N; j++) {
        lo[j] = gl_mul(x[j], lo[j]);
        hi[j] = gl_mul(x[j], hi[j]);
    }
    gl_ntt_inv(lo, T);
    gl_ntt_inv(hi, T);
    for (int j = 0; j < N; j++)
        out[j] = (int64_t)((uint64_t)gl_to_centered(lo[j]) + ((uint64_t)gl_to_centered(hi[j]) << 32));
    free(x);
}

/* t64tot32(d) = trunc(Int32, d / 2^32): Int64 -> Float64 (round-to-nearest-even), exact divide,
 * truncate toward zero.  J/numeric-functions.jl:109-111.  (Julia would throw InexactError when
 * the double reaches 2^31, i.e. d >= 2^63-512; we wrap that measure-zero case to INT32_MIN.) */
int32_t oracle_t64tot32(int64_t d) {
    double v = (double)d / 4294967296.0;
    v = trunc(v);
    if (v >= 2147483648.0) return INT32_MIN;
    return (int32_t)v;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* the caller knows the CPU share it really has (affinity mask AND cgroup quota); OpenMP only sees the affinity mask */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ============================================================================================
 * Engine 2: the reference's double-precision transform            J/polynomials.jl:81-247
 *   ForwardTransformPlan (:81-95): coeffs[k] = exp(-2 pi i k / (2N)), plan_fft of length N/2
 *   forward_transform (:208-214): buffer = (c[1:N/2] - i c[N/2+1:N]) .* coeffs ; fft(buffer)
 *   inverse_transform (:224-242): buffer = ifft(x) ; buffer = conj(buffer) .* coeffs ; to_int32 of real / imaginary parts
 *   to_int32(Float64) (:217-218): round(Int64, x) (ties to even), low 32 bits
 * fft = sum_j x_j exp(-2 pi i j k / M) (Julia / FFTW sign convention), ifft its inverse with 1/M.
 * ========================================================================================== */
typedef struct {
    double re, im;
} cd;
typedef struct {
    int M;      /* N / 2 */
    cd *w;      /* exp(-2 pi i t / M), t < M/2 */
    cd *twist;  /* exp(-2 pi i k / (2N)), k < M */
    int *rev;
} fft_plan;
static fft_plan g_fft[2];
static const fft_plan *fft_get_plan(int N) {
    fft_plan *P = &g_fft[N == 1024 ? 0 : 1];
    if (P->M) return P;
#pragma omp critical(fft_plan_init)
    if (!P->M) {
        const int M = N / 2;
        int lg = 0;
        while ((1 << lg) < M) lg++;
        cd *w = (cd *)malloc(sizeof(cd) * (size_t)(M / 2)), *tw = (cd *)malloc(sizeof(cd) * (size_t)M);
        int *rev = (int *)malloc(sizeof(int) * (size_t)M);
        const long double PI = 3.14159265358979323846264338327950288L;
        for (int t = 0; t < M / 2; t++) w[t] = (cd){(double)cosl(-2.0L * PI * t / M), (double)sinl(-2.0L * PI * t / M)};
        for (int k = 0; k < M; k++) tw[k] = (cd){(double)cosl(-PI * k / N), (double)sinl(-PI * k / N)};
        for (int k = 0; k < M; k++) {
            int r = 0;
            for (int b = 0; b < lg; b++) r |= ((k >> b) & 1) << (lg - 1 - b);
            rev[k] = r;
        }
        P->w = w, P->twist = tw, P->rev = rev;
#pragma omp flush
        P->M = M;
    }
    return P;
}
/* in-place radix-2 decimation in time; sign = -1: fft, +1: unnormalised ifft */
static void fft_run(cd *a, const fft_plan *P, int sign) {
    const int M = P->M;
    for (int k = 0; k < M; k++) {
        const int r = P->rev[k];
        if (r > k) {
            cd t = a[k];
            a[k] = a[r];
            a[r] = t;
        }
    }
    for (int half = 1; half < M; half <<= 1) {
        const int step = M / (2 * half);
        for (int base = 0; base < M; base += 2 * half)
            for (int j = 0; j < half; j++) {
                const cd w = P->w[j * step];
                const double wi = sign < 0 ? w.im : -w.im;
                cd *x = a + base + j, *y = x + half;
                const double tr = y->re * w.re - y->im * wi, ti = y->re * wi + y->im * w.re;
                y->re = x->re - tr, y->im = x->im - ti;
                x->re += tr, x->im += ti;
            }
    }
}
static void fft_forward_transform(const int32_t *c, int N, cd *out) { /* J/polynomials.jl:208-214 */
    const fft_plan *P = fft_get_plan(N);
    const int M = N / 2;
    for (int k = 0; k < M; k++) {
        const double re = (double)c[k], im = -(double)c[k + M];
        out[k] = (cd){re * P->twist[k].re - im * P->twist[k].im, re * P->twist[k].im + im * P->twist[k].re};
    }
    fft_run(out, P, -1);
}
static int32_t fft_to_int32(double x) { /* :217-218 */
    return (int32_t)(uint32_t)(uint64_t)(int64_t)nearbyint(x);
}
static void fft_inverse_transform(cd *x, int N, int32_t *out) { /* J/polynomials.jl:224-242; x is consumed */
    const fft_plan *P = fft_get_plan(N);
    const int M = N / 2;
    fft_run(x, P, +1);
    const double inv = 1.0 / M;
    for (int k = 0; k < M; k++) {
        const double re = x[k].re * inv, im = -x[k].im * inv; /* conj(buffer) */
        out[k] = fft_to_int32(re * P->twist[k].re - im * P->twist[k].im);
        out[k + M] = fft_to_int32(re * P->twist[k].im + im * P->twist[k].re);
    }
}

/* transformed_mul(x::IntPolynomial, y::TorusPolynomial): inverse_transform(forward_transform(x) * forward_transform(y))     J/polynomials.jl:245-247
 * (the product src/ntt-test.cpp:57-93 compares with the schoolbook one on its low 31 bits) */
void oracle_fft_polymul32(const int32_t *x, const int32_t *y, int32_t N, int32_t *out) {
    const int M = N / 2;
    cd *a = (cd *)malloc(sizeof(cd) * 2 * (size_t)M), *b = a + M;
    fft_forward_transform(x, N, a);
    fft_forward_transform(y, N, b);
    for (int j = 0; j < M; j++) a[j] = (cd){a[j].re * b[j].re - a[j].im * b[j].im, a[j].re * b[j].im + a[j].im * b[j].re};
    fft_inverse_transform(a, N, out);
    free(a);
}

/* ============================================================================================
 * single-key context
 * ========================================================================================== */
struct oracle_ctx {
    oracle_params p;
    const int32_t *bk;  /* borrowed */
    const int32_t *ksk; /* borrowed */
    uint64_t *bk_ntt;   /* [n][(k+1)l][k+1][N] NTT domain */
    void *bk_fft;       /* engine 2: [n][(k+1)l][k+1][N/2] Complex{Float64} = BootstrapKey's TransformedTGswSample (J/bootstrap.jl:11-12), built on first use */
};

oracle_ctx *oracle_ctx_create(const oracle_params *p, const int32_t *bk, const int32_t *ksk) {
    if (p->torus_bits != 32 || p->parties != 1) return NULL;
    oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof(*c));
    c->p = *p;
    c->bk = bk;
    c->ksk = ksk;
    const int N = p->N;
    const gl_tables *T = gl_get_tables(N);
    size_t polys = (size_t)p->n * (p->k + 1) * p->l * (p->k + 1);
    c->bk_ntt = (uint64_t *)malloc(polys * N * sizeof(uint64_t));
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)polys; q++) {
        uint64_t *dst = c->bk_ntt + (size_t)q * N;
        const int32_t *src = bk + (size_t)q * N;
        for (int j = 0; j < N; j++) dst[j] = gl_from_i64(src[j]);
        gl_ntt_fwd(dst, T);
    }
    return c;
}
void oracle_ctx_destroy(oracle_ctx *c) {
    if (!c) return;
    free(c->bk_ntt);
    free(c->bk_fft);
    free(c);
}
static const cd *ctx_bk_fft(const oracle_ctx *cc) {
    oracle_ctx *c = (oracle_ctx *)cc;
    if (c->bk_fft) return (const cd *)c->bk_fft;
#pragma omp critical(bk_fft_init)
    if (!c->bk_fft) {
        const int N = c->p.N;
        const size_t polys = (size_t)c->p.n * (c->p.k + 1) * c->p.l * (c->p.k + 1);
        cd *t = (cd *)malloc(polys * (size_t)(N / 2) * sizeof(cd));
        for (size_t q = 0; q < polys; q++) fft_forward_transform(c->bk + q * N, N, t + q * (size_t)(N / 2));
#pragma omp flush
        c->bk_fft = t;
    }
    return (const cd *)c->bk_fft;
}

/* tgsw_extern_mul: out[c] = sum_{j,p} digit_p(tmp[j]) (*) BK_i[j*l+p][c]     J/tgsw.jl:146-156 */
static void extern_mul32(const oracle_ctx *c, int32_t i, const int32_t *tmp, int32_t *out, int use_schoolbook) {
    const int N = c->p.N, k = c->p.k, l = c->p.l, rows = (k + 1) * l;
    int32_t *dig = (int32_t *)scr(SCR_EXT_DIG, sizeof(int32_t) * (size_t)rows * N);
    for (int j = 0; j <= k; j++) oracle_decompose32(tmp + (size_t)j * N, N, l, c->p.Bgbit, dig + (size_t)j * l * N);
    if (use_schoolbook == 2) { /* tgsw_extern_mul as the reference runs it: J/tgsw.jl:146-150 on J/polynomials.jl:208-247 */
        const int M = N / 2;
        const cd *bk = ctx_bk_fft(c) + (size_t)i * rows * (k + 1) * M;
        cd *d = (cd *)scr(SCR_EXT_FFT, sizeof(cd) * (size_t)(rows + 1) * M), *acc = d + (size_t)rows * M;
        for (int r = 0; r < rows; r++) fft_forward_transform(dig + (size_t)r * N, N, d + (size_t)r * M);
        for (int cc = 0; cc <= k; cc++) {
            for (int j = 0; j < M; j++) {
                double sr = 0.0, si = 0.0;
                for (int r = 0; r < rows; r++) {
                    const cd x = d[(size_t)r * M + j], y = bk[((size_t)r * (k + 1) + cc) * M + j];
                    sr += x.re * y.re - x.im * y.im;
                    si += x.re * y.im + x.im * y.re;
                }
                acc[j] = (cd){sr, si};
            }
            fft_inverse_transform(acc, N, out + (size_t)cc * N);
        }
        return;
    }
    if (use_schoolbook) {
        int32_t *prod = (int32_t *)scr(SCR_EXT_PROD, sizeof(int32_t) * N);
        memset(out, 0, sizeof(int32_t) * (size_t)(k + 1) * N);
        for (int r = 0; r < rows; r++)
            for (int cc = 0; cc <= k; cc++) {
                const int32_t *row = c->bk + (((size_t)i * rows + r) * (k + 1) + cc) * N;
                oracle_polymul_schoolbook32(dig + (size_t)r * N, row, N, prod);
                for (int j = 0; j < N; j++)
                    out[(size_t)cc * N + j] = (int32_t)((uint32_t)out[(size_t)cc * N + j] + (uint32_t)prod[j]);
            }
    } else {
        const gl_tables *T = gl_get_tables(N);
        uint64_t *d = (uint64_t *)scr(SCR_EXT_NTT, sizeof(uint64_t) * (size_t)(rows + k + 1) * N), *acc = d + (size_t)rows * N;
        for (int r = 0; r < rows; r++) {
            for (int j = 0; j < N; j++) d[(size_t)r * N + j] = gl_from_i64(dig[(size_t)r * N + j]);
            gl_ntt_fwd(d + (size_t)r * N, T);
        }
        for (int cc = 0; cc <= k; cc++) {
            uint64_t *a = acc + (size_t)cc * N;
            for (int j = 0; j < N; j++) {
                /* lazy accumulation of up to 8 products in 128 bits would overflow; reduce each */
                uint64_t s = 0;
                for (int r = 0; r < rows; r++)
                    s = gl_add(s, gl_mul(d[(size_t)r * N + j], c->bk_ntt[(((size_t)i * rows + r) * (k + 1) + cc) * N + j]));
                a[j] = s;
            }
            gl_ntt_inv(a, T);
            for (int j = 0; j < N; j++) out[(size_t)cc * N + j] = (int32_t)(uint32_t)(uint64_t)gl_to_centered(a[j]);
        }
    }
}

/* mux_rotate: acc += BK_i (.) (X^barai * acc - acc)        J/bootstrap.jl:19-23 */
void oracle_mux_rotate(const oracle_ctx *c, int32_t i, int32_t barai, int32_t *acc, int use_schoolbook) {
    const int N = c->p.N, k = c->p.k;
    size_t sz = (size_t)(k + 1) * N;
    int32_t *tmp = (int32_t *)scr(SCR_MUX, sizeof(int32_t) * 2 * sz), *ext = tmp + sz;
    for (int j = 0; j <= k; j++) {
        oracle_mul_by_monomial32(acc + (size_t)j * N, barai, N, tmp + (size_t)j * N);
        for (int q = 0; q < N; q++)
            tmp[(size_t)j * N + q] = (int32_t)((uint32_t)tmp[(size_t)j * N + q] - (uint32_t)acc[(size_t)j * N + q]);
    }
    extern_mul32(c, i, tmp, ext, use_schoolbook);
    for (size_t q = 0; q < sz; q++) acc[q] = (int32_t)((uint32_t)acc[q] + (uint32_t)ext[q]);
}

/* rlwe_extract_sample: a'_0 = a_0, a'_j = -a_{N-j}, b' = body_0     J/rlwe.jl:64-68, J/polynomials.jl:69-72 */
static void extract32(const int32_t *acc, int N, int k, int32_t *out) {
    for (int m = 0; m < k; m++) {
        const int32_t *a = acc + (size_t)m * N;
        out[(size_t)m * N] = a[0];
        for (int j = 1; j < N; j++) out[(size_t)m * N + j] = (int32_t)(0u - (uint32_t)a[N - j]);
    }
    out[(size_t)k * N] = acc[(size_t)k * N];
}

/* bootstrap_wo_keyswitch + blind_rotate_and_extract + blind_rotate     J/bootstrap.jl:38-88 */
void oracle_bootstrap_wo_keyswitch(const oracle_ctx *c, int32_t mu, const int32_t *x, int32_t *out, int use_schoolbook) {
    const int N = c->p.N, k = c->p.k, n = c->p.n;
    int32_t barb = oracle_modswitch(x[n], N);
    size_t sz = (size_t)(k + 1) * N;
    int32_t *acc = (int32_t *)scr(SCR_BOOT, sizeof(int32_t) * (sz + N)), *tv = acc + sz;
    memset(acc, 0, sizeof(int32_t) * sz);
    for (int j = 0; j < N; j++) tv[j] = mu;
    oracle_mul_by_monomial32(tv, -barb, N, acc + (size_t)k * N); /* acc = (0, X^{-barb} * testvect) */
    for (int i = 0; i < n; i++) {
        int32_t bara = oracle_modswitch(x[i], N);
        if (bara != 0) oracle_mux_rotate(c, i, bara, acc, use_schoolbook);
    }
    extract32(acc, N, k, out);
}

/* keyswitch      J/keyswitch.jl:45-80 */
static void keyswitch_with(const int32_t *ksk, int Nin, int n, int t, int basebit, const int32_t *in, int32_t b_init, int32_t *out) {
    const int base = 1 << basebit;
    const uint32_t mask = (uint32_t)base - 1u;
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    uint32_t *res = (uint32_t *)scr(SCR_KS, sizeof(uint32_t) * ((size_t)n + 1));
    memset(res, 0, sizeof(uint32_t) * (size_t)n);
    res[n] = (uint32_t)b_init;
    for (int i = 0; i < Nin; i++) {
        uint32_t aibar = (uint32_t)in[i] + prec_offset;
        for (int j = 1; j <= t; j++) {
            uint32_t d = (aibar >> (32 - j * basebit)) & mask;
            if (d != 0) {
                const int32_t *row = ksk + ((((size_t)i * t + (j - 1)) * (base - 1)) + (d - 1)) * ((size_t)n + 1);
                for (int q = 0; q <= n; q++) res[q] -= (uint32_t)row[q];
            }
        }
    }
    memcpy(out, res, sizeof(uint32_t) * ((size_t)n + 1));
}
void oracle_keyswitch(const oracle_ctx *c, const int32_t *in, int32_t *out) {
    const int Nin = c->p.N * c->p.k;
    keyswitch_with(c->ksk, Nin, c->p.n, c->p.ks_t, c->p.ks_basebit, in, in[Nin], out);
}

/* gate linear prologues: temp = (0, cb) + cx*x + cy*y      J/gates.jl:15-161 ; MUX J/gates.jl:163-177 */
typedef struct {
    int32_t cb, cx, cy;
} lin_t;
static int gate_lin(int op, int which, lin_t *L) {
    const int32_t E8 = 1 << 29, E4 = 1 << 30; /* encode_message(1,8), encode_message(1,4)   J/numeric-functions.jl:86-89 */
    switch (op) {
    case OR_GATE_NAND: *L = (lin_t){E8, -1, -1}; return 0;
    case OR_GATE_OR: *L = (lin_t){E8, 1, 1}; return 0;
    case OR_GATE_AND: *L = (lin_t){-E8, 1, 1}; return 0;
    case OR_GATE_XOR: *L = (lin_t){E4, 2, 2}; return 0;
    case OR_GATE_XNOR: *L = (lin_t){-E4, -2, -2}; return 0;
    case OR_GATE_NOR: *L = (lin_t){-E8, -1, -1}; return 0;
    case OR_GATE_ANDNY: *L = (lin_t){-E8, -1, 1}; return 0;
    case OR_GATE_ANDYN: *L = (lin_t){-E8, 1, -1}; return 0;
    case OR_GATE_ORNY: *L = (lin_t){E8, -1, 1}; return 0;
    case OR_GATE_ORYN: *L = (lin_t){E8, 1, -1}; return 0;
    case OR_GATE_MUX: *L = which == 0 ? (lin_t){-E8, 1, 1} /* AND(x,y) */ : (lin_t){-E8, -1, 1} /* AND(NOT x, z) */; return 0;
    default: return -1;
    }
}
int oracle_gate_prologue(const oracle_params *p, int op, int which, const int32_t *in0, const int32_t *in1,
                         const int32_t *in2, int32_t *tmp) {
    lin_t L;
    if (gate_lin(op, which, &L)) return -1;
    const int32_t *y = (op == OR_GATE_MUX && which == 1) ? in2 : in1;
    int words = p->n * p->parties;
    for (int q = 0; q <= words; q++) {
        uint32_t v = (uint32_t)L.cx * (uint32_t)in0[q] + (uint32_t)L.cy * (uint32_t)y[q];
        if (q == words) v += (uint32_t)L.cb;
        tmp[q] = (int32_t)v;
    }
    return 0;
}

int oracle_gates(const oracle_ctx *c, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                 int32_t *out, size_t count, int use_schoolbook) {
    const int n = c->p.n, Nk = c->p.N * c->p.k;
    const size_t rec = (size_t)n + 1;
    const int32_t MU = 1 << 29;
    if (op == OR_GATE_NOT || op == OR_GATE_COPY) { /* J/gates.jl:76-79: not bootstrapped */
        for (size_t g = 0; g < count * rec; g++) out[g] = op == OR_GATE_NOT ? (int32_t)(0u - (uint32_t)in0[g]) : in0[g];
        return 0;
    }
    lin_t L;
    if (gate_lin(op, 0, &L)) return -1;
    int32_t *res = (int32_t *)malloc(sizeof(int32_t) * count * rec); /* tolerate out aliasing an input */
    /* one thread per gate: the reference's only parallel pattern (src/KNN_medical_data.cpp:681) */
#pragma omp parallel for schedule(dynamic)
    for (long g = 0; g < (long)count; g++) {
        int32_t *tmp = (int32_t *)scr(SCR_GATE, sizeof(int32_t) * (rec + 2 * ((size_t)Nk + 1)));
        int32_t *u1 = tmp + rec, *u2 = u1 + Nk + 1;
        const int32_t *x = in0 + g * rec, *y = in1 + g * rec, *z = in2 ? in2 + g * rec : NULL;
        if (op != OR_GATE_MUX) {
            oracle_gate_prologue(&c->p, op, 0, x, y, z, tmp);
            oracle_bootstrap_wo_keyswitch(c, MU, tmp, u1, use_schoolbook);
            oracle_keyswitch(c, u1, res + g * rec); /* bootstrap = wo_keyswitch + keyswitch, J/bootstrap.jl:98-101 */
        } else {
            oracle_gate_prologue(&c->p, op, 0, x, y, z, tmp);
            oracle_bootstrap_wo_keyswitch(c, MU, tmp, u1, use_schoolbook);
            oracle_gate_prologue(&c->p, op, 1, x, y, z, tmp);
            oracle_bootstrap_wo_keyswitch(c, MU, tmp, u2, use_schoolbook);
            for (int q = 0; q <= Nk; q++) u1[q] = (int32_t)((uint32_t)u1[q] + (uint32_t)u2[q]); /* t3 = (0,1/8)+u1+u2 */
            u1[Nk] = (int32_t)((uint32_t)u1[Nk] + (uint32_t)MU);
            oracle_keyswitch(c, u1, res + g * rec);
        }
    }
    memcpy(out, res, sizeof(int32_t) * count * rec);
    free(res);
    return 0;
}

/* ============================================================================================
 * 3-gen multi-key context
 * ========================================================================================== */
struct oracle_mk_ctx {
    oracle_params p;
    const int64_t *bk;
    const int32_t *ksk;
    /* NTT of the limbs of every BK polynomial: two 32-bit limbs, or four 16-bit limbs when the digits are so wide (the l = 1, Bgbit 24-27
     * sets of 16+ parties, J/mk_api.jl:214-296) that a digit x 32-bit-limb sum would leave the +-p/2 range of the exact NTT product */
    int limbs, limb_bits;
    uint64_t *bk_limb[4];
};

oracle_mk_ctx *oracle_mk_ctx_create(const oracle_params *p, const int64_t *bk, const int32_t *ksk) {
    if (p->torus_bits != 64 || p->k != 1) return NULL;
    oracle_mk_ctx *c = (oracle_mk_ctx *)calloc(1, sizeof(*c));
    c->p = *p;
    c->bk = bk;
    c->ksk = ksk;
    const int N = p->N;
    const gl_tables *T = gl_get_tables(N);
    size_t polys = (size_t)p->parties * p->n * 4 * p->l;
    /* |sum| <= 2 l N 2^(Bgbit-1) 2^limb_bits must stay below 2^62 */
    int lg = 0;
    while ((1 << lg) < 2 * p->l * N) lg++;
    c->limb_bits = (lg + p->Bgbit - 1 + 32 <= 62) ? 32 : 16;
    c->limbs = 64 / c->limb_bits;
    if (lg + p->Bgbit - 1 + c->limb_bits > 62) {
        free(c);
        return NULL;
    }
    for (int h = 0; h < c->limbs; h++) c->bk_limb[h] = (uint64_t *)malloc(polys * N * sizeof(uint64_t));
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)polys; q++) {
        const int64_t *src = bk + (size_t)q * N;
        for (int h = 0; h < c->limbs; h++) {
            uint64_t *dst = c->bk_limb[h] + (size_t)q * N;
            const int sh = h * c->limb_bits;
            for (int j = 0; j < N; j++)   /* the top limb is signed, the others unsigned: v = sum_h limb_h 2^(h limb_bits) */
                dst[j] = h == c->limbs - 1 ? gl_from_i64(src[j] >> sh) : (((uint64_t)src[j] >> sh) & ((1ull << c->limb_bits) - 1ull));
            gl_ntt_fwd(dst, T);
        }
    }
    return c;
}
void oracle_mk_ctx_destroy(oracle_mk_ctx *c) {
    if (!c) return;
    for (int h = 0; h < 4; h++) free(c->bk_limb[h]);
    free(c);
}

/* tgsw_extern_mul_3gen       J/tgsw_3gen.jl:102-113
 * acc = [c1 (mask), c0 (body)];  c0' = S g(c0)_l*P1_l + g(c1)_l*P2_l ; c1' = S g(c0)_l*P4_l + g(c1)_l*P3_l */
static void mk_extern_mul(const oracle_mk_ctx *c, int party, int i, const int64_t *tmp, int64_t *out, int use_schoolbook) {
    const int N = c->p.N, l = c->p.l;
    const size_t key_off = ((size_t)party * c->p.n + i) * 4 * l; /* in polynomials */
    int64_t *dig = (int64_t *)scr(SCR_MKEXT_DIG, sizeof(int64_t) * 2 * (size_t)l * N);
    int64_t *g_c1 = dig, *g_c0 = dig + (size_t)l * N;
    oracle_decompose64(tmp, N, l, c->p.Bgbit, g_c1);     /* tmp[0] = c1 (mask) */
    oracle_decompose64(tmp + N, N, l, c->p.Bgbit, g_c0); /* tmp[1] = c0 (body) */
    /* which part multiplies which digit set, per output: out[1]=c0' : (g_c0,P1) (g_c1,P2); out[0]=c1' : (g_c0,P4) (g_c1,P3) */
    const int part_for[2][2] = {{3, 2}, {0, 1}}; /* [out poly][0: g_c0, 1: g_c1] -> part index 0..3 */
    if (use_schoolbook) {
        int64_t *prod = (int64_t *)scr(SCR_MKEXT_PROD, sizeof(int64_t) * N);
        memset(out, 0, sizeof(int64_t) * 2 * (size_t)N);
        for (int o = 0; o < 2; o++)
            for (int w = 0; w < 2; w++)
                for (int q = 0; q < l; q++) {
                    const int64_t *d = (w == 0 ? g_c0 : g_c1) + (size_t)q * N;
                    const int64_t *row = c->bk + (key_off + (size_t)part_for[o][w] * l + q) * N;
                    oracle_polymul_schoolbook64(d, row, N, prod);
                    for (int j = 0; j < N; j++) out[(size_t)o * N + j] = (int64_t)((uint64_t)out[(size_t)o * N + j] + (uint64_t)prod[j]);
                }
    } else {
        const gl_tables *T = gl_get_tables(N);
        uint64_t *d = (uint64_t *)scr(SCR_MKEXT_NTT, sizeof(uint64_t) * (2 * (size_t)l + 4) * N);
        uint64_t *al = d + 2 * (size_t)l * N;   /* one sum per key limb */
        const int H = c->limbs;
        for (int r = 0; r < 2 * l; r++) {
            for (int j = 0; j < N; j++) d[(size_t)r * N + j] = gl_from_i64(dig[(size_t)r * N + j]);
            gl_ntt_fwd(d + (size_t)r * N, T);
        }
        for (int o = 0; o < 2; o++) {
            for (int j = 0; j < N; j++) {
                uint64_t sum[4] = {0, 0, 0, 0};
                for (int w = 0; w < 2; w++)
                    for (int q = 0; q < l; q++) {
                        uint64_t dv = d[((size_t)(w == 0 ? l : 0) + q) * N + j]; /* g_c0 stored second */
                        size_t ro = (key_off + (size_t)part_for[o][w] * l + q) * N + j;
                        for (int h = 0; h < H; h++) sum[h] = gl_add(sum[h], gl_mul(dv, c->bk_limb[h][ro]));
                    }
                for (int h = 0; h < H; h++) al[(size_t)h * N + j] = sum[h];
            }
            for (int h = 0; h < H; h++) gl_ntt_inv(al + (size_t)h * N, T);
            for (int j = 0; j < N; j++) {
                uint64_t v = 0;
                for (int h = 0; h < H; h++) v += (uint64_t)gl_to_centered(al[(size_t)h * N + j]) << (h * c->limb_bits);
                out[(size_t)o * N + j] = (int64_t)v;
            }
        }
    }
}

/* mk_mux_rotate_3gen      J/3gen_mk_internals.jl:59-62 */
void oracle_mk_mux_rotate(const oracle_mk_ctx *c, int32_t party, int32_t i, int32_t barai, int64_t *acc, int use_schoolbook) {
    const int N = c->p.N;
    int64_t *tmp = (int64_t *)scr(SCR_MKMUX, sizeof(int64_t) * 4 * (size_t)N), *ext = tmp + 2 * (size_t)N;
    for (int m = 0; m < 2; m++) {
        oracle_mul_by_monomial64(acc + (size_t)m * N, barai, N, tmp + (size_t)m * N);
        for (int q = 0; q < N; q++) tmp[(size_t)m * N + q] = (int64_t)((uint64_t)tmp[(size_t)m * N + q] - (uint64_t)acc[(size_t)m * N + q]);
    }
    mk_extern_mul(c, party, i, tmp, ext, use_schoolbook);
    for (int q = 0; q < 2 * N; q++) acc[q] = (int64_t)((uint64_t)acc[q] + (uint64_t)ext[q]);
}

/* mk_bootstrap_wo_keyswitch_3gen / mk_blind_rotate_and_extract_3gen / mk_blind_rotate_3gen
 * J/3gen_mk_internals.jl:66-109 ; extraction rlwe_extract_sample_64 J/rlwe.jl:70-74 */
void oracle_mk_bootstrap_wo_keyswitch(const oracle_mk_ctx *c, int64_t mu, const int32_t *x, int32_t *out, int use_schoolbook) {
    const int N = c->p.N, n = c->p.n, P = c->p.parties;
    int32_t barb = oracle_modswitch(x[(size_t)n * P], N);
    int64_t *acc = (int64_t *)scr(SCR_MKBOOT, sizeof(int64_t) * 3 * (size_t)N), *tv = acc + 2 * (size_t)N;
    memset(acc, 0, sizeof(int64_t) * 2 * (size_t)N);
    for (int j = 0; j < N; j++) tv[j] = mu;
    oracle_mul_by_monomial64(tv, -barb, N, acc + N);
    for (int p = 0; p < P; p++) /* parties outer, key index inner */
        for (int i = 0; i < n; i++) {
            int32_t bara = oracle_modswitch(x[(size_t)p * n + i], N);
            if (bara != 0) oracle_mk_mux_rotate(c, p, i, bara, acc, use_schoolbook);
        }
    out[0] = oracle_t64tot32(acc[0]);
    for (int j = 1; j < N; j++) out[j] = oracle_t64tot32((int64_t)(0ull - (uint64_t)acc[N - j]));
    out[N] = oracle_t64tot32(acc[N]);
}

/* mk_keyswitch_3gen       J/mk_internals.jl:730-744 */
void oracle_mk_keyswitch(const oracle_mk_ctx *c, const int32_t *in, int32_t *out) {
    const int N = c->p.N, n = c->p.n, P = c->p.parties, t = c->p.ks_t, bb = c->p.ks_basebit;
    const size_t per_party = (size_t)N * t * ((1 << bb) - 1) * ((size_t)n + 1);
    int32_t *part = (int32_t *)scr(SCR_MKKS, sizeof(int32_t) * ((size_t)n + 1));
    uint32_t b = (uint32_t)in[N];
    for (int p = 0; p < P; p++) {
        keyswitch_with(c->ksk + (size_t)p * per_party, N, n, t, bb, in, 0, part); /* keyswitch(ks[p], (a, 0)) */
        memcpy(out + (size_t)p * n, part, sizeof(int32_t) * n);
        b += (uint32_t)part[n];
    }
    out[(size_t)n * P] = (int32_t)b;
}

int oracle_mk_gates(const oracle_mk_ctx *c, int op, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                    int32_t *out, size_t count, int use_schoolbook) {
    const int n = c->p.n, N = c->p.N, P = c->p.parties;
    const size_t rec = (size_t)n * P + 1;
    const int64_t MU = (int64_t)1 << 61; /* encode_message64(1, 8)   J/numeric-functions.jl:92-95 */
    const int32_t E8 = 1 << 29, E4 = 1 << 30;
    if (op == OR_GATE_NOT || op == OR_GATE_COPY) {
        for (size_t g = 0; g < count * rec; g++) out[g] = op == OR_GATE_NOT ? (int32_t)(0u - (uint32_t)in0[g]) : in0[g];
        return 0;
    }
    if (!(op == OR_GATE_NAND || op == OR_GATE_OR || op == OR_GATE_AND || op == OR_GATE_XOR || op == OR_GATE_AND3 || op == OR_GATE_MUX))
        return -1; /* only the gates J/3gen_mk_gates.jl defines */
    int32_t *res = (int32_t *)malloc(sizeof(int32_t) * count * rec);
#pragma omp parallel for schedule(dynamic)
    for (long g = 0; g < (long)count; g++) {
        int32_t *tmp = (int32_t *)scr(SCR_MKGATE, sizeof(int32_t) * (2 * rec + (size_t)N + 1));
        int32_t *t2 = tmp + rec, *u = t2 + rec;
        const int32_t *x = in0 + g * rec, *y = in1 + g * rec, *z = in2 ? in2 + g * rec : NULL;
        if (op == OR_GATE_AND3) { /* J/3gen_mk_gates.jl:55-64 */
            for (size_t q = 0; q < rec; q++) tmp[q] = (int32_t)((uint32_t)x[q] + (uint32_t)y[q] + (uint32_t)z[q]);
            tmp[rec - 1] = (int32_t)((uint32_t)tmp[rec - 1] - (uint32_t)E4);
            oracle_mk_bootstrap_wo_keyswitch(c, MU, tmp, u, use_schoolbook);
            oracle_mk_keyswitch(c, u, res + g * rec);
        } else if (op == OR_GATE_MUX) { /* J/3gen_mk_gates.jl:133-150: two full ANDs, then linear, no final bootstrap */
            for (size_t q = 0; q < rec; q++) tmp[q] = (int32_t)((uint32_t)x[q] + (uint32_t)y[q]);
            tmp[rec - 1] = (int32_t)((uint32_t)tmp[rec - 1] - (uint32_t)E8);
            oracle_mk_bootstrap_wo_keyswitch(c, MU, tmp, u, use_schoolbook);
            oracle_mk_keyswitch(c, u, t2);
            for (size_t q = 0; q < rec; q++) tmp[q] = (int32_t)((uint32_t)z[q] - (uint32_t)x[q]);
            tmp[rec - 1] = (int32_t)((uint32_t)tmp[rec - 1] - (uint32_t)E8);
            oracle_mk_bootstrap_wo_keyswitch(c, MU, tmp, u, use_schoolbook);
            oracle_mk_keyswitch(c, u, tmp);
            for (size_t q = 0; q < rec; q++) res[g * rec + q] = (int32_t)((uint32_t)tmp[q] + (uint32_t)t2[q]);
            res[g * rec + rec - 1] = (int32_t)((uint32_t)res[g * rec + rec - 1] + (uint32_t)E8);
        } else {
            lin_t L;
            gate_lin(op, 0, &L);
            for (size_t q = 0; q < rec; q++) tmp[q] = (int32_t)((uint32_t)L.cx * (uint32_t)x[q] + (uint32_t)L.cy * (uint32_t)y[q]);
            tmp[rec - 1] = (int32_t)((uint32_t)tmp[rec - 1] + (uint32_t)L.cb);
            oracle_mk_bootstrap_wo_keyswitch(c, MU, tmp, u, use_schoolbook);
            oracle_mk_keyswitch(c, u, res + g * rec);
        }
    }
    memcpy(out, res, sizeof(int32_t) * count * rec);
    free(res);
    return 0;
}

/* ============================================================================================
 * deterministic RNG, key generation, encryption (host side; mirrors the reference constructors
 * but with OUR OWN random streams -- Julia's MersenneTwister stream is not reproducible here)
 * ========================================================================================== */
typedef struct {
    uint64_t s[4];
    int have_spare;
    double spare;
} rng_t;
static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void rng_init(rng_t *r, uint64_t seed, uint64_t stream) {
    uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ULL + 0x2545F4914F6CDD1DULL);
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
    r->have_spare = 0;
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t rng_u64(rng_t *r) { /* xoshiro256** */
    uint64_t *s = r->s, result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}
static double rng_unit(rng_t *r) { return ((double)(rng_u64(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static double rng_gauss(rng_t *r) { /* Box-Muller */
    if (r->have_spare) {
        r->have_spare = 0;
        return r->spare;
    }
    double u = rng_unit(r), v = rng_unit(r), m = sqrt(-2.0 * log(u));
    r->spare = m * sin(6.283185307179586476925 * v);
    r->have_spare = 1;
    return m * cos(6.283185307179586476925 * v);
}
static int32_t dtot32(double d) { return (int32_t)(int64_t)trunc(d * 4294967296.0); } /* J/numeric-functions.jl:101-103 */
static int64_t dtot64(double d) { return (int64_t)trunc(d * 18446744073709551616.0); }  /* :105-107 */
static int32_t rng_ternary(rng_t *r) { /* rand_negative_binary: P(+-1) = 0.113546097609674   J/numeric-functions.jl:11-13 */
    double u = rng_unit(r);
    return u < 0.113546097609674 ? -1 : (u < 2 * 0.113546097609674 ? 1 : 0);
}

/* exact product of a small-coefficient key polynomial with a torus polynomial */
static void key_mul32(const int32_t *key, const int32_t *a, int N, int32_t *out) { oracle_polymul_ntt32(key, a, N, out); }
static void key_mul64(const int64_t *key, const int64_t *a, int N, int64_t *out) { oracle_polymul_ntt64(key, a, N, out); }

/* KeyswitchKey constructor      J/keyswitch.jl:14-41  (noise recentred over the whole table) */
static void gen_ksk(rng_t *r, const int32_t *in_key, int Nin, const int32_t *out_key, int n, int t, int basebit, double sigma, int32_t *ksk) {
    const int base = 1 << basebit;
    size_t cnt = (size_t)Nin * t * (base - 1);
    double *noise = (double *)malloc(sizeof(double) * cnt), mean = 0;
    for (size_t q = 0; q < cnt; q++) {
        noise[q] = rng_gauss(r) * sigma;
        mean += noise[q];
    }
    mean /= (double)cnt;
    for (size_t q = 0; q < cnt; q++) noise[q] -= mean;
    for (int i = 0; i < Nin; i++)
        for (int j = 1; j <= t; j++)
            for (int h = 1; h < base; h++) {
                size_t e = (((size_t)i * t + (j - 1)) * (base - 1)) + (h - 1);
                int32_t *row = ksk + e * ((size_t)n + 1);
                uint32_t msg = (uint32_t)(in_key[i] * h) << (32 - j * basebit);
                uint32_t b = msg + (uint32_t)dtot32(noise[e]);
                for (int q = 0; q < n; q++) {
                    row[q] = (int32_t)(uint32_t)rng_u64(r);
                    b += (uint32_t)row[q] * (uint32_t)out_key[q];
                }
                row[n] = (int32_t)b;
            }
    free(noise);
}

/* BootstrapKey constructor (coefficient domain)      J/bootstrap.jl:6-15, tgsw_encrypt J/tgsw.jl:88-101,
 * rlwe_encrypt_zero J/rlwe.jl:79-105, gadget J/tgsw.jl:65-85 */
void oracle_keygen_sk(const oracle_params *p, uint64_t seed, double sigma_bk, double sigma_ks,
                      const int32_t *lwe_key_in, int32_t *lwe_key, int32_t *rlwe_key, int32_t *bk, int32_t *ksk) {
    const int n = p->n, N = p->N, k = p->k, l = p->l, rows = (k + 1) * l;
    rng_t r;
    rng_init(&r, seed, 1);
    for (int i = 0; i < n; i++) lwe_key[i] = lwe_key_in ? lwe_key_in[i] : (int32_t)(rng_u64(&r) & 1);
    rng_init(&r, seed, 2);
    for (int i = 0; i < k * N; i++) rlwe_key[i] = (int32_t)(rng_u64(&r) & 1);
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; i++) {
        rng_t ri;
        rng_init(&ri, seed, 1000 + (uint64_t)i);
        int32_t *prod = (int32_t *)malloc(sizeof(int32_t) * N);
        for (int row = 0; row < rows; row++) {
            int j = row / l, q = row % l; /* block j, level q (0-based) */
            int32_t *base = bk + (((size_t)i * rows + row) * (k + 1)) * N;
            int32_t *body = base + (size_t)k * N;
            for (int t = 0; t < N; t++) body[t] = dtot32(rng_gauss(&ri) * sigma_bk);
            for (int m = 0; m < k; m++) {
                int32_t *a = base + (size_t)m * N;
                for (int t = 0; t < N; t++) a[t] = (int32_t)(uint32_t)rng_u64(&ri);
                key_mul32(rlwe_key + (size_t)m * N, a, N, prod);
                for (int t = 0; t < N; t++) body[t] = (int32_t)((uint32_t)body[t] + (uint32_t)prod[t]);
            }
            /* + message * gadget[q] on the constant coefficient of polynomial j */
            uint32_t g = (uint32_t)lwe_key[i] << (32 - (q + 1) * p->Bgbit);
            base[(size_t)j * N] = (int32_t)((uint32_t)base[(size_t)j * N] + g);
        }
        free(prod);
    }
    rng_init(&r, seed, 3);
    gen_ksk(&r, rlwe_key, k * N, lwe_key, n, p->ks_t, p->ks_basebit, sigma_ks, ksk);
}

/* 3-gen multi-key key material: flow of 3-gen-mk-tfhe/multikey_3gen.jl:15-30
 *   CRP_3gen(a_same=true) J/mk_internals.jl:120-137 ; PublicKey J/mk_internals.jl:209-245 ;
 *   CommonPubKey_3gen :266-298 ; tgsw_encrypt_3gen J/tgsw_3gen.jl:41-95 ; KeyswitchKey J/keyswitch.jl:14-41 */
void oracle_keygen_mk(const oracle_params *p, uint64_t seed, double sigma_bk, double sigma_ks,
                      int32_t *lwe_keys, int64_t *rlwe_keys, int64_t *bk, int32_t *ksk) {
    const int n = p->n, N = p->N, l = p->l, P = p->parties;
    rng_t r;
    rng_init(&r, seed, 11);
    for (int i = 0; i < P * n; i++) lwe_keys[i] = (int32_t)(rng_u64(&r) & 1); /* SecretKey_3gen: binary LWE keys */
    rng_init(&r, seed, 12);
    for (int i = 0; i < P * N; i++) rlwe_keys[i] = rng_ternary(&r); /* RLweKey(rng, params, true) */
    int64_t *crp = (int64_t *)malloc(sizeof(int64_t) * N);
    int64_t *B = (int64_t *)calloc((size_t)l * N, sizeof(int64_t));
    int64_t *prod = (int64_t *)malloc(sizeof(int64_t) * N);
    rng_init(&r, seed, 13);
    for (int t = 0; t < N; t++) crp[t] = (int64_t)rng_u64(&r); /* one common random polynomial, repeated l times */
    rng_init(&r, seed, 14);
    for (int q = 0; q < P; q++) { /* b_q[i] = z_q * a + e ; B[i] = sum_q b_q[i] */
        key_mul64(rlwe_keys + (size_t)q * N, crp, N, prod);
        for (int i = 0; i < l; i++)
            for (int t = 0; t < N; t++)
                B[(size_t)i * N + t] = (int64_t)((uint64_t)B[(size_t)i * N + t] + (uint64_t)prod[t] + (uint64_t)dtot64(rng_gauss(&r) * sigma_bk));
    }
    free(prod);
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int q = 0; q < P; q++)
        for (int i = 0; i < n; i++) {
            rng_t ri;
            rng_init(&ri, seed, 100000 + (uint64_t)q * 10000 + (uint64_t)i);
            int64_t *r1 = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)N), *r2 = r1 + N, *pr = r2 + N;
            int64_t *key = bk + (((size_t)q * n + i) * 4 * l) * N;
            int64_t m = lwe_keys[(size_t)q * n + i];
            for (int lv = 0; lv < l; lv++) {
                for (int t = 0; t < N; t++) r1[t] = rng_ternary(&ri);
                for (int t = 0; t < N; t++) r2[t] = rng_ternary(&ri);
                uint64_t g = (uint64_t)m << (64 - (lv + 1) * p->Bgbit);
                int64_t *P1 = key + ((size_t)0 * l + lv) * N, *P2 = key + ((size_t)1 * l + lv) * N;
                int64_t *P3 = key + ((size_t)2 * l + lv) * N, *P4 = key + ((size_t)3 * l + lv) * N;
                key_mul64(r1, B + (size_t)lv * N, N, pr);
                for (int t = 0; t < N; t++) P1[t] = (int64_t)((uint64_t)pr[t] + (uint64_t)dtot64(rng_gauss(&ri) * sigma_bk));
                P1[0] = (int64_t)((uint64_t)P1[0] + g);
                key_mul64(r2, B + (size_t)lv * N, N, pr);
                for (int t = 0; t < N; t++) P2[t] = (int64_t)((uint64_t)pr[t] + (uint64_t)dtot64(rng_gauss(&ri) * sigma_bk));
                key_mul64(r2, crp, N, pr);
                for (int t = 0; t < N; t++) P3[t] = (int64_t)((uint64_t)pr[t] + (uint64_t)dtot64(rng_gauss(&ri) * sigma_bk));
                P3[0] = (int64_t)((uint64_t)P3[0] + g);
                key_mul64(r1, crp, N, pr);
                for (int t = 0; t < N; t++) P4[t] = (int64_t)((uint64_t)pr[t] + (uint64_t)dtot64(rng_gauss(&ri) * sigma_bk));
            }
            free(r1);
        }
    const size_t per_party = (size_t)N * p->ks_t * ((1 << p->ks_basebit) - 1) * ((size_t)n + 1);
    int32_t *z32 = (int32_t *)malloc(sizeof(int32_t) * N);
    for (int q = 0; q < P; q++) {
        rng_init(&r, seed, 20 + (uint64_t)q);
        for (int t = 0; t < N; t++) z32[t] = (int32_t)rlwe_keys[(size_t)q * N + t];
        gen_ksk(&r, z32, N, lwe_keys + (size_t)q * n, n, p->ks_t, p->ks_basebit, sigma_ks, ksk + (size_t)q * per_party);
    }
    free(z32);
    free(crp);
    free(B);
}

/* lwe_encrypt: b = mu + e + <a, s>      J/lwe.jl:38-43 */
void oracle_lwe_encrypt(const int32_t *key, int32_t n, int32_t mu, double sigma, uint64_t seed, uint64_t idx, int32_t *rec) {
    rng_t r;
    rng_init(&r, seed, 0x10000000ULL + idx);
    uint32_t b = (uint32_t)mu + (uint32_t)dtot32(rng_gauss(&r) * sigma);
    for (int i = 0; i < n; i++) {
        rec[i] = (int32_t)(uint32_t)rng_u64(&r);
        b += (uint32_t)rec[i] * (uint32_t)key[i];
    }
    rec[n] = (int32_t)b;
}
/* lwe_phase = b - <a, s>      J/lwe.jl:59 */
int32_t oracle_lwe_phase(const int32_t *key, int32_t n, const int32_t *rec) {
    uint32_t ph = (uint32_t)rec[n];
    for (int i = 0; i < n; i++) ph -= (uint32_t)rec[i] * (uint32_t)key[i];
    return (int32_t)ph;
}
/* mk_encrypt_3gen      J/mk_api.jl:519-536 ; mk_lwe_phase J/mk_internals.jl:85-91 */
void oracle_mk_lwe_encrypt(const int32_t *keys, int32_t n, int32_t P, int32_t mu, double sigma, uint64_t seed, uint64_t idx, int32_t *rec) {
    oracle_lwe_encrypt(keys, n * P, mu, sigma, seed, idx, rec);
}
int32_t oracle_mk_lwe_phase(const int32_t *keys, int32_t n, int32_t P, const int32_t *rec) { return oracle_lwe_phase(keys, n * P, rec); }

/* ============================================================================================
 * LWE -> TLWE conversion and threshold partial / final decryption  (k = 1)
 * ========================================================================================== */
/* TLweFromLwe, src/libthfhe.cpp:340-348 (same in src/KNN_medical_data.cpp:492-500): LWE of dimension N -> ring sample */
void oracle_tlwe_from_lwe(const int32_t *lwe /*[N+1]*/, int32_t N, int32_t *tlwe_a /*[N]*/, int32_t *tlwe_b /*[N]*/) {
    tlwe_a[0] = lwe[0];
    for (int i = 1; i < N; i++) tlwe_a[i] = (int32_t)(0u - (uint32_t)lwe[N - i]);
    memset(tlwe_b, 0, sizeof(int32_t) * (size_t)N);
    tlwe_b[0] = lwe[N];
}
/* ThFHEKeyShare::PartialDecrypt, src/libthfhe.cpp:270-293: partial = key_share (*) a + smudging noise
 * (torusPolynomialAddMulR = exact negacyclic product mod 2^32; the Gaussian noise is an input here) */
void oracle_partial_decrypt(const int32_t *key_share, const int32_t *tlwe_a, const int32_t *noise /* NULL = none */, int32_t N,
                            int32_t *partial) {
    oracle_polymul_schoolbook32(key_share, tlwe_a, N, partial);
    if (noise)
        for (int j = 0; j < N; j++) partial[j] = (int32_t)((uint32_t)partial[j] + (uint32_t)noise[j]);
}
/* finalDecrypt, src/libthfhe.cpp:296-315: result = b - partial_0 + sum_{i>=1} partial_i ; message bit = result[0] > 0 */
int32_t oracle_final_decrypt(const int32_t *tlwe_b, const int32_t *partials /*[t][N]*/, int32_t t, int32_t N, int32_t *result /*[N] or NULL*/) {
    int32_t r0 = 0;
    for (int j = 0; j < N; j++) {
        uint32_t v = (uint32_t)tlwe_b[j];
        for (int i = 0; i < t; i++) v = i == 0 ? v - (uint32_t)partials[(size_t)i * N + j] : v + (uint32_t)partials[(size_t)i * N + j];
        if (result) result[j] = (int32_t)v;
        if (j == 0) r0 = (int32_t)v;
    }
    return r0 > 0 ? 1 : 0;
}

/* ============================================================================================
 * CCS multi-key scheme (Chen-Chillotti-Song): mk_bootstrap / mk_gate_nand of the reference      
 *   MKRLweSample acc = (a_0 .. a_{P-1}, b), Torus32        J/mk_internals.jl:104-152
 *   UniProduct_old                                        J/mk_internals.jl:477-536
 *   mk_mux_rotate / mk_blind_rotate / mk_bootstrap         J/mk_internals.jl:805-858
 *   mk_rlwe_extract_sample :141-148, mk_keyswitch :714-728, mk_gate_nand J/mk_gates.jl:7-13
 * Key tables (coefficient domain, Torus32):
 *   bk  int32[P][n][3][l][N]   per (party, j): d1[l], f0[l], f1[l] of MKTGswUESample (:338-388; c0, c1, d0 are not used by UniProduct_old)
 *   pk  int32[P][l][N]         PublicKey.b (:209-245);   crs int32[l][N]  SharedKey.a (:156-168)
 *   ksk int32[P][N][t][base-1][n+1]
 * Records: MK LWE int32[P*n+1]; extracted sample int32[P*N+1] = a[p*N + j], b.
 * ========================================================================================== */
struct oracle_ccs_ctx {
    oracle_params p;
    const int32_t *bk, *pk, *crs, *ksk; /* borrowed */
    uint64_t *bk_ntt, *pk_ntt, *crs_ntt;
};
static void ccs_ntt_table(const int32_t *src, size_t npolys, int N, uint64_t *dst) {
    const gl_tables *T = gl_get_tables(N);
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)npolys; q++) {
        for (int j = 0; j < N; j++) dst[(size_t)q * N + j] = gl_from_i64(src[(size_t)q * N + j]);
        gl_ntt_fwd(dst + (size_t)q * N, T);
    }
}
oracle_ccs_ctx *oracle_ccs_ctx_create(const oracle_params *p, const int32_t *bk, const int32_t *pk, const int32_t *crs, const int32_t *ksk) {
    if (p->torus_bits != 32 || p->k != 1) return NULL;
    oracle_ccs_ctx *c = (oracle_ccs_ctx *)calloc(1, sizeof(*c));
    c->p = *p;
    c->bk = bk, c->pk = pk, c->crs = crs, c->ksk = ksk;
    const size_t nb = (size_t)p->parties * p->n * 3 * p->l, np = (size_t)p->parties * p->l, nc = (size_t)p->l;
    c->bk_ntt = (uint64_t *)malloc(sizeof(uint64_t) * (nb + np + nc) * p->N);
    c->pk_ntt = c->bk_ntt + nb * p->N;
    c->crs_ntt = c->pk_ntt + np * p->N;
    ccs_ntt_table(bk, nb, p->N, c->bk_ntt);
    ccs_ntt_table(pk, np, p->N, c->pk_ntt);
    ccs_ntt_table(crs, nc, p->N, c->crs_ntt);
    return c;
}
void oracle_ccs_ctx_destroy(oracle_ccs_ctx *c) {
    if (!c) return;
    free(c->bk_ntt);
    free(c);
}
/* out = sum_l digits[l] (*) key[l], exact mod 2^32; dig_ntt: the digits' NTTs (shared by several products) */
static void ccs_dot(const oracle_ccs_ctx *c, const int32_t *digits, const uint64_t *dig_ntt, const int32_t *key, const uint64_t *key_ntt,
                    int use_schoolbook, int32_t *out) {
    const int N = c->p.N, l = c->p.l;
    if (use_schoolbook) {
        int32_t *prod = (int32_t *)malloc(sizeof(int32_t) * N);
        memset(out, 0, sizeof(int32_t) * N);
        for (int q = 0; q < l; q++) {
            oracle_polymul_schoolbook32(digits + (size_t)q * N, key + (size_t)q * N, N, prod);
            for (int j = 0; j < N; j++) out[j] = (int32_t)((uint32_t)out[j] + (uint32_t)prod[j]);
        }
        free(prod);
        return;
    }
    const gl_tables *T = gl_get_tables(N);
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * N);
    for (int j = 0; j < N; j++) {
        uint64_t s = 0;
        for (int q = 0; q < l; q++) s = gl_add(s, gl_mul(dig_ntt[(size_t)q * N + j], key_ntt[(size_t)q * N + j]));
        acc[j] = s;
    }
    gl_ntt_inv(acc, T);
    for (int j = 0; j < N; j++) out[j] = (int32_t)(uint32_t)(uint64_t)gl_to_centered(acc[j]);
    free(acc);
}
static void ccs_decompose_all(const oracle_ccs_ctx *c, const int32_t *polys /*[P+1][N]*/, int32_t *dig /*[P+1][l][N]*/, uint64_t *dig_ntt) {
    const int N = c->p.N, l = c->p.l, P = c->p.parties;
    const gl_tables *T = gl_get_tables(N);
    for (int i = 0; i <= P; i++) {
        oracle_decompose32(polys + (size_t)i * N, N, l, c->p.Bgbit, dig + (size_t)i * l * N);
        for (int q = 0; q < l; q++) {
            uint64_t *d = dig_ntt + ((size_t)i * l + q) * N;
            for (int j = 0; j < N; j++) d[j] = gl_from_i64(dig[((size_t)i * l + q) * N + j]);
            gl_ntt_fwd(d, T);
        }
    }
}
/* UniProduct_old(acc, bk[j, party], pk, crs, party)      J/mk_internals.jl:477-536 ; acc, out: int32[P+1][N] (a_0..a_{P-1}, b) */
void oracle_ccs_uniproduct(const oracle_ccs_ctx *c, int32_t party, int32_t j, const int32_t *acc, int32_t *out, int use_schoolbook) {
    const int N = c->p.N, l = c->p.l, P = c->p.parties, n = c->p.n;
    const size_t G = (size_t)(P + 1) * l * N;
    int32_t *dig = (int32_t *)malloc(sizeof(int32_t) * (G + (size_t)(P + 2) * N));
    int32_t *v = dig + G, *w = v + (size_t)(P + 1) * N;
    uint64_t *dn = (uint64_t *)malloc(sizeof(uint64_t) * G);
    const size_t ue = (((size_t)party * n + j) * 3) * l * N; /* d1 | f0 | f1 */
    const int32_t *d = c->bk + ue, *f0 = d + (size_t)l * N, *f1 = f0 + (size_t)l * N;
    const uint64_t *dN = c->bk_ntt + ue, *f0N = dN + (size_t)l * N, *f1N = f0N + (size_t)l * N;
    ccs_decompose_all(c, acc, dig, dn);
    for (int i = 0; i <= P; i++) { /* u_i (i < P), u0 (i = P) ; v_i, v0 */
        const int32_t *di = dig + (size_t)i * l * N;
        const uint64_t *dni = dn + (size_t)i * l * N;
        ccs_dot(c, di, dni, d, dN, use_schoolbook, out + (size_t)i * N);
        if (i < P) {
            ccs_dot(c, di, dni, c->pk + (size_t)i * l * N, c->pk_ntt + (size_t)i * l * N, use_schoolbook, v + (size_t)i * N);
        } else {
            ccs_dot(c, di, dni, c->crs, c->crs_ntt, use_schoolbook, v + (size_t)i * N);
            for (int t = 0; t < N; t++) v[(size_t)i * N + t] = (int32_t)(0u - (uint32_t)v[(size_t)i * N + t]); /* v0 = -sum dec_b * a */
        }
    }
    ccs_decompose_all(c, v, dig, dn); /* g^{-1}(v_i), g^{-1}(v0) */
    for (int i = 0; i <= P; i++) {
        const int32_t *di = dig + (size_t)i * l * N;
        const uint64_t *dni = dn + (size_t)i * l * N;
        ccs_dot(c, di, dni, f0, f0N, use_schoolbook, w); /* w0_i : into b */
        for (int t = 0; t < N; t++) out[(size_t)P * N + t] = (int32_t)((uint32_t)out[(size_t)P * N + t] + (uint32_t)w[t]);
        ccs_dot(c, di, dni, f1, f1N, use_schoolbook, w); /* w1_i : into a[party] */
        for (int t = 0; t < N; t++) out[(size_t)party * N + t] = (int32_t)((uint32_t)out[(size_t)party * N + t] + (uint32_t)w[t]);
    }
    free(dig);
    free(dn);
}
/* mk_mux_rotate: acc += UniProduct(X^barai acc - acc)      J/mk_internals.jl:805-812 */
void oracle_ccs_mux_rotate(const oracle_ccs_ctx *c, int32_t party, int32_t j, int32_t barai, int32_t *acc, int use_schoolbook) {
    const int N = c->p.N, P = c->p.parties;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(P + 1) * N), *up = tmp + (size_t)(P + 1) * N;
    for (int m = 0; m <= P; m++) {
        oracle_mul_by_monomial32(acc + (size_t)m * N, barai, N, tmp + (size_t)m * N);
        for (int q = 0; q < N; q++) tmp[(size_t)m * N + q] = (int32_t)((uint32_t)tmp[(size_t)m * N + q] - (uint32_t)acc[(size_t)m * N + q]);
    }
    oracle_ccs_uniproduct(c, party, j, tmp, up, use_schoolbook);
    for (size_t q = 0; q < (size_t)(P + 1) * N; q++) acc[q] = (int32_t)((uint32_t)acc[q] + (uint32_t)up[q]);
    free(tmp);
}
/* mk_bootstrap_wo_keyswitch: x int32[P*n+1] -> out int32[P*N+1]      J/mk_internals.jl:815-852 */
void oracle_ccs_bootstrap_wo_keyswitch(const oracle_ccs_ctx *c, int32_t mu, const int32_t *x, int32_t *out, int use_schoolbook) {
    const int N = c->p.N, n = c->p.n, P = c->p.parties;
    int32_t barb = oracle_modswitch(x[(size_t)n * P], N);
    int32_t *acc = (int32_t *)calloc((size_t)(P + 1) * N, sizeof(int32_t));
    int32_t *tv = (int32_t *)malloc(sizeof(int32_t) * N);
    for (int j = 0; j < N; j++) tv[j] = mu;
    oracle_mul_by_monomial32(tv, -barb, N, acc + (size_t)P * N);
    for (int p = 0; p < P; p++)
        for (int j = 0; j < n; j++) {
            int32_t bara = oracle_modswitch(x[(size_t)p * n + j], N);
            if (bara != 0) oracle_ccs_mux_rotate(c, p, j, bara, acc, use_schoolbook);
        }
    for (int p = 0; p < P; p++) { /* mk_rlwe_extract_sample: reverse_polynomial per party */
        const int32_t *a = acc + (size_t)p * N;
        out[(size_t)p * N] = a[0];
        for (int j = 1; j < N; j++) out[(size_t)p * N + j] = (int32_t)(0u - (uint32_t)a[N - j]);
    }
    out[(size_t)P * N] = acc[(size_t)P * N];
    free(tv);
    free(acc);
}
/* mk_keyswitch      J/mk_internals.jl:714-728: party p switches (a[:, p], 0) with ks[p]; b = u.b + sum of the parts' b */
void oracle_ccs_keyswitch(const oracle_ccs_ctx *c, const int32_t *in, int32_t *out) {
    const int N = c->p.N, n = c->p.n, P = c->p.parties, t = c->p.ks_t, bb = c->p.ks_basebit;
    const size_t per_party = (size_t)N * t * ((1 << bb) - 1) * ((size_t)n + 1);
    int32_t *part = (int32_t *)scr(SCR_MKKS, sizeof(int32_t) * ((size_t)n + 1));
    uint32_t b = (uint32_t)in[(size_t)P * N];
    for (int p = 0; p < P; p++) {
        keyswitch_with(c->ksk + (size_t)p * per_party, N, n, t, bb, in + (size_t)p * N, 0, part);
        memcpy(out + (size_t)p * n, part, sizeof(int32_t) * n);
        b += (uint32_t)part[n];
    }
    out[(size_t)n * P] = (int32_t)b;
}
/* mk_gate_nand (J/mk_gates.jl:7-13) and, with the same bootstrap, the other two-input linear prologues of J/gates.jl */
int oracle_ccs_gates(const oracle_ccs_ctx *c, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count, int use_schoolbook) {
    const int n = c->p.n, N = c->p.N, P = c->p.parties;
    const size_t rec = (size_t)n * P + 1;
    lin_t L;
    if (op == OR_GATE_MUX || op == OR_GATE_NOT || op == OR_GATE_COPY || gate_lin(op, 0, &L) != 0) return -1;
    int32_t *res = (int32_t *)malloc(sizeof(int32_t) * count * rec);
#pragma omp parallel for schedule(dynamic)
    for (long g = 0; g < (long)count; g++) {
        int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (rec + (size_t)P * N + 1)), *u = tmp + rec;
        const int32_t *x = in0 + g * rec, *y = in1 + g * rec;
        for (size_t q = 0; q < rec; q++) tmp[q] = (int32_t)((uint32_t)L.cx * (uint32_t)x[q] + (uint32_t)L.cy * (uint32_t)y[q]);
        tmp[rec - 1] = (int32_t)((uint32_t)tmp[rec - 1] + (uint32_t)L.cb);
        oracle_ccs_bootstrap_wo_keyswitch(c, 1 << 29, tmp, u, use_schoolbook);
        oracle_ccs_keyswitch(c, u, res + g * rec);
        free(tmp);
    }
    memcpy(out, res, sizeof(int32_t) * count * rec);
    free(res);
    return 0;
}
/* key material: SecretKey / SharedKey / CloudKeyPart (PublicKey, BootstrapKeyPart = mk_tgsw_encrypt of every key bit, KeyswitchKey)
 * J/mk_api.jl:368-384, J/mk_internals.jl:156-168,209-245,390-448,745-775 -- with OUR random streams */
void oracle_keygen_ccs(const oracle_params *p, uint64_t seed, double sigma_bk, double sigma_ks, int32_t *lwe_keys /*[P][n]*/, int32_t *rlwe_keys /*[P][N]*/,
                       int32_t *bk, int32_t *pk, int32_t *crs, int32_t *ksk) {
    const int n = p->n, N = p->N, l = p->l, P = p->parties;
    rng_t r;
    rng_init(&r, seed, 31);
    for (int i = 0; i < P * n; i++) lwe_keys[i] = (int32_t)(rng_u64(&r) & 1);
    rng_init(&r, seed, 32);
    for (int i = 0; i < P * N; i++) rlwe_keys[i] = (int32_t)(rng_u64(&r) & 1); /* RLweKey(rng, params): binary */
    rng_init(&r, seed, 33);
    for (int i = 0; i < l * N; i++) crs[i] = (int32_t)(uint32_t)rng_u64(&r); /* SharedKey.a */
    rng_init(&r, seed, 34);
    int32_t *prod = (int32_t *)malloc(sizeof(int32_t) * N);
    for (int q = 0; q < P; q++) /* PublicKey: b_i = s (*) a_i + e_i */
        for (int i = 0; i < l; i++) {
            key_mul32(rlwe_keys + (size_t)q * N, crs + (size_t)i * N, N, prod);
            for (int t = 0; t < N; t++) pk[((size_t)q * l + i) * N + t] = (int32_t)((uint32_t)prod[t] + (uint32_t)dtot32(rng_gauss(&r) * sigma_bk));
        }
    free(prod);
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int q = 0; q < P; q++)
        for (int j = 0; j < n; j++) { /* mk_tgsw_encrypt(lwe_key[j]) -> d1, f0, f1 */
            rng_t ri;
            rng_init(&ri, seed, 300000 + (uint64_t)q * 10000 + (uint64_t)j);
            int32_t *rr = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)N), *pr = rr + N;
            int32_t *ue = bk + (((size_t)q * n + j) * 3) * l * N, *d1 = ue, *f0 = ue + (size_t)l * N, *f1 = f0 + (size_t)l * N;
            const uint32_t m = (uint32_t)lwe_keys[(size_t)q * n + j];
            for (int t = 0; t < N; t++) rr[t] = (int32_t)(rng_u64(&ri) & 1); /* the shared randomness r */
            for (int i = 0; i < l; i++) {
                const uint32_t g = 1u << (32 - (i + 1) * p->Bgbit);
                key_mul32(rr, crs + (size_t)i * N, N, pr);
                for (int t = 0; t < N; t++) d1[(size_t)i * N + t] = (int32_t)((uint32_t)pr[t] + (uint32_t)dtot32(rng_gauss(&ri) * sigma_bk));
                d1[(size_t)i * N] = (int32_t)((uint32_t)d1[(size_t)i * N] + m * g);
                for (int t = 0; t < N; t++) f1[(size_t)i * N + t] = (int32_t)(uint32_t)rng_u64(&ri);
                key_mul32(rlwe_keys + (size_t)q * N, f1 + (size_t)i * N, N, pr);
                for (int t = 0; t < N; t++)
                    f0[(size_t)i * N + t] = (int32_t)((uint32_t)pr[t] + (uint32_t)dtot32(rng_gauss(&ri) * sigma_bk) + (uint32_t)rr[t] * g);
            }
            free(rr);
        }
    const size_t per_party = (size_t)N * p->ks_t * ((1 << p->ks_basebit) - 1) * ((size_t)n + 1);
    for (int q = 0; q < P; q++) {
        rng_init(&r, seed, 40 + (uint64_t)q);
        gen_ksk(&r, rlwe_keys + (size_t)q * N, N, lwe_keys + (size_t)q * n, n, p->ks_t, p->ks_basebit, sigma_ks, ksk + (size_t)q * per_party);
    }
}

/* ============================================================================================
 * KMS multi-key scheme: mk_bootstrap_new / mk_gate_nand_new      J/new_mk_internals.jl, J/tlev.jl, J/new_mk_gates.jl
 * (Torus64 ring, k = 1).  Three gadget families: gsw (the per-party single-key TGSW blind rotation of a TLev accumulator),
 * lev (the TLev accumulator itself) and uni (the uni-encryption that relinearises from the party's fresh key to the joint key).
 * Tables (coefficient domain, int64):  gsw [P][n][2 l_gsw][2][N] (row = block j * l + level, column 0 = mask, 1 = body);
 *   uni [P][3][l_uni][N] = d1, f0, f1 of mk_tgsw_encrypt (J/mk_internals.jl:390-446);  pk [P][l_uni][N];  crs [l_uni][N];
 *   ksk int32 [P][N][t][base-1][n+1].   A TLev sample is int64[l_lev][2][N]; an MKRLweSample int64[P+1][N] = a_0 .. a_{P-1}, b.
 * ========================================================================================== */
struct oracle_kms_ctx {
    oracle_kms_params p;
    const int64_t *gsw, *uni, *pk, *crs;
    const int32_t *ksk;
};
oracle_kms_ctx *oracle_kms_ctx_create(const oracle_kms_params *p, const int64_t *gsw, const int64_t *uni, const int64_t *pk, const int64_t *crs,
                                      const int32_t *ksk) {
    oracle_kms_ctx *c = (oracle_kms_ctx *)calloc(1, sizeof(*c));
    c->p = *p;
    c->gsw = gsw, c->uni = uni, c->pk = pk, c->crs = crs, c->ksk = ksk;
    return c;
}
void oracle_kms_ctx_destroy(oracle_kms_ctx *c) { free(c); }

/* acc += sign * small (*) torus    (exact; the reference's transformed_mul / IntPolynomial * TorusPolynomial) */
static void kms_mulacc(const int64_t *small, const int64_t *torus, int N, int64_t *acc, int sign, int use_schoolbook, int64_t *tmp) {
    if (use_schoolbook) oracle_polymul_schoolbook64(small, torus, N, tmp);
    else oracle_polymul_ntt64(small, torus, N, tmp);
    for (int q = 0; q < N; q++) acc[q] = (int64_t)(sign > 0 ? (uint64_t)acc[q] + (uint64_t)tmp[q] : (uint64_t)acc[q] - (uint64_t)tmp[q]);
}

/* mk_ith_blind_rotate: TLev accumulator of party `party`      J/new_mk_internals.jl:210-225, mk_mux_rotate_new :177-182,
 * tlev_trivial_int J/tlev.jl:37-45, tgsw_intern_mul J/tlev.jl:85-92 (tgsw_extern_mul per TLev sample, J/tgsw.jl:146-156) */
/* n CMuxes of ONE RLWE sample acc = (mask, body), int64[2][N], with party's TGSW key: mux_rotate (J/bootstrap.jl:19-23) under
   mk_single_blind_rotate (J/new_mk_internals.jl:226-238) and, per TLev sample, under mk_ith_blind_rotate (:210-223) */
void oracle_kms_rlwe_rotate(const oracle_kms_ctx *c, int32_t party, const int32_t *bara, int64_t *acc, int use_schoolbook) {
    const int N = c->p.N, n = c->p.n, lg = c->p.l_gsw, rows = 2 * lg;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 + rows + 2 + 1) * N);
    int64_t *dig = tmp + 2 * (size_t)N, *ext = dig + (size_t)rows * N, *pr = ext + 2 * (size_t)N;
    for (int j = 0; j < n; j++) {
        if (bara[j] == 0) continue;
        const int64_t *key = c->gsw + (((size_t)party * n + j) * rows) * 2 * N;
        for (int m = 0; m < 2; m++) {
            oracle_mul_by_monomial64(acc + (size_t)m * N, bara[j], N, tmp + (size_t)m * N);
            for (int q = 0; q < N; q++) tmp[(size_t)m * N + q] = (int64_t)((uint64_t)tmp[(size_t)m * N + q] - (uint64_t)acc[(size_t)m * N + q]);
            oracle_decompose64(tmp + (size_t)m * N, N, lg, c->p.bg_gsw, dig + (size_t)m * lg * N);
        }
        memset(ext, 0, sizeof(int64_t) * 2 * (size_t)N);
        for (int r = 0; r < rows; r++)
            for (int col = 0; col < 2; col++) kms_mulacc(dig + (size_t)r * N, key + ((size_t)r * 2 + col) * N, N, ext + (size_t)col * N, 1, use_schoolbook, pr);
        for (int q = 0; q < 2 * N; q++) acc[q] = (int64_t)((uint64_t)acc[q] + (uint64_t)ext[q]);
    }
    free(tmp);
}
void oracle_kms_tlev_rotate(const oracle_kms_ctx *c, int32_t party, const int32_t *bara, int64_t *lev, int use_schoolbook) {
    const int N = c->p.N, lv = c->p.l_lev;
    memset(lev, 0, sizeof(int64_t) * (size_t)lv * 2 * N);
    for (int s = 0; s < lv; s++) {   /* tgsw_intern_mul (J/tlev.jl:88-95) works sample by sample: the l_lev rotations are independent */
        lev[((size_t)s * 2 + 1) * N] = (int64_t)(1ull << (64 - (s + 1) * c->p.bg_lev)); /* body += 1 * gadget[s] */
        oracle_kms_rlwe_rotate(c, party, bara, lev + (size_t)s * 2 * N, use_schoolbook);
    }
}

/* UniProduct_new      J/new_mk_internals.jl:85-127.  e, out: int64[P+1][N] */
void oracle_kms_uniproduct(const oracle_kms_ctx *c, int32_t party, const int64_t *e, int64_t *out, int use_schoolbook) {
    const int N = c->p.N, P = c->p.parties, lu = c->p.l_uni;
    const int64_t *d = c->uni + (((size_t)party * 3 + 0) * lu) * N, *f0 = c->uni + (((size_t)party * 3 + 1) * lu) * N,
                  *f1 = c->uni + (((size_t)party * 3 + 2) * lu) * N;
    int64_t *dec = (int64_t *)malloc(sizeof(int64_t) * ((size_t)(P + 1) * lu + lu + 2) * N);
    int64_t *dec_v = dec + (size_t)(P + 1) * lu * N, *v = dec_v + (size_t)lu * N, *pr = v + N;
    for (int i = 0; i <= P; i++) oracle_decompose64(e + (size_t)i * N, N, lu, c->p.bg_uni, dec + (size_t)i * lu * N); /* i = P: dec_b */
    memset(out, 0, sizeof(int64_t) * (size_t)(P + 1) * N);
    memset(v, 0, sizeof(int64_t) * N);
    for (int i = 0; i <= P; i++)
        for (int l = 0; l < lu; l++) {
            const int64_t *di = dec + ((size_t)i * lu + l) * N;
            kms_mulacc(di, d + (size_t)l * N, N, out + (size_t)i * N, 1, use_schoolbook, pr); /* u_i (i < P), u0 (i = P) */
            if (i < P) kms_mulacc(di, c->pk + ((size_t)i * lu + l) * N, N, v, 1, use_schoolbook, pr);
            else kms_mulacc(di, c->crs + (size_t)l * N, N, v, -1, use_schoolbook, pr);
        }
    oracle_decompose64(v, N, lu, c->p.bg_uni, dec_v);
    for (int l = 0; l < lu; l++) {
        kms_mulacc(dec_v + (size_t)l * N, f0 + (size_t)l * N, N, out + (size_t)P * N, 1, use_schoolbook, pr);     /* bnew = u0 + w0 */
        kms_mulacc(dec_v + (size_t)l * N, f1 + (size_t)l * N, N, out + (size_t)party * N, 1, use_schoolbook, pr); /* anew[party] += w1 */
    }
    free(dec);
}

/* mk_lev_rlwe_mul: accum <- f - UniProduct_new(e), (e, f) = tlev_extern_mul of accum's polynomials      J/new_mk_internals.jl:185-207,
 * tlev_extern_mul J/tlev.jl:72-76 */
void oracle_kms_lev_rlwe_mul(const oracle_kms_ctx *c, int32_t party, int64_t *accum, const int64_t *lev, int use_schoolbook) {
    const int N = c->p.N, P = c->p.parties, lv = c->p.l_lev;
    int64_t *e = (int64_t *)calloc((size_t)(3 * (P + 1) + lv + 1) * N, sizeof(int64_t));
    int64_t *f = e + (size_t)(P + 1) * N, *up = f + (size_t)(P + 1) * N, *dec = up + (size_t)(P + 1) * N, *pr = dec + (size_t)lv * N;
    for (int i = 0; i <= P; i++) {
        if (i < P && i >= party) continue; /* only the parties before `party`, and b (i = P) */
        oracle_decompose64(accum + (size_t)i * N, N, lv, c->p.bg_lev, dec);
        for (int s = 0; s < lv; s++) {
            kms_mulacc(dec + (size_t)s * N, lev + ((size_t)s * 2 + 0) * N, N, e + (size_t)i * N, 1, use_schoolbook, pr);
            kms_mulacc(dec + (size_t)s * N, lev + ((size_t)s * 2 + 1) * N, N, f + (size_t)i * N, 1, use_schoolbook, pr);
        }
    }
    oracle_kms_uniproduct(c, party, e, up, use_schoolbook);
    for (size_t q = 0; q < (size_t)(P + 1) * N; q++) accum[q] = (int64_t)((uint64_t)f[q] - (uint64_t)up[q]);
    free(e);
}

/* mk_bootstrap_wo_keyswitch_new (fast_boot = false): x int32[P*n+1] -> out int32[P*N+1]      J/new_mk_internals.jl:254-262,276-283,
 * 303-314; extraction mk_rlwe_extract_sample_64 :295-300 */
/* fast_boot = 0: mk_blind_rotate_and_extract_new (:271-283); fast_boot = 1: ..._new_v2 (:255-269, 286-291), which replaces the first
   party's TLev rotation + relinearisation by ONE RLWE blind rotation (mask, body) of the test vector and accum = f - UniProduct_new(e),
   e / f = the trivial multi-key samples of mask / body */
void oracle_kms_bootstrap_wo_keyswitch_ex(const oracle_kms_ctx *c, int64_t mu, const int32_t *x, int32_t *out, int use_schoolbook, int fast_boot) {
    const int N = c->p.N, n = c->p.n, P = c->p.parties, lv = c->p.l_lev;
    int32_t barb = oracle_modswitch(x[(size_t)n * P], N);
    int64_t *accum = (int64_t *)calloc((size_t)(P + 1) * N + N + (size_t)lv * 2 * N, sizeof(int64_t));
    int64_t *tv = accum + (size_t)(P + 1) * N, *lev = tv + N;
    int32_t *bara = (int32_t *)malloc(sizeof(int32_t) * n);
    for (int j = 0; j < N; j++) tv[j] = mu;
    oracle_mul_by_monomial64(tv, -barb, N, accum + (size_t)P * N);
    int first = 0;
    if (fast_boot) {
        int64_t *acc1 = (int64_t *)calloc(2 * (size_t)N + 2 * (size_t)(P + 1) * N, sizeof(int64_t));
        int64_t *e = acc1 + 2 * (size_t)N, *up = e + (size_t)(P + 1) * N;
        memcpy(acc1 + N, accum + (size_t)P * N, sizeof(int64_t) * N);     /* rlwe_noiseless_trivial(testvectbis) */
        for (int j = 0; j < n; j++) bara[j] = oracle_modswitch(x[j], N);
        oracle_kms_rlwe_rotate(c, 0, bara, acc1, use_schoolbook);
        memcpy(e + (size_t)P * N, acc1, sizeof(int64_t) * N);              /* e = trivial(mask) */
        oracle_kms_uniproduct(c, 0, e, up, use_schoolbook);
        memset(accum, 0, sizeof(int64_t) * (size_t)(P + 1) * N);
        memcpy(accum + (size_t)P * N, acc1 + N, sizeof(int64_t) * N);      /* f = trivial(body) */
        for (size_t q = 0; q < (size_t)(P + 1) * N; q++) accum[q] = (int64_t)((uint64_t)accum[q] - (uint64_t)up[q]);
        free(acc1);
        first = 1;
    }
    for (int p = first; p < P; p++) {
        for (int j = 0; j < n; j++) bara[j] = oracle_modswitch(x[(size_t)p * n + j], N);
        oracle_kms_tlev_rotate(c, p, bara, lev, use_schoolbook);
        oracle_kms_lev_rlwe_mul(c, p, accum, lev, use_schoolbook);
    }
    for (int p = 0; p < P; p++) {
        const int64_t *a = accum + (size_t)p * N;
        out[(size_t)p * N] = oracle_t64tot32(a[0]);
        for (int j = 1; j < N; j++) out[(size_t)p * N + j] = oracle_t64tot32((int64_t)(0ull - (uint64_t)a[N - j]));
    }
    out[(size_t)P * N] = oracle_t64tot32(accum[(size_t)P * N]);
    free(bara);
    free(accum);
}
void oracle_kms_bootstrap_wo_keyswitch(const oracle_kms_ctx *c, int64_t mu, const int32_t *x, int32_t *out, int use_schoolbook) {
    oracle_kms_bootstrap_wo_keyswitch_ex(c, mu, x, out, use_schoolbook, 0);
}
/* mk_keyswitch (party p switches its own extracted mask)      J/mk_internals.jl:714-728 */
void oracle_kms_keyswitch(const oracle_kms_ctx *c, const int32_t *in, int32_t *out) {
    const int N = c->p.N, n = c->p.n, P = c->p.parties, t = c->p.ks_t, bb = c->p.ks_basebit;
    const size_t per_party = (size_t)N * t * ((1 << bb) - 1) * ((size_t)n + 1);
    int32_t *part = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    uint32_t b = (uint32_t)in[(size_t)P * N];
    for (int p = 0; p < P; p++) {
        keyswitch_with(c->ksk + (size_t)p * per_party, N, n, t, bb, in + (size_t)p * N, 0, part);
        memcpy(out + (size_t)p * n, part, sizeof(int32_t) * n);
        b += (uint32_t)part[n];
    }
    out[(size_t)n * P] = (int32_t)b;
    free(part);
}
/* mk_gate_nand_new (J/new_mk_gates.jl:1-7) and, on the same bootstrap, the other two-input linear prologues of J/gates.jl */
int oracle_kms_gates_ex(const oracle_kms_ctx *c, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count, int use_schoolbook,
                        int fast_boot) {
    const int n = c->p.n, N = c->p.N, P = c->p.parties;
    const size_t rec = (size_t)n * P + 1;
    lin_t L;
    if (op == OR_GATE_MUX || op == OR_GATE_NOT || op == OR_GATE_COPY || gate_lin(op, 0, &L) != 0) return -1;
    int32_t *res = (int32_t *)malloc(sizeof(int32_t) * count * rec);
#pragma omp parallel for schedule(dynamic)
    for (long g = 0; g < (long)count; g++) {
        int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (rec + (size_t)P * N + 1)), *u = tmp + rec;
        const int32_t *x = in0 + g * rec, *y = in1 + g * rec;
        for (size_t q = 0; q < rec; q++) tmp[q] = (int32_t)((uint32_t)L.cx * (uint32_t)x[q] + (uint32_t)L.cy * (uint32_t)y[q]);
        tmp[rec - 1] = (int32_t)((uint32_t)tmp[rec - 1] + (uint32_t)L.cb);
        oracle_kms_bootstrap_wo_keyswitch_ex(c, (int64_t)1 << 61, tmp, u, use_schoolbook, fast_boot); /* encode_message64(1, 8) */
        oracle_kms_keyswitch(c, u, res + g * rec);
        free(tmp);
    }
    memcpy(out, res, sizeof(int32_t) * count * rec);
    free(res);
    return 0;
}
int oracle_kms_gates(const oracle_kms_ctx *c, int op, const int32_t *in0, const int32_t *in1, int32_t *out, size_t count, int use_schoolbook) {
    return oracle_kms_gates_ex(c, op, in0, in1, out, count, use_schoolbook, 0);
}
