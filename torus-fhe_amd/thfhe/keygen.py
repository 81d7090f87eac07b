"""Host-side synthetic key generation, encryption and decryption for the bootstrapping engine.

Mirrors the reference constructors that produce the hot path's inputs (they run once, on the host):
  LweKey / RLweKey                      lwe.jl:11-19, rlwe.jl:12-30
  BootstrapKey (coefficient domain)     bootstrap.jl:6-15 -> tgsw_encrypt tgsw.jl:88-101 -> rlwe_encrypt_zero rlwe.jl:79-105
  KeyswitchKey                          keyswitch.jl:14-41 (noise recentred over the whole table)
  lwe_encrypt / lwe_phase               lwe.jl:38-59
  3-gen multi-key material              multikey_3gen.jl:15-30, mk_internals.jl:120-137,209-298, tgsw_3gen.jl:41-95
Randomness is numpy's PCG64 (the reference uses Julia's MersenneTwister, whose stream is not
reproducible outside Julia); the hot path itself consumes no randomness.  Ring products with the
small secret polynomials are computed EXACTLY with float64 GEMMs on <= 22-bit limbs (all partial
sums stay below 2^53), then reduced mod 2^32 / 2^64.
"""
import numpy as np

from . import MU8


def dtot32(d):
    """numeric-functions.jl:101-103: trunc(Int32, d * 2^32) (wrapping)."""
    return np.trunc(np.asarray(d, np.float64) * 4294967296.0).astype(np.int64).astype(np.uint32).view(np.int32)


def dtot64(d):
    """numeric-functions.jl:105-107 (|d| << 0.5 here, so the int64 conversion cannot overflow)."""
    return np.trunc(np.asarray(d, np.float64) * 18446744073709551616.0).astype(np.int64)


def negacyclic_matrix(z):
    """M[j, q] with (a (*) z)[q] = sum_j a[j] M[j, q]  (product mod X^N + 1), z small integers."""
    z = np.asarray(z, np.float64)
    N = z.shape[0]
    j = np.arange(N)[:, None]
    q = np.arange(N)[None, :]
    return z[(q - j) % N] * np.where(q >= j, 1.0, -1.0)


def polymul_small32(a, z):
    """rows of a (int32 torus polynomials) times the small polynomial z, exact mod 2^32."""
    M = negacyclic_matrix(z)
    a = np.ascontiguousarray(a, np.int32).astype(np.int64)
    lo, hi = (a & 0xFFFF).astype(np.float64), (a >> 16).astype(np.float64)
    r = (lo @ M).astype(np.int64) + ((hi @ M).astype(np.int64) << 16)
    return r.astype(np.uint32).view(np.int32)


def polymul_small64(a, z):
    """rows of a (int64 torus polynomials) times the small polynomial z, exact mod 2^64 (3 x 22-bit limbs)."""
    M = negacyclic_matrix(z)
    a = np.ascontiguousarray(a, np.int64).view(np.uint64)
    out = np.zeros(a.shape, np.uint64)
    for sh in (0, 22, 44):
        limb = ((a >> np.uint64(sh)) & np.uint64((1 << 22) - 1)).astype(np.float64)
        out += (limb @ M).astype(np.int64).view(np.uint64) << np.uint64(sh)
    return out.view(np.int64)


def gen_keyswitch_key(rng, in_key, out_key, t, basebit, sigma):
    """keyswitch.jl:14-41 -> int32[len(in_key)][t][base-1][n+1]."""
    in_key = np.asarray(in_key, np.int64)
    out_key = np.asarray(out_key, np.int32)
    Nin, n, base = in_key.shape[0], out_key.shape[0], 1 << basebit
    noise = rng.standard_normal((Nin, t, base - 1)) * sigma
    noise -= noise.mean()
    ksk = np.empty((Nin, t, base - 1, n + 1), np.int32)
    a = rng.integers(-2**31, 2**31, size=(Nin, t, base - 1, n), dtype=np.int64)
    ksk[..., :n] = a.astype(np.int32)
    h = np.arange(1, base, dtype=np.int64)[None, None, :]
    j = np.arange(1, t + 1, dtype=np.int64)[None, :, None]
    msg = (in_key[:, None, None] * h) << (32 - j * basebit)
    s = out_key.astype(np.float64)
    dot = ((a & 0xFFFF).astype(np.float64) @ s).astype(np.int64) + (((a >> 16).astype(np.float64) @ s).astype(np.int64) << 16)
    b = msg + dtot32(noise).astype(np.int64) + dot
    ksk[..., n] = b.astype(np.uint32).view(np.int32)
    return ksk


def seed_seq_generate(seeds, count):
    """std::seed_seq(seeds).generate: `count` 32-bit words ([rand.util.seedseq], as libstdc++ implements it)."""
    v = [int(x) & 0xFFFFFFFF for x in seeds]
    n, s = count, len(v)
    b = [0x8B8B8B8B] * n
    t = 11 if n >= 623 else 7 if n >= 68 else 5 if n >= 39 else 3 if n >= 7 else (n - 1) // 2
    p, q = (n - t) // 2, (n - t) // 2 + t
    T = lambda x: (x ^ (x >> 27)) & 0xFFFFFFFF
    for k in range(max(s + 1, n)):
        r1 = (1664525 * T(b[k % n] ^ b[(k + p) % n] ^ b[(k - 1) % n])) & 0xFFFFFFFF
        r2 = (r1 + (s if k == 0 else (k % n + v[k - 1]) if k <= s else k % n)) & 0xFFFFFFFF
        b[(k + p) % n] = (b[(k + p) % n] + r1) & 0xFFFFFFFF
        b[(k + q) % n] = (b[(k + q) % n] + r2) & 0xFFFFFFFF
        b[k % n] = r2
    for k in range(max(s + 1, n), max(s + 1, n) + n):
        r3 = (1566083941 * T((b[k % n] + b[(k + p) % n] + b[(k - 1) % n]) & 0xFFFFFFFF)) & 0xFFFFFFFF
        r4 = (r3 - k % n) & 0xFFFFFFFF
        b[(k + p) % n] ^= r3
        b[(k + q) % n] ^= r4
        b[k % n] = r4
    return b


def lwe_key_from_seed(seeds, n):
    """The LWE secret key a libtfhe program draws first after `tfhe_random_generator_setSeed(seeds, len)`: n draws of
    std::uniform_int_distribution<int32_t>(0, 1) on std::default_random_engine (minstd_rand0) seeded from std::seed_seq(seeds)
    -- libstdc++'s algorithms.  With the reference's seed {100, 20032, 21341} (src/bootstrap_modules.cpp:52-55, src/libthfhe.cpp:362-363)
    this is the key under which its committed fixtures test/bootstrap_modules/*.data decrypt."""
    M = 2147483647                                    # minstd_rand0: x <- 16807 x mod (2^31 - 1), values in [1, M-1]
    x = seed_seq_generate(seeds, 4)[3] % M            # linear_congruential_engine::seed(Sseq&): generates k + 3 = 4 words, uses the last
    if x == 0:
        x = 1
    urng_range = M - 2                                # max - min of the engine
    scaling = urng_range // 2                         # uniform_int_distribution, urange = 1 < urngrange: rejection + downscaling
    past = 2 * scaling
    key = np.empty(n, np.int32)
    for i in range(n):
        while True:
            x = (16807 * x) % M
            ret = x - 1
            if ret < past:
                break
        key[i] = ret // scaling
    return key


class SecretKeySet:
    """Single-key secret material + the cloud-key tables the engine consumes."""

    def __init__(self, params, seed=0x5EED0001, sigma_lwe=2.0**-15, sigma_bk=2.0**-25, sigma_ks=2.0**-15, lwe_key=None):
        p = self.params = params
        assert p.k == 1 and p.torus_bits == 32 and p.parties == 1
        rng = np.random.default_rng(seed)
        self.sigma_lwe = sigma_lwe
        self.lwe_key = (np.asarray(lwe_key, np.int32) if lwe_key is not None
                        else rng.integers(0, 2, p.n).astype(np.int32))          # LweKey: uniform binary
        self.rlwe_key = rng.integers(0, 2, p.N).astype(np.int32)                 # RLweKey: uniform binary
        rows = 2 * p.l
        # rlwe_encrypt_zero for every TGSW row: mask uniform, body = z (*) mask + gaussian
        mask = rng.integers(-2**31, 2**31, size=(p.n * rows, p.N), dtype=np.int64).astype(np.int32)
        body = polymul_small32(mask, self.rlwe_key).astype(np.int64) + dtot32(rng.standard_normal((p.n * rows, p.N)) * sigma_bk)
        bk = np.empty((p.n, rows, 2, p.N), np.int64)
        bk[:, :, 0, :] = mask.reshape(p.n, rows, p.N)
        bk[:, :, 1, :] = body.reshape(p.n, rows, p.N)
        # + message * gadget on the constant coefficient of polynomial j of row (j, level)   tgsw.jl:65-85
        for j in range(2):
            for lv in range(p.l):
                bk[:, j * p.l + lv, j, 0] += self.lwe_key.astype(np.int64) << (32 - (lv + 1) * p.Bgbit)
        self.bk = bk.astype(np.uint32).view(np.int32)
        self.ksk = gen_keyswitch_key(rng, self.rlwe_key, self.lwe_key, p.ks_t, p.ks_basebit, sigma_ks)

    def encrypt(self, bits, seed=0x5EED0002):
        """lwe_encrypt of +-1/8 per bit -> int32[len(bits)][n+1]."""
        bits = np.asarray(bits).astype(bool)
        rng = np.random.default_rng(seed)
        p = self.params
        a = rng.integers(-2**31, 2**31, size=(bits.shape[0], p.n), dtype=np.int64)
        e = dtot32(rng.standard_normal(bits.shape[0]) * self.sigma_lwe).astype(np.int64)
        mu = np.where(bits, MU8, -MU8).astype(np.int64)
        b = mu + e + (a * self.lwe_key.astype(np.int64)).sum(axis=1)
        out = np.empty((bits.shape[0], p.n + 1), np.int32)
        out[:, :p.n] = a.astype(np.int32)
        out[:, p.n] = b.astype(np.uint32).view(np.int32)
        return out

    def phase(self, recs):
        """lwe_phase = b - <a, s> as wrapping int32."""
        recs = np.asarray(recs, np.int32).reshape(-1, self.params.n + 1).astype(np.int64)
        ph = recs[:, -1] - (recs[:, :-1] * self.lwe_key.astype(np.int64)).sum(axis=1)
        return ph.astype(np.uint32).view(np.int32)

    def decrypt(self, recs):
        return self.phase(recs) > 0


def rand_ternary(rng, shape):
    """rand_negative_binary, numeric-functions.jl:11-13: P(+-1) = 0.113546097609674."""
    u = rng.random(shape)
    return np.where(u < 0.113546097609674, -1, np.where(u < 2 * 0.113546097609674, 1, 0)).astype(np.int64)


class MKSecretKeySet:
    """3-gen multi-key material following 3-gen-mk-tfhe/multikey_3gen.jl:15-30."""

    def __init__(self, params, seed=0x5EED0001, sigma_lwe=None, sigma_bk=2.0**-30.70, sigma_ks=None, device=None):
        """device = None: the ring products run on the host (float64 GEMMs on 22-bit limbs); device = d: on MI355X number d
        (thfhe.PolyMac / thfhe_pm_mac).  The random stream is drawn identically either way, so both give the same key material bit for bit."""
        p = self.params = params
        assert p.k == 1 and p.torus_bits == 64
        rng = np.random.default_rng(seed)
        pm = None
        if device is not None:
            from . import PolyMac
            pm = PolyMac(p.N, 64, device)
        P, n, N, l = p.parties, p.n, p.N, p.l
        self.sigma_lwe = sigma_lwe if sigma_lwe is not None else 2.0**-13.52
        sigma_ks = sigma_ks if sigma_ks is not None else self.sigma_lwe
        self.lwe_keys = rng.integers(0, 2, (P, n)).astype(np.int32)          # SecretKey_3gen
        self.rlwe_keys = rand_ternary(rng, (P, N))                            # RLweKey(rng, params, true)
        crp = rng.integers(-2**63, 2**63, size=N, dtype=np.int64)             # CRP_3gen(a_same = true)
        # PublicKey b_q[i] = z_q (*) a + e ; CommonPubKey B[i] = sum_q b_q[i]
        B = np.zeros((l, N), np.uint64)
        if pm is not None:   # z_q (*) a for all parties in one device call
            za_all = pm.mac(self.rlwe_keys, crp[None, :], [(q, q, 0, 1) for q in range(P)], P).view(np.uint64)
        for q in range(P):
            za = za_all[q] if pm is not None else polymul_small64(crp[None, :], self.rlwe_keys[q])[0].view(np.uint64)
            B += za[None, :] + dtot64(rng.standard_normal((l, N)) * sigma_bk).view(np.uint64)
        # tgsw_encrypt_3gen for every key bit
        bk = np.empty((P, n, 4, l, N), np.uint64)
        g = [np.uint64(64 - (lv + 1) * p.Bgbit) for lv in range(l)]
        for lv in range(l):
            r1 = rand_ternary(rng, (P * n, N))
            r2 = rand_ternary(rng, (P * n, N))
            e = [dtot64(rng.standard_normal((P * n, N)) * sigma_bk).view(np.uint64) for _ in range(4)]
            m = self.lwe_keys.reshape(-1).astype(np.uint64)
            if pm is not None:
                # tgsw_encrypt_3gen on the device: outputs (bit, part), small operands [r1 rows, r2 rows], torus operands [B_lv, A];
                # the noise and the message m g on coefficient 0 of part_1 / part_3 travel as the addend
                K = P * n
                add = np.stack(e, axis=1).copy()                      # [K][4][N]
                add[:, 0, 0] += m << g[lv]
                add[:, 2, 0] += m << g[lv]
                sm = np.concatenate([r1, r2]).astype(np.int32)
                terms = np.empty((K, 4, 4), np.int32)
                kk = np.arange(K)
                for part, (which, tor) in enumerate(((0, 0), (1, 0), (1, 1), (0, 1))):   # P1 = r1 B, P2 = r2 B, P3 = r2 A, P4 = r1 A
                    terms[:, part] = np.stack([kk * 4 + part, which * K + kk, np.full(K, tor), np.ones(K, np.int64)], axis=1)
                res = pm.mac(sm, np.stack([B[lv], crp.view(np.uint64)]), terms.reshape(-1, 4), 4 * K, add.reshape(-1, N)).view(np.uint64).reshape(K, 4, N)
                P1, P2, P3, P4 = (res[:, q] for q in range(4))
            else:
                MB = negacyclic_matrix_u64(B[lv])
                MA = negacyclic_matrix_u64(crp.view(np.uint64))
                P1 = small_times_torus64(r1, MB) + e[0]
                P2 = small_times_torus64(r2, MB) + e[1]
                P3 = small_times_torus64(r2, MA) + e[2]
                P4 = small_times_torus64(r1, MA) + e[3]
                P1[:, 0] += m << g[lv]
                P3[:, 0] += m << g[lv]
            for part, arr in enumerate((P1, P2, P3, P4)):
                bk[:, :, part, lv, :] = arr.reshape(P, n, N)
        self.bk = bk.view(np.int64)
        if pm is not None:
            pm.close()
        self.ksk = np.stack([gen_keyswitch_key(rng, self.rlwe_keys[q], self.lwe_keys[q], p.ks_t, p.ks_basebit, sigma_ks)
                             for q in range(P)])

    def encrypt(self, bits, seed=0x5EED0002):
        """mk_encrypt_3gen, mk_api.jl:519-536 -> int32[len(bits)][P*n+1]."""
        bits = np.asarray(bits).astype(bool)
        rng = np.random.default_rng(seed)
        p = self.params
        W = p.n * p.parties
        a = rng.integers(-2**31, 2**31, size=(bits.shape[0], W), dtype=np.int64)
        e = dtot32(rng.standard_normal(bits.shape[0]) * self.sigma_lwe).astype(np.int64)
        mu = np.where(bits, MU8, -MU8).astype(np.int64)
        b = mu + e + (a * self.lwe_keys.reshape(-1).astype(np.int64)).sum(axis=1)
        out = np.empty((bits.shape[0], W + 1), np.int32)
        out[:, :W] = a.astype(np.int32)
        out[:, W] = b.astype(np.uint32).view(np.int32)
        return out

    def phase(self, recs):
        """mk_lwe_phase, mk_internals.jl:85-91."""
        W = self.params.n * self.params.parties
        recs = np.asarray(recs, np.int32).reshape(-1, W + 1).astype(np.int64)
        ph = recs[:, -1] - (recs[:, :-1] * self.lwe_keys.reshape(-1).astype(np.int64)).sum(axis=1)
        return ph.astype(np.uint32).view(np.int32)

    def decrypt(self, recs):
        return self.phase(recs) > 0


class CCSSecretKeySet:
    """CCS multi-key material (SecretKey / SharedKey / CloudKeyPart, mk_api.jl:368-384): per party a binary LWE key, a binary RLWE
    key, PublicKey b_i = s (*) a_i + e (mk_internals.jl:209-245), the uni-encryption (d1, f0, f1) of every key bit
    (mk_tgsw_encrypt, :390-448; c0, c1, d0 are not read by UniProduct_old) and a KeyswitchKey."""

    def __init__(self, params, seed=0x5EED0001, sigma_lwe=3.05e-5, sigma_bk=3.72e-9, sigma_ks=3.05e-5, device=None):
        """device: as MKSecretKeySet (None = host products, d = MI355X number d; identical key material)."""
        p = self.params = params
        assert p.k == 1 and p.torus_bits == 32
        rng = np.random.default_rng(seed)
        pm = None
        if device is not None:
            from . import PolyMac
            pm = PolyMac(p.N, 32, device)
        P, n, N, l = p.parties, p.n, p.N, p.l
        self.sigma_lwe = sigma_lwe
        self.lwe_keys = rng.integers(0, 2, (P, n)).astype(np.int32)
        self.rlwe_keys = rng.integers(0, 2, (P, N)).astype(np.int32)
        self.crs = rng.integers(-2**31, 2**31, size=(l, N), dtype=np.int64).astype(np.int32)                 # SharedKey.a
        gauss = lambda shape: dtot32(rng.standard_normal(shape) * sigma_bk).astype(np.int64)
        if pm is not None:   # PublicKey b_i = s (*) a_i + e: outputs (party, level), small = the parties' RLWE keys, torus = the shared a_i
            noise = np.stack([gauss((l, N)) for q in range(P)])
            terms = [(q * l + i, q, i, 1) for q in range(P) for i in range(l)]
            self.pk = pm.mac(self.rlwe_keys, self.crs, terms, P * l, noise.astype(np.uint32).view(np.int32).reshape(-1, N)).reshape(P, l, N).astype(np.int64)
        else:
            self.pk = np.stack([(polymul_small32(self.crs, self.rlwe_keys[q]).astype(np.int64) + gauss((l, N))) for q in range(P)])
        self.pk = self.pk.astype(np.uint32).view(np.int32)
        g = np.array([1 << (32 - (i + 1) * p.Bgbit) for i in range(l)], np.int64)
        bk = np.empty((P, n, 3, l, N), np.int64)
        for q in range(P):
            r = rng.integers(0, 2, (n, N)).astype(np.int64)                                                   # the shared randomness r, one per key bit
            f1 = rng.integers(-2**31, 2**31, size=(n, l, N), dtype=np.int64).astype(np.int32)
            for i in range(l):
                e_d1, e_f0 = gauss((n, N)), gauss((n, N))
                if pm is not None:
                    # mk_tgsw_encrypt on the device: d1_i = r (*) a_i + (e + m g_i), f0_i = s (*) f1_i + (e + r g_i); small = [r rows, s],
                    # torus = [a_i, f1_i rows], outputs (bit, d1 | f0)
                    add = np.stack([e_d1, e_f0 + r * g[i]], axis=1)
                    add[:, 0, 0] += self.lwe_keys[q].astype(np.int64) * g[i]
                    kk = np.arange(n)
                    terms = np.empty((n, 2, 4), np.int32)
                    terms[:, 0] = np.stack([2 * kk, kk, np.zeros(n, np.int64), np.ones(n, np.int64)], axis=1)
                    terms[:, 1] = np.stack([2 * kk + 1, np.full(n, n), 1 + kk, np.ones(n, np.int64)], axis=1)
                    res = pm.mac(np.concatenate([r, self.rlwe_keys[q][None, :]]).astype(np.int32), np.concatenate([self.crs[i][None, :], f1[:, i, :]]),
                                 terms.reshape(-1, 4), 2 * n, add.astype(np.uint32).view(np.int32).reshape(-1, N)).reshape(n, 2, N)
                    bk[q, :, 0, i, :] = res[:, 0]
                    bk[q, :, 1, i, :] = res[:, 1]
                else:
                    # d1_i = e + r (*) a_i + m g_i ; r (*) a_i = rows of r times the fixed polynomial a_i
                    ra = (r.astype(np.float64) @ negacyclic_matrix((self.crs[i].astype(np.int64) & 0xFFFF).astype(np.float64))).astype(np.int64) \
                        + ((r.astype(np.float64) @ negacyclic_matrix((self.crs[i].astype(np.int64) >> 16).astype(np.float64))).astype(np.int64) << 16)
                    d1 = ra + e_d1
                    d1[:, 0] += self.lwe_keys[q].astype(np.int64) * g[i]
                    bk[q, :, 0, i, :] = d1
                    bk[q, :, 1, i, :] = polymul_small32(f1[:, i, :], self.rlwe_keys[q]).astype(np.int64) + e_f0 + r * g[i]   # f0_i = e + s (*) f1_i + r g_i
                bk[q, :, 2, i, :] = f1[:, i, :]
        self.bk = bk.astype(np.uint32).view(np.int32)
        if pm is not None:
            pm.close()
        self.ksk = np.stack([gen_keyswitch_key(rng, self.rlwe_keys[q], self.lwe_keys[q], p.ks_t, p.ks_basebit, sigma_ks) for q in range(P)])

    encrypt = MKSecretKeySet.encrypt
    phase = MKSecretKeySet.phase
    decrypt = MKSecretKeySet.decrypt


class KMSSecretKeySet:
    """Key material of the KMS multi-key scheme (mk_bootstrap_new): per party SecretKey_new (binary LWE key), RLweKey (binary),
    PublicKey b_l = s (*) a_l + e over the shared a (SharedKey_new: the uni gadget), and BootstrapKeyPart_new (new_mk_internals.jl:1-42):
    a fresh binary `rand_key`, gsw_key[j] = tgsw_encrypt(lwe_key[j]) under rand_key (tgsw.jl:88-101 -> rlwe_encrypt_zero rlwe.jl:79-105)
    and the uni-encryption (d1, f0, f1) of the POLYNOMIAL rand_key (mk_tgsw_encrypt, mk_internals.jl:390-446; c0, c1, d0 are never read
    by UniProduct_new), plus a KeyswitchKey from the RLWE key to the LWE key (mk_api.jl:418-436).
    Tables: gsw int64[P][n][2 l_gsw][2][N] (row = block * l + level, column 0 mask / 1 body), uni int64[P][3][l_uni][N], pk int64[P][l_uni][N],
    crs int64[l_uni][N], ksk int32[P][N][t][base-1][n+1]."""

    def __init__(self, params, seed=0x5EED0001, sigma_lwe=3.05e-5, sigma_gsw=4.63e-18, sigma_uni=4.63e-18, sigma_ks=3.05e-5):
        p = self.params = params
        rng = np.random.default_rng(seed)
        P, n, N = p.parties, p.n, p.N
        lg, lu = p.l_gsw, p.l_uni
        self.sigma_lwe = sigma_lwe
        self.lwe_keys = rng.integers(0, 2, (P, n)).astype(np.int32)
        self.rlwe_keys = rng.integers(0, 2, (P, N)).astype(np.int64)
        rand_keys = rng.integers(0, 2, (P, N)).astype(np.int64)
        self.crs = rng.integers(-2**63, 2**63, size=(lu, N), dtype=np.int64)
        gauss = lambda shape, sig: dtot64(rng.standard_normal(shape) * sig).view(np.uint64)
        u64 = lambda a: np.ascontiguousarray(a).view(np.uint64)
        g_gsw = [np.uint64(64 - (q + 1) * p.bg_gsw) for q in range(lg)]
        g_uni = [np.uint64(64 - (q + 1) * p.bg_uni) for q in range(lu)]
        self.pk = np.stack([u64(polymul_small64(self.crs, self.rlwe_keys[q])) + gauss((lu, N), sigma_uni) for q in range(P)]).view(np.int64)
        gsw = np.empty((P, n, 2 * lg, 2, N), np.uint64)
        uni = np.empty((P, 3, lu, N), np.uint64)
        for q in range(P):
            mask = rng.integers(-2**63, 2**63, size=(n * 2 * lg, N), dtype=np.int64)
            body = u64(polymul_small64(mask, rand_keys[q])) + gauss(mask.shape, sigma_gsw)
            gsw[q, :, :, 0, :] = u64(mask).reshape(n, 2 * lg, N)
            gsw[q, :, :, 1, :] = body.reshape(n, 2 * lg, N)
            m = self.lwe_keys[q].astype(np.uint64)
            for blk in range(2):               # + m * gadget on the constant coefficient of polynomial `blk` of row (blk, level)
                for lv in range(lg):
                    gsw[q, :, blk * lg + lv, blk, 0] += m << g_gsw[lv]
            r = rng.integers(0, 2, N).astype(np.int64)
            f1 = rng.integers(-2**63, 2**63, size=(lu, N), dtype=np.int64)
            uni[q, 0] = u64(polymul_small64(self.crs, r)) + gauss((lu, N), sigma_uni)
            uni[q, 1] = u64(polymul_small64(f1, self.rlwe_keys[q])) + gauss((lu, N), sigma_uni)
            for lv in range(lu):
                uni[q, 0, lv] += rand_keys[q].astype(np.uint64) << g_uni[lv]     # d1_l = e + r (*) a_l + rand_key * g_l
                uni[q, 1, lv] += r.astype(np.uint64) << g_uni[lv]                # f0_l = e + s (*) f1_l + r * g_l
            uni[q, 2] = u64(f1)
        self.gsw, self.uni = gsw.view(np.int64), uni.view(np.int64)
        self.ksk = np.stack([gen_keyswitch_key(rng, self.rlwe_keys[q], self.lwe_keys[q], p.ks_t, p.ks_basebit, sigma_ks) for q in range(P)])

    encrypt = MKSecretKeySet.encrypt
    phase = MKSecretKeySet.phase
    decrypt = MKSecretKeySet.decrypt


def negacyclic_matrix_u64(b):
    """Three float64 limb matrices of the torus polynomial b: small (*) b = sum_limb (small @ M_limb) << shift."""
    b = np.asarray(b).view(np.uint64)
    mats = []
    for sh in (0, 22, 44):
        limb = ((b >> np.uint64(sh)) & np.uint64((1 << 22) - 1)).astype(np.float64)
        mats.append((negacyclic_matrix(limb), np.uint64(sh)))
    return mats


def small_times_torus64(small, mats):
    """rows of small-coefficient polynomials (|s| <= 1) times the torus polynomial behind `mats`, exact mod 2^64."""
    s = np.asarray(small, np.float64)
    out = np.zeros(s.shape, np.uint64)
    for M, sh in mats:
        out += (s @ M).astype(np.int64).view(np.uint64) << sh
    return out
