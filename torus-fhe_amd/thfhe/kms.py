"""KMS multi-key scheme on the engine: mk_gate_nand_new / mk_bootstrap_new (3-gen-mk-tfhe/src/new_mk_gates.jl:1-7,
new_mk_internals.jl:303-325) with the reference's call surface.

Per gate and party the work is (new_mk_internals.jl:276-283):
    levkey_i = mk_ith_blind_rotate(...)        n x l_lev CMuxes on a Torus64 ring of degree 2048 -- > 99 % of the arithmetic:
                                               one fused HIP kernel (thfhe_kms_tlev_rotate, csrc/thfhe_kms.hip)
    accum    = mk_lev_rlwe_mul(accum, levkey_i, uni_key_i, ...)
                                               tlev_extern_mul + UniProduct_new: three rounds of gadget decomposition (integer bit
                                               fields, here) and exact polynomial multiply-accumulates (thfhe_pm_mac on the device)
and at the end mk_rlwe_extract_sample_64 (t64tot32 = trunc(Int32, Float64(d) / 2^32), numeric-functions.jl:109-111, mimicked exactly)
and mk_keyswitch (thfhe_kms_keyswitch).  Records are int32[P n + 1] = a[p n + i], b as for the other multi-key schemes.
There is no CPU fallback: every product and the rotation need the HIP device."""
import ctypes as C

import numpy as np

from . import MU8, MU8_64, NAND, PolyMac, ThfheError, _check, _i32p, _i64p, _p32, _rec, _same_count, _vp, lib

E8, E4 = 1 << 29, 1 << 30
_LIN = {0: (E8, -1, -1), 1: (E8, 1, 1), 2: (-E8, 1, 1), 3: (E4, 2, 2), 4: (-E4, -2, -2), 5: (-E8, -1, -1),
        6: (-E8, -1, 1), 7: (-E8, 1, -1), 8: (E8, -1, 1), 9: (E8, 1, -1)}   # gates.jl:15-161: (cb, cx, cy) by opcode NAND .. ORYN


def decompose64(polys, l, bg):
    """decompose (tgsw.jl:112-138) of Torus64 polynomials int64[..., N] -> signed digits int32[..., l, N], level 1 (most significant) first."""
    v = np.ascontiguousarray(polys, np.int64).view(np.uint64)
    half = np.uint64(1 << (bg - 1))
    offset = np.uint64(0)
    for q in range(1, l + 1):
        offset = offset + (half << np.uint64(64 - q * bg))
    v = v + offset
    out = np.empty(v.shape[:-1] + (l, v.shape[-1]), np.int32)
    for q in range(1, l + 1):
        out[..., q - 1, :] = ((v >> np.uint64(64 - q * bg)) & np.uint64((1 << bg) - 1)).astype(np.int64) - (1 << (bg - 1))
    return out


def t64tot32(d):
    """numeric-functions.jl:109-111: trunc(Int32, Float64(d) / 2^32) -- int64 -> float64 rounds to nearest even, the divide is exact,
    truncation goes toward zero; the measure-zero value 2^31 wraps to INT32_MIN as in the kernels."""
    v = np.trunc(np.asarray(d, np.int64).astype(np.float64) / 4294967296.0)
    return np.where(v >= 2147483648.0, -2147483648.0, v).astype(np.int64).astype(np.int32)


def modswitch(x, N):
    """decode_message(x, 2N) on Int32 words (numeric-functions.jl:70-73)."""
    lg = int(np.log2(2 * N))
    y = (np.asarray(x, np.int32).astype(np.int64) + (1 << (32 - lg - 1))).astype(np.uint32).view(np.int32)
    return (y >> (32 - lg)).astype(np.int32)


class KMSCloudKey:
    """MKCloudKey_new (mk_api.jl:440-455): the parties' TGSW bootstrapping keys and key-switch keys on one MI355X, plus the uni-encryptions,
    public keys and the shared key for the relinearisation products.  Tables as produced by thfhe.keygen.KMSSecretKeySet."""

    def __init__(self, params, gsw, uni, pk, crs, ksk, device=0):
        self.params = p = params
        gsw = np.ascontiguousarray(gsw, np.int64)
        ksk = np.ascontiguousarray(ksk, np.int32)
        if gsw.shape != (p.parties, p.n, 2 * p.l_gsw, 2, p.N):
            raise ValueError("gsw has the wrong shape for these parameters")
        h = _vp()
        _check(lib().thfhe_kms_ctx_create(C.byref(p), gsw.ctypes.data_as(_i64p), _p32(ksk), device, C.byref(h)))
        self.h, self._destroy = h, lib().thfhe_kms_ctx_destroy
        self.pm = PolyMac(p.N, 64, device)
        self.uni = np.ascontiguousarray(uni, np.int64).reshape(p.parties, 3, p.l_uni, p.N)
        self.pk = np.ascontiguousarray(pk, np.int64).reshape(p.parties, p.l_uni, p.N)
        self.crs = np.ascontiguousarray(crs, np.int64).reshape(p.l_uni, p.N)
        self.words = p.parties * p.n + 1

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h and getattr(self, "_destroy", None) is not None:
            self._destroy(h)
        if getattr(self, "pm", None) is not None:
            self.pm.close()
            self.pm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the pieces (same decomposition as the reference) --------------------------------------------------------------------------
    def tlev_rotate(self, party, bara):
        """mk_ith_blind_rotate for a batch: bara int32[count][n] -> int64[count][l_lev][2][N]."""
        p = self.params
        bara = np.ascontiguousarray(bara, np.int32).reshape(-1, p.n)
        lev = np.empty((bara.shape[0], p.l_lev, 2, p.N), np.int64)
        _check(lib().thfhe_kms_tlev_rotate(self.h, party, _p32(bara), lev.ctypes.data_as(_i64p), bara.shape[0]))
        return lev

    def rlwe_rotate(self, party, bara, acc):
        """mk_single_blind_rotate for a batch (new_mk_internals.jl:226-238): acc int64[count][2][N] (mask, body) -> rotated copy."""
        p = self.params
        bara = np.ascontiguousarray(bara, np.int32).reshape(-1, p.n)
        acc = np.array(acc, np.int64, order="C").reshape(bara.shape[0], 2, p.N)
        _check(lib().thfhe_kms_rlwe_rotate(self.h, party, _p32(bara), acc.ctypes.data_as(_i64p), bara.shape[0]))
        return acc

    def lev_rlwe_mul(self, party, accum, lev):
        """mk_lev_rlwe_mul for a batch: accum int64[count][P+1][N] (a_0 .. a_{P-1}, b), lev int64[count][l_lev][2][N] -> new accum."""
        p, pm = self.params, self.pm
        P, N, lv, lu = p.parties, p.N, p.l_lev, p.l_uni
        G = accum.shape[0]
        # (e, f) = tlev_extern_mul of a_0 .. a_{party-1} and b with the TLev sample (tlev.jl:72-76)
        src = list(range(party)) + [P]
        dec = decompose64(accum[:, src], lv, p.bg_lev)                                   # [G][len(src)][lv][N]
        ns = len(src)
        terms = [((g * ns + q) * 2 + w, (g * ns + q) * lv + s, (g * lv + s) * 2 + w, 1) for g in range(G) for q in range(ns) for w in range(2) for s in range(lv)]
        ef = pm.mac(dec.reshape(-1, N), lev.reshape(-1, N), terms, G * ns * 2).reshape(G, ns, 2, N)
        e = np.zeros((G, P + 1, N), np.int64)
        f = np.zeros((G, P + 1, N), np.int64)
        e[:, src], f[:, src] = ef[:, :, 0], ef[:, :, 1]
        up = self.uniproduct(party, e)
        return (f.view(np.uint64) - up.view(np.uint64)).view(np.int64)

    def uniproduct(self, party, e):
        """UniProduct_new (new_mk_internals.jl:85-127) for a batch: e int64[count][P+1][N] -> int64[count][P+1][N]."""
        p, pm = self.params, self.pm
        P, N, lu = p.parties, p.N, p.l_uni
        G = e.shape[0]
        dec = decompose64(e, lu, p.bg_uni)                                                # [G][P+1][lu][N]
        # torus table: d_l (lu), pk_i,l (P lu), a_l (lu)
        torus = np.concatenate([self.uni[party, 0], self.pk.reshape(-1, N), self.crs])
        T_D, T_PK, T_A = 0, lu, lu + P * lu
        sidx = lambda g, i, l: (g * (P + 1) + i) * lu + l
        terms = []
        for g in range(G):
            for i in range(P + 1):                                   # outputs u_0 .. u_{P-1}, u0
                terms += [(g * (P + 2) + i, sidx(g, i, l), T_D + l, 1) for l in range(lu)]
            o = g * (P + 2) + P + 1                                  # output v
            terms += [(o, sidx(g, i, l), T_PK + i * lu + l, 1) for i in range(P) for l in range(lu)]
            terms += [(o, sidx(g, P, l), T_A + l, -1) for l in range(lu)]
        uv = pm.mac(dec.reshape(-1, N), torus, terms, G * (P + 2)).reshape(G, P + 2, N)
        dec_v = decompose64(uv[:, P + 1], lu, p.bg_uni)                                   # [G][lu][N]
        terms = [(g * 2 + w, g * lu + l, w * lu + l, 1) for g in range(G) for w in range(2) for l in range(lu)]
        w01 = pm.mac(dec_v.reshape(-1, N), np.concatenate([self.uni[party, 1], self.uni[party, 2]]), terms, G * 2).reshape(G, 2, N)
        out = uv[:, :P + 1].copy().view(np.uint64)
        out[:, P] += w01[:, 0].view(np.uint64)                       # bnew = u0 + w0
        out[:, party] += w01[:, 1].view(np.uint64)                   # anew[party] += w1
        return out.view(np.int64)

    def bootstrap_wo_keyswitch(self, x, mu=MU8_64, fast_boot=False):
        """mk_bootstrap_wo_keyswitch_new (new_mk_internals.jl:303-314): int32[count][P n + 1] -> int32[count][P N + 1].
        fast_boot: mk_blind_rotate_new_v2 (:255-269) -- party 1 is ONE RLWE rotation of the test vector, accum = f - UniProduct_new(e)."""
        p = self.params
        P, N, n = p.parties, p.N, p.n
        x = _rec(x, self.words)
        G = x.shape[0]
        bar = modswitch(x, N)
        accum = np.zeros((G, P + 1, N), np.int64)
        k = (np.arange(N)[None, :] + bar[:, -1:].astype(np.int64)) % (2 * N)             # X^{-barb} (mu, ..., mu): coefficient q <- e = q + barb
        accum[:, P] = np.where(k >= N, -np.int64(mu), np.int64(mu))
        first = 0
        if fast_boot:
            acc1 = np.zeros((G, 2, N), np.int64)
            acc1[:, 1] = accum[:, P]                                  # rlwe_noiseless_trivial(testvectbis)
            acc1 = self.rlwe_rotate(0, bar[:, :n], acc1)
            e = np.zeros((G, P + 1, N), np.int64)
            e[:, P] = acc1[:, 0]                                      # e = mk_rlwe_noiseless_trivial(mask), f = ...(body)
            accum = np.zeros((G, P + 1, N), np.int64)
            accum[:, P] = acc1[:, 1]
            accum = (accum.view(np.uint64) - self.uniproduct(0, e).view(np.uint64)).view(np.int64)
            first = 1
        for party in range(first, P):
            lev = self.tlev_rotate(party, bar[:, party * n:(party + 1) * n])
            accum = self.lev_rlwe_mul(party, accum, lev)
        # mk_rlwe_extract_sample_64: a'_0 = a_0, a'_j = -a_{N-j} per party, b = body_0, each through t64tot32
        u = np.empty((G, P * N + 1), np.int32)
        a = accum[:, :P].view(np.uint64)
        rev = np.concatenate([a[:, :, :1], (np.uint64(0) - a[:, :, :0:-1])], axis=2).view(np.int64)
        u[:, :P * N] = t64tot32(rev).reshape(G, P * N)
        u[:, P * N] = t64tot32(accum[:, P, 0])
        return u

    def keyswitch(self, u):
        p = self.params
        u = np.ascontiguousarray(u, np.int32).reshape(-1, p.parties * p.N + 1)
        out = np.empty((u.shape[0], self.words), np.int32)
        _check(lib().thfhe_kms_keyswitch(self.h, _p32(u), _p32(out), u.shape[0]))
        return out

    def bootstrap(self, x, mu=MU8_64, fast_boot=False):
        return self.keyswitch(self.bootstrap_wo_keyswitch(x, mu, fast_boot))

    def gates(self, op, x, y, fast_boot=False):
        """Two-input bootstrapped gates on MKLweSample batches; the reference defines the NAND (new_mk_gates.jl:1-7), the other linear
        prologues are those of gates.jl on the same bootstrap."""
        if op not in _LIN:
            raise ThfheError("the KMS scheme evaluates two-input bootstrapped gates (opcodes NAND .. ORYN)")
        x, y = _rec(x, self.words), _rec(y, self.words)
        _same_count(x, y)
        cb, cx, cy = _LIN[op]
        t = (cx * x.astype(np.int64) + cy * y.astype(np.int64))
        t[:, -1] += cb
        return self.bootstrap(t.astype(np.uint32).view(np.int32), fast_boot=fast_boot)


def mk_gate_nand_new(ck, x, y, fast_boot=False):
    """new_mk_gates.jl:1-7; fast_boot selects mk_blind_rotate_new_v2 (the first party's TLev rotation replaced by one RLWE rotation)."""
    return ck.gates(NAND, x, y, fast_boot=fast_boot)


def mk_bootstrap_new(ck, mu, x, fast_boot=False):
    """new_mk_internals.jl:321-325 (bootstrap key and key-switch keys travel together in the KMSCloudKey)."""
    return ck.bootstrap(x, mu, fast_boot=fast_boot)
