"""KMS multi-key scheme on the engine: mk_gate_nand_new / mk_bootstrap_new (3-gen-mk-tfhe/src/new_mk_gates.jl:1-7,
new_mk_internals.jl:303-325) with the reference's call surface.

Per gate and party the work is (new_mk_internals.jl:276-283):
    levkey_i = mk_ith_blind_rotate(...)        n x l_lev CMuxes on a Torus64 ring of degree 2048 -- > 99 % of the arithmetic:
                                               one fused HIP kernel (kms_tlev_rotate_kernel, csrc/thfhe_kms.hip)
    accum    = mk_lev_rlwe_mul(accum, levkey_i, uni_key_i, ...)
                                               tlev_extern_mul + UniProduct_new: three rounds of gadget decomposition and exact polynomial
                                               multiply-accumulates, device-resident (kms_decompose_kernel, pm_mac_kernel)
and at the end mk_rlwe_extract_sample_64 (t64tot32 = trunc(Int32, Float64(d) / 2^32), numeric-functions.jl:109-111, mimicked exactly)
and mk_keyswitch.  A whole gate batch is ONE library call (thfhe_kms_gates): nothing but the input and output records crosses PCIe.
Records are int32[P n + 1] = a[p n + i], b as for the other multi-key schemes.  There is no CPU fallback."""
import ctypes as C

import numpy as np

from . import MU8_64, NAND, ThfheError, _check, _i64p, _p32, _rec, _same_count, _vp, lib

_TWO_INPUT = range(10)   # opcodes NAND .. ORYN (gates.jl:15-161)


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
    """MKCloudKey_new (mk_api.jl:440-455): the parties' TGSW bootstrapping keys, key-switch keys, uni-encryptions, public keys and the
    shared key on one MI355X (all as limb spectra).  Tables as produced by thfhe.keygen.KMSSecretKeySet."""

    def __init__(self, params, gsw, uni, pk, crs, ksk, device=0):
        self.params = p = params
        gsw = np.ascontiguousarray(gsw, np.int64)
        ksk = np.ascontiguousarray(ksk, np.int32)
        uni = np.ascontiguousarray(uni, np.int64)
        pk = np.ascontiguousarray(pk, np.int64)
        crs = np.ascontiguousarray(crs, np.int64)
        if gsw.shape != (p.parties, p.n, 2 * p.l_gsw, 2, p.N):
            raise ValueError("gsw has the wrong shape for these parameters")
        if uni.size != p.parties * 3 * p.l_uni * p.N or pk.size != p.parties * p.l_uni * p.N or crs.size != p.l_uni * p.N:
            raise ValueError("uni / pk / crs have the wrong size for these parameters")
        h = _vp()
        _check(lib().thfhe_kms_ctx_create(C.byref(p), gsw.ctypes.data_as(_i64p), _p32(ksk), device, C.byref(h)))
        self.h, self._destroy = h, lib().thfhe_kms_ctx_destroy
        _check(lib().thfhe_kms_set_relin_keys(h, uni.ctypes.data_as(_i64p), pk.ctypes.data_as(_i64p), crs.ctypes.data_as(_i64p)))
        self.words = p.parties * p.n + 1

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h and getattr(self, "_destroy", None) is not None:
            self._destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pair_threshold(self, max_single_jobs):
        """Launches of <= max_single_jobs rotations run one job per workgroup; larger ones two jobs per workgroup (shared key chunks)."""
        _check(lib().thfhe_kms_set_pair_threshold(self.h, int(max_single_jobs)))

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
        p = self.params
        accum = np.array(accum, np.int64, order="C").reshape(-1, p.parties + 1, p.N)
        lev = np.ascontiguousarray(lev, np.int64).reshape(accum.shape[0], p.l_lev, 2, p.N)
        _check(lib().thfhe_kms_lev_rlwe_mul(self.h, party, accum.ctypes.data_as(_i64p), lev.ctypes.data_as(_i64p), accum.shape[0]))
        return accum

    def _bootstrap(self, x, mu, fast_boot, want_u, want_out):
        p = self.params
        x = _rec(x, self.words)
        G = x.shape[0]
        u = np.empty((G, p.parties * p.N + 1), np.int32) if want_u else None
        out = np.empty((G, self.words), np.int32) if want_out else None
        _check(lib().thfhe_kms_bootstrap(self.h, int(mu), _p32(x), _p32(u) if want_u else None, _p32(out) if want_out else None, G, int(bool(fast_boot))))
        return u, out

    def bootstrap_wo_keyswitch(self, x, mu=MU8_64, fast_boot=False):
        """mk_bootstrap_wo_keyswitch_new (new_mk_internals.jl:303-314): int32[count][P n + 1] -> int32[count][P N + 1].
        fast_boot: mk_blind_rotate_new_v2 (:255-269) -- party 1 is ONE RLWE rotation of the test vector, accum = f - UniProduct_new(e)."""
        return self._bootstrap(x, mu, fast_boot, True, False)[0]

    def keyswitch(self, u):
        p = self.params
        u = np.ascontiguousarray(u, np.int32).reshape(-1, p.parties * p.N + 1)
        out = np.empty((u.shape[0], self.words), np.int32)
        _check(lib().thfhe_kms_keyswitch(self.h, _p32(u), _p32(out), u.shape[0]))
        return out

    def bootstrap(self, x, mu=MU8_64, fast_boot=False):
        return self._bootstrap(x, mu, fast_boot, False, True)[1]

    def gates(self, op, x, y, fast_boot=False):
        """Two-input bootstrapped gates on MKLweSample batches; the reference defines the NAND (new_mk_gates.jl:1-7), the other linear
        prologues are those of gates.jl on the same bootstrap."""
        if op not in _TWO_INPUT:
            raise ThfheError("the KMS scheme evaluates two-input bootstrapped gates (opcodes NAND .. ORYN)")
        x, y = _rec(x, self.words), _rec(y, self.words)
        _same_count(x, y)
        out = np.empty_like(x)
        _check(lib().thfhe_kms_gates(self.h, op, _p32(x), _p32(y), _p32(out), x.shape[0], int(bool(fast_boot))))
        return out


def mk_gate_nand_new(ck, x, y, fast_boot=False):
    """new_mk_gates.jl:1-7; fast_boot selects mk_blind_rotate_new_v2 (the first party's TLev rotation replaced by one RLWE rotation)."""
    return ck.gates(NAND, x, y, fast_boot=fast_boot)


def mk_bootstrap_new(ck, mu, x, fast_boot=False):
    """new_mk_internals.jl:321-325 (bootstrap key and key-switch keys travel together in the KMSCloudKey)."""
    return ck.bootstrap(x, mu, fast_boot=fast_boot)
