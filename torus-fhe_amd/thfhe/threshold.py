"""LWE -> TLWE conversion and threshold partial / final decryption on the GPU: the step after the gate
path in the reference's C++ applications (src/KNN_medical_data.cpp ciphertext_conversion_threshold_decryption, src/libthfhe.cpp).
Function names and argument meaning follow the reference; samples are numpy int32 arrays, batched over the leading axis."""
import ctypes as C

import numpy as np

from . import ThfheError, _check, _p32, _vp, lib


class PolyContext:
    """Device context for the ring operations (N = 1024, k = 1)."""

    def __init__(self, device=0, N=1024):
        self.N = N
        h = _vp()
        _check(lib().thfhe_poly_ctx_create(device, N, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().thfhe_poly_ctx_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def TLweFromLwe(ctx, cipher):
    """src/libthfhe.cpp:340-348: LWE records int32[count][N+1] -> (a int32[count][N], b int32[count][N])."""
    x = np.ascontiguousarray(cipher, np.int32).reshape(-1, ctx.N + 1)
    a, b = np.empty((x.shape[0], ctx.N), np.int32), np.empty((x.shape[0], ctx.N), np.int32)
    _check(lib().thfhe_tlwe_from_lwe(ctx.h, _p32(x), _p32(a), _p32(b), x.shape[0]))
    return a, b


def PartialDecrypt(ctx, key_share, tlwe_a, noise=None):
    """ThFHEKeyShare::PartialDecrypt, src/libthfhe.cpp:270-293: key_share (*) a + smudging noise, exact mod 2^32."""
    s = np.ascontiguousarray(key_share, np.int32).reshape(ctx.N)
    a = np.ascontiguousarray(tlwe_a, np.int32).reshape(-1, ctx.N)
    e = np.ascontiguousarray(noise, np.int32).reshape(a.shape) if noise is not None else None
    out = np.empty_like(a)
    _check(lib().thfhe_partial_decrypt(ctx.h, _p32(s), _p32(a), _p32(e), _p32(out), a.shape[0]))
    return out


def finalDecrypt(ctx, tlwe_b, partial_ciphertexts, want_result=False):
    """src/libthfhe.cpp:296-315: b - partial_0 + sum_{i>=1} partial_i; message bit = coefficient 0 > 0."""
    b = np.ascontiguousarray(tlwe_b, np.int32).reshape(-1, ctx.N)
    parts = np.ascontiguousarray(partial_ciphertexts, np.int32).reshape(-1, b.shape[0], ctx.N)
    bits = np.empty(b.shape[0], np.int32)
    res = np.empty_like(b) if want_result else None
    _check(lib().thfhe_final_decrypt(ctx.h, _p32(b), _p32(parts), parts.shape[0], _p32(res), _p32(bits), b.shape[0]))
    return (bits.astype(bool), res) if want_result else bits.astype(bool)
