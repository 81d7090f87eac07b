"""ctypes mirrors of the libtfhe structs in include/tfhe_shim.h, and a builder that lays a cloud key set out in memory
exactly as a program linked against libtfhe would hand it to bootsXXX (test infrastructure)."""
import ctypes as C

import numpy as np

i32p = C.POINTER(C.c_int32)


class LweParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("alpha_min", C.c_double), ("alpha_max", C.c_double)]


class LweSample(C.Structure):
    _fields_ = [("a", i32p), ("b", C.c_int32), ("current_variance", C.c_double)]


class LweKeySwitchKey(C.Structure):
    _fields_ = [("n", C.c_int32), ("t", C.c_int32), ("basebit", C.c_int32), ("base", C.c_int32),
                ("out_params", C.POINTER(LweParams)), ("ks0_raw", C.POINTER(LweSample)),
                ("ks1_raw", C.POINTER(C.POINTER(LweSample))), ("ks", C.POINTER(C.POINTER(C.POINTER(LweSample))))]


class TLweParams(C.Structure):
    _fields_ = [("N", C.c_int32), ("k", C.c_int32), ("alpha_min", C.c_double), ("alpha_max", C.c_double),
                ("extracted_lweparams", LweParams)]


class TorusPolynomial(C.Structure):
    _fields_ = [("N", C.c_int32), ("coefsT", i32p)]


class TLweSample(C.Structure):
    _fields_ = [("a", C.POINTER(TorusPolynomial)), ("b", C.POINTER(TorusPolynomial)), ("current_variance", C.c_double), ("k", C.c_int32)]


class TGswParams(C.Structure):
    _fields_ = [("l", C.c_int32), ("Bgbit", C.c_int32), ("Bg", C.c_int32), ("halfBg", C.c_int32), ("maskMod", C.c_uint32),
                ("tlwe_params", C.POINTER(TLweParams)), ("kpl", C.c_int32), ("h", i32p), ("offset", C.c_uint32)]


class TGswSample(C.Structure):
    _fields_ = [("all_sample", C.POINTER(TLweSample)), ("bloc_sample", C.POINTER(C.POINTER(TLweSample))), ("k", C.c_int32), ("l", C.c_int32)]


class LweBootstrappingKey(C.Structure):
    _fields_ = [("in_out_params", C.POINTER(LweParams)), ("bk_params", C.POINTER(TGswParams)), ("accum_params", C.POINTER(TLweParams)),
                ("extract_params", C.POINTER(LweParams)), ("bk", C.POINTER(TGswSample)), ("ks", C.POINTER(LweKeySwitchKey))]


class ParameterSet(C.Structure):
    _fields_ = [("ks_t", C.c_int32), ("ks_basebit", C.c_int32), ("in_out_params", C.POINTER(LweParams)), ("tgsw_params", C.POINTER(TGswParams))]


class CloudKeySet(C.Structure):
    _fields_ = [("params", C.POINTER(ParameterSet)), ("bk", C.POINTER(LweBootstrappingKey)), ("bkFFT", C.c_void_p)]


class TfheKeyImage:
    """Builds the pointer graph of a TFheGateBootstrappingCloudKeySet over flat numpy tables (bk: int32[n][(k+1)l][k+1][N],
    ksk: int32[N][t][base-1][n+1]); keeps every buffer alive."""

    def __init__(self, p, bk, ksk):
        self.keep = []
        n, N, k, l = p.n, p.N, p.k, p.l
        rows, base = (k + 1) * l, 1 << p.ks_basebit
        self.lwe_params = LweParams(n, 2.0**-15, 0.012467)
        self.tlwe_params = TLweParams(N, k, 2.0**-25, 0.012467, LweParams(N * k, 2.0**-25, 0.012467))
        self.tgsw_params = TGswParams(l, p.Bgbit, 1 << p.Bgbit, 1 << (p.Bgbit - 1), (1 << p.Bgbit) - 1, C.pointer(self.tlwe_params), rows, None, 0)
        self.bk_flat = np.ascontiguousarray(bk, np.int32)
        polys = (TorusPolynomial * (n * rows * (k + 1)))()
        rowsamples = (TLweSample * (n * rows))()
        gsw = (TGswSample * n)()
        base_ptr = self.bk_flat.ctypes.data
        for i in range(n):
            for r in range(rows):
                for c in range(k + 1):
                    q = (i * rows + r) * (k + 1) + c
                    polys[q].N = N
                    polys[q].coefsT = C.cast(base_ptr + q * N * 4, i32p)
                rs = rowsamples[i * rows + r]
                rs.a = C.cast(C.byref(polys, ((i * rows + r) * (k + 1)) * C.sizeof(TorusPolynomial)), C.POINTER(TorusPolynomial))
                rs.b = C.cast(C.byref(polys, ((i * rows + r) * (k + 1) + k) * C.sizeof(TorusPolynomial)), C.POINTER(TorusPolynomial))
                rs.k = k
            gsw[i].all_sample = C.cast(C.byref(rowsamples, i * rows * C.sizeof(TLweSample)), C.POINTER(TLweSample))
            gsw[i].k, gsw[i].l = k, l
        # key-switching key with libtfhe's h in [0, base) indexing (entry h = 0 is never read)
        Nin, t = N * k, p.ks_t
        self.ks_full = np.zeros((Nin, t, base, n + 1), np.int32)
        self.ks_full[:, :, 1:, :] = np.ascontiguousarray(ksk, np.int32).reshape(Nin, t, base - 1, n + 1)
        samples = (LweSample * (Nin * t * base))()
        kp = self.ks_full.ctypes.data
        for q in range(Nin * t * base):
            samples[q].a = C.cast(kp + q * (n + 1) * 4, i32p)
            samples[q].b = int(self.ks_full.reshape(-1, n + 1)[q, n])
        lvl1 = (C.POINTER(LweSample) * (Nin * t))()
        for q in range(Nin * t):
            lvl1[q] = C.cast(C.byref(samples, q * base * C.sizeof(LweSample)), C.POINTER(LweSample))
        lvl2 = (C.POINTER(C.POINTER(LweSample)) * Nin)()
        for i in range(Nin):
            lvl2[i] = C.cast(C.byref(lvl1, i * t * C.sizeof(C.POINTER(LweSample))), C.POINTER(C.POINTER(LweSample)))
        self.ks = LweKeySwitchKey(Nin, t, p.ks_basebit, base, C.pointer(self.lwe_params), samples, lvl1, lvl2)
        self.bkey = LweBootstrappingKey(C.pointer(self.lwe_params), C.pointer(self.tgsw_params), C.pointer(self.tlwe_params),
                                        C.pointer(self.tlwe_params.extracted_lweparams), gsw, C.pointer(self.ks))
        self.pset = ParameterSet(t, p.ks_basebit, C.pointer(self.lwe_params), C.pointer(self.tgsw_params))
        self.cloud = CloudKeySet(C.pointer(self.pset), C.pointer(self.bkey), None)
        self.keep += [polys, rowsamples, gsw, samples, lvl1, lvl2]


def make_samples(recs):
    """LweSample array over int32 records [count][n+1]; returns (array, backing buffer)."""
    recs = np.ascontiguousarray(recs, np.int32)
    count, n = recs.shape[0], recs.shape[1] - 1
    a = np.ascontiguousarray(recs[:, :n]).copy()
    arr = (LweSample * count)()
    for g in range(count):
        arr[g].a = C.cast(a.ctypes.data + g * n * 4, i32p)
        arr[g].b = int(recs[g, n])
    return arr, a


def read_samples(arr, a_buf):
    count, n = a_buf.shape
    out = np.empty((count, n + 1), np.int32)
    out[:, :n] = a_buf
    out[:, n] = [arr[g].b for g in range(count)]
    return out
