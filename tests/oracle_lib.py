"""ctypes binding of the CPU oracle (oracle/libthfhe_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under torus-fhe_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# gate opcodes (same numbering as include/thfhe_hip.h)
NAND, OR, AND, XOR, XNOR, NOR, ANDNY, ANDYN, ORNY, ORYN, MUX, NOT, COPY, AND3 = range(14)

# reference truth tables (3-gen-mk-tfhe/test/runtests.jl:10-42)
TRUTH = {
    NAND: lambda a, b: not (a and b), OR: lambda a, b: a or b, AND: lambda a, b: a and b,
    XOR: lambda a, b: a != b, XNOR: lambda a, b: a == b, NOR: lambda a, b: not (a or b),
    ANDNY: lambda a, b: (not a) and b, ANDYN: lambda a, b: a and (not b),
    ORNY: lambda a, b: (not a) or b, ORYN: lambda a, b: a or (not b),
}


class Params(C.Structure):
    _fields_ = [(f, C.c_int32) for f in
                ("n", "N", "k", "l", "Bgbit", "ks_t", "ks_basebit", "torus_bits", "parties")]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


# parameter sets (J/api.jl:76-115, src/libthfhe.cpp:316-338, J/mk_api.jl:32-146)
PARAM_SETS = {
    "SK-80": dict(n=500, N=1024, k=1, l=2, Bgbit=10, ks_t=8, ks_basebit=2, torus_bits=32, parties=1),
    "SK-128": dict(n=630, N=1024, k=1, l=3, Bgbit=7, ks_t=8, ks_basebit=2, torus_bits=32, parties=1),
    "SK-lib": dict(n=1024, N=1024, k=1, l=3, Bgbit=7, ks_t=8, ks_basebit=2, torus_bits=32, parties=1),
    "MK2": dict(n=520, N=1024, k=1, l=2, Bgbit=7, ks_t=3, ks_basebit=3, torus_bits=64, parties=2),
    "MK3": dict(n=510, N=1024, k=1, l=2, Bgbit=7, ks_t=5, ks_basebit=2, torus_bits=64, parties=3),
    "MK4": dict(n=510, N=1024, k=1, l=3, Bgbit=6, ks_t=5, ks_basebit=2, torus_bits=64, parties=4),
    "MK5": dict(n=520, N=1024, k=1, l=3, Bgbit=6, ks_t=5, ks_basebit=2, torus_bits=64, parties=5),   # mk_api.jl:98-104
    "MK8": dict(n=540, N=1024, k=1, l=4, Bgbit=4, ks_t=5, ks_basebit=2, torus_bits=64, parties=8),   # mk_api.jl:140-146
    "MK4-N2048": dict(n=510, N=2048, k=1, l=3, Bgbit=6, ks_t=5, ks_basebit=2, torus_bits=64, parties=4),
    # the 16 .. 128-party 3-gen sets: ring degree 2048, ONE decomposition level with a 24 .. 26-bit base (mk_api.jl:214-220, 246-252, 268-274, 292-298)
    "MK16": dict(n=590, N=2048, k=1, l=1, Bgbit=26, ks_t=4, ks_basebit=3, torus_bits=64, parties=16),
    "MK32": dict(n=620, N=2048, k=1, l=1, Bgbit=26, ks_t=4, ks_basebit=3, torus_bits=64, parties=32),
    "MK64": dict(n=650, N=2048, k=1, l=1, Bgbit=25, ks_t=4, ks_basebit=3, torus_bits=64, parties=64),
    "MK128": dict(n=670, N=2048, k=1, l=1, Bgbit=24, ks_t=5, ks_basebit=3, torus_bits=64, parties=128),
    "MK32-fft": dict(n=680, N=2048, k=1, l=1, Bgbit=25, ks_t=5, ks_basebit=3, torus_bits=64, parties=32),   # mktfhe_parameters_32party_3gen_for_fft, mk_api.jl:255-261
    # 256 parties: TWO levels with an 18-bit base (mk_api.jl:304-310) -> two 9-bit parts per level, eight row parts: the batched N = 2048 rotation
    "MK256": dict(n=740, N=2048, k=1, l=2, Bgbit=18, ks_t=8, ks_basebit=2, torus_bits=64, parties=256),
    # the ring of degree 4096: mktfhe_parameters_64party_3gen_for_fft, mktfhe_parameters_512party_3gen (mk_api.jl:277-283, 316-322); 27-bit base -> three 9-bit parts
    "MK64-fft": dict(n=720, N=4096, k=1, l=1, Bgbit=27, ks_t=5, ks_basebit=3, torus_bits=64, parties=64),
    "MK512": dict(n=730, N=4096, k=1, l=1, Bgbit=27, ks_t=5, ks_basebit=3, torus_bits=64, parties=512),
    # CCS scheme (mk_bootstrap / mk_gate_nand): mktfhe_parameters_2party / _4party, J/mk_api.jl:4-10,56-62
    "CCS2": dict(n=560, N=1024, k=1, l=3, Bgbit=9, ks_t=8, ks_basebit=2, torus_bits=32, parties=2),
    "CCS4": dict(n=560, N=1024, k=1, l=4, Bgbit=8, ks_t=8, ks_basebit=2, torus_bits=32, parties=4),
    "CCS8": dict(n=560, N=1024, k=1, l=5, Bgbit=6, ks_t=8, ks_basebit=2, torus_bits=32, parties=8),   # mktfhe_parameters_8party, mk_api.jl:111-117
    "CCS16": dict(n=560, N=1024, k=1, l=12, Bgbit=2, ks_t=8, ks_basebit=2, torus_bits=32, parties=16),   # mktfhe_parameters_16party, mk_api.jl:185-191
}
# noise standard deviations (torus units): J/api.jl:101-115 (SK-128: 2^-15 / 2^-25 per src/libthfhe.cpp:325-326),
# J/mk_api.jl:32-38 (MK2), :84-90 (MK4)
SIGMAS = {
    "SK-80": dict(lwe=2.0 ** -15, bk=9.0e-9, ks=2.44e-5),
    "SK-128": dict(lwe=2.0 ** -15, bk=2.0 ** -25, ks=2.0 ** -15),
    "SK-lib": dict(lwe=2.0 ** -15, bk=2.0 ** -25, ks=2.0 ** -15),
    "MK2": dict(lwe=2.0 ** -13.52, bk=2.0 ** -30.70, ks=2.0 ** -13.52),
    "MK3": dict(lwe=2.0 ** -13.26, bk=2.0 ** -30.70, ks=2.0 ** -13.26),
    "MK4": dict(lwe=2.0 ** -13.26, bk=2.0 ** -30.70, ks=2.0 ** -13.26),
    "MK5": dict(lwe=2.0 ** -13.52, bk=2.0 ** -30.70, ks=2.0 ** -13.52),
    "MK8": dict(lwe=2.0 ** -14.04, bk=2.0 ** -30.70, ks=2.0 ** -14.04),
    "MK4-N2048": dict(lwe=2.0 ** -13.26, bk=2.0 ** -30.70, ks=2.0 ** -13.26),
    "MK16": dict(lwe=2.0 ** -15.34, bk=2.0 ** -62.0, ks=2.0 ** -15.34),
    "MK32": dict(lwe=2.0 ** -16.12, bk=2.0 ** -62.0, ks=2.0 ** -16.12),
    "MK64": dict(lwe=2.0 ** -16.90, bk=2.0 ** -62.0, ks=2.0 ** -16.90),
    "MK128": dict(lwe=2.0 ** -17.42, bk=2.0 ** -62.0, ks=2.0 ** -17.42),
    "MK32-fft": dict(lwe=2.0 ** -17.68, bk=2.0 ** -62.0, ks=2.0 ** -17.68),
    "MK256": dict(lwe=2.0 ** -19.24, bk=2.0 ** -62.0, ks=2.0 ** -19.24),
    "MK64-fft": dict(lwe=2.0 ** -18.72, bk=2.0 ** -62.0, ks=2.0 ** -18.72),
    "MK512": dict(lwe=2.0 ** -18.98, bk=2.0 ** -62.0, ks=2.0 ** -18.98),
    "CCS2": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
    "CCS4": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
    "CCS8": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
    "CCS16": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
}


def make_params(name=None, **kw):
    d = dict(PARAM_SETS[name]) if name else {}
    d.update(kw)
    return Params(**d)


def build():
    """Compile the oracle in-tree (gcc only)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "libthfhe_oracle.so", "gen_fixture_key"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "libthfhe_oracle.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ORACLE_DIR, "thfhe_oracle.c")):
        build()
    L = C.CDLL(path)
    i32p, i64p, PP, vp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(Params), C.c_void_p
    sig = {
        "oracle_modswitch": (C.c_int32, [C.c_int32, C.c_int32]),
        "oracle_mul_by_monomial32": (None, [i32p, C.c_int32, C.c_int32, i32p]),
        "oracle_mul_by_monomial64": (None, [i64p, C.c_int32, C.c_int32, i64p]),
        "oracle_decompose32": (None, [i32p, C.c_int32, C.c_int32, C.c_int32, i32p]),
        "oracle_decompose64": (None, [i64p, C.c_int32, C.c_int32, C.c_int32, i64p]),
        "oracle_polymul_schoolbook32": (None, [i32p, i32p, C.c_int32, i32p]),
        "oracle_fft_polymul32": (None, [i32p, i32p, C.c_int32, i32p]),
        "oracle_polymul_schoolbook64": (None, [i64p, i64p, C.c_int32, i64p]),
        "oracle_polymul_ntt32": (None, [i32p, i32p, C.c_int32, i32p]),
        "oracle_polymul_ntt64": (None, [i64p, i64p, C.c_int32, i64p]),
        "oracle_t64tot32": (C.c_int32, [C.c_int64]),
        "oracle_ctx_create": (vp, [PP, i32p, i32p]),
        "oracle_ctx_destroy": (None, [vp]),
        "oracle_mux_rotate": (None, [vp, C.c_int32, C.c_int32, i32p, C.c_int]),
        "oracle_bootstrap_wo_keyswitch": (None, [vp, C.c_int32, i32p, i32p, C.c_int]),
        "oracle_keyswitch": (None, [vp, i32p, i32p]),
        "oracle_gates": (C.c_int, [vp, C.c_int, i32p, i32p, i32p, i32p, C.c_size_t, C.c_int]),
        "oracle_gate_prologue": (C.c_int, [PP, C.c_int, C.c_int, i32p, i32p, i32p, i32p]),
        "oracle_mk_ctx_create": (vp, [PP, i64p, i32p]),
        "oracle_mk_ctx_destroy": (None, [vp]),
        "oracle_mk_mux_rotate": (None, [vp, C.c_int32, C.c_int32, C.c_int32, i64p, C.c_int]),
        "oracle_mk_bootstrap_wo_keyswitch": (None, [vp, C.c_int64, i32p, i32p, C.c_int]),
        "oracle_mk_keyswitch": (None, [vp, i32p, i32p]),
        "oracle_mk_gates": (C.c_int, [vp, C.c_int, i32p, i32p, i32p, i32p, C.c_size_t, C.c_int]),
        "oracle_keygen_sk": (None, [PP, C.c_uint64, C.c_double, C.c_double, i32p, i32p, i32p, i32p, i32p]),
        "oracle_keygen_mk": (None, [PP, C.c_uint64, C.c_double, C.c_double, i32p, i64p, i64p, i32p]),
        "oracle_lwe_encrypt": (None, [i32p, C.c_int32, C.c_int32, C.c_double, C.c_uint64, C.c_uint64, i32p]),
        "oracle_lwe_phase": (C.c_int32, [i32p, C.c_int32, i32p]),
        "oracle_ccs_ctx_create": (vp, [PP, i32p, i32p, i32p, i32p]),
        "oracle_ccs_ctx_destroy": (None, [vp]),
        "oracle_ccs_uniproduct": (None, [vp, C.c_int32, C.c_int32, i32p, i32p, C.c_int]),
        "oracle_ccs_mux_rotate": (None, [vp, C.c_int32, C.c_int32, C.c_int32, i32p, C.c_int]),
        "oracle_ccs_bootstrap_wo_keyswitch": (None, [vp, C.c_int32, i32p, i32p, C.c_int]),
        "oracle_ccs_keyswitch": (None, [vp, i32p, i32p]),
        "oracle_ccs_gates": (C.c_int, [vp, C.c_int, i32p, i32p, i32p, C.c_size_t, C.c_int]),
        "oracle_keygen_ccs": (None, [PP, C.c_uint64, C.c_double, C.c_double, i32p, i32p, i32p, i32p, i32p, i32p]),
        "oracle_tlwe_from_lwe": (None, [i32p, C.c_int32, i32p, i32p]),
        "oracle_partial_decrypt": (None, [i32p, i32p, i32p, C.c_int32, i32p]),
        "oracle_final_decrypt": (C.c_int32, [i32p, i32p, C.c_int32, C.c_int32, i32p]),
        "oracle_kms_ctx_create": (vp, [C.c_void_p, i64p, i64p, i64p, i64p, i32p]),
        "oracle_kms_ctx_destroy": (None, [vp]),
        "oracle_kms_tlev_rotate": (None, [vp, C.c_int32, i32p, i64p, C.c_int]),
        "oracle_kms_uniproduct": (None, [vp, C.c_int32, i64p, i64p, C.c_int]),
        "oracle_kms_lev_rlwe_mul": (None, [vp, C.c_int32, i64p, i64p, C.c_int]),
        "oracle_kms_bootstrap_wo_keyswitch": (None, [vp, C.c_int64, i32p, i32p, C.c_int]),
        "oracle_kms_keyswitch": (None, [vp, i32p, i32p]),
        "oracle_kms_rlwe_rotate": (None, [vp, C.c_int32, i32p, i64p, C.c_int]),
        "oracle_kms_bootstrap_wo_keyswitch_ex": (None, [vp, C.c_int64, i32p, i32p, C.c_int, C.c_int]),
        "oracle_kms_gates_ex": (C.c_int, [vp, C.c_int, i32p, i32p, i32p, C.c_size_t, C.c_int, C.c_int]),
        "oracle_kms_gates": (C.c_int, [vp, C.c_int, i32p, i32p, i32p, C.c_size_t, C.c_int]),
        "oracle_max_threads": (C.c_int, []),
        "oracle_set_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    if "OMP_NUM_THREADS" not in os.environ:
        L.oracle_set_threads(usable_cpus())
    _lib = L
    return L


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota.  (The GPU boxes show 256 CPUs in the
    mask under a 16-CPU quota; an OpenMP team of 256 there just time-slices 16 cores.)"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64)) if a is not None else None


def bk_shape(p):
    return (p.n, (p.k + 1) * p.l, p.k + 1, p.N)


def ksk_shape(p):
    return (p.N * p.k, p.ks_t, (1 << p.ks_basebit) - 1, p.n + 1)


def mk_bk_shape(p):
    return (p.parties, p.n, 4, p.l, p.N)


def mk_ksk_shape(p):
    return (p.parties, p.N, p.ks_t, (1 << p.ks_basebit) - 1, p.n + 1)


class SKKeys:
    """Single-key key material generated by the oracle's deterministic keygen."""

    def __init__(self, params, seed, sigma_bk, sigma_ks, lwe_key=None):
        p = self.params = params
        self.lwe_key = np.zeros(p.n, np.int32)
        self.rlwe_key = np.zeros((p.k, p.N), np.int32)
        self.bk = np.zeros(bk_shape(p), np.int32)
        self.ksk = np.zeros(ksk_shape(p), np.int32)
        kin = np.ascontiguousarray(lwe_key, np.int32) if lwe_key is not None else None
        lib().oracle_keygen_sk(C.byref(p), seed, sigma_bk, sigma_ks, p32(kin), p32(self.lwe_key),
                               p32(self.rlwe_key), p32(self.bk), p32(self.ksk))

    def encrypt_bits(self, bits, sigma, seed):
        p = self.params
        out = np.zeros((len(bits), p.n + 1), np.int32)
        for i, b in enumerate(bits):
            lib().oracle_lwe_encrypt(p32(self.lwe_key), p.n, (1 << 29) if b else -(1 << 29), sigma, seed, i, p32(out[i]))
        return out

    def phases(self, recs):
        recs = np.ascontiguousarray(recs, np.int32).reshape(-1, self.params.n + 1)
        return np.array([lib().oracle_lwe_phase(p32(self.lwe_key), self.params.n, p32(r)) for r in recs], np.int32)

    def decrypt_bits(self, recs):
        return self.phases(recs) > 0


class MKKeys:
    def __init__(self, params, seed, sigma_bk, sigma_ks):
        p = self.params = params
        self.lwe_keys = np.zeros((p.parties, p.n), np.int32)
        self.rlwe_keys = np.zeros((p.parties, p.N), np.int64)
        self.bk = np.zeros(mk_bk_shape(p), np.int64)
        self.ksk = np.zeros(mk_ksk_shape(p), np.int32)
        lib().oracle_keygen_mk(C.byref(p), seed, sigma_bk, sigma_ks, p32(self.lwe_keys), p64(self.rlwe_keys),
                               p64(self.bk), p32(self.ksk))

    def encrypt_bits(self, bits, sigma, seed):
        p = self.params
        n = p.n * p.parties
        out = np.zeros((len(bits), n + 1), np.int32)
        for i, b in enumerate(bits):
            lib().oracle_lwe_encrypt(p32(self.lwe_keys), n, (1 << 29) if b else -(1 << 29), sigma, seed, i, p32(out[i]))
        return out

    def phases(self, recs):
        n = self.params.n * self.params.parties
        recs = np.ascontiguousarray(recs, np.int32).reshape(-1, n + 1)
        return np.array([lib().oracle_lwe_phase(p32(self.lwe_keys), n, p32(r)) for r in recs], np.int32)

    def decrypt_bits(self, recs):
        return self.phases(recs) > 0


def ccs_bk_shape(p):
    return (p.parties, p.n, 3, p.l, p.N)


class CCSKeys:
    """Key material of the CCS multi-key scheme (SecretKey / SharedKey / CloudKeyPart of J/mk_api.jl:368-384)."""

    def __init__(self, params, seed, sigma_bk, sigma_ks):
        p = self.params = params
        self.lwe_keys = np.zeros((p.parties, p.n), np.int32)
        self.rlwe_keys = np.zeros((p.parties, p.N), np.int32)
        self.bk = np.zeros(ccs_bk_shape(p), np.int32)
        self.pk = np.zeros((p.parties, p.l, p.N), np.int32)
        self.crs = np.zeros((p.l, p.N), np.int32)
        self.ksk = np.zeros(mk_ksk_shape(p), np.int32)
        lib().oracle_keygen_ccs(C.byref(p), seed, sigma_bk, sigma_ks, p32(self.lwe_keys), p32(self.rlwe_keys), p32(self.bk), p32(self.pk),
                                p32(self.crs), p32(self.ksk))

    encrypt_bits = MKKeys.encrypt_bits
    phases = MKKeys.phases
    decrypt_bits = MKKeys.decrypt_bits


class CCSOracle:
    def __init__(self, params, K):
        self.params, self.K = params, K
        self.h = lib().oracle_ccs_ctx_create(C.byref(params), p32(K.bk), p32(K.pk), p32(K.crs), p32(K.ksk))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_ccs_ctx_destroy(self.h)
            self.h = None

    def gates(self, op, in0, in1, schoolbook=False):
        in0, in1 = np.ascontiguousarray(in0, np.int32), np.ascontiguousarray(in1, np.int32)
        out = np.zeros_like(in0)
        assert lib().oracle_ccs_gates(self.h, op, p32(in0), p32(in1), p32(out), in0.shape[0], int(schoolbook)) == 0
        return out

    def bootstrap_wo_keyswitch(self, x, mu=1 << 29, schoolbook=False):
        x = np.ascontiguousarray(x, np.int32)
        out = np.zeros(self.params.N * self.params.parties + 1, np.int32)
        lib().oracle_ccs_bootstrap_wo_keyswitch(self.h, mu, p32(x), p32(out), int(schoolbook))
        return out

    def keyswitch(self, u):
        u = np.ascontiguousarray(u, np.int32)
        out = np.zeros(self.params.n * self.params.parties + 1, np.int32)
        lib().oracle_ccs_keyswitch(self.h, p32(u), p32(out))
        return out

    def mux_rotate(self, party, j, barai, acc, schoolbook=False):
        acc = np.ascontiguousarray(acc, np.int32).copy()
        lib().oracle_ccs_mux_rotate(self.h, party, j, barai, p32(acc), int(schoolbook))
        return acc


class Oracle:
    """Single-key oracle context over given (bk, ksk) tables."""

    def __init__(self, params, bk, ksk):
        self.params = params
        self.bk = np.ascontiguousarray(bk, np.int32)
        self.ksk = np.ascontiguousarray(ksk, np.int32)
        assert self.bk.shape == bk_shape(params) and self.ksk.shape == ksk_shape(params)
        self.h = lib().oracle_ctx_create(C.byref(params), p32(self.bk), p32(self.ksk))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_ctx_destroy(self.h)
            self.h = None

    def gates(self, op, in0, in1=None, in2=None, schoolbook=False):
        in0 = np.ascontiguousarray(in0, np.int32)
        in1 = np.ascontiguousarray(in1, np.int32) if in1 is not None else in0
        in2 = np.ascontiguousarray(in2, np.int32) if in2 is not None else None
        out = np.zeros_like(in0)
        rc = lib().oracle_gates(self.h, op, p32(in0), p32(in1), p32(in2), p32(out), in0.shape[0], int(schoolbook))
        assert rc == 0
        return out

    def bootstrap_wo_keyswitch(self, x, mu=1 << 29, schoolbook=False):
        x = np.ascontiguousarray(x, np.int32)
        out = np.zeros(self.params.N * self.params.k + 1, np.int32)
        lib().oracle_bootstrap_wo_keyswitch(self.h, mu, p32(x), p32(out), int(schoolbook))
        return out

    def keyswitch(self, u):
        u = np.ascontiguousarray(u, np.int32)
        out = np.zeros(self.params.n + 1, np.int32)
        lib().oracle_keyswitch(self.h, p32(u), p32(out))
        return out

    def mux_rotate(self, i, barai, acc, schoolbook=False):
        acc = np.ascontiguousarray(acc, np.int32).copy()
        lib().oracle_mux_rotate(self.h, i, barai, p32(acc), int(schoolbook))
        return acc


class KmsParams(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "N", "parties", "l_gsw", "bg_gsw", "l_lev", "bg_lev", "l_uni", "bg_uni", "ks_t", "ks_basebit")]


class KMSOracle:
    """KMS scheme (mk_bootstrap_new) oracle context over given key tables (layouts: thfhe_oracle.c, section KMS)."""

    def __init__(self, params, gsw, uni, pk, crs, ksk):
        self.params = p = KmsParams(**{f: getattr(params, f) for f, _ in KmsParams._fields_})
        self.tabs = [np.ascontiguousarray(a, np.int64) for a in (gsw, uni, pk, crs)] + [np.ascontiguousarray(ksk, np.int32)]
        assert self.tabs[0].shape == (p.parties, p.n, 2 * p.l_gsw, 2, p.N) and self.tabs[1].shape == (p.parties, 3, p.l_uni, p.N)
        self.h = lib().oracle_kms_ctx_create(C.byref(p), *[p64(a) for a in self.tabs[:4]], p32(self.tabs[4]))
        self.words = p.parties * p.n + 1

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_kms_ctx_destroy(self.h)
            self.h = None

    def gates(self, op, in0, in1, schoolbook=False, fast_boot=False):
        in0, in1 = np.ascontiguousarray(in0, np.int32), np.ascontiguousarray(in1, np.int32)
        out = np.zeros_like(in0)
        assert lib().oracle_kms_gates_ex(self.h, op, p32(in0), p32(in1), p32(out), in0.shape[0], int(schoolbook), int(fast_boot)) == 0
        return out

    def rlwe_rotate(self, party, bara, acc, schoolbook=False):
        acc = np.ascontiguousarray(acc, np.int64).copy()
        lib().oracle_kms_rlwe_rotate(self.h, party, p32(np.ascontiguousarray(bara, np.int32)), p64(acc), int(schoolbook))
        return acc

    def uniproduct(self, party, e, schoolbook=False):
        e = np.ascontiguousarray(e, np.int64)
        out = np.zeros_like(e)
        lib().oracle_kms_uniproduct(self.h, party, p64(e), p64(out), int(schoolbook))
        return out

    def tlev_rotate(self, party, bara, schoolbook=False):
        p = self.params
        lev = np.zeros((p.l_lev, 2, p.N), np.int64)
        lib().oracle_kms_tlev_rotate(self.h, party, p32(np.ascontiguousarray(bara, np.int32)), p64(lev), int(schoolbook))
        return lev

    def lev_rlwe_mul(self, party, accum, lev, schoolbook=False):
        accum = np.ascontiguousarray(accum, np.int64).copy()
        lib().oracle_kms_lev_rlwe_mul(self.h, party, p64(accum), p64(np.ascontiguousarray(lev, np.int64)), int(schoolbook))
        return accum

    def bootstrap_wo_keyswitch(self, x, mu=1 << 61, schoolbook=False, fast_boot=False):
        p = self.params
        out = np.zeros(p.parties * p.N + 1, np.int32)
        lib().oracle_kms_bootstrap_wo_keyswitch_ex(self.h, mu, p32(np.ascontiguousarray(x, np.int32)), p32(out), int(schoolbook), int(fast_boot))
        return out

    def keyswitch(self, u):
        out = np.zeros(self.words, np.int32)
        lib().oracle_kms_keyswitch(self.h, p32(np.ascontiguousarray(u, np.int32)), p32(out))
        return out


class MKOracle:
    def __init__(self, params, bk, ksk):
        self.params = params
        self.bk = np.ascontiguousarray(bk, np.int64)
        self.ksk = np.ascontiguousarray(ksk, np.int32)
        assert self.bk.shape == mk_bk_shape(params) and self.ksk.shape == mk_ksk_shape(params)
        self.h = lib().oracle_mk_ctx_create(C.byref(params), p64(self.bk), p32(self.ksk))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_mk_ctx_destroy(self.h)
            self.h = None

    def gates(self, op, in0, in1=None, in2=None, schoolbook=False):
        in0 = np.ascontiguousarray(in0, np.int32)
        in1 = np.ascontiguousarray(in1, np.int32) if in1 is not None else in0
        in2 = np.ascontiguousarray(in2, np.int32) if in2 is not None else None
        out = np.zeros_like(in0)
        rc = lib().oracle_mk_gates(self.h, op, p32(in0), p32(in1), p32(in2), p32(out), in0.shape[0], int(schoolbook))
        assert rc == 0
        return out

    def bootstrap_wo_keyswitch(self, x, mu=1 << 61, schoolbook=False):
        x = np.ascontiguousarray(x, np.int32)
        out = np.zeros(self.params.N + 1, np.int32)
        lib().oracle_mk_bootstrap_wo_keyswitch(self.h, mu, p32(x), p32(out), int(schoolbook))
        return out

    def keyswitch(self, u):
        u = np.ascontiguousarray(u, np.int32)
        out = np.zeros(self.params.n * self.params.parties + 1, np.int32)
        lib().oracle_mk_keyswitch(self.h, p32(u), p32(out))
        return out

    def mux_rotate(self, party, i, barai, acc, schoolbook=False):
        acc = np.ascontiguousarray(acc, np.int64).copy()
        lib().oracle_mk_mux_rotate(self.h, party, i, barai, p64(acc), int(schoolbook))
        return acc


# ---- reference fixtures (tests/golden/*.data = /root/reference/test/bootstrap_modules/*.data) --------
def load_fixture_records(name, n=630):
    """libtfhe ciphertext file: 32 records of (int32 type=42, int32 a[n], int32 b, double variance)."""
    raw = np.fromfile(os.path.join(GOLDEN, name), dtype=np.uint8)
    rec_bytes = 4 + 4 * n + 4 + 8
    assert raw.size == 32 * rec_bytes, raw.size
    raw = raw.reshape(32, rec_bytes)
    types = raw[:, :4].copy().view(np.int32).ravel()
    words = raw[:, 4:4 + 4 * (n + 1)].copy().view(np.int32)
    var = raw[:, 4 + 4 * (n + 1):].copy().view(np.float64).ravel()
    return types, words, var


def fixture_key():
    with open(os.path.join(GOLDEN, "fixture_lwe_key.txt")) as f:
        s = f.read().strip()
    return np.array([int(ch) for ch in s], np.int32)


def bits_to_int_msb_first(bits):
    v = 0
    for b in bits:
        v = (v << 1) | int(bool(b))
    return v
