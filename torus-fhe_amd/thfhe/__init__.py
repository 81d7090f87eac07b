"""thfhe -- Python host layer over the C ABI of libthfhe_hip.so (include/thfhe_hip.h).

Mirrors the reference's gate/bootstrapping interface for the hot path (3-gen-mk-tfhe/src/gates.jl,
bootstrap.jl, keyswitch.jl, 3gen_mk_gates.jl, 3gen_mk_internals.jl): same function names, argument
order and meaning, with numpy int32 LWE records ([..., n+1] = a..., b) instead of Julia structs.
There is no CPU fallback: importing works anywhere (so the ABI can be inspected), but creating a
context or evaluating a gate without a usable HIP device raises ThfheError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("THFHE_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libthfhe_hip.so"))

# gate opcodes (include/thfhe_hip.h enum thfhe_gate)
NAND, OR, AND, XOR, XNOR, NOR, ANDNY, ANDYN, ORNY, ORYN, MUX, NOT, COPY, AND3 = range(14)

MU8 = 1 << 29     # encode_message(1, 8), Torus32      (numeric-functions.jl:86-89)
MU8_64 = 1 << 61  # encode_message64(1, 8), Torus64    (numeric-functions.jl:92-95)


class ThfheError(RuntimeError):
    pass


class Params(C.Structure):
    """thfhe_params (include/thfhe_hip.h)."""
    _fields_ = [(f, C.c_int32) for f in
                ("n", "N", "k", "l", "Bgbit", "ks_t", "ks_basebit", "torus_bits", "parties")]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


# the reference's parameter tables for this path (api.jl:76-115, src/libthfhe.cpp:316-338, mk_api.jl:32-146)
PARAM_SETS = {
    "SK-80": dict(n=500, N=1024, k=1, l=2, Bgbit=10, ks_t=8, ks_basebit=2, torus_bits=32, parties=1),
    "SK-128": dict(n=630, N=1024, k=1, l=3, Bgbit=7, ks_t=8, ks_basebit=2, torus_bits=32, parties=1),
    "SK-lib": dict(n=1024, N=1024, k=1, l=3, Bgbit=7, ks_t=8, ks_basebit=2, torus_bits=32, parties=1),
    "MK2": dict(n=520, N=1024, k=1, l=2, Bgbit=7, ks_t=3, ks_basebit=3, torus_bits=64, parties=2),
    "MK3": dict(n=510, N=1024, k=1, l=2, Bgbit=7, ks_t=5, ks_basebit=2, torus_bits=64, parties=3),
    "MK4": dict(n=510, N=1024, k=1, l=3, Bgbit=6, ks_t=5, ks_basebit=2, torus_bits=64, parties=4),
    "MK5": dict(n=520, N=1024, k=1, l=3, Bgbit=6, ks_t=5, ks_basebit=2, torus_bits=64, parties=5),   # mk_api.jl:98-104
    "MK8": dict(n=540, N=1024, k=1, l=4, Bgbit=4, ks_t=5, ks_basebit=2, torus_bits=64, parties=8),   # mk_api.jl:140-146
    # BASELINE.json configs[4] wording ("4-party 3-gen MK-TFHE, N=2048 l=3"): the reference's 4-party set on the larger ring
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
    # CCS scheme (mk_bootstrap / mk_gate_nand): mktfhe_parameters_2party / _4party, mk_api.jl:4-10,56-62
    "CCS2": dict(n=560, N=1024, k=1, l=3, Bgbit=9, ks_t=8, ks_basebit=2, torus_bits=32, parties=2),
    "CCS4": dict(n=560, N=1024, k=1, l=4, Bgbit=8, ks_t=8, ks_basebit=2, torus_bits=32, parties=4),
    "CCS8": dict(n=560, N=1024, k=1, l=5, Bgbit=6, ks_t=8, ks_basebit=2, torus_bits=32, parties=8),   # mktfhe_parameters_8party, mk_api.jl:111-117
    "CCS16": dict(n=560, N=1024, k=1, l=12, Bgbit=2, ks_t=8, ks_basebit=2, torus_bits=32, parties=16),   # mktfhe_parameters_16party, mk_api.jl:185-191
}


# Noise standard deviations of the reference's parameter sets (lwe = fresh ciphertexts and key-switch key, bk = bootstrapping key):
# api.jl:76-115, mk_api.jl:4-10,32-38,44-50,56-62,84-90,98-104,111-117,140-146,214-220,246-252,268-274,292-298; SK-lib = libthfhe.cpp:316-338.
# MK4-N2048 (BASELINE.json configs[4] wording) inherits the 4-party set's.  Indexed with [] on purpose: an unknown set must raise.
SIGMAS = {
    "SK-80": dict(lwe=2.0**-15, bk=9.0e-9, ks=2.44e-5),
    "SK-128": dict(lwe=2.0**-15, bk=2.0**-25, ks=2.0**-15),
    "SK-lib": dict(lwe=2.0**-15, bk=2.0**-25, ks=2.0**-15),
    "MK2": dict(lwe=2.0**-13.52, bk=2.0**-30.70, ks=2.0**-13.52),
    "MK3": dict(lwe=2.0**-13.26, bk=2.0**-30.70, ks=2.0**-13.26),
    "MK4": dict(lwe=2.0**-13.26, bk=2.0**-30.70, ks=2.0**-13.26),
    "MK5": dict(lwe=2.0**-13.52, bk=2.0**-30.70, ks=2.0**-13.52),
    "MK8": dict(lwe=2.0**-14.04, bk=2.0**-30.70, ks=2.0**-14.04),
    "MK4-N2048": dict(lwe=2.0**-13.26, bk=2.0**-30.70, ks=2.0**-13.26),
    "MK16": dict(lwe=2.0**-15.34, bk=2.0**-62.0, ks=2.0**-15.34),
    "MK32": dict(lwe=2.0**-16.12, bk=2.0**-62.0, ks=2.0**-16.12),
    "MK64": dict(lwe=2.0**-16.90, bk=2.0**-62.0, ks=2.0**-16.90),
    "MK128": dict(lwe=2.0**-17.42, bk=2.0**-62.0, ks=2.0**-17.42),
    "MK32-fft": dict(lwe=2.0**-17.68, bk=2.0**-62.0, ks=2.0**-17.68),
    "MK256": dict(lwe=2.0**-19.24, bk=2.0**-62.0, ks=2.0**-19.24),
    "MK64-fft": dict(lwe=2.0**-18.72, bk=2.0**-62.0, ks=2.0**-18.72),
    "MK512": dict(lwe=2.0**-18.98, bk=2.0**-62.0, ks=2.0**-18.98),
    "CCS2": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
    "CCS4": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
    "CCS8": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
    "CCS16": dict(lwe=3.05e-5, bk=3.72e-9, ks=3.05e-5),
}


class KmsParams(C.Structure):
    """thfhe_kms_params (include/thfhe_hip.h): the KMS scheme (mk_bootstrap_new) has three gadget families -- gsw (per-party TGSW blind
    rotation of the TLev accumulator), lev (the TLev accumulator), uni (uni-encryption / public keys) -- on a Torus64 ring."""
    _fields_ = [(f, C.c_int32) for f in ("n", "N", "parties", "l_gsw", "bg_gsw", "l_lev", "bg_lev", "l_uni", "bg_uni", "ks_t", "ks_basebit")]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


# mktfhe_parameters_{2,4,8}party_new and _fast, mk_api.jl:12-30, 64-82, 120-138 (noise: lwe / ks 3.05e-5, gsw / uni 4.63e-18)
KMS_PARAM_SETS = {
    "KMS2": dict(n=560, N=2048, parties=2, l_gsw=3, bg_gsw=13, l_lev=2, bg_lev=7, l_uni=2, bg_uni=13, ks_t=8, ks_basebit=2),
    "KMS2-fast": dict(n=560, N=2048, parties=2, l_gsw=3, bg_gsw=13, l_lev=2, bg_lev=7, l_uni=3, bg_uni=10, ks_t=8, ks_basebit=2),
    "KMS4": dict(n=560, N=2048, parties=4, l_gsw=5, bg_gsw=8, l_lev=2, bg_lev=8, l_uni=5, bg_uni=8, ks_t=8, ks_basebit=2),
    "KMS8": dict(n=560, N=2048, parties=8, l_gsw=4, bg_gsw=11, l_lev=3, bg_lev=6, l_uni=8, bg_uni=4, ks_t=8, ks_basebit=2),
    "KMS16": dict(n=560, N=2048, parties=16, l_gsw=5, bg_gsw=9, l_lev=3, bg_lev=6, l_uni=9, bg_uni=4, ks_t=8, ks_basebit=2),    # mk_api.jl:194-202
    "KMS32": dict(n=560, N=2048, parties=32, l_gsw=6, bg_gsw=8, l_lev=3, bg_lev=7, l_uni=16, bg_uni=2, ks_t=8, ks_basebit=2),   # mk_api.jl:225-233
    # the `_fast` twins (mk_api.jl:74-82, 130-138, 204-212, 235-243): other uni-encryption gadgets, same rotation; the 32-party one equals `_new`
    "KMS4-fast": dict(n=560, N=2048, parties=4, l_gsw=5, bg_gsw=8, l_lev=2, bg_lev=8, l_uni=7, bg_uni=6, ks_t=8, ks_basebit=2),
    "KMS8-fast": dict(n=560, N=2048, parties=8, l_gsw=4, bg_gsw=11, l_lev=3, bg_lev=6, l_uni=7, bg_uni=4, ks_t=8, ks_basebit=2),
    "KMS16-fast": dict(n=560, N=2048, parties=16, l_gsw=5, bg_gsw=9, l_lev=3, bg_lev=6, l_uni=7, bg_uni=4, ks_t=8, ks_basebit=2),
    "KMS32-fast": dict(n=560, N=2048, parties=32, l_gsw=6, bg_gsw=8, l_lev=3, bg_lev=7, l_uni=16, bg_uni=2, ks_t=8, ks_basebit=2),
}


def make_kms_params(name=None, **kw):
    d = dict(KMS_PARAM_SETS[name]) if name else {}
    d.update(kw)
    return KmsParams(**d)


def make_params(name=None, **kw):
    d = dict(PARAM_SETS[name]) if name else {}
    d.update(kw)
    return Params(**d)


_lib = None

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# symbol -> (restype, argtypes); tests/test_abi.py checks every symbol declared in include/*.h is exported
SIGNATURES = {
    "thfhe_last_error": (C.c_char_p, []),
    "thfhe_device_count": (C.c_int, []),
    "thfhe_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "thfhe_ctx_create": (C.c_int, [C.POINTER(Params), _i32p, _i32p, C.c_int, C.POINTER(_vp)]),
    "thfhe_ctx_destroy": (None, [_vp]),
    "thfhe_ctx_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "thfhe_gates": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p, _i32p, C.c_size_t]),
    "thfhe_gates_mixed": (C.c_int, [_vp, _i32p, _i32p, _i32p, _i32p, C.c_size_t]),
    "thfhe_dag_run": (C.c_int, [_vp, _i32p, C.c_size_t, _i32p, C.c_size_t, _i64p]),
    "thfhe_set_dag_slice": (C.c_int, [_vp, C.c_size_t]),
    "thfhe_mk_set_dag_slice": (C.c_int, [_vp, C.c_size_t]),
    "thfhe_dag_run_batch": (C.c_int, [_vp, _i32p, C.c_size_t, _i32p, C.c_size_t, C.c_size_t, _i32p, C.c_size_t, _i32p, _i64p]),
    "thfhe_bootstrap": (C.c_int, [_vp, C.c_int32, _i32p, _i32p, C.c_size_t]),
    "thfhe_bootstrap_wo_keyswitch": (C.c_int, [_vp, C.c_int32, _i32p, _i32p, C.c_size_t]),
    "thfhe_keyswitch": (C.c_int, [_vp, _i32p, _i32p, C.c_size_t]),
    "thfhe_dev_alloc": (_vp, [_vp, C.c_size_t]),
    "thfhe_dev_free": (None, [_vp, _vp]),
    "thfhe_copy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "thfhe_copy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "thfhe_reserve": (C.c_int, [_vp, C.c_size_t]),
    "thfhe_gates_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t]),
    "thfhe_sync": (C.c_int, [_vp]),
    "thfhe_set_coop_threshold": (C.c_int, [_vp, C.c_int]),
    "thfhe_set_ring4_threshold": (C.c_int, [_vp, C.c_int]),
    "thfhe_set_profiling": (C.c_int, [_vp, C.c_int]),
    "thfhe_last_timings": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "thfhe_ccs_ctx_create": (C.c_int, [C.POINTER(Params), _i32p, _i32p, _i32p, _i32p, C.c_int, C.POINTER(_vp)]),
    "thfhe_ccs_ctx_destroy": (None, [_vp]),
    "thfhe_ccs_gates": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p, C.c_size_t]),
    "thfhe_ccs_bootstrap": (C.c_int, [_vp, C.c_int32, _i32p, _i32p, C.c_size_t]),
    "thfhe_poly_ctx_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "thfhe_poly_ctx_destroy": (None, [_vp]),
    "thfhe_tlwe_from_lwe": (C.c_int, [_vp, _i32p, _i32p, _i32p, C.c_size_t]),
    "thfhe_partial_decrypt": (C.c_int, [_vp, _i32p, _i32p, _i32p, _i32p, C.c_size_t]),
    "thfhe_final_decrypt": (C.c_int, [_vp, _i32p, _i32p, C.c_int, _i32p, _i32p, C.c_size_t]),
    "thfhe_kms_ctx_create": (C.c_int, [_vp, _i64p, _i32p, C.c_int, C.POINTER(_vp)]),
    "thfhe_kms_ctx_destroy": (None, [_vp]),
    "thfhe_kms_tlev_rotate": (C.c_int, [_vp, C.c_int, _i32p, _i64p, C.c_size_t]),
    "thfhe_kms_rlwe_rotate": (C.c_int, [_vp, C.c_int, _i32p, _i64p, C.c_size_t]),
    "thfhe_kms_set_relin_keys": (C.c_int, [_vp, _i64p, _i64p, _i64p]),
    "thfhe_kms_lev_rlwe_mul": (C.c_int, [_vp, C.c_int, _i64p, _i64p, C.c_size_t]),
    "thfhe_kms_bootstrap": (C.c_int, [_vp, C.c_int64, _i32p, _i32p, _i32p, C.c_size_t, C.c_int]),
    "thfhe_kms_gates": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p, C.c_size_t, C.c_int]),
    "thfhe_kms_rotate_parties_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t]),
    "thfhe_kms_finish_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t]),
    "thfhe_kms_set_stream": (C.c_int, [_vp, _vp]),
    "thfhe_kms_set_pair_threshold": (C.c_int, [_vp, C.c_long]),
    "thfhe_kms_keyswitch": (C.c_int, [_vp, _i32p, _i32p, C.c_size_t]),
    "thfhe_pm_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "thfhe_pm_ctx_destroy": (None, [_vp]),
    "thfhe_pm_mac": (C.c_int, [_vp, _i32p, C.c_size_t, _vp, C.c_size_t, _i32p, C.c_size_t, _vp, _vp, C.c_size_t]),
    "thfhe_mk_ctx_create": (C.c_int, [C.POINTER(Params), _i64p, _i32p, C.c_int, C.POINTER(_vp)]),
    "thfhe_mk_ctx_destroy": (None, [_vp]),
    "thfhe_mk_gates": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p, _i32p, C.c_size_t]),
    "thfhe_mk_gates_mixed": (C.c_int, [_vp, _i32p, _i32p, _i32p, _i32p, C.c_size_t]),
    "thfhe_mk_dag_run": (C.c_int, [_vp, _i32p, C.c_size_t, _i32p, C.c_size_t, _i64p]),
    "thfhe_mk_dag_run_batch": (C.c_int, [_vp, _i32p, C.c_size_t, _i32p, C.c_size_t, C.c_size_t, _i32p, C.c_size_t, _i32p, _i64p]),
    "thfhe_mk_bootstrap": (C.c_int, [_vp, C.c_int64, _i32p, _i32p, C.c_size_t]),
    "thfhe_mk_prologue_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_size_t]),
    "thfhe_mk_set_stream": (C.c_int, [_vp, _vp]),
    "thfhe_mk_set_pair_threshold": (C.c_int, [_vp, C.c_long]),
    "thfhe_mk_rotate_partial_dev": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_size_t]),
    "thfhe_mk_extract_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "thfhe_mk_keyswitch_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "thfhe_mk_dev_alloc": (_vp, [_vp, C.c_size_t]),
    "thfhe_mk_dev_free": (None, [_vp, _vp]),
    "thfhe_mk_copy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "thfhe_mk_copy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "thfhe_mk_reserve": (C.c_int, [_vp, C.c_size_t]),
    "thfhe_mk_gates_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t]),
    "thfhe_mk_sync": (C.c_int, [_vp]),
    "thfhe_mk_set_profiling": (C.c_int, [_vp, C.c_int]),
    "thfhe_mk_last_timings": (C.c_int, [_vp, C.POINTER(C.c_float)]),
}


torch_loaded_first = None  # set by lib(): whether torch's HIP runtime was already in the process when the engine was loaded


def _needs_torch_first():
    """Multi-rank jobs run under torch.distributed (RCCL).  torch bundles its own HIP runtime; if libthfhe_hip.so (rpath
    /opt/rocm/lib) is loaded BEFORE torch, torch's later-loaded runtime sees no GPU.  So in a multi-rank job -- or whenever the
    caller asks with THFHE_TORCH_FIRST=1 -- torch is imported first, and both share torch's runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return False
    want = os.environ.get("THFHE_TORCH_FIRST")
    if want is None:
        want = "1" if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 else "0"
    return want == "1" and importlib.util.find_spec("torch") is not None


def lib():
    """Load libthfhe_hip.so (built in-tree by __graft_entry__.build()); fail loudly if it is missing."""
    global _lib, torch_loaded_first
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ThfheError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        if _needs_torch_first():
            import torch  # noqa: F401  (runtime-order rule above)
        import sys
        torch_loaded_first = "torch" in sys.modules
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise ThfheError(f"libthfhe_hip error {rc}: {lib().thfhe_last_error().decode()}")


def _p32(a):
    return a.ctypes.data_as(_i32p) if a is not None else None


def _rec(a, words):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if a.shape[-1] != words:
        raise ValueError(f"expected records of {words} int32 words, got shape {a.shape}")
    return a.reshape(-1, words)


def _same_count(x, *others):
    """The C side copies x.shape[0] records from every operand: a shorter one would be read out of bounds."""
    for o in others:
        if o is not None and o.shape[0] != x.shape[0]:
            raise ValueError(f"operand batches differ in length: {x.shape[0]} vs {o.shape[0]} records")


class DeviceBuffer:
    """A device allocation owned by a context (records resident in HBM)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        self.ptr = ctx._alloc(nbytes)
        if not self.ptr:
            raise ThfheError("device allocation failed")

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._h2d(self.ptr, arr)
        return self

    def download(self, shape, dtype=np.int32):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._d2h(out, self.ptr)
        return out

    def free(self):
        if self.ptr:
            self.ctx._free(self.ptr)
            self.ptr = None


def _dag_run_batch(fn, h, words, input_records, gates, out_wires):
    """Shared body of CloudKey.dag_run_batch / MKCloudKey.dag_run_batch."""
    x = np.ascontiguousarray(input_records, np.int32)
    if x.ndim != 3 or x.shape[2] != words:
        raise ValueError("dag_run_batch: input records must be int32[instances][n_inputs][%d]" % words)
    g = np.ascontiguousarray(gates, np.int32).reshape(-1, 4)
    q, n_in = x.shape[0], x.shape[1]
    sel = None if out_wires is None else np.ascontiguousarray(out_wires, np.int32).reshape(-1)
    out = np.zeros((q, g.shape[0] if sel is None else sel.shape[0], words), np.int32)
    st = np.zeros(4, np.int64)
    _check(fn(h, _p32(x), n_in, _p32(g), g.shape[0], q, _p32(sel), 0 if sel is None else sel.shape[0], _p32(out), st.ctypes.data_as(_i64p)))
    return out, dict(levels=int(st[0]), launches=int(st[1]), rotations=int(st[2]) * q, widest_level=int(st[3]) * q, instances=q)



class CloudKey:
    """Single-key evaluation context = the reference's CloudKey (api.jl:215-231): bootstrap key + keyswitch key,
    held on one MI355X in the engine's transformed layout.

    bk_coeff: int32[n][(k+1)l][k+1][N] coefficient-domain TGSW rows; ksk: int32[N][t][base-1][n+1].
    """

    def __init__(self, params, bk_coeff, ksk, device=0):
        self.params = params
        bk = np.ascontiguousarray(bk_coeff, np.int32)
        ks = np.ascontiguousarray(ksk, np.int32)
        p = params
        if bk.size != p.n * (p.k + 1) * p.l * (p.k + 1) * p.N:
            raise ValueError("bk_coeff has the wrong size for these parameters")
        if ks.size != p.N * p.k * p.ks_t * ((1 << p.ks_basebit) - 1) * (p.n + 1):
            raise ValueError("ksk has the wrong size for these parameters")
        h = _vp()
        _check(lib().thfhe_ctx_create(C.byref(p), _p32(bk), _p32(ks), device, C.byref(h)))
        self.h, self._destroy = h, lib().thfhe_ctx_destroy
        self.words = p.n + 1

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h and getattr(self, "_destroy", None) is not None:
            self._destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter teardown
            pass

    # -- host-buffer calls -------------------------------------------------------------------------
    def gates(self, op, x, y=None, z=None):
        x = _rec(x, self.words)
        y = _rec(y, self.words) if y is not None else None
        z = _rec(z, self.words) if z is not None else None
        _same_count(x, y, z)
        out = np.empty_like(x)
        _check(lib().thfhe_gates(self.h, op, _p32(x), _p32(y), _p32(z), _p32(out), x.shape[0]))
        return out

    def gates_mixed(self, ops, x, y):
        """One launch for a DAG level: gate g applies ops[g] (two-input bootstrapped gates) to (x[g], y[g])."""
        x, y = _rec(x, self.words), _rec(y, self.words)
        ops = np.ascontiguousarray(ops, np.int32)
        _same_count(x, y, ops)
        out = np.empty_like(x)
        _check(lib().thfhe_gates_mixed(self.h, _p32(ops), _p32(x), _p32(y), _p32(out), x.shape[0]))
        return out

    def dag_run(self, input_records, gates):
        """Native levelising scheduler + device-resident executor.  gates: int32[n_gates][4] = (op, in0, in1, in2).
        Returns (wires int32[n_inputs + n_gates][n+1], stats dict)."""
        x = _rec(input_records, self.words)
        g = np.ascontiguousarray(gates, np.int32).reshape(-1, 4)
        wires = np.zeros((x.shape[0] + g.shape[0], self.words), np.int32)
        wires[:x.shape[0]] = x
        st = np.zeros(4, np.int64)
        _check(lib().thfhe_dag_run(self.h, _p32(wires), x.shape[0], _p32(g), g.shape[0], st.ctypes.data_as(_i64p)))
        return wires, dict(levels=int(st[0]), launches=int(st[1]), rotations=int(st[2]), widest_level=int(st[3]))

    def dag_run_batch(self, input_records, gates, out_wires=None):
        """`instances` evaluations of one gate list side by side (thfhe_dag_run_batch; the reference's loop over test records,
        src/KNN_medical_data.cpp:676-691).  input_records: int32[instances][n_inputs][n+1]; out_wires: wire ids to return (None = every gate
        wire).  Returns (int32[instances][len(out_wires) or n_gates][n+1], stats)."""
        return _dag_run_batch(lib().thfhe_dag_run_batch, self.h, self.words, input_records, gates, out_wires)

    def bootstrap(self, x, mu=MU8):
        x = _rec(x, self.words)
        out = np.empty_like(x)
        _check(lib().thfhe_bootstrap(self.h, mu, _p32(x), _p32(out), x.shape[0]))
        return out

    def bootstrap_wo_keyswitch(self, x, mu=MU8):
        x = _rec(x, self.words)
        out = np.empty((x.shape[0], self.params.N + 1), np.int32)
        _check(lib().thfhe_bootstrap_wo_keyswitch(self.h, mu, _p32(x), _p32(out), x.shape[0]))
        return out

    def keyswitch(self, u):
        u = _rec(u, self.params.N + 1)
        out = np.empty((u.shape[0], self.words), np.int32)
        _check(lib().thfhe_keyswitch(self.h, _p32(u), _p32(out), u.shape[0]))
        return out

    # -- device-buffer calls -----------------------------------------------------------------------
    def _alloc(self, n):
        return lib().thfhe_dev_alloc(self.h, n)

    def _free(self, p):
        lib().thfhe_dev_free(self.h, p)

    def _h2d(self, dptr, arr):
        _check(lib().thfhe_copy_h2d(self.h, dptr, arr.ctypes.data_as(_vp), arr.nbytes))

    def _d2h(self, arr, dptr):
        _check(lib().thfhe_copy_d2h(self.h, arr.ctypes.data_as(_vp), dptr, arr.nbytes))

    def device_records(self, count):
        return DeviceBuffer(self, count * self.words * 4)

    def reserve(self, max_count):
        _check(lib().thfhe_reserve(self.h, max_count))

    def gates_dev(self, op, dx, dy, dz, dout, count):
        _check(lib().thfhe_gates_dev(self.h, op, dx.ptr, dy.ptr if dy else None, dz.ptr if dz else None, dout.ptr, count))

    def sync(self):
        _check(lib().thfhe_sync(self.h))

    def set_dag_slice(self, max_gates):
        """Gates per launch of a DAG level (dag_run_batch cuts wider levels into slices)."""
        _check(lib().thfhe_set_dag_slice(self.h, int(max_gates)))

    def set_ring4_threshold(self, max_jobs):
        """Remainders (batch mod 2048) above the cooperative threshold and <= max_jobs rotations use the four-wave ring kernel; 0 disables it."""
        _check(lib().thfhe_set_ring4_threshold(self.h, int(max_jobs)))
        self._ring4_threshold = int(max_jobs)

    def set_coop_threshold(self, max_jobs):
        """Remainders (batch mod 2048) of <= max_jobs rotations use the cooperative latency kernel; 0 disables it."""
        _check(lib().thfhe_set_coop_threshold(self.h, int(max_jobs)))
        self._coop_threshold = int(max_jobs)

    def rotation_kernel_name(self, rotations):
        """The blind-rotation kernel that does most of a batch of `rotations` (thfhe_sk.hip launch_br; for profiles and bench.py)."""
        l = self.params.l
        coop, ring4 = getattr(self, "_coop_threshold", 768), getattr(self, "_ring4_threshold", 1024)
        r = rotations % 2048 if (coop or ring4) else 0
        if rotations >= 2048 or r == 0:
            return f"sk_blind_rotate_ring_kernel<{l}>"
        if r <= coop:
            return f"sk_blind_rotate_coop_kernel<{l}>"
        return f"sk_blind_rotate_ring_kernel<{l}, 4 waves>" if r <= ring4 + min(coop, 256) and ring4 else f"sk_blind_rotate_ring_kernel<{l}>"

    def set_profiling(self, on):
        _check(lib().thfhe_set_profiling(self.h, int(bool(on))))

    def last_timings(self):
        ms = (C.c_float * 4)()
        _check(lib().thfhe_last_timings(self.h, ms))
        return dict(prologue_ms=ms[0], blind_rotate_ms=ms[1], keyswitch_ms=ms[2], total_ms=ms[3])


# ---- the reference's single-key gate API (gates.jl:15-177), batched over the leading axis -------------
def gate_nand(ck, x, y): return ck.gates(NAND, x, y)
def gate_or(ck, x, y): return ck.gates(OR, x, y)
def gate_and(ck, x, y): return ck.gates(AND, x, y)
def gate_xor(ck, x, y): return ck.gates(XOR, x, y)
def gate_xnor(ck, x, y): return ck.gates(XNOR, x, y)
def gate_nor(ck, x, y): return ck.gates(NOR, x, y)
def gate_andny(ck, x, y): return ck.gates(ANDNY, x, y)
def gate_andyn(ck, x, y): return ck.gates(ANDYN, x, y)
def gate_orny(ck, x, y): return ck.gates(ORNY, x, y)
def gate_oryn(ck, x, y): return ck.gates(ORYN, x, y)
def gate_mux(ck, x, y, z): return ck.gates(MUX, x, y, z)
def gate_not(ck, x): return ck.gates(NOT, x)


def gate_constant(ck, value):
    """gates.jl:91-93: noiseless trivial sample of +-1/8 (not encrypted)."""
    r = np.zeros(ck.words, np.int32)
    r[-1] = MU8 if value else -MU8
    return r


def bootstrap(ck, mu, x): return ck.bootstrap(x, mu)                              # bootstrap.jl:98-101
def bootstrap_wo_keyswitch(ck, mu, x): return ck.bootstrap_wo_keyswitch(x, mu)    # bootstrap.jl:75-88
def keyswitch(ck, u): return ck.keyswitch(u)                                      # keyswitch.jl:45-80


# ---- 3-gen multi-key -------------------------------------------------------------------------------------
class MKCloudKey:
    """Evaluation context of the 3rd-generation multi-key scheme: the parties' TransformedBootstrapKeyPart_3gen
    (3gen_mk_internals.jl:45-56) and KeyswitchKey tables on one MI355X.

    bk_coeff: int64[P][n][4][l][N] (part_1..part_4 of every TGswSample_3gen, coefficient domain);
    ksk: int32[P][N][t][base-1][n+1].  Records are int32[P*n+1] = a[p*n+i], b (MKLweSample, mk_internals.jl:23-37).
    """

    def __init__(self, params, bk_coeff, ksk, device=0):
        self.params = p = params
        bk = np.ascontiguousarray(bk_coeff, np.int64)
        ks = np.ascontiguousarray(ksk, np.int32)
        if bk.size != p.parties * p.n * 4 * p.l * p.N:
            raise ValueError("bk_coeff has the wrong size for these parameters")
        if ks.size != p.parties * p.N * p.ks_t * ((1 << p.ks_basebit) - 1) * (p.n + 1):
            raise ValueError("ksk has the wrong size for these parameters")
        h = _vp()
        _check(lib().thfhe_mk_ctx_create(C.byref(p), bk.ctypes.data_as(_i64p), _p32(ks), device, C.byref(h)))
        self.h, self._destroy = h, lib().thfhe_mk_ctx_destroy
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

    def gates(self, op, x, y=None, z=None):
        x = _rec(x, self.words)
        y = _rec(y, self.words) if y is not None else None
        z = _rec(z, self.words) if z is not None else None
        _same_count(x, y, z)
        out = np.empty_like(x)
        _check(lib().thfhe_mk_gates(self.h, op, _p32(x), _p32(y), _p32(z), _p32(out), x.shape[0]))
        return out

    def gates_mixed(self, ops, x, y):
        """One launch for a DAG level of two-input 3-gen gates with per-gate opcodes."""
        x, y = _rec(x, self.words), _rec(y, self.words)
        ops = np.ascontiguousarray(ops, np.int32)
        _same_count(x, y, ops)
        out = np.empty_like(x)
        _check(lib().thfhe_mk_gates_mixed(self.h, _p32(ops), _p32(x), _p32(y), _p32(out), x.shape[0]))
        return out

    def dag_run(self, input_records, gates):
        """Native levelising scheduler + device-resident executor for 3-gen circuits (thfhe_mk_dag_run)."""
        x = _rec(input_records, self.words)
        g = np.ascontiguousarray(gates, np.int32).reshape(-1, 4)
        wires = np.zeros((x.shape[0] + g.shape[0], self.words), np.int32)
        wires[:x.shape[0]] = x
        st = np.zeros(4, np.int64)
        _check(lib().thfhe_mk_dag_run(self.h, _p32(wires), x.shape[0], _p32(g), g.shape[0], st.ctypes.data_as(_i64p)))
        return wires, dict(levels=int(st[0]), launches=int(st[1]), rotations=int(st[2]), widest_level=int(st[3]))

    def dag_run_batch(self, input_records, gates, out_wires=None):
        """`instances` evaluations of one 3-gen gate list side by side (thfhe_mk_dag_run_batch)."""
        return _dag_run_batch(lib().thfhe_mk_dag_run_batch, self.h, self.words, input_records, gates, out_wires)

    def bootstrap(self, x, mu=MU8_64):
        x = _rec(x, self.words)
        out = np.empty_like(x)
        _check(lib().thfhe_mk_bootstrap(self.h, mu, _p32(x), _p32(out), x.shape[0]))
        return out

    def _alloc(self, n):
        return lib().thfhe_mk_dev_alloc(self.h, n)

    def _free(self, p):
        lib().thfhe_mk_dev_free(self.h, p)

    def _h2d(self, dptr, arr):
        _check(lib().thfhe_mk_copy_h2d(self.h, dptr, arr.ctypes.data_as(_vp), arr.nbytes))

    def _d2h(self, arr, dptr):
        _check(lib().thfhe_mk_copy_d2h(self.h, arr.ctypes.data_as(_vp), dptr, arr.nbytes))

    def device_records(self, count):
        return DeviceBuffer(self, count * self.words * 4)

    def reserve(self, max_count):
        _check(lib().thfhe_mk_reserve(self.h, max_count))

    def gates_dev(self, op, dx, dy, dz, dout, count):
        _check(lib().thfhe_mk_gates_dev(self.h, op, dx.ptr, dy.ptr if dy else None, dz.ptr if dz else None, dout.ptr, count))

    def sync(self):
        _check(lib().thfhe_mk_sync(self.h))

    def set_profiling(self, on):
        _check(lib().thfhe_mk_set_profiling(self.h, int(bool(on))))

    def set_dag_slice(self, max_gates):
        _check(lib().thfhe_mk_set_dag_slice(self.h, int(max_gates)))

    def set_pair_threshold(self, max_single_jobs):
        """Batches of <= max_single_jobs rotations run one gate per workgroup; larger ones two gates per workgroup."""
        _check(lib().thfhe_mk_set_pair_threshold(self.h, int(max_single_jobs)))
        self._pair_threshold = int(max_single_jobs)

    def rotation_kernel_name(self, rotations):
        """The blind-rotation kernel a batch of `rotations` is dispatched to (mk_launch_rotation in thfhe_mk.hip)."""
        p = self.params
        if p.N == 4096:
            return "r4k_rotate_kernel"
        if p.N == 2048:
            le = p.l * (((p.Bgbit + 8) // 9) if p.Bgbit > 10 else 1)
            pair = rotations > getattr(self, "_pair_threshold", 256)
            if le > 3:
                return "kms_tlev_rotate_pair_kernel" if pair else "kms_tlev_rotate_kernel"
            return f"mk_blind_rotate_{'pair2k' if pair else 'coop2k'}_kernel<{le}>"
        pair = p.l <= 3 and rotations > getattr(self, "_pair_threshold", 256)
        return f"mk_blind_rotate_{'pair' if pair else 'coop'}_kernel<{p.l}>"

    def last_timings(self):
        ms = (C.c_float * 4)()
        _check(lib().thfhe_mk_last_timings(self.h, ms))
        return dict(prologue_ms=ms[0], blind_rotate_ms=ms[1], keyswitch_ms=ms[2], total_ms=ms[3])


class PolyMac:
    """Device engine for the key-generation products (thfhe_pm_mac): out[j] = addend[j] + sum_terms sign * small[s] (*) torus[t], exact."""

    def __init__(self, N, torus_bits, device=0):
        h = _vp()
        _check(lib().thfhe_pm_ctx_create(device, N, torus_bits, C.byref(h)))
        self.h, self._destroy, self.N, self.dtype = h, lib().thfhe_pm_ctx_destroy, N, (np.int32 if torus_bits == 32 else np.int64)

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h and getattr(self, "_destroy", None) is not None:
            self._destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def mac(self, small, torus, terms, n_out, addend=None):
        small = np.ascontiguousarray(small, np.int32).reshape(-1, self.N)
        torus = np.ascontiguousarray(torus).view(self.dtype).reshape(-1, self.N)
        terms = np.ascontiguousarray(terms, np.int32).reshape(-1, 4)
        out = np.empty((n_out, self.N), self.dtype)
        if addend is not None:
            addend = np.ascontiguousarray(addend).view(self.dtype).reshape(n_out, self.N)
        _check(lib().thfhe_pm_mac(self.h, _p32(small), small.shape[0], torus.ctypes.data_as(_vp), torus.shape[0], _p32(terms), terms.shape[0],
                                  addend.ctypes.data_as(_vp) if addend is not None else None, out.ctypes.data_as(_vp), n_out))
        return out


class CCSCloudKey:
    """MKCloudKey of the CCS scheme (mk_api.jl:392-408): MKBootstrapKey (uni-encrypted key bits, public keys, shared key) and the
    parties' KeyswitchKeys on one MI355X.  bk int32[P][n][3][l][N] (d1, f0, f1), pk int32[P][l][N], crs int32[l][N], ksk as MKCloudKey."""

    def __init__(self, params, bk, pk, crs, ksk, device=0):
        self.params = p = params
        arrs = [np.ascontiguousarray(a, np.int32) for a in (bk, pk, crs, ksk)]
        sizes = (p.parties * p.n * 3 * p.l * p.N, p.parties * p.l * p.N, p.l * p.N, p.parties * p.N * p.ks_t * ((1 << p.ks_basebit) - 1) * (p.n + 1))
        if any(a.size != s for a, s in zip(arrs, sizes)):
            raise ValueError("key table has the wrong size for these parameters")
        h = _vp()
        _check(lib().thfhe_ccs_ctx_create(C.byref(p), *[_p32(a) for a in arrs], device, C.byref(h)))
        self.h, self._destroy = h, lib().thfhe_ccs_ctx_destroy
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

    def gates(self, op, x, y):
        x, y = _rec(x, self.words), _rec(y, self.words)
        _same_count(x, y)
        out = np.empty_like(x)
        _check(lib().thfhe_ccs_gates(self.h, op, _p32(x), _p32(y), _p32(out), x.shape[0]))
        return out

    def bootstrap(self, x, mu=MU8):
        x = _rec(x, self.words)
        out = np.empty_like(x)
        _check(lib().thfhe_ccs_bootstrap(self.h, mu, _p32(x), _p32(out), x.shape[0]))
        return out


def mk_gate_nand(ck, x, y): return ck.gates(NAND, x, y)          # mk_gates.jl:7-13 (CCS scheme)
def mk_bootstrap(ck, mu, x): return ck.bootstrap(x, mu)         # mk_internals.jl:855-858


# the reference's 3-gen gate API (3gen_mk_gates.jl:8-150); `bk` is the MKCloudKey (it holds bk and ks together)
def mk_gate_nand_3gen(bk, x, y): return bk.gates(NAND, x, y)
def mk_gate_or_3gen(bk, x, y): return bk.gates(OR, x, y)
def mk_gate_and_3gen(bk, x, y): return bk.gates(AND, x, y)
def mk_gate_xor_3gen(bk, x, y): return bk.gates(XOR, x, y)
def mk_gate_3and_3gen(bk, x, y, z): return bk.gates(AND3, x, y, z)
def mk_gate_mux_3gen(bk, x, y, z): return bk.gates(MUX, x, y, z)
def mk_gate_not_3gen(bk, x): return bk.gates(NOT, x)
def mk_bootstrap_3gen(bk, mu, x): return bk.bootstrap(x, mu)   # 3gen_mk_internals.jl:112-116
