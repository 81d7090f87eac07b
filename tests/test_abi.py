"""The C-ABI library loads on a machine without a GPU, exports every symbol include/*.h declares, and refuses to
compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(only=None):
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h") or (only and fn != only):
            continue
        src = open(os.path.join(inc, fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}()]*\)\s*;", src):
            if m.group(1) not in ("defined", "sizeof"):
                names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    import thfhe
    L = thfhe.lib()
    decl = declared_symbols()
    assert "thfhe_gates" in decl and "thfhe_ctx_create" in decl and len(decl) >= 30
    missing = [n for n in sorted(decl) if not hasattr(L, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    # the Python host layer binds exactly the thfhe_* symbols of thfhe_hip.h
    assert {"bootsNAND", "bootsAND", "bootsOR", "bootsXOR", "bootsMUX", "bootsNOT"} <= decl   # the libtfhe names the reference calls
    assert set(thfhe.SIGNATURES) == declared_symbols("thfhe_hip.h")


def test_julia_layer_binds_existing_symbols_with_matching_arity():
    # Julia is not installed in the build image: what can be checked without it is that every ccall of torus-fhe_amd/julia/TFHE_HIP.jl names a
    # symbol the library exports and passes as many arguments as the C prototype (and the Python binding) declares
    import thfhe
    L = thfhe.lib()
    src = open(os.path.join(ROOT, "torus-fhe_amd", "julia", "TFHE_HIP.jl")).read()
    calls = re.findall(r"ccall\(\(:([A-Za-z0-9_]+), LIB\), *[A-Za-z]+, *\(([^()]*(?:\{[^()]*\}[^()]*)*)\)", src)
    assert len(calls) >= 20
    for name, argtypes in calls:
        assert hasattr(L, name), name
        n_jl = len([a for a in argtypes.split(",") if a.strip()])
        if name in thfhe.SIGNATURES:
            assert n_jl == len(thfhe.SIGNATURES[name][1]), (name, n_jl, len(thfhe.SIGNATURES[name][1]))
    assert {"thfhe_kms_gates", "thfhe_kms_bootstrap", "thfhe_kms_set_relin_keys", "thfhe_pm_mac"} <= {c[0] for c in calls}


def test_params_struct_layout():
    import thfhe
    assert C.sizeof(thfhe.Params) == 36
    p = thfhe.make_params("SK-128")
    assert (p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit, p.torus_bits, p.parties) == (630, 1024, 1, 3, 7, 8, 2, 32, 1)


def test_no_cpu_fallback_without_device():
    import thfhe
    if thfhe.lib().thfhe_device_count() > 0:
        pytest.skip("a HIP device is present")
    p = thfhe.make_params("SK-128")
    with pytest.raises(thfhe.ThfheError, match="no usable HIP device"):
        thfhe.CloudKey(p, np.zeros(630 * 6 * 2 * 1024, np.int32), np.zeros(1024 * 8 * 3 * 631, np.int32))


def test_argument_validation_messages():
    import thfhe
    L = thfhe.lib()
    h = C.c_void_p()
    bad = thfhe.make_params("SK-128", N=512)
    z = np.zeros(8, np.int32)
    rc = L.thfhe_ctx_create(C.byref(bad), z.ctypes.data_as(C.POINTER(C.c_int32)), z.ctypes.data_as(C.POINTER(C.c_int32)), 0, C.byref(h))
    assert rc == -2 and b"N = 1024" in L.thfhe_last_error()
    rc = L.thfhe_ctx_create(None, None, None, 0, C.byref(h))
    assert rc == -1
    assert L.thfhe_gates(None, 0, None, None, None, None, 1) == -1


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under torus-fhe_amd/ may reference it
    pkg = os.path.join(ROOT, "torus-fhe_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".jl")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle_lib" not in txt and "thfhe_oracle" not in txt and "lane_emu" not in txt.replace("tests/emu/lane_emu.cpp", ""), os.path.join(dp, fn)


def test_new_entry_points_validate_arguments_without_a_device():
    # party-sharded blocks, DAG executor, threshold decryption, N = 2048: argument checks run before any device work
    import thfhe
    L = thfhe.lib()
    h = C.c_void_p()
    assert L.thfhe_dag_run(None, None, 0, None, 0, None) == -1
    assert L.thfhe_mk_set_pair_threshold(None, 0) == -1 and L.thfhe_mk_set_stream(None, None) == -1
    assert L.thfhe_mk_rotate_partial_dev(None, None, None, 0, None, None, 1) == -1
    assert L.thfhe_mk_prologue_dev(None, 0, 0, None, None, None, 0, 0, None, None, 1) == -1
    assert L.thfhe_partial_decrypt(None, None, None, None, None, 1) == -1 and L.thfhe_final_decrypt(None, None, None, 0, None, None, 1) == -1
    assert L.thfhe_poly_ctx_create(0, 512, C.byref(h)) == -2 and b"N = 1024" in L.thfhe_last_error()
    z64, z32 = np.zeros(8, np.int64), np.zeros(8, np.int32)
    p64, p32 = z64.ctypes.data_as(C.POINTER(C.c_int64)), z32.ctypes.data_as(C.POINTER(C.c_int32))
    # N = 2048 is a multi-key ring degree only, and only up to l = 3
    rc = L.thfhe_mk_ctx_create(C.byref(thfhe.make_params("MK4-N2048", l=4, Bgbit=4)), p64, p32, 0, C.byref(h))
    assert rc == -2 and b"l <= 3" in L.thfhe_last_error()
    rc = L.thfhe_mk_ctx_create(C.byref(thfhe.make_params("MK4", N=8192)), p64, p32, 0, C.byref(h))
    assert rc == -2
    # N = 4096 holds six row parts: l x ceil(Bgbit / 9) <= 3
    rc = L.thfhe_mk_ctx_create(C.byref(thfhe.make_params("MK64-fft", l=2)), p64, p32, 0, C.byref(h))
    assert rc == -2 and b"N = 4096 needs" in L.thfhe_last_error()
    if L.thfhe_device_count() == 0:
        rc = L.thfhe_mk_ctx_create(C.byref(thfhe.make_params("MK4-N2048")), p64, p32, 0, C.byref(h))
        assert rc == -3 and b"no usable HIP device" in L.thfhe_last_error()
        assert L.thfhe_poly_ctx_create(0, 1024, C.byref(h)) == -3
        with pytest.raises(thfhe.ThfheError):
            from thfhe import threshold
            threshold.PolyContext(0)


def test_multi_rank_processes_load_torch_before_the_engine():
    """Runtime-order rule: torch bundles its own HIP runtime, and loading libthfhe_hip.so first hides the GPU
    from torch.  thfhe.lib() therefore imports torch itself in a multi-rank job (WORLD_SIZE > 1) -- and does not in a single process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import thfhe; assert 'torch' not in sys.modules; thfhe.lib(); "
            "print(int('torch' in sys.modules), int(bool(thfhe.torch_loaded_first)))") % os.path.join(ROOT, "torus-fhe_amd")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "THFHE_TORCH_FIRST")}
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, WORLD_SIZE="2"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["1", "1"]
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["0", "0"]


def test_operand_batches_of_different_length_are_rejected():
    # the C side copies x.shape[0] records from every operand: the host layer must refuse a shorter one
    import thfhe
    x, y = np.zeros((3, 631), np.int32), np.zeros((2, 631), np.int32)
    with pytest.raises(ValueError):
        thfhe._same_count(x, y)
    with pytest.raises(ValueError):
        thfhe._same_count(x, x, np.zeros(2, np.int32))
    thfhe._same_count(x, x, None)
