"""CPU check of the GPU kernel's lane-level code (torus-fhe_amd/csrc/thfhe_lane.h) replayed on the host by
tests/emu/lane_emu.cpp: transform correctness, exactness margin of the split-limb FP64 product, CMux and blind
rotation bit-for-bit against the oracle's schoolbook path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def E():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu")], check=True)
    L = C.CDLL(os.path.join(HERE, "emu", "liblane_emu.so"))
    L.emu_polymul.restype = C.c_double
    return L


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_forward_transform_matches_definition(E):
    # P_k = sum_j z_j zeta^(j(4k+1)), zeta = exp(i pi / N); lane (k1 + 8 k0), register k2 holds k = k0 + 8 k1 + 64 k2
    rng = np.random.default_rng(0)
    z = rng.standard_normal(512) + 1j * rng.standard_normal(512)
    zin = np.ascontiguousarray(np.stack([z.real, z.imag], -1)).ravel()
    out = np.zeros(64 * 8 * 2)
    E.emu_fwd_raw(dptr(zin), dptr(out))
    got = out.reshape(64, 8, 2)
    got = got[..., 0] + 1j * got[..., 1]
    j = np.arange(512)
    k = np.arange(512)
    ang = (np.outer(4 * k + 1, j) % 2048) * (np.pi / 1024)
    P = (np.exp(1j * ang) * z[None, :]).sum(axis=1)
    exp = np.zeros((64, 8), complex)
    for k0 in range(8):
        for k1 in range(8):
            for k2 in range(8):
                exp[k1 + 8 * k0, k2] = P[k0 + 8 * k1 + 64 * k2]
    assert np.abs(got - exp).max() < 1e-10
    back = np.zeros(1024)
    E.emu_inv_raw(dptr(out), dptr(back))
    zb = back.reshape(512, 2)
    assert np.abs((zb[:, 0] + 1j * zb[:, 1]) / 512 - z).max() < 1e-13


@pytest.mark.parametrize("case", ["random7", "random10", "worst_neg", "worst_alt"])
def test_polymul_exact_with_margin(E, O, case):
    # exactness claim of DESIGN.md section 3: every inverse-transform output is within << 1/2 of an integer
    N = 1024
    rng = np.random.default_rng(1)
    if case == "random7":
        a = rng.integers(-64, 64, N); b = rng.integers(-2**31, 2**31, N)
    elif case == "random10":
        a = rng.integers(-512, 512, N); b = rng.integers(-2**31, 2**31, N)
    elif case == "worst_neg":
        a = np.full(N, -512); b = np.full(N, -2**31)
    else:
        a = 511 * (-1) ** np.arange(N); b = np.full(N, 2**31 - 1)
    a = a.astype(np.int32); b = b.astype(np.int32)
    ref, got = np.zeros(N, np.int32), np.zeros(N, np.int32)
    O.lib().oracle_polymul_schoolbook32(O.p32(a), O.p32(b), N, O.p32(ref))
    margin = E.emu_polymul(O.p32(a), O.p32(b), O.p32(got))
    assert np.array_equal(ref, got)
    assert margin < 1e-3


def test_cmux_and_blind_rotate_bit_exact(E, O, sk_small):
    p, K, orc = sk_small
    npolys = K.bk.size // 1024
    spec = np.zeros(npolys * 2 * 512 * 2, np.float64)
    E.emu_transform_key_polys(O.p32(K.bk), C.c_int64(npolys), dptr(spec))
    rng = np.random.default_rng(3)
    acc = rng.integers(-2**31, 2**31, (2, 1024)).astype(np.int32)
    for i, a in [(0, 5), (3, -1000), (7, 1023), (2, -1024), (15, 1)]:
        ref = orc.mux_rotate(i, a, acc, schoolbook=True)
        got = acc.copy()
        E.emu_mux_rotate(dptr(spec), p.l, p.Bgbit, i, a, O.p32(got))
        assert np.array_equal(ref, got)
        acc = ref
    x = K.encrypt_bits([1], 2.0**-15, 3)[0]
    ref = orc.bootstrap_wo_keyswitch(x)
    bara = np.array([O.lib().oracle_modswitch(int(v), 1024) for v in x[:p.n]], np.int32)
    barb = O.lib().oracle_modswitch(int(x[p.n]), 1024)
    out = np.zeros(1025, np.int32)
    E.emu_blind_rotate(dptr(spec), p.n, p.l, p.Bgbit, O.p32(bara), barb, 1 << 29, O.p32(out))
    assert np.array_equal(ref, out)


def test_swizzled_variant_matches_padded(E, O):
    # the LDS-ring kernel's unpadded / XOR-swizzled buffer and power-twiddles give the same spectra and exact products
    E.emu_variant_crosscheck.restype = C.c_double
    rng = np.random.default_rng(8)
    a = rng.integers(-512, 512, 1024).astype(np.int32); b = rng.integers(-2**31, 2**31, 1024).astype(np.int32)
    ref, got = np.zeros(1024, np.int32), np.zeros(1024, np.int32)
    O.lib().oracle_polymul_schoolbook32(O.p32(a), O.p32(b), 1024, O.p32(ref))
    d = E.emu_variant_crosscheck(O.p32(a), O.p32(b), O.p32(got))
    assert np.array_equal(ref, got) and d < 1e-9


@pytest.mark.parametrize("fn", ["emu_roots_variant_crosscheck", "emu_regtranspose_variant_crosscheck", "emu_regtranspose_swizzled_variant_crosscheck"])
@pytest.mark.parametrize("case", ["random", "adversarial"])
def test_roots_variant_exact_with_margin(E, O, case, fn):
    # second-generation ring kernel: pass-1 twiddles rebuilt from two per-lane roots (b * s^k0) instead of the T1 table.  The
    # products must stay exact with a wide margin, on random digits and on the worst case of the SK-80 shape (|digit| = 512,
    # every key word at +-2^31) where the limb sums are largest.
    # (second variant: the same with the first transpose modelled as the in-register lane exchange of the third-generation kernel;
    # third: that exchange followed by the XOR-swizzled second transpose, the combination the multi-key kernels use)
    getattr(E, fn).restype = C.c_double
    rng = np.random.default_rng(18)
    if case == "random":
        a = rng.integers(-512, 512, 1024).astype(np.int32); b = rng.integers(-2**31, 2**31, 1024).astype(np.int32)
    else:
        a = (rng.integers(0, 2, 1024) * 1023 - 512).astype(np.int32); b = np.where(rng.integers(0, 2, 1024) == 1, 2**31 - 1, -2**31).astype(np.int32)
    ref, got = np.zeros(1024, np.int32), np.zeros(1024, np.int32)
    O.lib().oracle_polymul_schoolbook32(O.p32(a), O.p32(b), 1024, O.p32(ref))
    dmax = C.c_double(0)
    margin = getattr(E, fn)(O.p32(a), O.p32(b), O.p32(got), C.byref(dmax))
    assert np.array_equal(ref, got)
    assert margin < 1e-4 and dmax.value < 1e-8, (margin, dmax.value)


def test_fused_rotated_digits_match_reference_form(E):
    # rotated_digits_z (one signed bit-field extract per digit, sign by xor/subtract) == load_rotated16 + digits_to_z on random and
    # extreme accumulators, every rotation class (0, < N, = N, > N, 2N - 1) and the three gadget shapes in use
    rng = np.random.default_rng(33)
    for acc in (rng.integers(-2**31, 2**31, 1024).astype(np.int32), np.full(1024, -2**31, np.int32), np.full(1024, 2**31 - 1, np.int32)):
        for a2n in (0, 1, 63, 64, 1023, 1024, 1025, 2047, int(rng.integers(0, 2048))):
            for l, bg in ((3, 7), (2, 10), (4, 8), (1, 4)):
                assert E.emu_rotated_digits_crosscheck(acc.ctypes.data_as(C.POINTER(C.c_int32)), a2n, l, bg) == 0, (a2n, l, bg)


def test_mk_cmux_and_extract_bit_exact(E, O):
    # Torus64 3-gen CMux through the lane code (four 16-bit limbs, hi-word digits) vs the MK oracle's schoolbook path
    for name, n, parties in (("MK2", 6, 2), ("MK4", 3, 2)):
        p = O.make_params(name, n=n, parties=parties)
        K = O.MKKeys(p, 3, 2.0**-30.70, 2.0**-13.52)
        orc = O.MKOracle(p, K.bk, K.ksk)
        PN = p.parties * p.n
        spec = np.zeros(PN * 2 * p.l * 8 * 512 * 2, np.float64)
        E.emu_mk_transform_key(O.p64(K.bk), C.c_long(PN), p.l, dptr(spec))
        acc = np.random.default_rng(5).integers(-2**63, 2**63, (2, 1024)).astype(np.int64)
        for party, i, a in [(0, 0, 5), (1, n - 1, -1000), (0, 1, 1023), (1, 2, -1024)]:
            ref = orc.mux_rotate(party, i, a, acc, schoolbook=True)
            got = acc.copy()
            E.emu_mk_mux_rotate(dptr(spec), p.l, p.Bgbit, C.c_long(party * p.n + i), a, O.p64(got))
            assert np.array_equal(ref, got)
            acc = ref
        acc[0, 5] = -2**63
        out = np.zeros(1025, np.int32)
        E.emu_mk_extract(O.p64(acc), O.p32(out))
        f = O.lib().oracle_t64tot32
        exp = [f(int(acc[0, 0]))] + [f(int(((-int(acc[0, 1024 - q])) + 2**63) % 2**64 - 2**63)) for q in range(1, 1024)] + [f(int(acc[1, 0]))]
        assert np.array_equal(out, np.array(exp, np.int32))


def test_cmux_l2_bgbit10(E, O):
    # the SK-80 shape (l = 2, Bgbit = 10) exercises the largest digits the Torus32 engine accepts
    p = O.make_params("SK-80", n=4)
    K = O.SKKeys(p, 5, 9.0e-9, 2.44e-5)
    orc = O.Oracle(p, K.bk, K.ksk)
    npolys = K.bk.size // 1024
    spec = np.zeros(npolys * 2 * 512 * 2, np.float64)
    E.emu_transform_key_polys(O.p32(K.bk), C.c_int64(npolys), dptr(spec))
    acc = np.random.default_rng(4).integers(-2**31, 2**31, (2, 1024)).astype(np.int32)
    ref = orc.mux_rotate(1, 777, acc, schoolbook=True)
    got = acc.copy()
    E.emu_mux_rotate(dptr(spec), p.l, p.Bgbit, 1, 777, O.p32(got))
    assert np.array_equal(ref, got)


def test_2048_transform_matches_definition(E):
    # P_k = sum_{j<1024} z_j zeta^(j(4k+1)), zeta = exp(i pi/2048); half = k & 1, k'' = k >> 1 in the 512-point register order
    rng = np.random.default_rng(2)
    z = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    zin = np.ascontiguousarray(np.stack([z.real, z.imag], -1)).ravel()
    out, back = np.zeros(2 * 64 * 8 * 2), np.zeros(2048)
    E.emu_fwd_raw_2k(dptr(zin), dptr(out), dptr(back))
    got = out.reshape(2, 64, 8, 2)
    got = got[..., 0] + 1j * got[..., 1]
    j = np.arange(1024)
    k = np.arange(1024)
    ang = (np.outer(4 * k + 1, j) % 4096) * (np.pi / 2048)
    P = (np.exp(1j * ang) * z[None, :]).sum(axis=1)
    exp = np.zeros((2, 64, 8), complex)
    for half in range(2):
        for k0 in range(8):
            for k1 in range(8):
                for k2 in range(8):
                    exp[half, k1 + 8 * k0, k2] = P[2 * (k0 + 8 * k1 + 64 * k2) + half]
    assert np.abs(got - exp).max() < 1e-9
    zb = back.reshape(1024, 2)
    assert np.abs((zb[:, 0] + 1j * zb[:, 1]) / 1024 - z).max() < 1e-13


def test_mk_cmux_2048_bit_exact(E, O):
    # BASELINE config 5 shape: 3-gen MK, N = 2048, l = 3, Bgbit = 6 -- lane code vs the oracle's schoolbook path
    E.emu_mk_mux_rotate_2k.restype = C.c_double
    p = O.make_params("MK4", n=2, parties=2, N=2048)
    K = O.MKKeys(p, 3, 2.0**-30.70, 2.0**-13.52)
    orc = O.MKOracle(p, K.bk, K.ksk)
    PN = p.parties * p.n
    spec = np.zeros(PN * 2 * p.l * 8 * 1024 * 2, np.float64)
    E.emu_mk_transform_key_2k(O.p64(K.bk), C.c_long(PN), p.l, dptr(spec))
    acc = np.random.default_rng(6).integers(-2**63, 2**63, (2, 2048)).astype(np.int64)
    for party, i, a in [(0, 0, 5), (1, 1, -2047), (0, 1, 2047), (1, 0, -2048), (1, 1, 1024)]:
        ref = orc.mux_rotate(party, i, a, acc, schoolbook=True)
        got = acc.copy()
        margin = E.emu_mk_mux_rotate_2k(dptr(spec), p.l, p.Bgbit, C.c_long(party * p.n + i), a, O.p64(got))
        assert np.array_equal(ref, got)
        assert margin < 1e-3
        acc = ref
    out = np.zeros(2049, np.int32)
    E.emu_mk_extract_2k(O.p64(acc), O.p32(out))
    x = np.zeros(p.n * p.parties + 1, np.int32)      # all-zero mask words: bootstrap_wo_keyswitch only initialises + extracts
    ref = orc.bootstrap_wo_keyswitch(x)
    f = O.lib().oracle_t64tot32
    exp = [f(int(acc[0, 0]))] + [f(int(((-int(acc[0, 2048 - q])) + 2**63) % 2**64 - 2**63)) for q in range(1, 2048)] + [f(int(acc[1, 0]))]
    assert np.array_equal(out, np.array(exp, np.int32)) and ref.shape == (2049,)


def test_table_free_twisted_transforms_equal_the_table_form(E):
    # "tq" form (two-gate N = 2048 kernels, no T1 tables): same spectra in the same order as the table form, and its inverse pair returns 1024 z
    rng = np.random.default_rng(4)
    z = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    zin = np.ascontiguousarray(np.stack([z.real, z.imag], -1)).ravel()
    E.emu_tq_vs_table_2k.restype = C.c_double
    assert E.emu_tq_vs_table_2k(dptr(zin)) < 1e-10


def test_ring_4096_product_exact_with_margin(E, O):
    # the arithmetic of r4k_rotate_kernel (radix-4 split, four twisted 512-point quarter transforms, four 16-bit limbs, 1/2048 in the key
    # spectra): digits of the 9-bit parts (|d| <= 256) times a Torus64 polynomial == the exact negacyclic convolution mod 2^64, and every
    # inverse output within 1e-3 of an integer on an adversarial input (all digits +-256, key words +-2^63 pattern) as well as a random one
    E.emu_polymul_4k.restype = C.c_double
    rng = np.random.default_rng(6)
    N = 4096
    cases = [(rng.integers(-256, 257, N).astype(np.int32), rng.integers(-2**63, 2**63, N, dtype=np.int64)),
             (np.where(rng.integers(0, 2, N) == 1, 256, -256).astype(np.int32), np.where(rng.integers(0, 2, N) == 1, 2**63 - 1, -2**63).astype(np.int64))]
    for d, k in cases:
        out = np.zeros(N, np.int64)
        margin = E.emu_polymul_4k(O.p32(d), O.p64(k), O.p64(out))
        # exact reference: negacyclic convolution with Python integers reduced mod 2^64 (numpy object arithmetic is too slow at N = 4096:
        # use the oracle's exact 64-bit product)
        ref = np.zeros(N, np.int64)
        O.lib().oracle_polymul_schoolbook64(O.p64(d.astype(np.int64)), O.p64(k), N, O.p64(ref))
        assert np.array_equal(out, ref)
        assert margin < 1e-3, margin
