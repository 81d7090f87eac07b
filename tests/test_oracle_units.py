"""CPU tests of the oracle's primitives against independent numpy restatements and algebraic properties.
Reference semantics cited per test (paths relative to /root/reference, J/ = 3-gen-mk-tfhe/src/)."""
import ctypes as C

import numpy as np
import pytest


def test_modswitch_matches_definition(O):
    # decode_message(x, 2N) = round(x * 2N / 2^32) in [-N, N)     J/numeric-functions.jl:70-73
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.integers(-2**31, 2**31, 2000), [0, -1, 1, 2**31 - 1, -2**31, 2**20, -2**20, 2**20 - 1]])
    for N in (1024, 2048):
        for x in xs:
            got = O.lib().oracle_modswitch(int(x), N)
            v = (int(x) + 2**31 + (1 << (31 - N.bit_length()))) % 2**32 - 2**31  # wrap the addition to int32
            exp = v >> (32 - N.bit_length())
            assert got == exp and -N <= got < N


@pytest.mark.parametrize("shift", [0, 1, 5, 1023, 1024, 1025, 2047, -1, -1000, 4096 + 3])
def test_mul_by_monomial(O, shift):
    # X^s * p mod X^N+1                                             J/rlwe.jl:130-131
    N = 1024
    rng = np.random.default_rng(shift & 0xFF)
    p = rng.integers(-2**31, 2**31, N).astype(np.int32)
    out = np.zeros(N, np.int32)
    O.lib().oracle_mul_by_monomial32(O.p32(p), shift, N, O.p32(out))
    exp = np.zeros(N, np.int64)
    s = shift % (2 * N)
    for j in range(N):
        d = (j + s) % (2 * N)
        exp[d % N] = -int(p[j]) if d >= N else int(p[j])
    assert np.array_equal(out, (exp % 2**32).astype(np.uint32).view(np.int32))
    p64 = rng.integers(-2**63, 2**63, N).astype(np.int64)
    out64 = np.zeros(N, np.int64)
    O.lib().oracle_mul_by_monomial64(O.p64(p64), shift, N, O.p64(out64))
    back = np.zeros(N, np.int64)
    O.lib().oracle_mul_by_monomial64(O.p64(out64), -shift, N, O.p64(back))
    assert np.array_equal(back, p64)


@pytest.mark.parametrize("l,Bgbit", [(3, 7), (2, 10), (3, 6), (2, 7), (4, 4)])
def test_decompose_reconstructs(O, l, Bgbit):
    # digits in [-Bg/2, Bg/2), sum_p d_p 2^(32 - p Bgbit) == x rounded to l*Bgbit bits      J/tgsw.jl:112-138
    N = 1024
    rng = np.random.default_rng(l * 100 + Bgbit)
    p = rng.integers(-2**31, 2**31, N).astype(np.int32)
    p[:4] = [0, -1, 2**31 - 1, -2**31]
    d = np.zeros((l, N), np.int32)
    O.lib().oracle_decompose32(O.p32(p), N, l, Bgbit, O.p32(d))
    half = 1 << (Bgbit - 1)
    assert d.min() >= -half and d.max() < half
    rec = sum(d[q].astype(np.int64) << (32 - (q + 1) * Bgbit) for q in range(l))
    # "floors it to a multiple of 1/B^l" (J/tgsw.jl:104-110): 0 <= x - rec < 2^(32 - l*Bgbit)
    err = (p.astype(np.int64) - rec) % 2**32
    assert err.max() < (1 << (32 - l * Bgbit))
    p64 = rng.integers(-2**63, 2**63, N).astype(np.int64)
    d64 = np.zeros((l, N), np.int64)
    O.lib().oracle_decompose64(O.p64(p64), N, l, Bgbit, O.p64(d64))
    assert d64.min() >= -half and d64.max() < half
    rec = sum(int(d64[q][7]) << (64 - (q + 1) * Bgbit) for q in range(l))
    assert (int(p64[7]) - rec) % 2**64 < (1 << (64 - l * Bgbit))


def test_polymul_ntt_equals_schoolbook(O):
    # src/ntt-test.cpp:29-93 (FFT product vs schoolbook) restated as exact equality
    N = 1024
    rng = np.random.default_rng(5)
    for trial in range(4):
        a = rng.integers(-512, 512, N).astype(np.int32)
        b = rng.integers(-2**31, 2**31, N).astype(np.int32)
        if trial == 0:
            a[:] = -512
            b[:] = -2**31
        o1, o2 = np.zeros(N, np.int32), np.zeros(N, np.int32)
        O.lib().oracle_polymul_schoolbook32(O.p32(a), O.p32(b), N, O.p32(o1))
        O.lib().oracle_polymul_ntt32(O.p32(a), O.p32(b), N, O.p32(o2))
        assert np.array_equal(o1, o2)
        a6 = a.astype(np.int64)
        b6 = rng.integers(-2**63, 2**63, N).astype(np.int64)
        if trial == 0:
            b6[:] = -2**63
        o1, o2 = np.zeros(N, np.int64), np.zeros(N, np.int64)
        O.lib().oracle_polymul_schoolbook64(O.p64(a6), O.p64(b6), N, O.p64(o1))
        O.lib().oracle_polymul_ntt64(O.p64(a6), O.p64(b6), N, O.p64(o2))
        assert np.array_equal(o1, o2)


def test_polymul_schoolbook_small_case_by_hand(O):
    # (1 + 2X)(3 + X^{N-1}) mod X^N+1 = 3 + 6X + X^{N-1} - 2
    N = 8
    a = np.zeros(N, np.int32); a[0], a[1] = 1, 2
    b = np.zeros(N, np.int32); b[0], b[N - 1] = 3, 1
    o = np.zeros(N, np.int32)
    O.lib().oracle_polymul_schoolbook32(O.p32(a), O.p32(b), N, O.p32(o))
    exp = np.zeros(N, np.int32); exp[0], exp[1], exp[N - 1] = 3 - 2, 6, 1
    assert np.array_equal(o, exp)


def test_t64tot32(O):
    # trunc(Int32, d / 2^32) through Float64                         J/numeric-functions.jl:109-111
    f = O.lib().oracle_t64tot32
    assert f(0) == 0 and f(1 << 32) == 1 and f(-(1 << 32)) == -1
    assert f((1 << 32) - 1) == 0 and f(-(1 << 32) + 1) == 0          # toward zero
    assert f((5 << 32) + 123) == 5 and f(-(5 << 32) - 123) == -5
    assert f(-(1 << 63)) == -(1 << 31)
    rng = np.random.default_rng(3)
    for d in rng.integers(-2**63, 2**63 - 1024, 2000):
        exp = int(np.trunc(np.float64(int(d)) / 4294967296.0))
        assert f(int(d)) == exp


def test_keyswitch_against_numpy(O, sk_small):
    # res = (0, b) - sum_{i,j: d != 0} KS[d, j, i]                   J/keyswitch.jl:45-80
    p, K, orc = sk_small
    rng = np.random.default_rng(9)
    u = rng.integers(-2**31, 2**31, p.N + 1).astype(np.int32)
    u[:3] = [0, -1, 2**31 - 1]
    got = orc.keyswitch(u)
    t, bb = p.ks_t, p.ks_basebit
    res = np.zeros(p.n + 1, np.int64)
    res[p.n] = u[p.N]
    off = 1 << (32 - (1 + bb * t))
    for i in range(p.N):
        ab = (int(u[i]) + off) % 2**32
        for j in range(1, t + 1):
            d = (ab >> (32 - j * bb)) & ((1 << bb) - 1)
            if d:
                res -= K.ksk[i, j - 1, d - 1].astype(np.int64)
    assert np.array_equal(got, (res % 2**32).astype(np.uint32).view(np.int32))
    # semantic check: key switching preserves the phase up to small noise
    ph_in = (int(u[p.N]) - int((u[:p.N].astype(np.int64) * K.rlwe_key[0].astype(np.int64)).sum())) % 2**32
    ph_out = int(K.phases(got)[0]) % 2**32
    diff = ((ph_out - ph_in + 2**31) % 2**32 - 2**31) / 2**32
    assert abs(diff) < 0.02


def test_cmux_schoolbook_equals_ntt_and_selects(O, sk_small):
    # mux_rotate: acc += BK_i (.) (X^a acc - acc)  == X^{a s_i} acc + noise      J/bootstrap.jl:19-29
    p, K, orc = sk_small
    rng = np.random.default_rng(11)
    acc = rng.integers(-2**31, 2**31, (2, p.N)).astype(np.int32)
    # make acc a valid RLWE sample of a message so that the selection property is observable
    mu = np.zeros(p.N, np.int32); mu[:] = 1 << 29
    body = np.zeros(p.N, np.int32)
    O.lib().oracle_polymul_ntt32(O.p32(K.rlwe_key[0]), O.p32(np.ascontiguousarray(acc[0])), p.N, O.p32(body))
    acc[1] = ((body.astype(np.int64) + mu.astype(np.int64)) % 2**32).astype(np.uint32).view(np.int32)
    for i, a in [(0, 17), (1, -300), (2, 1023), (3, -1024)]:
        r1 = orc.mux_rotate(i, a, acc, schoolbook=True)
        r2 = orc.mux_rotate(i, a, acc, schoolbook=False)
        assert np.array_equal(r1, r2)
        # phase of result = X^{a*s_i} * mu (+ noise)
        prod = np.zeros(p.N, np.int32)
        O.lib().oracle_polymul_ntt32(O.p32(K.rlwe_key[0]), O.p32(np.ascontiguousarray(r1[0])), p.N, O.p32(prod))
        phase = ((r1[1].astype(np.int64) - prod.astype(np.int64) + 2**31) % 2**32 - 2**31)
        exp = np.zeros(p.N, np.int32)
        O.lib().oracle_mul_by_monomial32(O.p32(mu), a * int(K.lwe_key[i]), p.N, O.p32(exp))
        err = ((phase - exp.astype(np.int64) + 2**31) % 2**32 - 2**31) / 2**32
        assert np.abs(err).max() < 1e-3


def test_gate_prologue_constants(O):
    # J/gates.jl:15-161: NAND (0,1/8)-x-y ; XOR (0,1/4)+2(x+y) ; ...
    p = O.make_params("SK-128")
    rng = np.random.default_rng(2)
    x = rng.integers(-2**31, 2**31, p.n + 1).astype(np.int32)
    y = rng.integers(-2**31, 2**31, p.n + 1).astype(np.int32)
    z = rng.integers(-2**31, 2**31, p.n + 1).astype(np.int32)
    tmp = np.zeros(p.n + 1, np.int32)
    E8, E4 = 1 << 29, 1 << 30
    table = {O.NAND: (E8, -1, -1), O.OR: (E8, 1, 1), O.AND: (-E8, 1, 1), O.XOR: (E4, 2, 2), O.XNOR: (-E4, -2, -2),
             O.NOR: (-E8, -1, -1), O.ANDNY: (-E8, -1, 1), O.ANDYN: (-E8, 1, -1), O.ORNY: (E8, -1, 1), O.ORYN: (E8, 1, -1)}
    for op, (cb, cx, cy) in table.items():
        assert O.lib().oracle_gate_prologue(C.byref(p), op, 0, O.p32(x), O.p32(y), O.p32(z), O.p32(tmp)) == 0
        exp = cx * x.astype(np.int64) + cy * y.astype(np.int64)
        exp[p.n] += cb
        assert np.array_equal(tmp, (exp % 2**32).astype(np.uint32).view(np.int32))
    # MUX second rotation uses (-1/8) - x + z                         J/gates.jl:169-170
    O.lib().oracle_gate_prologue(C.byref(p), O.MUX, 1, O.p32(x), O.p32(y), O.p32(z), O.p32(tmp))
    exp = -x.astype(np.int64) + z.astype(np.int64); exp[p.n] -= E8
    assert np.array_equal(tmp, (exp % 2**32).astype(np.uint32).view(np.int32))
    assert O.lib().oracle_gate_prologue(C.byref(p), 99, 0, O.p32(x), O.p32(y), O.p32(z), O.p32(tmp)) == -1


def test_gate_truth_tables_sk128(O, sk128):
    # 3-gen-mk-tfhe/test/runtests.jl:10-42: every gate over every input combination decrypts to the truth table
    p, K, orc = sk128
    s = O.SIGMAS["SK-128"]
    a = np.array([0, 0, 1, 1]); b = np.array([0, 1, 0, 1])
    ca, cb = K.encrypt_bits(a, s["lwe"], 101), K.encrypt_bits(b, s["lwe"], 102)
    for op, fn in O.TRUTH.items():
        out = orc.gates(op, ca, cb)
        assert np.array_equal(K.decrypt_bits(out), [bool(fn(bool(x), bool(y))) for x, y in zip(a, b)])
        ph = K.phases(out) / 2.0**32
        assert np.abs(np.abs(ph) - 0.125).max() < 0.03   # reference envelope: <= 0.0085 on 96 fixture bits
    # NOT / COPY are not bootstrapped (J/gates.jl:76-79)
    assert np.array_equal(orc.gates(O.NOT, ca), (-ca.astype(np.int64) % 2**32).astype(np.uint32).view(np.int32))
    assert np.array_equal(orc.gates(O.COPY, ca), ca)


def test_mux_truth_table_sk128(O, sk128):
    # gate_mux = 2 rotations + 1 key switch                           J/gates.jl:163-177
    p, K, orc = sk128
    s = O.SIGMAS["SK-128"]
    bits = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)])
    cx, cy, cz = (K.encrypt_bits(bits[:, q], s["lwe"], 200 + q) for q in range(3))
    out = orc.gates(O.MUX, cx, cy, cz)
    assert np.array_equal(K.decrypt_bits(out), np.where(bits[:, 0] == 1, bits[:, 1], bits[:, 2]).astype(bool))


def test_full_gate_schoolbook_equals_ntt_small(O, sk_small):
    p, K, orc = sk_small
    ca = K.encrypt_bits([1, 0], 2.0**-15, 1); cb = K.encrypt_bits([1, 1], 2.0**-15, 2)
    assert np.array_equal(orc.gates(O.NAND, ca, cb, schoolbook=True), orc.gates(O.NAND, ca, cb, schoolbook=False))


def test_blind_rotate_skips_zero_bara(O, sk_small):
    # J/bootstrap.jl:40: `if bara[i] != 0` -- a mask word that mod-switches to 0 must leave acc untouched
    p, K, orc = sk_small
    x = K.encrypt_bits([1], 2.0**-15, 5)[0].copy()
    x[:p.n] = 0            # every bara_i == 0: output = extraction of X^{-barb} * testvector
    u = orc.bootstrap_wo_keyswitch(x)
    barb = O.lib().oracle_modswitch(int(x[p.n]), p.N)
    tv = np.full(p.N, 1 << 29, np.int32); rot = np.zeros(p.N, np.int32)
    O.lib().oracle_mul_by_monomial32(O.p32(tv), -barb, p.N, O.p32(rot))
    assert np.all(u[:p.N] == 0) and u[p.N] == rot[0]


def test_reference_fft_path_equals_the_exact_product_on_its_low_bits(O):
    # src/ntt-test.cpp:57-93 in the reference: the FFT product (torusPolynomialAddMulRFFT) must equal the schoolbook negacyclic product on the low
    # 31 bits.  Here: engine 2 of the oracle = the reference's Complex{Float64} transform restated (J/polynomials.jl:208-247, transformed_mul :245-247)
    # against the exact schoolbook product.  With gadget-sized small operands (|d| <= 64) and random Torus32 words the sums stay near 2^43 and the
    # double transform is exact; with every word at +-2^31 and |d| = 512 (2^50) its low bits are rounding noise, the top 22 bits still agree.
    L = O.lib()
    rng = np.random.default_rng(44)
    for dmax, exact in ((64, True), (512, False)):
        a = rng.integers(-dmax, dmax, 1024).astype(np.int32)
        b = (rng.integers(-2**31, 2**31, 1024) if exact else np.where(rng.integers(0, 2, 1024) == 1, 2**31 - 1, -2**31)).astype(np.int32)
        if not exact:
            a = (rng.integers(0, 2, 1024) * (2 * dmax - 1) - dmax).astype(np.int32)
        ref, got = np.zeros(1024, np.int32), np.zeros(1024, np.int32)
        L.oracle_polymul_schoolbook32(O.p32(a), O.p32(b), 1024, O.p32(ref))
        L.oracle_fft_polymul32(O.p32(a), O.p32(b), 1024, O.p32(got))
        diff = (got.astype(np.int64) - ref.astype(np.int64) + 2**31) % 2**32 - 2**31
        if exact:
            assert np.array_equal(got, ref)
        else:
            assert np.abs(diff).max() < 2**10, np.abs(diff).max()


def test_reference_fft_engine_gate_decrypts_like_the_exact_engines(O, sk_small):
    # gate level: bootstrap with tgsw_extern_mul as the reference runs it (engine 2) against the exact engine -- same decryptions, phases equal up to
    # FFT rounding noise far below the ciphertext noise (observed: word for word equal at these magnitudes)
    p, K, orc = sk_small
    rng = np.random.default_rng(45)
    a, b = rng.integers(0, 2, 6), rng.integers(0, 2, 6)
    ca, cb = K.encrypt_bits(a, 2.0**-15, 71), K.encrypt_bits(b, 2.0**-15, 72)
    for op in (O.NAND, O.XOR):
        ex = orc.gates(op, ca, cb)
        ff = orc.gates(op, ca, cb, schoolbook=2)
        assert np.array_equal(K.decrypt_bits(ff), K.decrypt_bits(ex))
        d = (K.phases(ff).astype(np.int64) - K.phases(ex).astype(np.int64) + 2**31) % 2**32 - 2**31
        assert np.abs(d).max() < 2**12      # 2^-20 of the torus; the ciphertext noise is ~2^-7.6 after bootstrapping
    ff = orc.gates(O.MUX, ca, cb, cb[::-1].copy(), schoolbook=2)
    assert np.array_equal(K.decrypt_bits(ff), K.decrypt_bits(orc.gates(O.MUX, ca, cb, cb[::-1].copy())))
