"""CPU tests of the 3-gen multi-key oracle (J/tgsw_3gen.jl, J/3gen_mk_internals.jl, J/3gen_mk_gates.jl,
J/mk_internals.jl:730-744).  The reference holds no MK ciphertext fixtures and Julia is not installed here, so MK
ciphertext parity with the reference is UNPINNED; what the reference does pin -- decrypt-equality on random trials
(test/runtests.jl:62-102, multikey_3gen.jl) and the post-bootstrap noise sample it committed
(noise_results/mk-noises__parties-2_lambda-1001_pi-2_qw-2.dat, copied to tests/golden/) -- is checked here."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def mk2(O):
    p = O.make_params("MK2")
    s = O.SIGMAS["MK2"]
    K = O.MKKeys(p, 0x5EED0001, s["bk"], s["ks"])
    return p, K, O.MKOracle(p, K.bk, K.ksk)


@pytest.fixture(scope="module")
def mk_small(O):
    p = O.make_params("MK2", n=6)
    K = O.MKKeys(p, 9, 2.0**-30.70, 2.0**-13.52)
    return p, K, O.MKOracle(p, K.bk, K.ksk)


def test_mk_cmux_schoolbook_equals_ntt_and_selects(O, mk_small):
    # acc += BK[p][i] (.)_3 (X^a acc - acc)  (J/3gen_mk_internals.jl:59-62); phase under Z = sum_p z_p rotates by a*s
    p, K, orc = mk_small
    rng = np.random.default_rng(1)
    N = p.N
    Z = K.rlwe_keys.sum(axis=0).astype(np.int64)
    mask = rng.integers(-2**63, 2**63, N).astype(np.int64)
    mu = np.full(N, 1 << 61, np.int64)
    body = np.zeros(N, np.int64)
    O.lib().oracle_polymul_ntt64(O.p64(Z), O.p64(mask), N, O.p64(body))
    acc = np.stack([mask, (body.view(np.uint64) + mu.view(np.uint64)).view(np.int64)])
    for party, i, a in [(0, 0, 17), (1, 3, -300), (0, 5, 1023), (1, 2, -1024)]:
        r1 = orc.mux_rotate(party, i, a, acc, schoolbook=True)
        r2 = orc.mux_rotate(party, i, a, acc, schoolbook=False)
        assert np.array_equal(r1, r2)
        prod = np.zeros(N, np.int64)
        O.lib().oracle_polymul_ntt64(O.p64(Z), O.p64(np.ascontiguousarray(r1[0])), N, O.p64(prod))
        phase = (r1[1].view(np.uint64) - prod.view(np.uint64)).view(np.int64)
        exp = np.zeros(N, np.int64)
        O.lib().oracle_mul_by_monomial64(O.p64(mu), a * int(K.lwe_keys[party, i]), N, O.p64(exp))
        err = (phase.view(np.uint64) - exp.view(np.uint64)).view(np.int64) / 2.0**64
        assert np.abs(err).max() < 1e-3


def test_mk_gates_truth_tables_and_noise(O, mk2):
    # multikey NAND trials of runtests.jl:62-102 on the 3-gen gates; noise inside the reference's committed envelope
    p, K, orc = mk2
    s = O.SIGMAS["MK2"]
    a = np.array([0, 0, 1, 1, 1, 0, 1, 0]); b = np.array([0, 1, 0, 1, 1, 1, 0, 0]); c = np.array([1, 1, 1, 1, 0, 0, 1, 0])
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 300 + q) for q, v in enumerate((a, b, c)))
    assert np.array_equal(K.decrypt_bits(ca), a.astype(bool))
    ref = np.loadtxt(os.path.join(O.GOLDEN, "mk_noise_2party_reference.dat"))
    assert 0.03 < ref.std() < 0.06 and np.abs(ref).max() < 0.35       # the reference's own sample: sigma ~ 0.046
    for op, fn in ((O.NAND, lambda x, y: ~(x & y)), (O.XOR, lambda x, y: x ^ y)):
        out = orc.gates(op, ca, cb)
        assert np.array_equal(K.decrypt_bits(out), fn(a.astype(bool), b.astype(bool)))
        noise = np.abs(K.phases(out) / 2.0**32) - 0.125
        assert np.abs(noise).max() < 0.125 and np.abs(noise).max() < 4 * ref.std()
    out = orc.gates(O.AND3, ca[:4], cb[:4], cc[:4])
    assert np.array_equal(K.decrypt_bits(out), (a & b & c)[:4].astype(bool))
    # the reference's gate as written (J/3gen_mk_gates.jl:55-64): three false operands -> phase -1/4 - 3/8 = +3/8 (mod 1) -> TRUE
    assert bool(K.decrypt_bits(orc.gates(O.AND3, ca[7:8], cb[7:8], cc[7:8]))[0])
    out = orc.gates(O.MUX, ca[:4], cb[:4], cc[:4])                        # J/3gen_mk_gates.jl:133-150
    assert np.array_equal(K.decrypt_bits(out), np.where(a == 1, b, c)[:4].astype(bool))
    assert np.array_equal(orc.gates(O.NOT, ca), (-ca.astype(np.int64) % 2**32).astype(np.uint32).view(np.int32))


def test_mk_full_gate_schoolbook_equals_ntt(O, mk_small):
    p, K, orc = mk_small
    ca = K.encrypt_bits([1, 0], 2.0**-13.52, 1); cb = K.encrypt_bits([1, 1], 2.0**-13.52, 2)
    assert np.array_equal(orc.gates(O.NAND, ca, cb, schoolbook=True), orc.gates(O.NAND, ca, cb, schoolbook=False))


def test_mk_keyswitch_combines_parties(O, mk_small):
    # out.a[:, p] = keyswitch(ks[p], (a, 0)).a ; out.b = b + sum_p part_p.b     (J/mk_internals.jl:730-744)
    p, K, orc = mk_small
    rng = np.random.default_rng(2)
    u = rng.integers(-2**31, 2**31, p.N + 1).astype(np.int32)
    got = orc.keyswitch(u)
    t, bb = p.ks_t, p.ks_basebit
    off = 1 << (32 - (1 + bb * t))
    b = int(u[p.N])
    for q in range(p.parties):
        res = np.zeros(p.n + 1, np.int64)
        for i in range(p.N):
            ab = (int(u[i]) + off) % 2**32
            for j in range(1, t + 1):
                d = (ab >> (32 - j * bb)) & ((1 << bb) - 1)
                if d:
                    res -= K.ksk[q, i, j - 1, d - 1].astype(np.int64)
        assert np.array_equal(got[q * p.n:(q + 1) * p.n], (res[:p.n] % 2**32).astype(np.uint32).view(np.int32))
        b += int(res[p.n])
    assert int(got[-1]) == ((b + 2**31) % 2**32) - 2**31


def test_mk_product_keygen_decrypts(O):
    # the product's host keygen (thfhe/keygen.py, mirrors multikey_3gen.jl:15-30) drives the oracle correctly
    import thfhe
    from thfhe import keygen
    pm = thfhe.make_params("MK2")
    MK = keygen.MKSecretKeySet(pm, seed=3, sigma_lwe=2.0**-13.52, sigma_bk=2.0**-30.70)
    orc = O.MKOracle(O.make_params("MK2"), MK.bk, MK.ksk)
    a = np.array([0, 1, 1, 0]); b = np.array([1, 1, 0, 0])
    out = orc.gates(O.NAND, MK.encrypt(a, 1), MK.encrypt(b, 2))
    assert np.array_equal(MK.decrypt(out), ~(a.astype(bool) & b.astype(bool)))


def test_mk_wide_base_sets_16_plus_parties(O):
    # the 16 .. 128-party 3-gen sets use ONE decomposition level with a 24 .. 26-bit base on the ring of degree 2048 (mk_api.jl:214-298):
    # a digit (2^25) times a 32-bit key limb summed over 2 l N terms leaves the exact range of the NTT product, so the oracle switches to
    # four 16-bit limbs; schoolbook (wrapping, always exact) and NTT engines must agree, and the gates must decrypt.  Reduced n and P.
    p = O.make_params("MK16", n=5, parties=3)
    s = O.SIGMAS["MK16"]
    K = O.MKKeys(p, 77, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    a, b = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1])
    ca, cb = K.encrypt_bits(a, s["lwe"], 5), K.encrypt_bits(b, s["lwe"], 6)
    out = orc.gates(O.NAND, ca, cb)
    assert np.array_equal(out[:2], orc.gates(O.NAND, ca[:2], cb[:2], schoolbook=True))
    assert np.array_equal(K.decrypt_bits(out), ~(a.astype(bool) & b.astype(bool)))
    assert np.abs(np.abs(K.phases(out) / 2.0**32) - 0.125).max() < 0.06
    p24 = O.make_params("MK128", n=4, parties=2)
    K24 = O.MKKeys(p24, 78, 2.0**-62, 2.0**-17.42)
    o24 = O.MKOracle(p24, K24.bk, K24.ksk)
    c1, c2 = K24.encrypt_bits([1, 0], 2.0**-17.42, 1), K24.encrypt_bits([1, 1], 2.0**-17.42, 2)
    r = o24.gates(O.XOR, c1, c2)
    assert np.array_equal(r, o24.gates(O.XOR, c1, c2, schoolbook=True)) and np.array_equal(K24.decrypt_bits(r), [False, True])
