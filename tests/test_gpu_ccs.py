"""GPU parity of the CCS multi-key scheme (the reference's mk_bootstrap / mk_gate_nand; pytest -m gpu): thfhe_ccs_* vs the CCS
oracle bit for bit, and the reference's own multi-key test (test/runtests.jl:62-102: random NANDs must decrypt correctly)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(O, name, **over):
    import thfhe
    p = O.make_params(name, **over)
    s = O.SIGMAS[name]
    K = O.CCSKeys(p, 0x5EED0001, s["bk"], s["ks"])
    ck = thfhe.CCSCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.pk, K.crs, K.ksk, device=0)
    return p, s, K, O.CCSOracle(p, K), ck


@pytest.mark.parametrize("name,over", [("CCS2", dict(n=20)),                                   # l = 3: batches of 2 + 1 polynomial groups
                                       ("CCS2", dict(n=9, l=2, Bgbit=8, parties=3)),          # l = 2: 4 groups per batch, two output tasks per wave
                                       ("CCS4", dict(n=6)),                                    # l = 4, 4 parties: one group per batch... (G = 2)
                                       ("CCS2", dict(n=7, l=5, Bgbit=6, parties=1)),          # one group per batch, single party
                                       ("CCS8", dict(n=3)),                                    # the reference's 8-party set: 45 digit rows per stage
                                       ("CCS16", dict(n=2)),                                   # the reference's 16-party set (ccs_blind_rotate_wide_kernel): l = 12, 204 digit rows per stage
                                       ("CCS16", dict(n=3, parties=3)),                        # twelve levels on four polynomials: two level batches per polynomial
                                       ("CCS8", dict(n=2, l=3, parties=11))])                  # more than eight parties with a short gadget (one level batch)
def test_ccs_reduced_bit_exact(O, name, over):
    import thfhe
    p, s, K, orc, ck = make(O, name, **over)
    rng = np.random.default_rng(2)
    a, b = rng.integers(0, 2, 5), rng.integers(0, 2, 5)
    ca, cb = K.encrypt_bits(a, s["lwe"], 31), K.encrypt_bits(b, s["lwe"], 32)
    ca[0, 2] = cb[0, 2] = 0                       # a zero rotation is skipped (J/mk_internals.jl:821-823)
    for op in (O.NAND, O.AND, O.OR, O.XOR):
        assert np.array_equal(ck.gates(op, ca, cb), orc.gates(op, ca, cb)), (name, over, op)
    assert np.array_equal(K.decrypt_bits(thfhe.mk_gate_nand(ck, ca[1:], cb[1:])), ~(a[1:].astype(bool) & b[1:].astype(bool)))
    ref = np.stack([orc.keyswitch(orc.bootstrap_wo_keyswitch(r)) for r in ca[:2]])
    assert np.array_equal(thfhe.mk_bootstrap(ck, thfhe.MU8, ca[:2]), ref)
    with pytest.raises(thfhe.ThfheError):
        ck.gates(thfhe.XNOR, ca, cb)
    assert ck.gates(thfhe.NAND, ca[:0], cb[:0]).shape == (0, p.n * p.parties + 1)
    ck.close()


def test_ccs_2party_full_size_like_runtests(O):
    # mktfhe_parameters_2party (J/mk_api.jl:4-10): n = 560, N = 1024, l = 3, Bgbit = 9, ks 8/2 -- the reference's "multikey NAND" test
    import thfhe
    p, s, K, orc, ck = make(O, "CCS2")
    rng = np.random.default_rng(3)
    B = 64
    m1, m2 = rng.integers(0, 2, B), rng.integers(0, 2, B)
    c1, c2 = K.encrypt_bits(m1, s["lwe"], 41), K.encrypt_bits(m2, s["lwe"], 42)
    t0 = time.time()
    out = thfhe.mk_gate_nand(ck, c1, c2)
    dt = time.time() - t0
    assert np.array_equal(K.decrypt_bits(out), ~(m1.astype(bool) & m2.astype(bool)))
    dev = np.abs(np.abs(K.phases(out) / 2.0**32) - 0.125)
    assert dev.max() < 0.125 and dev.mean() < 0.04     # the 2-party CCS set is noisy by design (about 0.03 rms after one bootstrap)
    assert np.array_equal(out[:2], orc.gates(O.NAND, c1[:2], c2[:2]))
    # 200 gates: the key switch (ks 8/2, one mask per party) goes through the staged kernel; a ragged last workgroup
    B2 = 200
    m1, m2 = rng.integers(0, 2, B2), rng.integers(0, 2, B2)
    c1, c2 = K.encrypt_bits(m1, s["lwe"], 43), K.encrypt_bits(m2, s["lwe"], 44)
    out = thfhe.mk_gate_nand(ck, c1, c2)
    pick = [0, 1, 31, 32, 191, 192, 199]
    assert np.array_equal(out[pick], orc.gates(O.NAND, c1[pick], c2[pick]))
    wrong = int((K.decrypt_bits(out) != ~(m1.astype(bool) & m2.astype(bool))).sum())
    assert wrong <= 2, wrong     # this set decrypts with ~4 sigma of margin (above): a flipped bit in 200 gates is noise, not arithmetic
    assert np.array_equal(out, thfhe.mk_gate_nand(ck, c1, c2))       # deterministic
    print(f"CCS 2-party NAND: {B} gates in {dt * 1e3:.1f} ms")
    ck.close()
