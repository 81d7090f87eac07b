"""CPU oracle of the CCS multi-key scheme (mk_bootstrap / mk_gate_nand; J/mk_internals.jl:477-536,805-858, J/mk_gates.jl:7-13).
The reference's own multi-key test is exactly this gate (test/runtests.jl:62-102: 2 parties, mktfhe_parameters_2party, ten random
NAND trials must decrypt correctly); Julia's RNG stream cannot be reproduced, so ciphertext-level parity is unpinned and the pin is
the truth table + the noise envelope, as for the 3-gen scheme."""
import numpy as np


def test_ccs_mux_rotate_ntt_equals_schoolbook(O):
    p = O.make_params("CCS2", n=4)
    K = O.CCSKeys(p, 5, 3.72e-9, 3.05e-5)
    orc = O.CCSOracle(p, K)
    acc = np.random.default_rng(0).integers(-2**31, 2**31, (p.parties + 1, p.N)).astype(np.int32)
    for party, j, a in [(0, 0, 5), (1, 3, -1000), (0, 2, 1023), (1, 1, -1024)]:
        ref = orc.mux_rotate(party, j, a, acc, schoolbook=True)
        assert np.array_equal(ref, orc.mux_rotate(party, j, a, acc, schoolbook=False))
        acc = ref


def test_ccs_nand_truth_table_like_runtests(O):
    # test/runtests.jl:62-102 with the LWE dimension reduced for CPU time (ring, decomposition and key-switch shape as the reference)
    p = O.make_params("CCS2", n=24)
    s = O.SIGMAS["CCS2"]
    K = O.CCSKeys(p, 0x5EED0001, s["bk"], s["ks"])
    orc = O.CCSOracle(p, K)
    rng = np.random.default_rng(1)
    m1, m2 = rng.integers(0, 2, 10), rng.integers(0, 2, 10)
    c1, c2 = K.encrypt_bits(m1, s["lwe"], 11), K.encrypt_bits(m2, s["lwe"], 12)
    assert np.array_equal(K.decrypt_bits(c1), m1.astype(bool)) and np.array_equal(K.decrypt_bits(c2), m2.astype(bool))
    out = orc.gates(O.NAND, c1, c2)
    assert np.array_equal(K.decrypt_bits(out), ~(m1.astype(bool) & m2.astype(bool)))
    ph = K.phases(out) / 2.0**32
    assert np.abs(np.abs(ph) - 0.125).max() < 0.05          # bootstrapped phase sits at +-1/8
    for op, f in ((O.AND, lambda a, b: a & b), (O.OR, lambda a, b: a | b), (O.XOR, lambda a, b: a ^ b)):
        assert np.array_equal(K.decrypt_bits(orc.gates(op, c1[:4], c2[:4])), f(m1[:4].astype(bool), m2[:4].astype(bool)))
    # composition: gate = keyswitch(bootstrap_wo_keyswitch(prologue))
    tmp = (-(c1[0].astype(np.int64)) - c2[0]).astype(np.int64)
    tmp[-1] += 1 << 29
    tmp = tmp.astype(np.uint32).view(np.int32)
    assert np.array_equal(orc.keyswitch(orc.bootstrap_wo_keyswitch(tmp)), out[0])


def test_ccs_full_size_gate(O):
    p = O.make_params("CCS2")
    s = O.SIGMAS["CCS2"]
    K = O.CCSKeys(p, 3, s["bk"], s["ks"])
    orc = O.CCSOracle(p, K)
    m1, m2 = np.array([1, 0]), np.array([1, 1])
    c1, c2 = K.encrypt_bits(m1, s["lwe"], 1), K.encrypt_bits(m2, s["lwe"], 2)
    out = orc.gates(O.NAND, c1, c2)
    assert np.array_equal(K.decrypt_bits(out), [False, True])
    assert np.abs(np.abs(K.phases(out) / 2.0**32) - 0.125).max() < 0.06
