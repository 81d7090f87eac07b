"""KMS multi-key scheme on the GPU (mk_gate_nand_new / mk_bootstrap_new, 3-gen-mk-tfhe/src/new_mk_gates.jl:1-7, new_mk_internals.jl):
the fused TLev blind-rotation kernel, the relinearisation products (thfhe_pm_mac) and the key switch against the CPU oracle, bit for bit,
on the gadgets of the reference's 2-, 4- and 8-party sets (mk_api.jl:12-20, 64-72, 120-128; ring degree 2048, Torus64) with a reduced LWE
dimension.  Ciphertext parity with the reference itself is UNPINNED (no KMS fixtures, Julia RNG); the pin is the reference's own
multi-key test pattern, the NAND truth table (test/runtests.jl:62-102)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def setup(O, name, n, parties=None, seed=7):
    import thfhe
    from thfhe import keygen, kms
    over = dict(n=n)
    if parties:
        over["parties"] = parties
    p = thfhe.make_kms_params(name, **over)
    K = keygen.KMSSecretKeySet(p, seed=seed)
    orc = O.KMSOracle(p, K.gsw, K.uni, K.pk, K.crs, K.ksk)
    ck = kms.KMSCloudKey(p, K.gsw, K.uni, K.pk, K.crs, K.ksk, device=0)
    return p, K, orc, ck


@pytest.mark.parametrize("name,n,parties", [("KMS2", 12, None), ("KMS2-fast", 5, None), ("KMS4", 6, None), ("KMS8", 4, 3), ("KMS16", 3, 3), ("KMS32", 3, 2),
                                            ("KMS4-fast", 4, 3), ("KMS8-fast", 3, 2), ("KMS16-fast", 3, 2)])   # the `_fast` twins: uni gadgets 7/6, 7/4, 7/4
def test_kms_pieces_and_gates_bit_exact(O, name, n, parties):
    # KMS2: l_gsw = 3, Bgbit 13 -> two-part digits, 12 row parts (two LDS batches); KMS4: l_gsw = 5, Bgbit 8 -> 10 row parts;
    # KMS8 gadgets (l_gsw = 4, Bgbit 11 -> 16 row parts, three batches; l_lev = 3; l_uni = 8) on three parties; KMS16 / KMS32 gadgets
    # (gsw 5/9, lev 3/6, uni 9/4 and gsw 6/8, lev 3/7, uni 16/2: the 16- and 32-party sets, mk_api.jl:194-202, 225-233) on three / two parties
    from thfhe import kms
    p, K, orc, ck = setup(O, name, n, parties)
    a, b = np.array([0, 0, 1, 1, 1]), np.array([0, 1, 0, 1, 1])
    ca, cb = K.encrypt(a, 21), K.encrypt(b, 22)
    # mk_ith_blind_rotate: the TLev accumulator of every party
    bar = kms.modswitch(ca, p.N)
    for party in range(p.parties):
        lev = ck.tlev_rotate(party, bar[:2, party * n:(party + 1) * n])
        for g in range(2):
            assert np.array_equal(lev[g], orc.tlev_rotate(party, bar[g, party * n:(party + 1) * n])), (name, party, g)
    # mk_lev_rlwe_mul on a random accumulator
    rng = np.random.default_rng(3)
    acc = rng.integers(-2**63, 2**63, size=(2, p.parties + 1, p.N), dtype=np.int64)
    party = p.parties - 1
    got = ck.lev_rlwe_mul(party, acc, lev)
    for g in range(2):
        assert np.array_equal(got[g], orc.lev_rlwe_mul(party, acc[g], lev[g])), (name, g)
    # whole bootstrap and the gate
    u = ck.bootstrap_wo_keyswitch(ca[:2])
    assert np.array_equal(u[0], orc.bootstrap_wo_keyswitch(ca[0]))
    assert np.array_equal(ck.keyswitch(u), np.stack([orc.keyswitch(r) for r in u]))
    out = kms.mk_gate_nand_new(ck, ca, cb)
    assert np.array_equal(out, orc.gates(O.NAND, ca, cb))
    assert np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))
    assert np.array_equal(ck.gates(O.XOR, ca, cb), orc.gates(O.XOR, ca, cb))
    # fast_boot = true (mk_blind_rotate_new_v2, new_mk_internals.jl:255-269): one RLWE rotation for the first party
    acc1 = rng.integers(-2**63, 2**63, size=(2, 2, p.N), dtype=np.int64)
    rot = ck.rlwe_rotate(0, bar[:2, :n], acc1)
    for g in range(2):
        assert np.array_equal(rot[g], orc.rlwe_rotate(0, bar[g, :n], acc1[g])), (name, g)
    fast = kms.mk_gate_nand_new(ck, ca, cb, fast_boot=True)
    assert np.array_equal(fast, orc.gates(O.NAND, ca, cb, fast_boot=True))
    assert np.array_equal(K.decrypt(fast), ~(a.astype(bool) & b.astype(bool)))
    assert np.array_equal(kms.mk_bootstrap_new(ck, 1 << 61, ca[:1], fast_boot=True), ck.keyswitch(orc.bootstrap_wo_keyswitch(ca[0], fast_boot=True)[None]))
    ck.close()


def test_kms2_truth_table_larger_n(O):
    # the reference's NAND test pattern with a larger LWE dimension (n = 96 of 560): decrypt-level check, two gates against the oracle
    from thfhe import kms
    p, K, orc, ck = setup(O, "KMS2", 96, seed=9)
    a, b = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1])
    ca, cb = K.encrypt(a, 31), K.encrypt(b, 32)
    out = kms.mk_gate_nand_new(ck, ca, cb)
    assert np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))
    assert np.abs(np.abs(K.phase(out).astype(np.float64) / 2.0**32) - 0.125).max() < 0.06
    assert np.array_equal(out[:2], orc.gates(O.NAND, ca[:2], cb[:2]))
    ck.close()


def test_kms2_keyswitch_full_n_both_kernels(O):
    # mk_keyswitch (mk_internals.jl:714-728) at the reference's n = 560 (640-word rows): 100 samples one workgroup per (sample, party), 260 through
    # the staged kernel; P N + 1 = 4097-word extracted samples with one mask per party
    p, K, orc, ck = setup(O, "KMS2", 560, seed=10)
    rng = np.random.default_rng(14)
    for count in (100, 260):
        u = rng.integers(-2**31, 2**31, (count, p.parties * p.N + 1), dtype=np.int64).astype(np.int32)
        got = ck.keyswitch(u)
        for g in list(range(8)) + list(range(count - 40, count)):
            assert np.array_equal(got[g], orc.keyswitch(u[g])), (count, g)
    ck.close()


@pytest.mark.parametrize("name,n,parties", [("KMS16", 2, 16), ("KMS32", 2, 32)])
def test_kms_16_and_32_party_sets_at_the_real_party_count(O, name, n, parties):
    # mktfhe_parameters_16party_new / _32party_new (mk_api.jl:194-202, 225-233) with ALL 16 / 32 parties (17 / 33 accumulator polynomials, the
    # relinearisation index tables at their real width); only the LWE dimension is reduced.  Gate, fast_boot gate and the pre-key-switch sample.
    from thfhe import kms
    p, K, orc, ck = setup(O, name, n, parties, seed=11)
    a, b = np.array([0, 1, 1]), np.array([1, 1, 0])
    ca, cb = K.encrypt(a, 41), K.encrypt(b, 42)
    out = kms.mk_gate_nand_new(ck, ca, cb)
    assert np.array_equal(out, orc.gates(O.NAND, ca, cb)), name
    assert np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))
    u = ck.bootstrap_wo_keyswitch(ca[:1])
    assert np.array_equal(u[0], orc.bootstrap_wo_keyswitch(ca[0]))
    fast = kms.mk_gate_nand_new(ck, ca[:2], cb[:2], fast_boot=True)
    assert np.array_equal(fast, orc.gates(O.NAND, ca[:2], cb[:2], fast_boot=True))
    ck.close()


def test_kms4_moderate_size_buffers_regrow_between_routes(O):
    # KMS4 with all four parties, n = 32, 40 gates: the G-dependent index tables of the relinearisation (terms / first / pos behind the index
    # list, W_EF / W_INDEX regrowing between parties and between the fast_boot and the normal route) at a batch where offsets matter.
    # One context: fast_boot = True first, then False, then a LARGER batch again; sampled gates and pre-key-switch samples word for word.
    from thfhe import kms
    p, K, orc, ck = setup(O, "KMS4", 32, seed=13)
    rng = np.random.default_rng(17)
    G = 40
    a, b = rng.integers(0, 2, G), rng.integers(0, 2, G)
    ca, cb = K.encrypt(a, 51), K.encrypt(b, 52)
    idx = np.array([0, 7, 19, 26, 39])
    fast = kms.mk_gate_nand_new(ck, ca[:12], cb[:12], fast_boot=True)          # small batch through the fast route first
    assert np.array_equal(fast[[0, 7, 11]], orc.gates(O.NAND, ca[[0, 7, 11]], cb[[0, 7, 11]], fast_boot=True))
    out = kms.mk_gate_nand_new(ck, ca, cb)                                      # buffers regrow: 12 -> 40 gates, other route
    assert np.array_equal(out[idx], orc.gates(O.NAND, ca[idx], cb[idx]))
    assert np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))
    fast = kms.mk_gate_nand_new(ck, ca, cb, fast_boot=True)                     # and back, at the full batch
    assert np.array_equal(fast[idx], orc.gates(O.NAND, ca[idx], cb[idx], fast_boot=True))
    assert np.array_equal(K.decrypt(fast), ~(a.astype(bool) & b.astype(bool)))
    u = ck.bootstrap_wo_keyswitch(ca)
    for g in (3, 38):
        assert np.array_equal(u[g], orc.bootstrap_wo_keyswitch(ca[g])), g
    assert np.array_equal(ck.gates(O.XOR, ca[:9], cb[:9])[[2, 8]], orc.gates(O.XOR, ca[[2, 8]], cb[[2, 8]]))   # shrinks again
    ck.close()


@pytest.mark.parametrize("name,n,parties", [("KMS2", 12, None), ("KMS4", 6, None), ("KMS8", 4, 3), ("KMS32", 3, 2)])
def test_kms_two_jobs_per_workgroup_bit_exact(O, name, n, parties):
    # kms_tlev_rotate_pair_kernel (two jobs share every key chunk; row parts in batches of six with the partial spectra parked in between):
    # forced for small batches.  KMS2: 12 row parts, l_lev = 2 (the two TLev samples of a gate form a pair); KMS4: 10 row parts; KMS8: 16 row
    # parts in three batches, l_lev = 3 (pairs straddle gates: different rotations in one workgroup); KMS32: l_lev = 7, odd job counts.
    # Against the oracle and against the one-job kernel, incl. zero mask words in one job of a pair and the RLWE rotation of fast_boot.
    from thfhe import kms
    p, K, orc, ck = setup(O, name, n, parties)
    a, b = np.array([0, 0, 1, 1, 1]), np.array([0, 1, 0, 1, 1])
    ca, cb = K.encrypt(a, 21), K.encrypt(b, 22)
    ca[1, 2] = cb[1, 2] = 0
    ca[4, n + 1] = cb[4, n + 1] = 0
    bar = kms.modswitch(ca, p.N)
    single_lev = [ck.tlev_rotate(party, bar[:3, party * n:(party + 1) * n]) for party in range(p.parties)]
    single_gate = kms.mk_gate_nand_new(ck, ca, cb)
    rng = np.random.default_rng(3)
    acc1 = rng.integers(-2**63, 2**63, size=(3, 2, p.N), dtype=np.int64)
    single_rot = ck.rlwe_rotate(0, bar[:3, :n], acc1)
    ck.set_pair_threshold(0)
    for party in range(p.parties):
        lev = ck.tlev_rotate(party, bar[:3, party * n:(party + 1) * n])
        assert np.array_equal(lev, single_lev[party]), (name, party)
        for g in range(3):
            assert np.array_equal(lev[g], orc.tlev_rotate(party, bar[g, party * n:(party + 1) * n])), (name, party, g)
    assert np.array_equal(ck.rlwe_rotate(0, bar[:3, :n], acc1), single_rot)
    out = kms.mk_gate_nand_new(ck, ca, cb)
    assert np.array_equal(out, single_gate)
    assert np.array_equal(out, orc.gates(O.NAND, ca, cb))
    fast = kms.mk_gate_nand_new(ck, ca, cb, fast_boot=True)
    assert np.array_equal(fast, orc.gates(O.NAND, ca, cb, fast_boot=True))
    fa, fb = K.encrypt(a, 23), K.encrypt(b, 24)
    assert np.array_equal(K.decrypt(kms.mk_gate_nand_new(ck, fa, fb)), ~(a.astype(bool) & b.astype(bool)))
    ck.close()
