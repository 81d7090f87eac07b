"""GPU parity tests of the 3-gen multi-key path (pytest -m gpu): C ABI vs the MK oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk2gpu(O):
    import thfhe
    p = O.make_params("MK2")
    s = O.SIGMAS["MK2"]
    K = O.MKKeys(p, 0x5EED0001, s["bk"], s["ks"])
    ck = thfhe.MKCloudKey(thfhe.make_params("MK2"), K.bk, K.ksk, device=0)
    yield p, K, O.MKOracle(p, K.bk, K.ksk), ck
    ck.close()


def check_mk_keyswitch(ck, orc, p, counts, seed):
    """mk_keyswitch_3gen (J/mk_internals.jl:730-744) of random extracted samples through thfhe_mk_keyswitch_dev: below 192 samples one workgroup
    per (sample, party), from 192 on the staged kernel (rows in LDS, the digit selects an address); sampled rows against the oracle."""
    import thfhe
    rng = np.random.default_rng(seed)
    for count in counts:
        u = rng.integers(-2**31, 2**31, (count, p.N + 1), dtype=np.int64).astype(np.int32)
        u[0, :] = 0
        u[count - 1, :] = -1
        du = thfhe.DeviceBuffer(ck, u.nbytes).upload(u)
        do = thfhe.DeviceBuffer(ck, count * (p.parties * p.n + 1) * 4)
        assert thfhe.lib().thfhe_mk_keyswitch_dev(ck.h, du.ptr, do.ptr, count) == 0
        ck.sync()
        got = do.download((count, p.parties * p.n + 1))
        du.free(); do.free()
        rows = sorted(set(list(range(min(count, 12))) + list(range(max(0, count - 36), count)) + list(range(0, count, 97))))
        for g in rows:
            assert np.array_equal(got[g], orc.keyswitch(u[g])), (count, g)


def test_mk2_keyswitch_kernels_bit_exact(O, mk2gpu):
    # ks 3/3: seven rows per (i, j), three levels -- stages straddle coordinates; ragged last workgroups
    p, K, orc, ck = mk2gpu
    check_mk_keyswitch(ck, orc, p, (5, 191, 192, 333, 1025), 21)


@pytest.mark.parametrize("name", ["MK64", "MK256"])
def test_mk_keyswitch_768_word_rows(O, name):
    # n = 650 (ks 4/3: seven rows per (i, j), ONE (i, j) per stage) and n = 740 (ks 8/2, two per stage) at the sets' own LWE dimension, two parties
    import thfhe
    p = O.make_params(name, parties=2)
    s = O.SIGMAS[name]
    K = O.MKKeys(p, 79, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params(name, parties=2), K.bk, K.ksk, device=0)
    check_mk_keyswitch(ck, orc, p, (64, 230), 24)
    ck.close()


def test_mk2_gates_bit_exact(O, mk2gpu):
    # mk_gate_{nand,or,and,xor,3and,mux,not}_3gen, J/3gen_mk_gates.jl:8-150
    import thfhe
    p, K, orc, ck = mk2gpu
    s = O.SIGMAS["MK2"]
    a = np.array([0, 0, 1, 1, 1, 0]); b = np.array([0, 1, 0, 1, 1, 0]); c = np.array([1, 0, 1, 0, 1, 1])
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 400 + q) for q, v in enumerate((a, b, c)))
    api = {O.NAND: (thfhe.mk_gate_nand_3gen, lambda x, y: ~(x & y)), O.OR: (thfhe.mk_gate_or_3gen, lambda x, y: x | y),
           O.AND: (thfhe.mk_gate_and_3gen, lambda x, y: x & y), O.XOR: (thfhe.mk_gate_xor_3gen, lambda x, y: x ^ y)}
    for op, (fn, truth) in api.items():
        got = fn(ck, ca, cb)
        assert np.array_equal(got, orc.gates(op, ca, cb)), f"gate {op}"
        assert np.array_equal(K.decrypt_bits(got), truth(a.astype(bool), b.astype(bool)))
    got = thfhe.mk_gate_3and_3gen(ck, ca, cb, cc)
    assert np.array_equal(got, orc.gates(O.AND3, ca, cb, cc))
    assert np.array_equal(K.decrypt_bits(got), (a & b & c).astype(bool))
    got = thfhe.mk_gate_mux_3gen(ck, ca, cb, cc)
    assert np.array_equal(got, orc.gates(O.MUX, ca, cb, cc))
    assert np.array_equal(K.decrypt_bits(got), np.where(a == 1, b, c).astype(bool))
    assert np.array_equal(thfhe.mk_gate_not_3gen(ck, ca), orc.gates(O.NOT, ca))
    # mk_bootstrap_3gen(bk, ks, mu, x)  J/3gen_mk_internals.jl:112-116
    x = ca[:2]
    ref = np.stack([orc.keyswitch(orc.bootstrap_wo_keyswitch(r)) for r in x])
    assert np.array_equal(thfhe.mk_bootstrap_3gen(ck, thfhe.MU8_64, x), ref)
    # gates the 3-gen scheme does not define are refused
    with pytest.raises(thfhe.ThfheError):
        ck.gates(thfhe.XNOR, ca, cb)
    assert ck.gates(thfhe.NAND, ca[:0], cb[:0]).shape == (0, p.n * p.parties + 1)


def test_mk2_batch_1024_properties(O, mk2gpu):
    # BASELINE.json configs[2]: 2-party gate bootstrap, 1024-gate batch, device-resident records
    import thfhe
    p, K, orc, ck = mk2gpu
    s = O.SIGMAS["MK2"]
    B = 1024
    rng = np.random.default_rng(5)
    a, b = rng.integers(0, 2, B), rng.integers(0, 2, B)
    xa, xb = K.encrypt_bits(a, s["lwe"], 501), K.encrypt_bits(b, s["lwe"], 502)
    da, db, do = ck.device_records(B), ck.device_records(B), ck.device_records(B)
    da.upload(xa); db.upload(xb); ck.reserve(B)
    ck.gates_dev(thfhe.NAND, da, db, None, do, B); ck.sync()
    out1 = do.download((B, p.n * p.parties + 1))
    ck.gates_dev(thfhe.NAND, da, db, None, do, B); ck.sync()
    assert np.array_equal(out1, do.download((B, p.n * p.parties + 1)))          # deterministic
    assert np.array_equal(K.decrypt_bits(out1), ~(a.astype(bool) & b.astype(bool)))
    assert np.abs(np.abs(K.phases(out1) / 2.0**32) - 0.125).max() < 0.125
    idx = np.sort(rng.choice(B, 160, replace=False))   # 160 of the 1024 gates against the MK oracle, bit for bit
    assert np.array_equal(out1[idx], orc.gates(O.NAND, xa[idx], xb[idx]))
    # odd batch size (not a multiple of the 4 gates a workgroup holds)
    assert np.array_equal(ck.gates(thfhe.XOR, xa[:5], xb[:5]), orc.gates(O.XOR, xa[:5], xb[:5]))


def test_mk4_bit_exact(O):
    # mktfhe_parameters_4party_3gen: P = 4, n = 510, l = 3, Bgbit = 6, ks 5/2    (J/mk_api.jl:84-90)
    import thfhe
    p = O.make_params("MK4")
    s = O.SIGMAS["MK4"]
    K = O.MKKeys(p, 77, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params("MK4"), K.bk, K.ksk, device=0)
    a = np.array([0, 1, 1, 0]); b = np.array([1, 1, 0, 0])
    ca, cb = K.encrypt_bits(a, s["lwe"], 1), K.encrypt_bits(b, s["lwe"], 2)
    got = ck.gates(thfhe.NAND, ca, cb)
    assert np.array_equal(got, orc.gates(O.NAND, ca, cb))
    assert np.array_equal(K.decrypt_bits(got), ~(a.astype(bool) & b.astype(bool)))
    check_mk_keyswitch(ck, orc, p, (100, 230), 22)   # ks 5/2: three rows per (i, j), five levels
    ck.close()


@pytest.mark.parametrize("name,n", [("MK3", 510), ("MK5", 520), ("MK8", 96)])
def test_mk5_mk8_bit_exact(O, name, n):
    # mktfhe_parameters_3party_3gen (P = 3, l = 2, Bgbit = 7; full size, J/mk_api.jl:44-50), mktfhe_parameters_5party_3gen (P = 5, l = 3, Bgbit = 6; full size) and mktfhe_parameters_8party_3gen (P = 8, l = 4, Bgbit = 4:
    # eight digit rows, key rows streamed through a register window; LWE dimension reduced to keep the 8-party oracle in seconds)
    # J/mk_api.jl:98-104, 140-146.  Small batches take the one-gate-per-workgroup kernel, 300 gates the two-gate kernel (l <= 3).
    import thfhe
    p = O.make_params(name, n=n)
    s = O.SIGMAS[name]
    K = O.MKKeys(p, 79, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params(name, n=n), K.bk, K.ksk, device=0)
    a = np.array([0, 1, 1, 0, 1]); b = np.array([1, 1, 0, 0, 1]); c = np.array([1, 0, 1, 1, 0])
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 11 + q) for q, v in enumerate((a, b, c)))
    for op in (O.NAND, O.XOR):
        got = ck.gates(op, ca, cb)
        assert np.array_equal(got, orc.gates(op, ca, cb)), (name, op)
    got = ck.gates(O.NAND, ca, cb)
    assert np.array_equal(K.decrypt_bits(got), ~(a.astype(bool) & b.astype(bool)))
    assert np.array_equal(ck.gates(O.AND3, ca, cb, cc), orc.gates(O.AND3, ca, cb, cc))
    assert np.array_equal(ck.gates(O.MUX, ca, cb, cc), orc.gates(O.MUX, ca, cb, cc))
    if name == "MK5":   # the throughput kernel on an odd batch
        rng = np.random.default_rng(4)
        B = 301
        xa, xb = K.encrypt_bits(rng.integers(0, 2, B), s["lwe"], 21), K.encrypt_bits(rng.integers(0, 2, B), s["lwe"], 22)
        got = ck.gates(O.NAND, xa, xb)
        idx = np.sort(rng.choice(B, 24, replace=False))
        assert np.array_equal(got[idx], orc.gates(O.NAND, xa[idx], xb[idx]))
        assert np.array_equal(K.decrypt_bits(got), ~(K.decrypt_bits(xa) & K.decrypt_bits(xb)))
    ck.close()


def test_mk_n2048_reduced_n_all_gates_bit_exact(O):
    # ring degree 2048 (BASELINE configs[4]): radix-2 split + two twisted 512-point transforms; l = 3 and l = 2 shapes
    import thfhe
    for name, over in (("MK4-N2048", dict(n=40, parties=2)), ("MK2", dict(n=33, N=2048))):
        p = O.make_params(name, **over)
        s = O.SIGMAS[name]
        K = O.MKKeys(p, 11, s["bk"], s["ks"])
        orc = O.MKOracle(p, K.bk, K.ksk)
        ck = thfhe.MKCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.ksk, device=0)
        a = np.array([0, 0, 1, 1, 1]); b = np.array([0, 1, 0, 1, 1]); c = np.array([1, 0, 1, 0, 0])
        ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 40 + q) for q, v in enumerate((a, b, c)))
        for op, args in ((O.NAND, (ca, cb)), (O.XOR, (ca, cb)), (O.OR, (ca, cb)), (O.AND3, (ca, cb, cc)), (O.MUX, (ca, cb, cc))):
            got = ck.gates(op, *args)
            assert np.array_equal(got, orc.gates(op, *args)), (name, op)
        assert np.array_equal(K.decrypt_bits(ck.gates(thfhe.NAND, ca, cb)), ~(a.astype(bool) & b.astype(bool)))
        ref = np.stack([orc.keyswitch(orc.bootstrap_wo_keyswitch(r)) for r in ca[:2]])
        assert np.array_equal(thfhe.mk_bootstrap_3gen(ck, thfhe.MU8_64, ca[:2]), ref)
        ck.close()


@pytest.mark.parametrize("name,n,parties", [("MK16", 12, 3), ("MK64", 9, 2), ("MK128", 10, 2)])
def test_mk_wide_base_16_plus_party_sets_bit_exact(O, name, n, parties):
    # mktfhe_parameters_{16,32,64,128}party_3gen (J/mk_api.jl:214-298): N = 2048, ONE level with a 26 / 25 / 24-bit base.  A 26-bit digit times a
    # 16-bit key limb is outside the FP64 exactness bound, so the kernel cuts every digit into three balanced parts of 9 / 9 / 8 bit and
    # multiplies part w into the key row shifted left by 9 w bits (three copies in the key table): six row parts, the l = 3 shape.
    # Gadgets, ring and key-switch shape of the reference sets, LWE dimension and party count reduced to keep the oracle in seconds.
    import thfhe
    p = O.make_params(name, n=n, parties=parties)
    s = O.SIGMAS[name]
    K = O.MKKeys(p, 83, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.ksk, device=0)
    a = np.array([0, 0, 1, 1, 1]); b = np.array([0, 1, 0, 1, 1]); c = np.array([1, 0, 1, 0, 0])
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 50 + q) for q, v in enumerate((a, b, c)))
    for op, args in ((O.NAND, (ca, cb)), (O.XOR, (ca, cb)), (O.AND3, (ca, cb, cc)), (O.MUX, (ca, cb, cc))):
        assert np.array_equal(ck.gates(op, *args), orc.gates(op, *args)), (name, op)
    assert np.array_equal(K.decrypt_bits(ck.gates(thfhe.NAND, ca, cb)), ~(a.astype(bool) & b.astype(bool)))
    ck.close()


@pytest.mark.parametrize("n,parties", [(10, 2), (3, 9)])
def test_mk256_two_level_wide_base_gadget_bit_exact(O, n, parties):
    # mktfhe_parameters_256party_3gen (J/mk_api.jl:304-310): N = 2048, l = 2, Bgbit = 18, ks 8/2.  Two levels of two 9-bit digit parts = eight row
    # parts per CMux: more than the one-pass N = 2048 kernel holds in LDS, so the rotation goes through the batched kernel shared with the KMS scheme
    # (thfhe_rot2k.h) on a key table in its layout, accumulators in global memory, digits taken from all 36 bits.  Gadget, ring and key-switch shape of
    # the reference set; LWE dimension and party count reduced (9 parties: an odd count, more than one batch of chunks per party).
    import thfhe
    p = O.make_params("MK256", n=n, parties=parties)
    s = O.SIGMAS["MK256"]
    K = O.MKKeys(p, 87, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.ksk, device=0)
    assert ck.rotation_kernel_name(4) == "kms_tlev_rotate_kernel"
    a = np.array([0, 0, 1, 1, 1]); b = np.array([0, 1, 0, 1, 1]); c = np.array([1, 0, 1, 0, 0])
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 60 + q) for q, v in enumerate((a, b, c)))
    for op, args in ((O.NAND, (ca, cb)), (O.XOR, (ca, cb)), (O.AND3, (ca, cb, cc)), (O.MUX, (ca, cb, cc))):
        assert np.array_equal(ck.gates(op, *args), orc.gates(op, *args)), op
    assert np.array_equal(K.decrypt_bits(ck.gates(thfhe.NAND, ca, cb)), ~(a.astype(bool) & b.astype(bool)))
    ref = np.stack([orc.keyswitch(orc.bootstrap_wo_keyswitch(r)) for r in ca[:2]])
    assert np.array_equal(thfhe.mk_bootstrap_3gen(ck, thfhe.MU8_64, ca[:2]), ref)
    # the same through two jobs per workgroup (kms_tlev_rotate_pair_kernel: eight row parts = two batches, partial spectra parked in between)
    ck.set_pair_threshold(0)
    assert ck.rotation_kernel_name(4) == "kms_tlev_rotate_pair_kernel"
    for op, args in ((O.NAND, (ca, cb)), (O.AND3, (ca, cb, cc)), (O.MUX, (ca, cb, cc))):
        assert np.array_equal(ck.gates(op, *args), orc.gates(op, *args)), op
    ck.close()


@pytest.mark.parametrize("name,n,parties", [("MK64-fft", 6, 2), ("MK512", 4, 3), ("MK64-fft", 2, 64)])   # the last one: all 64 parties
def test_mk_ring_4096_sets_bit_exact(O, name, n, parties):
    # mktfhe_parameters_64party_3gen_for_fft / _512party_3gen (J/mk_api.jl:277-283, 316-322): ring of degree 4096, one level with a 27-bit base
    # = three 9-bit digit parts, six row parts; r4k_rotate_kernel (radix-4 split into four twisted 512-point transforms, two passes per step).
    # Gadget, ring and key-switch shape of the reference sets; LWE dimension and party count reduced.  Zero mask words included.
    import thfhe
    p = O.make_params(name, n=n, parties=parties)
    s = O.SIGMAS[name]
    K = O.MKKeys(p, 91, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.ksk, device=0)
    assert ck.rotation_kernel_name(4) == "r4k_rotate_kernel"
    a = np.array([0, 0, 1, 1, 1]); b = np.array([0, 1, 0, 1, 1]); c = np.array([1, 0, 1, 0, 0])
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 70 + q) for q, v in enumerate((a, b, c)))
    assert np.array_equal(K.decrypt_bits(ck.gates(thfhe.NAND, ca, cb)), ~(a.astype(bool) & b.astype(bool)))
    ca[1, 2] = cb[1, 2] = 0
    for op, args in ((O.NAND, (ca, cb)), (O.XOR, (ca, cb)), (O.AND3, (ca, cb, cc)), (O.MUX, (ca, cb, cc))):
        assert np.array_equal(ck.gates(op, *args), orc.gates(op, *args)), op
    ref = np.stack([orc.keyswitch(orc.bootstrap_wo_keyswitch(r)) for r in ca[:2]])
    assert np.array_equal(thfhe.mk_bootstrap_3gen(ck, thfhe.MU8_64, ca[:2]), ref)
    ck.close()


def test_mk16_full_size_bit_exact(O):
    # mktfhe_parameters_16party_3gen AS WRITTEN (J/mk_api.jl:214-220): P = 16, n = 590, N = 2048, l = 1, Bgbit = 26, ks 4/3 -- 9 440 sequential
    # CMuxes per gate through the three-part digit split and the shifted key-row copies of mk_expand_parts_kernel at the REAL party count
    # (the reduced gadgets above stop at P = 3).  Four gates, every output word against the oracle.
    import thfhe
    p = O.make_params("MK16")
    s = O.SIGMAS["MK16"]
    K = O.MKKeys(p, 85, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params("MK16"), K.bk, K.ksk, device=0)
    a = np.array([0, 1, 1, 0]); b = np.array([1, 1, 0, 0])
    ca, cb = K.encrypt_bits(a, s["lwe"], 1), K.encrypt_bits(b, s["lwe"], 2)
    got = ck.gates(thfhe.NAND, ca, cb)
    assert np.array_equal(got, orc.gates(O.NAND, ca, cb))
    assert np.array_equal(K.decrypt_bits(got), ~(a.astype(bool) & b.astype(bool)))
    ck.close()


def test_mk4_full_n_pair_kernel_32_gates_bit_exact(O):
    # the l = 3 two-gates-per-workgroup kernel (the one bench.py --set MK4 measures) at the reference's full 4-party size, forced for a
    # 33-gate batch (odd: a lone last gate): every output word of every gate against the oracle
    import thfhe
    p = O.make_params("MK4")
    s = O.SIGMAS["MK4"]
    K = O.MKKeys(p, 77, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params("MK4"), K.bk, K.ksk, device=0)
    ck.set_pair_threshold(0)
    rng = np.random.default_rng(21)
    B = 33
    a, b = rng.integers(0, 2, B), rng.integers(0, 2, B)
    ca, cb = K.encrypt_bits(a, s["lwe"], 31), K.encrypt_bits(b, s["lwe"], 32)
    got = ck.gates(thfhe.NAND, ca, cb)
    assert ck.rotation_kernel_name(B) == "mk_blind_rotate_pair_kernel<3>"
    assert np.array_equal(got, orc.gates(O.NAND, ca, cb))
    assert np.array_equal(K.decrypt_bits(got), ~(a.astype(bool) & b.astype(bool)))
    ck.close()


def test_mk4_n2048_full_size(O):
    # BASELINE configs[4]: 4 parties, N = 2048, l = 3 at the reference's 4-party LWE dimension (n = 510)
    import thfhe
    p = O.make_params("MK4-N2048")
    s = O.SIGMAS["MK4-N2048"]
    K = O.MKKeys(p, 78, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params("MK4-N2048"), K.bk, K.ksk, device=0)
    rng = np.random.default_rng(9)
    B = 96
    a, b = rng.integers(0, 2, B), rng.integers(0, 2, B)
    ca, cb = K.encrypt_bits(a, s["lwe"], 1), K.encrypt_bits(b, s["lwe"], 2)
    got = ck.gates(thfhe.NAND, ca, cb)
    assert np.array_equal(got, orc.gates(O.NAND, ca, cb))      # all 96 gates against the MK oracle, bit for bit
    assert np.array_equal(K.decrypt_bits(got), ~(a.astype(bool) & b.astype(bool)))
    assert np.array_equal(got, ck.gates(thfhe.NAND, ca, cb))   # deterministic
    ck.set_pair_threshold(0)                                     # the same 96 gates two per workgroup (mk_blind_rotate_pair2k_kernel) at full size
    assert ck.rotation_kernel_name(B) == "mk_blind_rotate_pair2k_kernel<3>"
    assert np.array_equal(got, ck.gates(thfhe.NAND, ca, cb))
    check_mk_keyswitch(ck, orc, p, (200,), 23)                   # N = 2048 coordinates per party through the staged key switch
    ck.close()


def test_mk_n2048_pair_kernel_bit_exact(O):
    # two gates per workgroup on the ring of degree 2048 (mk_blind_rotate_pair2k_kernel: two half passes per step, digits extracted twice):
    # forced for small odd batches, zero mask words in one gate of a pair, a lone last gate; l = 3 and l = 2 shapes; the wide-base shape
    # (MK16 gadget: one level, three digit parts) as well.  Against the oracle AND against the one-gate kernel.
    import thfhe
    for name, over in (("MK4-N2048", dict(n=40, parties=2)), ("MK2", dict(n=33, N=2048)), ("MK16", dict(n=12, parties=3))):
        p = O.make_params(name, **over)
        s = O.SIGMAS[name]
        K = O.MKKeys(p, 21, s["bk"], s["ks"])
        orc = O.MKOracle(p, K.bk, K.ksk)
        ck = thfhe.MKCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.ksk, device=0)
        rng = np.random.default_rng(5)
        a, b, c = (rng.integers(0, 2, 7) for _ in range(3))
        ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 300 + q) for q, v in enumerate((a, b, c)))
        ca[0, 3] = cb[0, 3] = 0               # bara = 0 for gate 0 at i = 3, not for gate 1
        ca[3, p.n + 2] = cb[3, p.n + 2] = 0   # second party's range, odd gate of a pair
        ca[6, 0] = cb[6, 0] = 0               # the lone gate of the last workgroup
        single = {op: ck.gates(op, *args) for op, args in ((O.NAND, (ca, cb)), (O.AND3, (ca, cb, cc)))}
        ck.set_pair_threshold(0)
        assert "pair2k" in ck.rotation_kernel_name(7)
        for op, args in ((O.NAND, (ca, cb)), (O.XOR, (ca, cb)), (O.AND3, (ca, cb, cc)), (O.MUX, (ca, cb, cc))):
            got = ck.gates(op, *args)
            assert np.array_equal(got, orc.gates(op, *args)), (name, op)
            if op in single:
                assert np.array_equal(got, single[op]), (name, op)
        fa, fb = K.encrypt_bits(a, s["lwe"], 310), K.encrypt_bits(b, s["lwe"], 311)   # untouched ciphertexts for the truth table
        assert np.array_equal(K.decrypt_bits(ck.gates(thfhe.NAND, fa, fb)), ~(a.astype(bool) & b.astype(bool)))
        ck.close()


def test_mk_pair_kernel_bit_exact(O, mk2gpu):
    # throughput kernel (two gates per workgroup share every key chunk): forced for small odd batches, incl. a zero mod-switched
    # mask word in one gate of a pair (that gate skips the CMux, its partner does not) and a lone last gate
    import thfhe
    p, K, orc, ck = mk2gpu
    s = O.SIGMAS["MK2"]
    rng = np.random.default_rng(12)
    a, b, c = (rng.integers(0, 2, 7) for _ in range(3))
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 700 + q) for q, v in enumerate((a, b, c)))
    ca[0, 3] = cb[0, 3] = 0           # bara = 0 for gate 0 at i = 3, not for gate 1
    ca[2, 600] = cb[2, 600] = 0       # second party's range
    ca[6, 0] = cb[6, 0] = 0           # the lone gate of the last workgroup
    ck.set_pair_threshold(0)
    try:
        for op, args in ((O.NAND, (ca, cb)), (O.XOR, (ca, cb)), (O.AND3, (ca, cb, cc)), (O.MUX, (ca, cb, cc))):
            assert np.array_equal(ck.gates(op, *args), orc.gates(op, *args)), op
        ops = np.array([O.NAND, O.OR, O.AND, O.XOR, O.NAND, O.XOR, O.OR], np.int32)
        ref = np.stack([orc.gates(int(o), ca[i:i + 1], cb[i:i + 1])[0] for i, o in enumerate(ops)])
        assert np.array_equal(ck.gates_mixed(ops, ca, cb), ref)
    finally:
        ck.set_pair_threshold(256)


def test_mk4_pair_kernel_bit_exact(O):
    import thfhe
    p = O.make_params("MK4", n=48)     # l = 3: twelve forward transforms per step on eight waves
    s = O.SIGMAS["MK4"]
    K = O.MKKeys(p, 79, s["bk"], s["ks"])
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.ksk, device=0)
    ck.set_pair_threshold(0)
    a = np.array([0, 1, 1, 0, 1]); b = np.array([1, 1, 0, 0, 1])
    ca, cb = K.encrypt_bits(a, s["lwe"], 1), K.encrypt_bits(b, s["lwe"], 2)
    got = ck.gates(thfhe.NAND, ca, cb)
    assert np.array_equal(got, orc.gates(O.NAND, ca, cb))
    assert np.array_equal(K.decrypt_bits(got), ~(a.astype(bool) & b.astype(bool)))
    ck.close()
