"""GPU parity tests (pytest -m gpu, on the MI355X box): every call goes through the C ABI of libthfhe_hip.so
and is compared bit-for-bit with the CPU oracle on the same key tables and ciphertexts; full-size batches are
checked through size-independent properties (decryption, truth-table identities, determinism)."""
import numpy as np
import pytest

from conftest import full_adder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu128(O, sk128):
    import thfhe
    p, K, orc = sk128
    ck = thfhe.CloudKey(thfhe.make_params("SK-128"), K.bk, K.ksk, device=0)
    yield ck
    ck.close()


def enc(O, K, bits, seed, name="SK-128"):
    return K.encrypt_bits(bits, O.SIGMAS[name]["lwe"], seed)


def test_library_is_native_and_sees_the_gpu():
    import thfhe
    assert thfhe.lib().thfhe_device_count() >= 1


def test_bootstrap_pieces_bit_exact(O, sk128, gpu128):
    # bootstrap_wo_keyswitch (J/bootstrap.jl:75-88), keyswitch (J/keyswitch.jl:45-80), bootstrap (:98-101)
    p, K, orc = sk128
    x = enc(O, K, [0, 1, 1, 0, 1, 0], 31)
    u_ref = np.stack([orc.bootstrap_wo_keyswitch(r) for r in x])
    assert np.array_equal(gpu128.bootstrap_wo_keyswitch(x), u_ref)
    ks_ref = np.stack([orc.keyswitch(r) for r in u_ref])
    assert np.array_equal(gpu128.keyswitch(u_ref), ks_ref)
    assert np.array_equal(gpu128.bootstrap(x), ks_ref)
    # a different output message mu                                   (bootstrap's mu argument)
    mu = 1 << 28
    assert np.array_equal(gpu128.bootstrap_wo_keyswitch(x[:2], mu), np.stack([orc.bootstrap_wo_keyswitch(r, mu) for r in x[:2]]))


def test_all_gates_bit_exact_and_truth_tables(O, sk128, gpu128):
    # 3-gen-mk-tfhe/test/runtests.jl:10-42 through the drop-in API, plus ciphertext equality with the oracle
    import thfhe
    p, K, orc = sk128
    a = np.array([0, 0, 1, 1, 1, 0]); b = np.array([0, 1, 0, 1, 1, 0])
    ca, cb = enc(O, K, a, 41), enc(O, K, b, 42)
    api = {O.NAND: thfhe.gate_nand, O.OR: thfhe.gate_or, O.AND: thfhe.gate_and, O.XOR: thfhe.gate_xor, O.XNOR: thfhe.gate_xnor,
           O.NOR: thfhe.gate_nor, O.ANDNY: thfhe.gate_andny, O.ANDYN: thfhe.gate_andyn, O.ORNY: thfhe.gate_orny, O.ORYN: thfhe.gate_oryn}
    for op, fn in api.items():
        got = fn(gpu128, ca, cb)
        assert np.array_equal(got, orc.gates(op, ca, cb)), f"gate {op}"
        assert np.array_equal(K.decrypt_bits(got), [bool(O.TRUTH[op](bool(x), bool(y))) for x, y in zip(a, b)])
    assert np.array_equal(thfhe.gate_not(gpu128, ca), orc.gates(O.NOT, ca))
    assert np.array_equal(gpu128.gates(thfhe.COPY, ca), ca)


def test_every_blind_rotate_kernel_bit_exact(O, sk128, gpu128):
    # the same 12 gates through every kernel shape: eight-wave LDS ring (both thresholds 0), four-wave LDS ring, cooperative latency kernel
    import thfhe
    p, K, orc = sk128
    a = np.array([0, 1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 0]); b = np.array([1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1])
    ca, cb = enc(O, K, a, 45), enc(O, K, b, 46)
    ref = orc.gates(O.XOR, ca, cb)
    try:
        for coop, ring4 in ((0, 0), (0, 1024), (1 << 20, 1024), (5, 6)):     # (5, 6): 6 gates on the four-wave ring + 6 cooperative (12 <= ring4 + 256)
            gpu128.set_coop_threshold(coop)
            gpu128.set_ring4_threshold(ring4)
            assert np.array_equal(gpu128.gates(thfhe.XOR, ca, cb), ref), (coop, ring4)   # partially filled workgroups in both ring shapes
    finally:
        gpu128.set_coop_threshold(768)
        gpu128.set_ring4_threshold(1024)


def test_mux_bit_exact(O, sk128, gpu128):
    import thfhe
    p, K, orc = sk128
    bits = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)])
    cx, cy, cz = (enc(O, K, bits[:, q], 50 + q) for q in range(3))
    got = thfhe.gate_mux(gpu128, cx, cy, cz)
    assert np.array_equal(got, orc.gates(O.MUX, cx, cy, cz))
    assert np.array_equal(K.decrypt_bits(got), np.where(bits[:, 0] == 1, bits[:, 1], bits[:, 2]).astype(bool))


def test_edge_cases(O, sk128, gpu128):
    import thfhe
    p, K, orc = sk128
    # empty batch
    assert gpu128.gates(thfhe.NAND, np.zeros((0, p.n + 1), np.int32), np.zeros((0, p.n + 1), np.int32)).shape == (0, p.n + 1)
    # unknown opcode
    x = enc(O, K, [1], 60)
    with pytest.raises(thfhe.ThfheError):
        gpu128.gates(77, x, x)
    # mask words that mod-switch to zero take the `bara == 0` skip of J/bootstrap.jl:40
    y = x.copy()
    y[0, : p.n : 2] = 0
    y[0, 1 : p.n : 2] = 1 << 19          # rounds to 0 mod 2N as well (2^19 * 2048 / 2^32 < 0.5)
    got = gpu128.bootstrap_wo_keyswitch(y)
    assert np.array_equal(got[0], orc.bootstrap_wo_keyswitch(y[0]))
    assert np.all(got[0, : p.N] == 0)
    # extreme words
    z = np.full((1, p.n + 1), -2**31, np.int32)
    w = np.full((1, p.n + 1), 2**31 - 1, np.int32)
    assert np.array_equal(gpu128.gates(thfhe.XOR, z, w), orc.gates(O.XOR, z, w))
    # in-place use: output array aliasing an input, as the reference's callers do (src/KNN_medical_data.cpp:256,395)
    import ctypes as C
    a = enc(O, K, [1, 0, 1], 61); b = enc(O, K, [1, 1, 0], 62)
    exp = orc.gates(O.AND, a, b)
    buf = a.copy()
    i32p = C.POINTER(C.c_int32)
    rc = thfhe.lib().thfhe_gates(gpu128.h, thfhe.AND, buf.ctypes.data_as(i32p), b.ctypes.data_as(i32p), None, buf.ctypes.data_as(i32p), 3)
    assert rc == 0 and np.array_equal(buf, exp)


def test_reference_adder_and_subtractor_on_gpu(O, sk128, gpu128):
    """Full 32-bit FullAdder / subtractor of src/bootstrap_modules.cpp:20-44,412-482 on the reference's own input
    ciphertexts (tests/golden/cloud1.data, cloud2.data, allOne.data, lsbOne.data): results must decrypt to the
    reference's sum.txt / carry.txt / diff.txt; sampled gates are compared bit-for-bit with the oracle."""
    p, K, orc = sk128
    _, c1, _ = O.load_fixture_records("cloud1.data")
    _, c2, _ = O.load_fixture_records("cloud2.data")
    _, all_one, _ = O.load_fixture_records("allOne.data")
    _, lsb_one, _ = O.load_fixture_records("lsbOne.data")
    zero = K.encrypt_bits([0], 2.0**-15, 999)[0]
    log = []

    def multi(jobs):
        outs = [gpu128.gates(op, x, y) for op, x, y in jobs]
        log.extend((op, x.copy(), y.copy(), o.copy()) for (op, x, y), o in zip(jobs, outs))
        return outs

    s, c = full_adder(multi, c1, c2, zero)
    assert O.bits_to_int_msb_first(K.decrypt_bits(s)) == 10562
    assert O.bits_to_int_msb_first(K.decrypt_bits(c)) == 3448
    ones = gpu128.gates(O.XOR, all_one, c2)
    twos, _ = full_adder(multi, ones, lsb_one, zero)
    d, _ = full_adder(multi, c1, twos, zero)
    assert O.bits_to_int_msb_first(K.decrypt_bits(d)) == 9190
    for arr in (s, d):
        assert np.abs(np.abs(K.phases(arr) / 2.0**32) - 0.125).max() < 0.03
    # bit-for-bit: the two 32-gate first levels and a sample of the ripple gates
    picks = [0, 1] + list(range(2, len(log), max(1, len(log) // 12)))
    for q in picks[:14]:
        op, x, y, o = log[q]
        assert np.array_equal(o, orc.gates(op, x, y))


def test_full_batch_4096_properties(O, sk128, gpu128):
    """BASELINE.json configs[1]: 4096 independent NANDs, device-resident records.  Size-independent checks:
    all outputs decrypt to NAND, noise inside the envelope, NOT(NAND) == AND as plaintexts, determinism
    (a second run gives the identical bytes) -- and all 4096 outputs equal the oracle bit for bit."""
    import thfhe
    p, K, orc = sk128
    B = 4096
    rng = np.random.default_rng(70)
    a, b = rng.integers(0, 2, B), rng.integers(0, 2, B)
    xa, xb = enc(O, K, a, 71), enc(O, K, b, 72)
    da, db, do = gpu128.device_records(B), gpu128.device_records(B), gpu128.device_records(B)
    da.upload(xa); db.upload(xb)
    gpu128.reserve(B)
    gpu128.gates_dev(thfhe.NAND, da, db, None, do, B); gpu128.sync()
    out1 = do.download((B, p.n + 1))
    gpu128.gates_dev(thfhe.NAND, da, db, None, do, B); gpu128.sync()
    out2 = do.download((B, p.n + 1))
    assert np.array_equal(out1, out2)
    nand = ~(a.astype(bool) & b.astype(bool))
    assert np.array_equal(K.decrypt_bits(out1), nand)
    assert np.abs(np.abs(K.phases(out1) / 2.0**32) - 0.125).max() < 0.04
    gpu128.gates_dev(thfhe.AND, da, db, None, do, B); gpu128.sync()
    assert np.array_equal(K.decrypt_bits(do.download((B, p.n + 1))), ~nand)
    # EVERY output word of the full batch against the oracle (NTT engine, OpenMP over gates on the box's CPU share)
    ref = orc.gates(O.NAND, xa, xb)
    bad = np.flatnonzero((out1 != ref).any(axis=1))
    assert bad.size == 0, f"{bad.size} of {B} gates differ from the oracle, first at {bad[:8]}"
    # host-buffer API gives the same bytes as the device-buffer API
    assert np.array_equal(gpu128.gates(thfhe.NAND, xa[:64], xb[:64]), out1[:64])
    for d in (da, db, do):
        d.free()


def _check_keyswitch(O, ck, orc, p, batch, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(-2**31, 2**31, (batch, p.N + 1), dtype=np.int64).astype(np.int32)
    u[0, :] = 0                      # every digit zero but the rounding offset's carry
    u[min(1, batch - 1), :] = -1     # all digits 3
    u[min(2, batch - 1), :] = 2**31 - 1
    got = ck.keyswitch(u)
    rows = sorted(set(list(range(min(batch, 40))) + list(range(max(0, batch - 40), batch)) + list(range(0, batch, 61))))
    for g in rows:
        assert np.array_equal(got[g], orc.keyswitch(u[g])), (batch, g)


def test_every_keyswitch_kernel_bit_exact(O, sk128, gpu128):
    """keyswitch (J/keyswitch.jl:45-80) through all three kernels: one gate per workgroup (< 192 gates), rows staged in LDS for 32 gates with the
    digit selecting an address (from 192 gates on; 16 coordinate ranges, 8 from 2 048 gates on), also with a ragged last workgroup, and the
    two-rotation input of the MUX epilogue (J/gates.jl:172-176) at a staged batch size."""
    import thfhe
    p, K, orc = sk128
    for batch, seed in ((1, 1), (191, 2), (192, 3), (223, 4), (1000, 5), (2051, 6)):
        _check_keyswitch(O, gpu128, orc, p, batch, seed)
    B = 200
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2, (3, B))
    cx, cy, cz = (enc(O, K, bits[q], 300 + q) for q in range(3))
    got = thfhe.gate_mux(gpu128, cx, cy, cz)
    assert np.array_equal(got, orc.gates(O.MUX, cx, cy, cz))
    assert np.array_equal(K.decrypt_bits(got), np.where(bits[0] == 1, bits[1], bits[2]).astype(bool))


@pytest.mark.parametrize("name,kw", [("SK-80", {}), ("SK-128", dict(ks_t=4)), ("SK-128", dict(ks_t=6)), ("SK-lib", {})])
def test_keyswitch_shapes(O, name, kw):
    """the staged kernel's other row length (n = 500: 8 words per lane), a key-switch depth of 4, and the shapes it hands to the other kernels
    (t = 6: not a multiple of 4; n = 1024: 18 words per lane)"""
    import thfhe
    p = O.make_params(name, **kw)
    s = O.SIGMAS[name]
    K = O.SKKeys(p, 83, s["bk"], s["ks"])
    orc = O.Oracle(p, K.bk, K.ksk)
    ck = thfhe.CloudKey(thfhe.make_params(name, **kw), K.bk, K.ksk, device=0)
    for batch, seed in ((37, 11), (230, 12), (1030, 13)):
        _check_keyswitch(O, ck, orc, p, batch, seed)
    ck.close()


@pytest.mark.parametrize("name", ["SK-80", "SK-lib"])
def test_other_parameter_sets(O, name):
    # J/api.jl:76-91 (l = 2, Bgbit = 10, n = 500) and src/libthfhe.cpp:316-338 (n = 1024)
    import thfhe
    p = O.make_params(name)
    s = O.SIGMAS[name]
    K = O.SKKeys(p, 81, s["bk"], s["ks"])
    orc = O.Oracle(p, K.bk, K.ksk)
    ck = thfhe.CloudKey(thfhe.make_params(name), K.bk, K.ksk, device=0)
    a = np.array([0, 1, 1, 0]); b = np.array([1, 1, 0, 0]); c = np.array([1, 0, 1, 0])
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 90 + q) for q, v in enumerate((a, b, c)))
    for op in (O.NAND, O.XOR, O.ORYN):
        got = ck.gates(op, ca, cb)
        assert np.array_equal(got, orc.gates(op, ca, cb))
        assert np.array_equal(K.decrypt_bits(got), [bool(O.TRUTH[op](bool(x), bool(y))) for x, y in zip(a, b)])
    got = ck.gates(O.MUX, ca, cb, cc)
    assert np.array_equal(got, orc.gates(O.MUX, ca, cb, cc))
    ck.close()


def test_product_keygen_matches_oracle_semantics(O):
    # keys made by the product's host keygen (torus-fhe_amd/thfhe/keygen.py) drive GPU and oracle identically
    import thfhe
    from thfhe import keygen
    p = thfhe.make_params("SK-128")
    K = keygen.SecretKeySet(p, seed=5)
    ck = thfhe.CloudKey(p, K.bk, K.ksk, device=0)
    orc = O.Oracle(O.make_params("SK-128"), K.bk, K.ksk)
    a = np.array([0, 1, 1, 0]); b = np.array([1, 1, 0, 0])
    ca, cb = K.encrypt(a, 1), K.encrypt(b, 2)
    got = ck.gates(thfhe.NAND, ca, cb)
    assert np.array_equal(got, orc.gates(O.NAND, ca, cb))
    assert np.array_equal(K.decrypt(got), ~(a.astype(bool) & b.astype(bool)))
    ck.close()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()
