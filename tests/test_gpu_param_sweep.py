"""GPU parity over a sweep of parameter shapes (small n keeps the oracle fast): decomposition lengths 1..4, gadget bases up to
2^10, several key-switch shapes, 1..4 parties -- every shape through both the cooperative and the ring kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SK_SHAPES = [  # (n, l, Bgbit, ks_t, ks_basebit)
    (24, 1, 8, 8, 2), (24, 2, 10, 8, 2), (37, 3, 7, 8, 2), (16, 4, 8, 5, 3), (33, 3, 6, 3, 5), (1, 2, 7, 15, 1), (64, 4, 4, 4, 4),
]
MK_SHAPES = [  # (parties, n, l, Bgbit, ks_t, ks_basebit)
    (1, 12, 2, 7, 3, 3), (2, 10, 2, 7, 3, 3), (3, 9, 2, 7, 5, 2), (4, 7, 3, 6, 5, 2), (2, 8, 4, 4, 5, 2), (2, 11, 1, 9, 4, 4),
]


@pytest.mark.parametrize("shape", SK_SHAPES)
def test_single_key_shapes(O, shape):
    import thfhe
    n, l, Bgbit, t, bb = shape
    kw = dict(n=n, N=1024, k=1, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, torus_bits=32, parties=1)
    p = O.make_params(**kw)
    K = O.SKKeys(p, 1000 + n, 2.0**-25, 2.0**-15)
    orc = O.Oracle(p, K.bk, K.ksk)
    ck = thfhe.CloudKey(thfhe.make_params(**kw), K.bk, K.ksk, device=0)
    rng = np.random.default_rng(n)
    G = 11
    a, b, c = (rng.integers(0, 2, G) for _ in range(3))
    ca, cb, cc = (K.encrypt_bits(v, 2.0**-15, 5 + q) for q, v in enumerate((a, b, c)))
    ref_x, ref_m = orc.gates(O.XNOR, ca, cb), orc.gates(O.MUX, ca, cb, cc)
    for thr, ring4 in ((0, 0), (0, 1024), (1 << 20, 1024)):      # eight-wave ring, four-wave ring, cooperative kernel
        ck.set_coop_threshold(thr)
        ck.set_ring4_threshold(ring4)
        assert np.array_equal(ck.gates(thfhe.XNOR, ca, cb), ref_x), (shape, thr)
        assert np.array_equal(ck.gates(thfhe.MUX, ca, cb, cc), ref_m), (shape, thr)
    ops = rng.integers(0, 10, G).astype(np.int32)
    got = ck.gates_mixed(ops, ca, cb)
    for op in set(ops.tolist()):
        idx = np.nonzero(ops == op)[0]
        assert np.array_equal(got[idx], orc.gates(int(op), ca[idx], cb[idx]))
    ck.close()


@pytest.mark.parametrize("shape", MK_SHAPES)
def test_multi_key_shapes(O, shape, monkeypatch):
    import thfhe
    P, n, l, Bgbit, t, bb = shape
    kw = dict(n=n, N=1024, k=1, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, torus_bits=64, parties=P)
    p = O.make_params(**kw)
    K = O.MKKeys(p, 2000 + n, 2.0**-30.70, 2.0**-13.52)
    orc = O.MKOracle(p, K.bk, K.ksk)
    ck = thfhe.MKCloudKey(thfhe.make_params(**kw), K.bk, K.ksk, device=0)
    rng = np.random.default_rng(n)
    G = 5
    a, b, c = (rng.integers(0, 2, G) for _ in range(3))
    ca, cb, cc = (K.encrypt_bits(v, 2.0**-13.52, 5 + q) for q, v in enumerate((a, b, c)))
    assert np.array_equal(ck.gates(thfhe.XOR, ca, cb), orc.gates(O.XOR, ca, cb)), shape
    assert np.array_equal(ck.gates(thfhe.AND3, ca, cb, cc), orc.gates(O.AND3, ca, cb, cc)), shape
    assert np.array_equal(ck.gates(thfhe.MUX, ca, cb, cc), orc.gates(O.MUX, ca, cb, cc)), shape
    ck.close()


@pytest.mark.parametrize("name", ["SK-80", "SK-128", "SK-lib"])
def test_multi_gate_keyswitch_kernel(O, name):
    # sk_keyswitch_multi_kernel (batches >= 1024 gates, basebit 2; 8 / 4 gates per workgroup share every row load, coordinate range
    # cut in four, atomics): every output word against the oracle for a ragged batch, and the MUX combine (two rotations per gate)
    import thfhe
    p = O.make_params(name)
    s = O.SIGMAS[name]
    K = O.SKKeys(p, 31, s["bk"], s["ks"])
    orc = O.Oracle(p, K.bk, K.ksk)
    ck = thfhe.CloudKey(thfhe.make_params(name), K.bk, K.ksk, device=0)
    rng = np.random.default_rng(17)
    B = 1030 + 3
    u = rng.integers(-2**31, 2**31, size=(B, p.N + 1), dtype=np.int64).astype(np.int32)
    u[5, :p.N] = 0                                   # all digits zero except the rounding offset
    got = ck.keyswitch(u)
    idx = np.r_[0:40, 5, B - 9:B]                    # first workgroups, the ragged tail
    assert np.array_equal(got[idx], np.stack([orc.keyswitch(u[i]) for i in idx]))
    assert np.array_equal(got, ck.keyswitch(u))      # atomics: order-independent integer adds
    a, b, c = (rng.integers(0, 2, 1024) for _ in range(3))
    ca, cb, cc = (K.encrypt_bits(v, s["lwe"], 80 + q) for q, v in enumerate((a, b, c)))
    out = ck.gates(thfhe.MUX, ca, cb, cc)
    assert np.array_equal(K.decrypt_bits(out), np.where(a == 1, b, c).astype(bool))
    pick = [0, 511, 1023]
    assert np.array_equal(out[pick], orc.gates(O.MUX, ca[pick], cb[pick], cc[pick]))
    ck.close()
