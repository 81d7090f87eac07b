"""LWE -> TLWE conversion and threshold partial / final decryption (src/libthfhe.cpp:270-348).
CPU: the oracle restatement against the defining property (phase of the ring sample's constant coefficient = LWE phase) and
an additive t-out-of-t sharing.  GPU (-m gpu): thfhe_tlwe_from_lwe / thfhe_partial_decrypt / thfhe_final_decrypt bit for bit
against the oracle, and the reference's application flow gate -> conversion -> partial decryptions -> final decryption."""
import numpy as np
import pytest

N = 1024


def lwe_encrypt(rng, key, bits, sigma=2.0**-15):
    a = rng.integers(-2**31, 2**31, size=(len(bits), N), dtype=np.int64)
    e = np.trunc(rng.standard_normal(len(bits)) * sigma * 2.0**32).astype(np.int64)
    mu = np.where(np.asarray(bits, bool), 1 << 29, -(1 << 29)).astype(np.int64)
    b = mu + e + (a * key.astype(np.int64)).sum(axis=1)
    out = np.empty((len(bits), N + 1), np.int32)
    out[:, :N] = a.astype(np.int32)
    out[:, N] = b.astype(np.uint32).view(np.int32)
    return out


def additive_shares(rng, key, t):
    """finalDecrypt computes b - partial_0 + sum_{i>=1} partial_i, so the shares must satisfy key = s_0 - s_1 - ... - s_{t-1}."""
    others = [rng.integers(-3, 4, N).astype(np.int32) for _ in range(t - 1)]
    s0 = key.astype(np.int32) + sum(others, np.zeros(N, np.int32))
    return [s0] + others


def oracle_flow(O, lwe, shares, noises):
    L = O.lib()
    bits = []
    for c in range(lwe.shape[0]):
        ta, tb = np.zeros(N, np.int32), np.zeros(N, np.int32)
        L.oracle_tlwe_from_lwe(O.p32(np.ascontiguousarray(lwe[c])), N, O.p32(ta), O.p32(tb))
        parts = np.zeros((len(shares), N), np.int32)
        for i, s in enumerate(shares):
            e = np.ascontiguousarray(noises[i][c]) if noises is not None else None
            L.oracle_partial_decrypt(O.p32(s), O.p32(ta), O.p32(e), N, O.p32(parts[i]))
        bits.append(L.oracle_final_decrypt(O.p32(tb), O.p32(parts), len(shares), N, None))
    return np.array(bits, bool)


def test_oracle_conversion_and_threshold_flow(O):
    rng = np.random.default_rng(0)
    key = rng.integers(0, 2, N).astype(np.int32)
    bits = rng.integers(0, 2, 6)
    lwe = lwe_encrypt(rng, key, bits)
    # defining property of TLweFromLwe: (b' - key (*) a')[0] = b - <a, key>
    L = O.lib()
    ta, tb, prod = np.zeros(N, np.int32), np.zeros(N, np.int32), np.zeros(N, np.int32)
    L.oracle_tlwe_from_lwe(O.p32(np.ascontiguousarray(lwe[0])), N, O.p32(ta), O.p32(tb))
    L.oracle_polymul_schoolbook32(O.p32(key), O.p32(ta), N, O.p32(prod))
    phase = np.int64(lwe[0, N]) - (lwe[0, :N].astype(np.int64) * key).sum()
    assert np.uint32(np.int64(tb[0]) - np.int64(prod[0])) == np.uint32(phase)
    for t in (1, 2, 3, 5):
        shares = additive_shares(rng, key, t)
        noises = [np.trunc(rng.standard_normal((len(bits), N)) * 2.0**-20 * 2.0**32).astype(np.int32) for _ in range(t)]
        assert np.array_equal(oracle_flow(O, lwe, shares, noises), bits.astype(bool)), t


@pytest.mark.gpu
def test_gpu_threshold_ops_bit_exact(O):
    import thfhe
    from thfhe import threshold as T
    ctx = T.PolyContext(0)
    rng = np.random.default_rng(1)
    L = O.lib()
    cnt = 37
    lwe = rng.integers(-2**31, 2**31, size=(cnt, N + 1), dtype=np.int64).astype(np.int32)
    ta, tb = T.TLweFromLwe(ctx, lwe)
    ra, rb = np.zeros_like(ta), np.zeros_like(tb)
    for c in range(cnt):
        L.oracle_tlwe_from_lwe(O.p32(np.ascontiguousarray(lwe[c])), N, O.p32(ra[c]), O.p32(rb[c]))
    assert np.array_equal(ta, ra) and np.array_equal(tb, rb)
    for name, share in (("binary", rng.integers(0, 2, N)), ("ternary", rng.integers(-1, 2, N)), ("max", rng.integers(-512, 513, N)),
                        ("worst", np.where(np.arange(N) % 2 == 0, 512, -512))):
        share = share.astype(np.int32)
        noise = rng.integers(-2**20, 2**20, size=(cnt, N)).astype(np.int32) if name != "binary" else None
        got = T.PartialDecrypt(ctx, share, ta, noise)
        ref = np.zeros_like(got)
        for c in range(cnt):
            e = np.ascontiguousarray(noise[c]) if noise is not None else None
            L.oracle_partial_decrypt(O.p32(share), O.p32(np.ascontiguousarray(ta[c])), O.p32(e), N, O.p32(ref[c]))
        assert np.array_equal(got, ref), name
    worst_a = np.full((2, N), -2**31, np.int32)       # largest limbs everywhere
    share = np.full(N, -512, np.int32)
    ref = np.zeros_like(worst_a)
    for c in range(2):
        L.oracle_partial_decrypt(O.p32(share), O.p32(worst_a[c]), None, N, O.p32(ref[c]))
    assert np.array_equal(T.PartialDecrypt(ctx, share, worst_a), ref)
    parts = rng.integers(-2**31, 2**31, size=(4, cnt, N), dtype=np.int64).astype(np.int32)
    bits, res = T.finalDecrypt(ctx, tb, parts, want_result=True)
    for c in range(cnt):
        r = np.zeros(N, np.int32)
        bit = L.oracle_final_decrypt(O.p32(np.ascontiguousarray(tb[c])), O.p32(np.ascontiguousarray(parts[:, c])), 4, N, O.p32(r))
        assert bool(bit) == bool(bits[c]) and np.array_equal(r, res[c])
    with pytest.raises(thfhe.ThfheError):
        T.PartialDecrypt(ctx, np.full(N, 513, np.int32), ta)
    assert T.PartialDecrypt(ctx, share, ta[:0]).shape == (0, N)
    ctx.close()


@pytest.mark.gpu
def test_gate_then_threshold_decryption_flow(O):
    # src/KNN_medical_data.cpp:722-745: a bootstrapped result bit (libthfhe parameters, n = N = 1024, src/libthfhe.cpp:316-338)
    # is converted to a ring sample and decrypted by t parties holding shares of the key
    import thfhe
    from thfhe import threshold as T
    p = O.make_params("SK-lib")
    s = O.SIGMAS["SK-lib"]
    K = O.SKKeys(p, 21, s["bk"], s["ks"])
    ck = thfhe.CloudKey(thfhe.make_params("SK-lib"), K.bk, K.ksk, device=0)
    a = np.array([0, 0, 1, 1, 1, 0, 1, 0]); b = np.array([0, 1, 0, 1, 1, 0, 0, 1])
    ca, cb = K.encrypt_bits(a, s["lwe"], 1), K.encrypt_bits(b, s["lwe"], 2)
    out = thfhe.gate_nand(ck, ca, cb)
    orc = O.Oracle(p, K.bk, K.ksk)
    assert np.array_equal(out[:2], orc.gates(O.NAND, ca[:2], cb[:2]))
    ctx = T.PolyContext(0)
    ta, tb = T.TLweFromLwe(ctx, out)
    rng = np.random.default_rng(3)
    for t in (2, 3):
        shares = additive_shares(rng, K.lwe_key, t)
        noises = [np.trunc(rng.standard_normal((len(a), N)) * 2.0**-20 * 2.0**32).astype(np.int32) for _ in range(t)]
        parts = np.stack([T.PartialDecrypt(ctx, shares[i], ta, noises[i]) for i in range(t)])
        bits = T.finalDecrypt(ctx, tb, parts)
        assert np.array_equal(bits, ~(a.astype(bool) & b.astype(bool)))
        assert np.array_equal(bits, oracle_flow(O, out, shares, noises))
    ctx.close(); ck.close()
