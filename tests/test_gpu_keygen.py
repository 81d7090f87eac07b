"""Multi-key key generation on the device: thfhe_pm_mac, the exact small x torus multiply-accumulate behind
tgsw_encrypt_3gen (J/tgsw_3gen.jl:41-95), PublicKey / CommonPubKey_3gen (J/mk_internals.jl:266-345) and the CCS mk_tgsw_encrypt
(J/mk_internals.jl:390-446).  The randomness is an input, so the device result must equal the oracle's exact products and the host
(numpy) key generation bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def schoolbook(O, small, torus, bits):
    out = np.zeros(small.shape[0], np.int32 if bits == 32 else np.int64)
    N = small.shape[0]
    if bits == 32:
        O.lib().oracle_polymul_schoolbook32(O.p32(small.astype(np.int32)), O.p32(torus), N, O.p32(out))
    else:
        O.lib().oracle_polymul_schoolbook64(O.p64(small.astype(np.int64)), O.p64(torus), N, O.p64(out))
    return out


@pytest.mark.parametrize("N,bits", [(1024, 32), (1024, 64), (2048, 64)])
def test_polymac_equals_schoolbook(O, N, bits):
    import thfhe
    rng = np.random.default_rng(N + bits)
    dt = np.int32 if bits == 32 else np.int64
    lim = 2 ** (bits - 1)
    small = np.stack([rng.integers(-4096, 4097, N), rng.integers(-1, 2, N), rng.integers(0, 2, N),
                      np.where(rng.integers(0, 2, N) == 1, 4096, -4096)]).astype(np.int32)          # 13-bit digits, ternary, binary, extreme
    torus = np.stack([rng.integers(-lim, lim, N, dtype=np.int64), np.where(rng.integers(0, 2, N) == 1, lim - 1, -lim)]).astype(dt)
    addend = rng.integers(-lim, lim, (3, N), dtype=np.int64).astype(dt)
    terms = [(0, 0, 0, 1), (0, 1, 1, -1), (0, 3, 1, 1), (1, 2, 0, 1), (2, 3, 1, -1)]   # output 0: three terms with signs; 1, 2: one term each
    pm = thfhe.PolyMac(N, bits, 0)
    got = pm.mac(small, torus, terms, 3, addend)
    exp = addend.astype(np.uint64 if bits == 64 else np.uint32).copy()
    for j, s, t, sg in terms:
        pr = schoolbook(O, small[s], torus[t], bits).astype(exp.dtype)
        exp[j] = exp[j] + pr if sg > 0 else exp[j] - pr
    assert np.array_equal(got.view(exp.dtype), exp)
    # no addend, an output without terms
    got = pm.mac(small, torus, [(1, 1, 0, 1)], 2)
    assert not got[0].any() and np.array_equal(got[1], schoolbook(O, small[1], torus[0], bits))
    # a coefficient outside the bound is refused, not silently mis-multiplied
    bad = small.copy(); bad[0, 5] = 5000
    with pytest.raises(thfhe.ThfheError):
        pm.mac(bad, torus, [(0, 0, 0, 1)], 1)
    with pytest.raises(thfhe.ThfheError):
        pm.mac(small, torus, [(1, 0, 0, 1), (0, 0, 0, 1)], 2)   # outputs must ascend
    pm.close()


@pytest.mark.parametrize("name,over", [("MK2", dict(n=6)), ("MK4", dict(n=3)), ("MK4-N2048", dict(n=3, parties=2))])
def test_3gen_keygen_on_device_equals_host(name, over):
    import thfhe
    from thfhe import keygen
    p = thfhe.make_params(name, **over)
    host = keygen.MKSecretKeySet(p, seed=41)
    dev = keygen.MKSecretKeySet(p, seed=41, device=0)
    assert np.array_equal(host.bk, dev.bk) and np.array_equal(host.ksk, dev.ksk)
    assert np.array_equal(host.lwe_keys, dev.lwe_keys) and np.array_equal(host.rlwe_keys, dev.rlwe_keys)
    # and the device-made key bootstraps: one NAND truth table through the engine
    ck = thfhe.MKCloudKey(p, dev.bk, dev.ksk, device=0)
    a, b = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1])
    out = ck.gates(thfhe.NAND, dev.encrypt(a, 5), dev.encrypt(b, 6))
    assert np.array_equal(dev.decrypt(out), ~(a.astype(bool) & b.astype(bool)))
    ck.close()


def test_ccs_keygen_on_device_equals_host():
    import thfhe
    from thfhe import keygen
    p = thfhe.make_params("CCS2", n=5)
    host = keygen.CCSSecretKeySet(p, seed=43)
    dev = keygen.CCSSecretKeySet(p, seed=43, device=0)
    for f in ("bk", "pk", "crs", "ksk", "lwe_keys", "rlwe_keys"):
        assert np.array_equal(getattr(host, f), getattr(dev, f)), f
    ck = thfhe.CCSCloudKey(p, dev.bk, dev.pk, dev.crs, dev.ksk, device=0)
    a, b = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1])
    out = ck.gates(thfhe.NAND, dev.encrypt(a, 5), dev.encrypt(b, 6))
    assert np.array_equal(dev.decrypt(out), ~(a.astype(bool) & b.astype(bool)))
    ck.close()
