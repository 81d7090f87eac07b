"""The libtfhe-named entry points (include/tfhe_shim.h) called exactly as the reference's C++ programs call them
(bootsXOR(&result[i], &a[i], &b[i], cloud_key), src/KNN_medical_data.cpp:142-151): a TFheGateBootstrappingCloudKeySet
pointer graph is laid out in host memory with libtfhe's struct layouts and the shim must read it, bootstrap on the GPU
and write the LweSample back -- bit-exact with the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_boots_symbols_on_libtfhe_structs(O):
    import thfhe
    import tfhe_structs as T
    L = thfhe.lib()
    p = O.make_params("SK-128", n=40)      # struct marshalling is independent of n; keeps the host-side key image small
    K = O.SKKeys(p, 31, 2.0**-25, 2.0**-15)
    orc = O.Oracle(p, K.bk, K.ksk)
    img = T.TfheKeyImage(p, K.bk, K.ksk)
    a = np.array([0, 0, 1, 1]); b = np.array([0, 1, 0, 1]); c = np.array([1, 0, 0, 1])
    ca, cb, cc = (K.encrypt_bits(v, 2.0**-15, 10 + q) for q, v in enumerate((a, b, c)))
    ck = C.byref(img.cloud)
    for name, op in (("bootsNAND", O.NAND), ("bootsAND", O.AND), ("bootsOR", O.OR), ("bootsXOR", O.XOR), ("bootsXNOR", O.XNOR),
                     ("bootsNOR", O.NOR), ("bootsANDNY", O.ANDNY), ("bootsANDYN", O.ANDYN), ("bootsORNY", O.ORNY), ("bootsORYN", O.ORYN)):
        f = getattr(L, name)
        f.restype = None
        sa, ba = T.make_samples(ca); sb, bb = T.make_samples(cb); sr, br = T.make_samples(np.zeros_like(ca))
        for g in range(4):
            f(C.byref(sr, g * C.sizeof(T.LweSample)), C.byref(sa, g * C.sizeof(T.LweSample)), C.byref(sb, g * C.sizeof(T.LweSample)), ck)
        got = T.read_samples(sr, br)
        assert np.array_equal(got, orc.gates(op, ca, cb)), name
        assert np.array_equal(K.decrypt_bits(got), [bool(O.TRUTH[op](bool(x), bool(y))) for x, y in zip(a, b)])
    # bootsMUX(result, a, b, c, bk) and in-place result (src/KNN_medical_data.cpp:227,256)
    sa, ba = T.make_samples(ca); sb, bb = T.make_samples(cb); sc, bc = T.make_samples(cc)
    L.bootsMUX.restype = None
    for g in range(4):
        L.bootsMUX(C.byref(sa, g * C.sizeof(T.LweSample)), C.byref(sa, g * C.sizeof(T.LweSample)), C.byref(sb, g * C.sizeof(T.LweSample)),
                   C.byref(sc, g * C.sizeof(T.LweSample)), ck)
    assert np.array_equal(T.read_samples(sa, ba), orc.gates(O.MUX, ca, cb, cc))
    # bootsNOT / bootsCOPY / bootsCONSTANT
    sa, ba = T.make_samples(ca); sr, br = T.make_samples(np.zeros_like(ca))
    L.bootsNOT.restype = None
    L.bootsNOT(sr, sa, ck)
    assert np.array_equal(T.read_samples(sr, br)[0], orc.gates(O.NOT, ca[:1])[0])
    L.bootsCONSTANT.restype = None
    L.bootsCONSTANT(sr, 1, ck)
    assert sr[0].b == 1 << 29 and not br[0].any()
    # batched helper
    sa, ba = T.make_samples(ca); sb, bb = T.make_samples(cb); sr, br = T.make_samples(np.zeros_like(ca))
    assert L.thfhe_tfhe_gate_batch(O.XOR, sr, sa, sb, None, 4, ck) == 0
    assert np.array_equal(T.read_samples(sr, br), orc.gates(O.XOR, ca, cb))
    L.thfhe_tfhe_forget_key.restype = None
    L.thfhe_tfhe_forget_key(ck)


def test_concurrent_boots_calls_are_batched_and_correct(O):
    """Eight host threads call bootsXOR / bootsAND / bootsMUX concurrently on a shared key set (the reference's OpenMP pattern,
    src/KNN_medical_data.cpp:681-691); the shim's combining batcher evaluates queued calls together; results stay bit-exact."""
    import threading
    import thfhe
    import tfhe_structs as T
    L = thfhe.lib()
    p = O.make_params("SK-128", n=40)
    K = O.SKKeys(p, 32, 2.0**-25, 2.0**-15)
    orc = O.Oracle(p, K.bk, K.ksk)
    img = T.TfheKeyImage(p, K.bk, K.ksk)
    ck = C.byref(img.cloud)
    G = 24
    rng = np.random.default_rng(3)
    a, b, c = (rng.integers(0, 2, G) for _ in range(3))
    ca, cb, cc = (K.encrypt_bits(v, 2.0**-15, 40 + q) for q, v in enumerate((a, b, c)))
    sa, ba = T.make_samples(ca); sb, bb = T.make_samples(cb); sc, bc = T.make_samples(cc); sr, br = T.make_samples(np.zeros_like(ca))
    ops = [O.XOR, O.AND, O.MUX]
    names = {O.XOR: "bootsXOR", O.AND: "bootsAND", O.MUX: "bootsMUX"}
    for nm in names.values():
        getattr(L, nm).restype = None
    SZ = C.sizeof(T.LweSample)

    def worker(tid):
        for g in range(tid, G, 8):
            op = ops[g % 3]
            f = getattr(L, names[op])
            if op == O.MUX:
                f(C.byref(sr, g * SZ), C.byref(sa, g * SZ), C.byref(sb, g * SZ), C.byref(sc, g * SZ), ck)
            else:
                f(C.byref(sr, g * SZ), C.byref(sa, g * SZ), C.byref(sb, g * SZ), ck)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    got = T.read_samples(sr, br)
    for op in ops:
        idx = [g for g in range(G) if ops[g % 3] == op]
        ref = orc.gates(op, ca[idx], cb[idx], cc[idx] if op == O.MUX else None)
        assert np.array_equal(got[idx], ref), names[op]
    L.thfhe_tfhe_forget_key.restype = None
    L.thfhe_tfhe_forget_key(ck)
