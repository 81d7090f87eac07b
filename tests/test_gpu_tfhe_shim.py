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


def test_key_swap_at_one_address_under_concurrent_callers(O):
    """A libtfhe client may delete a key set and load another one at the SAME address.  One CloudKeySet struct is overwritten in place
    (A -> B -> A) while eight threads call bootsNAND / bootsXOR through it and a straggler's 96-gate batch call on the previous key is
    still in flight; every result must equal the batch API (thfhe.CloudKey on the same device) under the key the call was issued
    against.  CPU twin of the lifetime logic under ThreadSanitizer: tests/test_keyslot.py."""
    import threading
    import time
    import thfhe
    import tfhe_structs as T
    L = thfhe.lib()
    for nm in ("bootsNAND", "bootsXOR", "thfhe_tfhe_forget_key"):
        getattr(L, nm).restype = None
    p = O.make_params("SK-128", n=40)
    tp = thfhe.make_params(**p.as_dict())
    keys = [O.SKKeys(p, 71 + q, 2.0**-25, 2.0**-15) for q in range(2)]
    imgs = [T.TfheKeyImage(p, K.bk, K.ksk) for K in keys]
    cks = [thfhe.CloudKey(tp, K.bk, K.ksk, device=0) for K in keys]
    slot = T.CloudKeySet()                       # THE address every call goes through
    ck = C.byref(slot)
    SZ = C.sizeof(T.LweSample)
    G = 32
    rng = np.random.default_rng(5)
    for phase, which in enumerate((0, 1, 0)):
        K = keys[which]
        a, b = rng.integers(0, 2, G), rng.integers(0, 2, G)
        ca, cb = K.encrypt_bits(a, 2.0**-15, 300 + phase), K.encrypt_bits(b, 2.0**-15, 400 + phase)
        # straggler: a batch call issued against the key that lives at the address NOW, still running when the address is overwritten
        prev = keys[1 - which] if phase else None
        if prev is not None:
            pa = prev.encrypt_bits(rng.integers(0, 2, 96), 2.0**-15, 500 + phase)
            spa, sba = T.make_samples(pa); spr, sbr = T.make_samples(np.zeros_like(pa))
            strag = threading.Thread(target=lambda: L.thfhe_tfhe_gate_batch(O.NAND, spr, spa, spa, None, 96, ck))
            strag.start()
            time.sleep(0.002)
        C.memmove(C.addressof(slot), C.addressof(imgs[which].cloud), C.sizeof(T.CloudKeySet))   # "delete + load at the same address"
        sa, ba = T.make_samples(ca); sb, bb = T.make_samples(cb); sr, br = T.make_samples(np.zeros_like(ca))

        def worker(tid):
            for g in range(tid, G, 8):
                f = L.bootsNAND if g % 2 == 0 else L.bootsXOR
                f(C.byref(sr, g * SZ), C.byref(sa, g * SZ), C.byref(sb, g * SZ), ck)

        ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        got = T.read_samples(sr, br)
        even, odd = np.arange(0, G, 2), np.arange(1, G, 2)
        assert np.array_equal(got[even], cks[which].gates(O.NAND, ca[even], cb[even])), f"phase {phase}"
        assert np.array_equal(got[odd], cks[which].gates(O.XOR, ca[odd], cb[odd])), f"phase {phase}"
        assert np.array_equal(K.decrypt_bits(got[even]), ~(a[even].astype(bool) & b[even].astype(bool)))
        if prev is not None:
            strag.join()
            # the straggler read the address either before or after the overwrite; its result must be a correct NAND under one of the two keys
            res = T.read_samples(spr, sbr)
            ok_prev = np.array_equal(res, cks[1 - which].gates(O.NAND, pa, pa))
            ok_new = np.array_equal(res, cks[which].gates(O.NAND, pa, pa))
            assert ok_prev or ok_new, f"phase {phase}: straggler batch matches neither key"
    L.thfhe_tfhe_forget_key(ck)
    for c in cks:
        c.close()


def test_cpp_client_links_and_matches_oracle(O, tmp_path):
    """tests/cpp/evaluate_demo.cpp: a plain g++ program written like the reference's C++ callers (Evaluate of src/Convert.cpp:28-33 from
    OpenMP threads, FullAdder of src/KNN_medical_data.cpp:134-157 with in-place carries), built against include/tfhe_shim.h and LINKED
    with libthfhe_hip.so in the place of libtfhe; outputs must equal the oracle bit for bit."""
    import os
    import subprocess
    from conftest import full_adder, threaded_multi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-B", "-C", os.path.join(root, "tests", "cpp")], check=True)
    p = O.make_params("SK-128", n=40)
    K = O.SKKeys(p, 33, 2.0**-25, 2.0**-15)
    orc = O.Oracle(p, K.bk, K.ksk)
    nbits = 8
    x, y = 0xB5, 0x6E
    bits = lambda v: [(v >> (nbits - 1 - i)) & 1 for i in range(nbits)]          # MSB first, as the reference
    c1, c2 = K.encrypt_bits(bits(x), 2.0**-15, 61), K.encrypt_bits(bits(y), 2.0**-15, 62)
    cin = K.encrypt_bits([0], 2.0**-15, 63)[0]
    hdr = np.array([p.n, p.N, p.l, p.Bgbit, p.ks_t, p.ks_basebit, nbits], np.int32)
    blob_in, blob_out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(blob_in, "wb") as f:
        for arr in (hdr, K.bk, K.ksk, c1, c2, cin):
            f.write(np.ascontiguousarray(arr, np.int32).tobytes())
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([os.path.join(root, "tests", "cpp", "evaluate_demo"), str(blob_in), str(blob_out)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.fromfile(blob_out, np.int32).reshape(3, nbits, p.n + 1)
    assert np.array_equal(out[0], orc.gates(O.AND, c1, c2))
    ref_sum, ref_carry = full_adder(threaded_multi(orc.gates), c1, c2, cin)
    assert np.array_equal(out[1], ref_sum) and np.array_equal(out[2], ref_carry)
    val = lambda rec: int("".join("1" if b else "0" for b in K.decrypt_bits(rec)), 2)
    assert val(out[0]) == (x & y) and val(out[1]) == (x + y) & 0xFF
