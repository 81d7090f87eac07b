"""Gate-DAG front end (torus-fhe_amd/thfhe/circuits.py): the reference's KNN building blocks as static gate lists.
CPU part: gate census against SURVEY.md appendix D / src/KNN_medical_data.cpp, plaintext simulation of the wiring,
level structure.  GPU part: the circuits on the reference's fixture ciphertexts."""
import numpy as np
import pytest


def bits_msb(v, nb=32):
    return [(v >> (nb - 1 - i)) & 1 for i in range(nb)]


def from_bits(b):
    v = 0
    for x in b:
        v = (v << 1) | int(bool(x))
    return v


def build_distance():
    from thfhe import circuits as Cc
    cir = Cc.Circuit()
    x, y, all_one, lsb_one = (cir.inputs(32) for _ in range(4))
    zero = cir.inputs(1)[0]
    out = Cc.distance(cir, x, y, all_one, lsb_one, zero)
    return cir, out


def test_census_matches_reference_structure():
    from thfhe import circuits as Cc
    import thfhe
    cir = Cc.Circuit()
    a, b = cir.inputs(32), cir.inputs(32)
    zero = cir.inputs(1)[0]
    Cc.full_adder(cir, a, b, zero)
    ops = [g[0] for g in cir.gates]
    assert (ops.count(thfhe.XOR), ops.count(thfhe.AND), ops.count(thfhe.OR)) == (64, 64, 31)      # FullAdder(32) = 159 gates
    cir, _ = build_distance()
    cs = cir.census()
    assert cs["gates"] == 2 * (32 + 2 * 159) + 32 and cs["mux"] == 32 and cs["rotations"] == 700 + 64   # distance = 2 difference + 32 MUX
    # ripple carry costs 2 dependent gate levels per bit (AND, OR); the second adder pipelines behind the first, so the
    # ASAP depth is ~2*32 + a few, against 732 sequential boots* calls in the reference
    assert 64 <= cs["depth"] <= 72
    widths = [len(l) for l in cir.levels()]
    assert max(widths) >= 128       # the independent first levels of both differences are batched together


def test_plaintext_simulation_of_reference_circuits():
    from thfhe import circuits as Cc
    rng = np.random.default_rng(0)
    cir, out = build_distance()
    for xa, ya in [(9876, 686), (686, 9876), (0, 0), (123456789, 987654321), (2**31 - 1, 1)] + [tuple(int(v) for v in rng.integers(0, 2**31, 2)) for _ in range(5)]:
        bits = bits_msb(xa) + bits_msb(ya) + [1] * 32 + bits_msb(1) + [0]
        v = Cc.simulate(cir, bits)
        assert from_bits(v[out]) == abs(xa - ya)
    # distance_bw_data over 3 columns (column 0 is skipped by the reference, :257)
    cir = Cc.Circuit()
    ra = [cir.inputs(16) for _ in range(3)]
    rb = [cir.inputs(16) for _ in range(3)]
    all_zero, all_one, lsb_one = cir.inputs(16), cir.inputs(16), cir.inputs(16)
    zero = cir.inputs(1)[0]
    res = Cc.distance_bw_data(cir, ra, rb, all_zero, all_one, lsb_one, zero)
    va, vb = [7, 1000, 30], [9, 400, 75]
    bits = sum((bits_msb(x, 16) for x in va), []) + sum((bits_msb(x, 16) for x in vb), []) + [0] * 16 + [1] * 16 + bits_msb(1, 16) + [0]
    assert from_bits(Cc.simulate(cir, bits)[res]) == abs(1000 - 400) + abs(30 - 75)


@pytest.mark.gpu
def test_distance_circuit_on_reference_ciphertexts(O, sk128):
    """distance(cloud1, cloud2) (src/KNN_medical_data.cpp:217-236) on the reference's fixture ciphertexts and constants
    (allOne.data, lsbOne.data): |9876 - 686| = 9190 = diff.txt; levelised evaluation; sampled gates bit-exact vs the oracle."""
    import thfhe
    from thfhe import circuits as Cc
    p, K, orc = sk128
    ck = thfhe.CloudKey(thfhe.make_params("SK-128"), K.bk, K.ksk, device=0)
    _, c1, _ = O.load_fixture_records("cloud1.data")
    _, c2, _ = O.load_fixture_records("cloud2.data")
    _, all_one, _ = O.load_fixture_records("allOne.data")
    _, lsb_one, _ = O.load_fixture_records("lsbOne.data")
    _, lsb_zero, _ = O.load_fixture_records("lsbZero.data")
    cir, out = build_distance()
    inputs = np.concatenate([c1, c2, all_one, lsb_one, lsb_zero[31:32]])     # carry-in = LSB of lsbZero.data = Enc(0)
    stats = {}
    vals = Cc.evaluate(ck, cir, inputs, stats)
    assert O.bits_to_int_msb_first(K.decrypt_bits(vals[out])) == 9190
    assert stats["launches"] < stats["gates"] / 3        # batching really happened
    rng = np.random.default_rng(1)
    picks = rng.choice(len(cir.gates), 24, replace=False)
    for op in set(cir.gates[g][0] for g in picks):
        gs = [g for g in picks if cir.gates[g][0] == op]
        a = vals[[cir.gates[g][1] for g in gs]]; b = vals[[cir.gates[g][2] for g in gs]]
        c = vals[[cir.gates[g][3] for g in gs]] if op == thfhe.MUX else None
        assert np.array_equal(vals[cir.n_inputs + np.array(gs)], orc.gates(op, a, b, c))
    ck.close()


def test_mk_integer_circuits_plaintext():
    from thfhe import circuits as Cc
    cir = Cc.Circuit()
    a, b = cir.inputs(8), cir.inputs(8)
    zero, one = cir.inputs(2)
    s = Cc.mk_add_3gen(cir, a, b, zero)
    d = Cc.mk_sub_3gen(cir, a, b, one)
    lt = Cc.mk_less_3gen(cir, a, b, one)
    lsb = lambda v: [(v >> i) & 1 for i in range(8)]
    val = lambda w, v: sum(int(v[x]) << i for i, x in enumerate(w))
    for x, y in [(3, 9), (10, 10), (7, 1), (100, 27), (1, 2)]:
        v = Cc.simulate(cir, lsb(x) + lsb(y) + [0, 1])
        assert val(s, v) == (x + y) % 256 and val(d, v) == (x - y) % 256 and bool(v[lt]) == (((x - y) % 256) >= 128)


@pytest.mark.gpu
def test_mk_adder_demo_on_gpu(O):
    """The reference's multi-key demo (3-gen-mk-tfhe/multikey_3gen.jl:66-92): two parties, 8-bit encrypted integers,
    mk_add_3gen_v2 -- as one levelised DAG on the GPU (thfhe_mk_gates_mixed)."""
    import thfhe
    from thfhe import circuits as Cc
    p = O.make_params("MK2")
    sg = O.SIGMAS["MK2"]
    K = O.MKKeys(p, 0x5EED0001, sg["bk"], sg["ks"])
    ck = thfhe.MKCloudKey(thfhe.make_params("MK2"), K.bk, K.ksk, device=0)
    cir = Cc.Circuit()
    a, b = cir.inputs(8), cir.inputs(8)
    zero = cir.inputs(1)[0]
    s = Cc.mk_add_3gen(cir, a, b, zero)
    rng = np.random.default_rng(9)
    for trial in range(3):
        m1, m2 = int(rng.integers(1, 11)), int(rng.integers(1, 11))
        bits = [(m1 >> i) & 1 for i in range(8)] + [(m2 >> i) & 1 for i in range(8)] + [0]
        vals = Cc.evaluate(ck, cir, K.encrypt_bits(bits, sg["lwe"], 60 + trial))
        out = K.decrypt_bits(vals[s])
        assert sum(int(x) << i for i, x in enumerate(out)) == m1 + m2
    ck.close()


def build_knn(nb, ncol, ntrain):
    from thfhe import circuits as Cc
    cir = Cc.Circuit()
    test_row = [cir.inputs(nb) for _ in range(ncol)]
    train = [[cir.inputs(nb) for _ in range(ncol)] for _ in range(ntrain)]
    threshold, all_zero, all_one, lsb_one = (cir.inputs(nb) for _ in range(4))
    zero = cir.inputs(1)[0]
    decision, count, sdists = Cc.knn_classify(cir, test_row, train, threshold, all_zero, all_one, lsb_one, zero, zero)
    return cir, decision, count, sdists


def knn_plain_inputs(nb, test_row, train, threshold):
    bits = []
    for w in test_row:
        bits += bits_msb(w, nb)
    for r in train:
        for w in r:
            bits += bits_msb(w, nb)
    return bits + bits_msb(threshold, nb) + [0] * nb + [1] * nb + bits_msb(1, nb) + [0]


def test_full_knn_circuit_plaintext_and_census():
    # src/KNN_medical_data.cpp:676-732 on a reduced shape: distances over columns 1..ncol-2, MUX copy, bubble sort with
    # the records as payload, vote over the label column, decision = sign(threshold - count)
    from thfhe import circuits as Cc
    import thfhe
    nb, ncol, ntrain = 8, 4, 3
    cir, decision, count, sdists = build_knn(nb, ncol, ntrain)
    rng = np.random.default_rng(1)
    for trial in range(6):
        test_row = [0] + [int(v) for v in rng.integers(0, 50, ncol - 2)] + [int(rng.integers(0, 2))]
        train = [[i] + [int(v) for v in rng.integers(0, 50, ncol - 2)] + [int(rng.integers(0, 2))] for i in range(ntrain)]
        threshold = ntrain // 2
        v = Cc.simulate(cir, knn_plain_inputs(nb, test_row, train, threshold))
        d = [sum(abs(test_row[c] - r[c]) for c in range(1, ncol - 1)) for r in train]
        assert [from_bits(v[w]) for w in sdists] == sorted(d)
        votes = sum(r[ncol - 1] for r in train)          # K = all train rows, as in the reference (K = train_row_size = 5)
        assert from_bits(v[count]) == votes
        assert bool(v[decision]) == (votes > threshold)
    # census of the reference-sized circuit: 5 train rows, 14 columns, 32 bit
    cir, *_ = build_knn(32, 14, 5)
    cs = cir.census()
    fa, dif = 159, 32 + 2 * 159
    dist_bw = 12 * (2 * dif + 32 + fa)                   # two-input gates + 32 MUX per column
    exp_two = 5 * (dist_bw - 12 * 32) + 20 * (dif + 2 * 15 * 32) + 5 * fa + dif + 1
    exp_mux = 5 * 12 * 32 + 5 * 14 * 32 + 20 * 2 * 15 * 32
    assert cs["mux"] == exp_mux and cs["gates"] == exp_two + exp_mux
    assert cs["rotations"] == exp_two + 2 * exp_mux      # ~1.26e5 blind rotations
    assert 1.2e5 < cs["rotations"] < 1.3e5


@pytest.mark.gpu
def test_reduced_knn_decision_on_gpu():
    # the whole KNN decision DAG (distances, copy, sort with payload, vote, decision) at 8 bit x 4 columns x 3 train rows
    import thfhe
    from thfhe import keygen, circuits as Cc
    nb, ncol, ntrain = 8, 4, 3
    cir, decision, count, sdists = build_knn(nb, ncol, ntrain)
    test_row = [0, 17, 40, 1]
    train = [[1, 20, 35, 1], [2, 3, 44, 0], [3, 16, 41, 1]]
    p = thfhe.make_params("SK-128")
    K = keygen.SecretKeySet(p, seed=0x5EED0001)
    ck = thfhe.CloudKey(p, K.bk, K.ksk, device=0)
    plain = knn_plain_inputs(nb, test_row, train, ntrain // 2)
    stats = {}
    vals = Cc.evaluate(ck, cir, K.encrypt(np.array(plain), seed=5), stats)
    sim = Cc.simulate(cir, plain)
    assert np.array_equal(K.decrypt(vals), sim)                       # every wire of the DAG decrypts to the plaintext simulation
    d = sorted(sum(abs(test_row[c] - r[c]) for c in range(1, ncol - 1)) for r in train)
    assert [from_bits(K.decrypt(vals[w])) for w in sdists] == d
    assert from_bits(K.decrypt(vals[count])) == 2 and bool(K.decrypt(vals[[decision]])[0])
    assert stats["launches"] < stats["gates"] / 4                      # levelised: far fewer launches than gates
    # the native scheduler / device-resident executor (thfhe_dag_run) and the host-driven level loop evaluate the same gates
    assert np.array_equal(vals, Cc.evaluate_levels(ck, cir, K.encrypt(np.array(plain), seed=5)))
    # NOT / COPY gates ride on their operand's level (no bootstrap), including NOT of an input and NOT of a NOT
    c2 = Cc.Circuit()
    x, y = c2.inputs(1)[0], c2.inputs(1)[0]
    nx = c2.gate(thfhe.NOT, x); nnx = c2.gate(thfhe.NOT, nx); g1 = c2.gate(thfhe.NAND, nnx, y); ng = c2.gate(thfhe.NOT, g1)
    g2 = c2.gate(thfhe.MUX, ng, x, y); cp = c2.gate(thfhe.COPY, g2)
    for bits in ([0, 0], [0, 1], [1, 0], [1, 1]):
        v = Cc.evaluate(ck, c2, K.encrypt(np.array(bits), seed=9))
        assert np.array_equal(K.decrypt(v), Cc.simulate_ext(c2, bits))
        assert np.array_equal(v, Cc.evaluate_levels(ck, c2, K.encrypt(np.array(bits), seed=9)))
    ck.close()


@pytest.mark.gpu
def test_reference_sized_knn_decision_on_gpu(O):
    """BASELINE.json configs[3] at the reference's size: the whole KNN decision of src/KNN_medical_data.cpp:676-732 for one test
    record -- 5 train rows x 14 columns x 32 bit, the first six records of the reference's data1.csv (tests/golden/data1.csv):
    ~1.26e5 blind rotations in ~1e3 levels through the native DAG executor.  Every wire must decrypt to the plaintext simulation,
    >= 200 sampled gates (two-input and MUX) must equal the oracle bit for bit, and the sorted distances / vote count / decision
    must equal the plaintext KNN."""
    import os
    import thfhe
    from thfhe import keygen, circuits as Cc
    nb, ncol, ntrain = 32, 14, 5
    rows = []
    with open(os.path.join(O.GOLDEN, "data1.csv")) as f:
        next(f)
        for line in f:
            rows.append([int(float(w)) & 0xFFFFFFFF for w in line.strip().split(",")][:ncol])   # the reference reads every field into an int
            if len(rows) == ntrain + 1:
                break
    train, test_row = rows[:ntrain], rows[ntrain]
    threshold = ntrain // 2
    cir, decision, count, sdists = build_knn(nb, ncol, ntrain)
    cs = cir.census()
    assert 1.2e5 < cs["rotations"] < 1.3e5
    p = thfhe.make_params("SK-128")
    K = keygen.SecretKeySet(p, seed=0x5EED0001)
    ck = thfhe.CloudKey(p, K.bk, K.ksk, device=0)
    plain = knn_plain_inputs(nb, test_row, train, threshold)
    stats = {}
    vals = Cc.evaluate(ck, cir, K.encrypt(np.array(plain), seed=0x5EED0002), stats)
    sim = Cc.simulate(cir, plain)
    assert np.array_equal(K.decrypt(vals), sim)                       # all ~1.02e5 wires
    d = [sum(abs(test_row[c] - r[c]) for c in range(1, ncol - 1)) & 0xFFFFFFFF for r in train]
    assert [from_bits(K.decrypt(vals[w])) for w in sdists] == sorted(d)
    votes = sum(r[ncol - 1] for r in train)
    assert from_bits(K.decrypt(vals[count])) == votes
    assert bool(K.decrypt(vals[[decision]])[0]) == (votes > threshold)
    assert stats["levels"] > 900 and stats["rotations"] == cs["rotations"]
    # sampled gates against the oracle, grouped by opcode so that every group is one OpenMP batch
    orc = O.Oracle(O.make_params("SK-128"), K.bk, K.ksk)
    rng = np.random.default_rng(11)
    gates = np.array(cir.gates, np.int64)
    two = np.flatnonzero(gates[:, 0] != thfhe.MUX)
    mux = np.flatnonzero(gates[:, 0] == thfhe.MUX)
    pick = np.concatenate([rng.choice(two, 176, replace=False), rng.choice(mux, 48, replace=False)])
    base = cir.n_inputs
    for op in np.unique(gates[pick, 0]):
        g = pick[gates[pick, 0] == op]
        x, y = vals[gates[g, 1]], vals[gates[g, 2]]
        z = vals[gates[g, 3]] if op == thfhe.MUX else None
        ref = orc.gates(int(op), x, y, z)
        assert np.array_equal(vals[base + g], ref), f"opcode {op}: sampled DAG gates differ from the oracle"
    ck.close()


def test_mk_comparison_and_multiplier_circuits_plaintext():
    # J/3gen_mk_gates.jl:258-362 (bit vectors LSB-first, two's complement comparisons)
    from thfhe import circuits as Cc
    W = 6
    cir = Cc.Circuit()
    a, b = cir.inputs(W), cir.inputs(W)
    one, zero = cir.inputs(1)[0], cir.inputs(1)[0]
    grt, leq, geq = Cc.mk_grt_3gen(cir, a, b, one), Cc.mk_leq_3gen(cir, a, b, one), Cc.mk_geq_3gen(cir, a, b, one)
    add = Cc.mk_int_add_with_carry_3gen(cir, a, b, zero)
    mul = Cc.mk_int_mul_3gen(cir, a, b, zero)
    lsb = lambda v, w: [(v >> i) & 1 for i in range(w)]
    val = lambda bits: sum(int(x) << i for i, x in enumerate(bits))

    def ref_mul(x, y):   # the reference's dataflow on plain integers (1-based rows; the last addition re-uses row ctr)
        rows = [x if (y >> i) & 1 else 0 for i in range(W)]
        result = [0] * (2 * W + 1)
        result[0] = rows[0] & 1
        tmp = rows[0] >> 1
        ctr = 1
        for i in range(2, W):
            t = tmp + rows[i - 1]
            result[i - 1] = t & 1
            tmp = t >> 1
            ctr = i
        t = tmp + rows[ctr - 1]
        for i in range(W + 1):
            result[i + ctr] = (t >> i) & 1
        return val(result[:W])

    rng = np.random.default_rng(4)
    for _ in range(40):
        x, y = (int(v) for v in rng.integers(0, 1 << (W - 1), 2))       # non-negative, so the sign bit of x - y is the comparison
        v = Cc.simulate(cir, lsb(x, W) + lsb(y, W) + [1, 0])
        assert bool(v[grt]) == (x > y) and bool(v[leq]) == (x <= y) and bool(v[geq]) == (x >= y)
        assert val(v[add]) == x + y
        assert val(v[mul]) == ref_mul(x, y)
    assert ref_mul(3, 1) == 3 and ref_mul(5, 2) == 10                   # agrees with true multiplication when the re-used row is zero


@pytest.mark.gpu
def test_mk_dag_executor_on_gpu(O):
    # thfhe_mk_dag_run: comparison and multiplier circuits of J/3gen_mk_gates.jl:258-362 plus AND3 / MUX / NOT, natively scheduled;
    # every wire must decrypt to the plaintext simulation and equal the host-driven level loop bit for bit
    import thfhe
    from thfhe import circuits as Cc
    p = O.make_params("MK2", n=96)      # reduced LWE dimension keeps the noise budget of these depths comfortable
    sg = O.SIGMAS["MK2"]
    K = O.MKKeys(p, 0x5EED0001, sg["bk"], sg["ks"])
    ck = thfhe.MKCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.ksk, device=0)
    W = 4
    cir = Cc.Circuit()
    a, b = cir.inputs(W), cir.inputs(W)
    one, zero = cir.inputs(1)[0], cir.inputs(1)[0]
    outs = dict(grt=Cc.mk_grt_3gen(cir, a, b, one), geq=Cc.mk_geq_3gen(cir, a, b, one))
    mul = Cc.mk_int_mul_3gen(cir, a, b, zero)
    extra = cir.gate(thfhe.MUX, cir.gate(thfhe.NOT, a[0]), cir.gate(thfhe.AND3, a[1], b[1], one), b[0])
    for x, y in ((5, 3), (2, 7), (6, 6)):
        bits = [(x >> i) & 1 for i in range(W)] + [(y >> i) & 1 for i in range(W)] + [1, 0]
        enc = K.encrypt_bits(bits, sg["lwe"], 70 + x)
        stats = {}
        vals = Cc.evaluate(ck, cir, enc, stats)
        sim = Cc.simulate_mk(cir, bits)
        assert np.array_equal(K.decrypt_bits(vals), sim)
        assert bool(K.decrypt_bits(vals[[outs["grt"]]])[0]) == (x > y) and bool(K.decrypt_bits(vals[[outs["geq"]]])[0]) == (x >= y)
        assert stats["launches"] < stats["gates"] / 2
    # two-input-only circuit: the native executor equals the host-driven loop (thfhe_mk_gates_mixed per level) bit for bit
    c2 = Cc.Circuit()
    a2, b2 = c2.inputs(W), c2.inputs(W)
    z2 = c2.inputs(1)[0]
    Cc.mk_int_add_with_carry_3gen(c2, a2, b2, z2)
    enc = K.encrypt_bits([1, 0, 1, 1, 0, 1, 1, 0, 0], sg["lwe"], 99)
    assert np.array_equal(Cc.evaluate(ck, c2, enc), Cc.evaluate_levels(ck, c2, enc))
    ck.close()


def random_dag(rng, n_inputs, n_gates, ops):
    from thfhe import circuits as Cc
    import thfhe
    cir = Cc.Circuit()
    cir.inputs(n_inputs)
    for _ in range(n_gates):
        op = int(rng.choice(ops))
        w = cir.n_wires()
        pick = lambda: int(rng.integers(max(0, w - 12), w))       # mostly recent wires: deep chains and wide levels both occur
        if op in (thfhe.NOT, thfhe.COPY):
            cir.gate(op, pick())
        elif op in (thfhe.MUX, thfhe.AND3):
            cir.gate(op, pick(), pick(), pick())
        else:
            cir.gate(op, pick(), pick())
    return cir


def test_scheduler_levels_respect_dependencies_on_random_dags():
    # the Python ASAP schedule (the native one is checked against it on the GPU): every gate runs after its operands
    import thfhe
    rng = np.random.default_rng(11)
    ops = [thfhe.NAND, thfhe.XOR, thfhe.OR, thfhe.ANDNY, thfhe.MUX, thfhe.NOT, thfhe.COPY]
    for trial in range(5):
        cir = random_dag(rng, 5, 120, ops)
        seen = set(range(cir.n_inputs))
        for level in cir.levels():
            linear = cir.gates[level[0]][0] in (thfhe.NOT, thfhe.COPY)
            done_now = set()
            for g in level:
                for w in cir.gates[g][1:]:
                    assert w < 0 or w in seen or (linear and w in done_now), (trial, g)
                done_now.add(cir.n_inputs + g)
            seen |= done_now
        assert len(seen) == cir.n_wires()


@pytest.mark.gpu
def test_native_dag_executors_on_random_dags(O):
    # thfhe_dag_run / thfhe_mk_dag_run on random gate DAGs (all gate classes, NOT / COPY chains, shared operands): every wire decrypts to
    # the plaintext simulation, and the single-key result equals the host-driven level loop bit for bit
    import thfhe
    from thfhe import keygen, circuits as Cc
    rng = np.random.default_rng(12)
    p = thfhe.make_params("SK-128", n=64)
    K = keygen.SecretKeySet(p, seed=4)
    ck = thfhe.CloudKey(p, K.bk, K.ksk, device=0)
    sk_ops = [thfhe.NAND, thfhe.OR, thfhe.AND, thfhe.XOR, thfhe.XNOR, thfhe.NOR, thfhe.ANDNY, thfhe.ANDYN, thfhe.ORNY, thfhe.ORYN,
              thfhe.MUX, thfhe.NOT, thfhe.COPY]
    for trial in range(3):
        cir = random_dag(rng, 6, 150, sk_ops)
        bits = rng.integers(0, 2, 6)
        enc = K.encrypt(bits, seed=100 + trial)
        stats = {}
        vals = Cc.evaluate(ck, cir, enc, stats)
        assert np.array_equal(K.decrypt(vals), Cc.simulate(cir, bits)), trial
        assert np.array_equal(vals, Cc.evaluate_levels(ck, cir, enc)), trial
        assert stats["levels"] == cir.census()["depth"]
    ck.close()
    pm = O.make_params("MK2", n=64)
    sg = O.SIGMAS["MK2"]
    KM = O.MKKeys(pm, 5, sg["bk"], sg["ks"])
    mk = thfhe.MKCloudKey(thfhe.make_params(**pm.as_dict()), KM.bk, KM.ksk, device=0)
    mk_ops = [thfhe.NAND, thfhe.OR, thfhe.AND, thfhe.XOR, thfhe.AND3, thfhe.MUX, thfhe.NOT, thfhe.COPY]
    cir = random_dag(rng, 6, 80, mk_ops)
    bits = rng.integers(0, 2, 6)
    vals = Cc.evaluate(mk, cir, KM.encrypt_bits(bits, sg["lwe"], 7))
    assert np.array_equal(KM.decrypt_bits(vals), Cc.simulate_mk(cir, bits))
    mk.close()


@pytest.mark.gpu
def test_batched_dag_instances_equal_single_runs(O):
    """thfhe_dag_run_batch / thfhe_mk_dag_run_batch: Q instances of one gate list walk the levels side by side.  Every instance must
    equal its own thfhe_dag_run bit for bit -- with the level cut into slices that straddle instances, with selected output wires and
    with all wires -- and sampled gates must equal the oracle."""
    import thfhe
    from thfhe import keygen, circuits as Cc
    rng = np.random.default_rng(21)
    p = thfhe.make_params("SK-128", n=64)
    K = keygen.SecretKeySet(p, seed=4)
    ck = thfhe.CloudKey(p, K.bk, K.ksk, device=0)
    orc = O.Oracle(O.make_params("SK-128", n=64), K.bk, K.ksk)
    sk_ops = [thfhe.NAND, thfhe.OR, thfhe.AND, thfhe.XOR, thfhe.XNOR, thfhe.NOR, thfhe.ANDNY, thfhe.ANDYN, thfhe.ORNY, thfhe.ORYN,
              thfhe.MUX, thfhe.NOT, thfhe.COPY]
    Q = 7
    cir = random_dag(rng, 6, 160, sk_ops)
    gates = np.array(cir.gates, np.int32)
    bits = rng.integers(0, 2, (Q, 6))
    enc = np.stack([K.encrypt(bits[q], seed=300 + q) for q in range(Q)])
    single = np.stack([ck.dag_run(enc[q], gates)[0] for q in range(Q)])
    for slice_gates in (28672, 5, 1):      # whole levels; slices of 5 and of 1 gate: a slice then straddles instances / classes
        ck.set_dag_slice(slice_gates)
        st = {}
        allw = Cc.evaluate_batch(ck, cir, enc, None, st)
        assert np.array_equal(allw, single), slice_gates
        assert st["instances"] == Q and st["rotations"] == Q * cir.census()["rotations"]
        sel = rng.choice(cir.n_wires(), 37, replace=False)
        assert np.array_equal(Cc.evaluate_batch(ck, cir, enc, sel), single[:, sel]), slice_gates
    ck.set_dag_slice(28672)
    for q in range(Q):
        assert np.array_equal(K.decrypt(single[q]), Cc.simulate(cir, bits[q]))
    g = rng.choice(np.flatnonzero(~np.isin(gates[:, 0], (thfhe.NOT, thfhe.COPY))), 24, replace=False)   # sampled gates vs the oracle, all instances
    for gi in g:
        op, a, b, c = (int(v) for v in gates[gi])
        ref = orc.gates(op, single[:, a], single[:, b], single[:, c] if op == thfhe.MUX else None)
        assert np.array_equal(single[:, cir.n_inputs + gi], ref), gi
    # argument checks of the C entry point
    with pytest.raises(thfhe.ThfheError):
        ck.dag_run_batch(enc, gates, np.array([cir.n_wires()], np.int32))          # output wire out of range
    out, _ = ck.dag_run_batch(enc[:0], gates)                                       # zero instances: nothing to do
    assert out.shape == (0, len(cir.gates), p.n + 1)
    ck.close()
    # 3-gen engine: all its gate classes, 5 instances, slices of 3 gates
    pm = O.make_params("MK2", n=64)
    sg = O.SIGMAS["MK2"]
    KM = O.MKKeys(pm, 5, sg["bk"], sg["ks"])
    mk = thfhe.MKCloudKey(thfhe.make_params(**pm.as_dict()), KM.bk, KM.ksk, device=0)
    mk_ops = [thfhe.NAND, thfhe.OR, thfhe.AND, thfhe.XOR, thfhe.AND3, thfhe.MUX, thfhe.NOT, thfhe.COPY]
    cir = random_dag(rng, 6, 60, mk_ops)
    bits = rng.integers(0, 2, (5, 6))
    enc = np.stack([KM.encrypt_bits(bits[q], sg["lwe"], 40 + q) for q in range(5)])
    single = np.stack([mk.dag_run(enc[q], np.array(cir.gates, np.int32))[0] for q in range(5)])
    for slice_gates in (8192, 3):
        mk.set_dag_slice(slice_gates)
        assert np.array_equal(Cc.evaluate_batch(mk, cir, enc), single), slice_gates
    for q in range(5):
        assert np.array_equal(KM.decrypt_bits(single[q]), Cc.simulate_mk(cir, bits[q]))
    mk.close()


@pytest.mark.gpu
def test_batched_knn_decisions_on_gpu(O):
    """The reference's loop over test records (src/KNN_medical_data.cpp:676-691) as one batched evaluation at the reference's size:
    4 test records x (5 train rows x 14 columns x 32 bit) = 5.0e5 blind rotations through thfhe_dag_run_batch.  Every decision, vote
    count and sorted distance list must equal the plaintext KNN; >= 200 sampled gates of the batch (operands and outputs fetched with
    out_wires, all four instances) must equal the oracle bit for bit; record 0 must equal the single-decision path bit for bit."""
    import os
    import thfhe
    from thfhe import keygen, circuits as Cc
    nb, ncol, ntrain, Q = 32, 14, 5, 4
    rows = []
    with open(os.path.join(O.GOLDEN, "data1.csv")) as f:
        next(f)
        for line in f:
            rows.append([int(float(w)) & 0xFFFFFFFF for w in line.strip().split(",")][:ncol])
            if len(rows) == ntrain + Q:
                break
    train, tests = rows[:ntrain], rows[ntrain:]
    threshold = ntrain // 2
    p = thfhe.make_params("SK-128")
    K = keygen.SecretKeySet(p, seed=0x5EED0001)
    ck = thfhe.CloudKey(p, K.bk, K.ksk, device=0)
    words = p.n + 1
    enc = lambda vals, seed: K.encrypt(np.array(sum((bits_msb(v, nb) for v in vals), [])), seed=seed).reshape(len(vals), nb, words)
    e_tests = np.stack([enc(t, 0x5EED0100 + q) for q, t in enumerate(tests)])
    e_train = np.stack([enc(r, 0x5EED0200 + j) for j, r in enumerate(train)])
    thr, az, ao, lo = (enc([v], 0x5EED0300 + q)[0] for q, v in enumerate((threshold, 0, 0xFFFFFFFF, 1)))
    zero = K.encrypt(np.array([0]), seed=0x5EED0400)[0]
    plan = Cc.KnnPlan(nb, ncol, ntrain)
    st = {}
    res = Cc.knn_decisions_batched(ck, plan, e_tests, e_train, thr, az, ao, lo, zero, stats=st)
    votes = sum(r[ncol - 1] for r in train)
    for q, t in enumerate(tests):
        d = [sum(abs(t[c] - r[c]) for c in range(1, ncol - 1)) & 0xFFFFFFFF for r in train]
        assert [from_bits(K.decrypt(w)) for w in res["dists"][q]] == d, q
        assert [from_bits(K.decrypt(w)) for w in res["sorted_dists"][q]] == sorted(d), q
        assert from_bits(K.decrypt(res["count"][q])) == votes
        assert bool(K.decrypt(res["decision"][q][None])[0]) == (votes > threshold)
    assert st["phase1"]["rotations"] + st["phase2"]["rotations"] > 4 * 1.2e5
    one = Cc.knn_decision_sharded(ck, plan, e_tests[0], e_train, thr, az, ao, lo, zero)
    for k in ("decision", "count", "sorted_dists", "dists"):
        assert np.array_equal(one[k], res[k][0]), k
    # sampled gates of both phases (the wide distance levels, the deep sort / vote chain), all four instances, against the oracle
    orc = O.Oracle(O.make_params("SK-128"), K.bk, K.ksk)
    rng = np.random.default_rng(5)

    def sample_against_oracle(cir, inputs, extra_sel, n_two, n_mux):
        gates = np.array(cir.gates, np.int64)
        two = np.flatnonzero(gates[:, 0] != thfhe.MUX)
        mux = np.flatnonzero(gates[:, 0] == thfhe.MUX)
        pick = np.concatenate([rng.choice(two, n_two, replace=False), rng.choice(mux, n_mux, replace=False)])
        need = sorted(set(int(w) for g in pick for w in gates[g, 1:] if w >= 0) | set(int(cir.n_inputs + g) for g in pick))
        pos = {w: i for i, w in enumerate(need)}
        got_all = Cc.evaluate_batch(ck, cir, inputs, list(extra_sel) + need)
        got = got_all[:, len(extra_sel):]
        for op in np.unique(gates[pick, 0]):
            g = pick[gates[pick, 0] == op]
            col = lambda k: got[:, [pos[int(w)] for w in gates[g, k]]].reshape(-1, words)
            ref = orc.gates(int(op), col(1), col(2), col(3) if op == thfhe.MUX else None)
            assert np.array_equal(got[:, [pos[int(cir.n_inputs + x)] for x in g]].reshape(-1, words), ref), f"opcode {op}"
        return got_all[:, :len(extra_sel)]

    c1, dist, copies = plan.phase1(ntrain)
    shared = np.concatenate([e_train.reshape(-1, words), az, ao, lo, zero[None]])
    in1 = np.stack([np.concatenate([e_tests[q].reshape(-1, words), shared]) for q in range(Q)])
    sel1 = [w for r in range(ntrain) for col in range(ncol) for w in copies[r][col]] + [w for r in range(ntrain) for w in dist[r]]
    o1 = sample_against_oracle(c1, in1, sel1, 28, 8)
    c2, *_ = plan.phase2()
    tail = np.concatenate([thr, az, ao, lo, zero[None]])
    in2 = np.concatenate([o1, np.broadcast_to(tail, (Q,) + tail.shape)], axis=1)
    sample_against_oracle(c2, in2, [], 28, 8)            # (36 + 36) gates x 4 instances = 288 oracle comparisons
    ck.close()
