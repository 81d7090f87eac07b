"""Gate-DAG front end: the reference's circuits as static gate lists + a levelising evaluator.

The reference's applications issue long dependent streams of two-input gates, one `boots*` call at a time
(FullAdder / difference / distance / distance_bw_data, src/KNN_medical_data.cpp:127-263; mk_add_3gen,
3gen_mk_gates.jl:183-220).  Here the same wiring is recorded as a DAG, scheduled ASAP into levels, and every level is
evaluated as ONE batched launch (thfhe_gates_mixed for the two-input gates, thfhe_gates for the MUXes) so that the
independent gates of a level -- and of independent sub-circuits placed in the same DAG -- fill the GPU.

Bit vectors are MSB-first lists of wire ids, as in the reference (index nbits-1 = least significant bit,
src/bootstrap_modules.cpp:95).
"""
import time

import numpy as np

from . import AND, COPY, MUX, NOT, OR, XOR


class Circuit:
    """A static gate list.  Wires are integers; inputs are declared first, every gate defines one new wire."""

    def __init__(self):
        self.n_inputs = 0
        self.gates = []   # (op, a, b, c) ; c = -1 unless MUX ; output wire id = n_inputs + index
        self.outputs = {}

    def inputs(self, count):
        assert not self.gates, "declare all inputs before the first gate"
        ids = list(range(self.n_inputs, self.n_inputs + count))
        self.n_inputs += count
        return ids

    def gate(self, op, a, b=-1, c=-1):
        self.gates.append((op, a, b, c))
        return self.n_inputs + len(self.gates) - 1

    def n_wires(self):
        return self.n_inputs + len(self.gates)

    def levels(self):
        """ASAP schedule: list of lists of gate indices; NOT costs no level (it is not bootstrapped, gates.jl:76-79)."""
        depth = np.zeros(self.n_wires(), np.int64)
        lv = {}
        for gi, (op, a, b, c) in enumerate(self.gates):
            d = max(depth[w] for w in (a, b, c) if w >= 0)
            if op not in (NOT, COPY):
                d += 1
            depth[self.n_inputs + gi] = d
            lv.setdefault((int(d), op in (NOT, COPY)), []).append(gi)
        keys = sorted(lv)  # (depth, is_not): bootstrapped gates of depth d first, then the free NOTs that read them
        return [lv[k] for k in keys]

    def census(self):
        ops = [g[0] for g in self.gates]
        boot = sum(1 for o in ops if o not in (NOT, COPY))
        return dict(gates=len(ops), bootstrapped=boot, mux=ops.count(MUX), rotations=boot + ops.count(MUX),
                    depth=len([l for l in self.levels() if self.gates[l[0]][0] not in (NOT, COPY)]))


# ---- the reference's building blocks (src/KNN_medical_data.cpp) ---------------------------------------------------
def ones_comp(cir, all_one, x):
    """onesComp, :127-132."""
    return [cir.gate(XOR, all_one[i], x[i]) for i in range(len(x))]


def full_adder(cir, a, b, carry_in):
    """FullAdder, :134-157 (same wiring as src/bootstrap_modules.cpp:20-44).  Returns (sum, carry) MSB-first; carry[nb-1] = carry_in."""
    nb = len(a)
    sum2 = [None] * nb
    carry = [None] * nb
    carry[nb - 1] = carry_in
    for i in range(nb - 1, -1, -1):
        s1 = cir.gate(XOR, a[i], b[i])
        c1 = cir.gate(AND, a[i], b[i])
        sum2[i] = cir.gate(XOR, s1, carry[i])
        c2 = cir.gate(AND, s1, carry[i])
        if i != 0:
            carry[i - 1] = cir.gate(OR, c1, c2)
    return sum2, carry


def difference(cir, x, y, all_one, lsb_one, zero):
    """difference = x - y via two's complement, :161-213."""
    ones = ones_comp(cir, all_one, y)
    twos, _ = full_adder(cir, ones, lsb_one, zero)
    diff, _ = full_adder(cir, x, twos, zero)
    return diff


def distance(cir, x, y, all_one, lsb_one, zero):
    """|x - y|: dist[i] = MUX(d1[0], d2[i], d1[i]), :217-236."""
    d1 = difference(cir, x, y, all_one, lsb_one, zero)
    d2 = difference(cir, y, x, all_one, lsb_one, zero)
    return [cir.gate(MUX, d1[0], d2[i], d1[i]) for i in range(len(x))]


def distance_bw_data(cir, row_a, row_b, all_zero, all_one, lsb_one, zero):
    """Manhattan distance of two records, columns 1..end, :239-263."""
    result = list(all_zero)
    for col in range(1, len(row_a)):
        dist = distance(cir, row_a[col], row_b[col], all_one, lsb_one, zero)
        result, _ = full_adder(cir, result, dist, zero)
    return result


def copy_through_mux(cir, all_one, x):
    """The reference copies a word with bootsMUX(allOne[l], x[l], x[l]), :685-689 (a bootstrapped refresh)."""
    return [cir.gate(MUX, all_one[i], x[i], x[i]) for i in range(len(x))]


def compare_swap(cir, key_a, key_b, payload_a, payload_b, all_zero, all_one, lsb_one, zero):
    """One step of sort_with_distance, :443-481: diff = key_a - key_b; its sign bit routes the smaller key (and its
    payload words) to position a and the bigger to position b through MUXes; every routed bit is then refreshed with
    XOR(., allZero).  Returns (key_a', key_b', payload_a', payload_b')."""
    diff = difference(cir, key_a, key_b, all_one, lsb_one, zero)
    s = diff[0]
    nb = len(key_a)
    big = [cir.gate(MUX, s, key_b[j], key_a[j]) for j in range(nb)]
    small = [cir.gate(MUX, s, key_a[j], key_b[j]) for j in range(nb)]
    pay_big = [[cir.gate(MUX, s, wb[j], wa[j]) for j in range(nb)] for wa, wb in zip(payload_a, payload_b)]
    pay_small = [[cir.gate(MUX, s, wa[j], wb[j]) for j in range(nb)] for wa, wb in zip(payload_a, payload_b)]
    refresh = lambda w: [cir.gate(XOR, w[j], all_zero[j]) for j in range(nb)]
    return refresh(small), refresh(big), [refresh(w) for w in pay_small], [refresh(w) for w in pay_big]


def sort_with_distance(cir, rows, dists, all_zero, all_one, lsb_one, zero):
    """sort_with_distance, :410-489: n passes of adjacent compare-swaps (bubble sort) on the distances, the train records
    moving with them.  rows[i] = list of words, dists[i] = word.  Returns (rows, dists) sorted by ascending distance."""
    rows, dists = [list(r) for r in rows], list(dists)
    n = len(dists)
    for _ in range(n):
        for i in range(1, n):
            dists[i - 1], dists[i], rows[i - 1], rows[i] = compare_swap(cir, dists[i - 1], dists[i], rows[i - 1], rows[i],
                                                                        all_zero, all_one, lsb_one, zero)
    return rows, dists


def knn_classify(cir, test_row, train_rows, threshold, all_zero, all_one, lsb_one, lsb_zero_carry, zero, K=None):
    """The reference's KNN decision for one test record, :676-732: Manhattan distances to every train row (columns
    1..col_size-2), a MUX copy of the train rows, sort by distance, count = sum of the label column of the K nearest,
    decision = XOR(sign(threshold - count), 0).  Returns (decision_wire, count_word, sorted_dists)."""
    ncol = len(test_row)
    n = len(train_rows)
    K = n if K is None else K
    dists = [distance_bw_data(cir, test_row[:ncol - 1], tr[:ncol - 1], all_zero, all_one, lsb_one, zero) for tr in train_rows]
    copies = [[copy_through_mux(cir, all_one, w) for w in tr] for tr in train_rows]
    srows, sdists = sort_with_distance(cir, copies, dists, all_zero, all_one, lsb_one, zero)
    count = list(all_zero)
    for j in range(K):
        count, _ = full_adder(cir, count, srows[j][ncol - 1], lsb_zero_carry)
    diff = difference(cir, threshold, count, all_one, lsb_one, zero)
    return cir.gate(XOR, diff[0], all_zero[0]), count, sdists


# ---- the KNN decision sharded over ranks (BASELINE.json configs[3]; src/KNN_medical_data.cpp:676-732) ------------------------------
class KnnPlan:
    """The reference's KNN decision for one test record, cut the way its `#pragma omp parallel for` over train rows (:681-691) cuts it:
      phase 1  per train row: distance_bw_data to the test record + the MUX copy of the row -- independent rows, sharded over ranks;
      phase 2  sort_with_distance, vote over the label column of the K = n_train nearest, decision bit -- one sequential chain,
               evaluated by every rank on the gathered rows (replicated keys; 855 of its levels hold 1-3 gates, nothing to shard).
    Inputs (MSB-first bit records): the test record, the train rows, and the constants (threshold, allZero, allOne, lsbOne, zero).
    One rank (world = 1) evaluates the very same two DAGs, so the sharded result equals the single-rank result bit for bit."""

    def __init__(self, nb, ncol, ntrain):
        self.nb, self.ncol, self.ntrain = nb, ncol, ntrain

    def phase1(self, n_rows):
        """DAG for n_rows train rows: inputs = test[ncol], rows[n_rows][ncol], allZero, allOne, lsbOne, zero."""
        c = Circuit()
        test = [c.inputs(self.nb) for _ in range(self.ncol)]
        rows = [[c.inputs(self.nb) for _ in range(self.ncol)] for _ in range(n_rows)]
        all_zero, all_one, lsb_one = c.inputs(self.nb), c.inputs(self.nb), c.inputs(self.nb)
        zero = c.inputs(1)[0]
        dist = [distance_bw_data(c, test[:self.ncol - 1], r[:self.ncol - 1], all_zero, all_one, lsb_one, zero) for r in rows]
        copies = [[copy_through_mux(c, all_one, w) for w in r] for r in rows]
        return c, dist, copies

    def phase1_distances(self, n_rows):
        """phase1 without the MUX copies (they do not depend on the test record): inputs as phase1."""
        c = Circuit()
        test = [c.inputs(self.nb) for _ in range(self.ncol)]
        rows = [[c.inputs(self.nb) for _ in range(self.ncol)] for _ in range(n_rows)]
        all_zero, all_one, lsb_one = c.inputs(self.nb), c.inputs(self.nb), c.inputs(self.nb)
        zero = c.inputs(1)[0]
        dist = [distance_bw_data(c, test[:self.ncol - 1], r[:self.ncol - 1], all_zero, all_one, lsb_one, zero) for r in rows]
        return c, dist

    def copies(self, n_rows):
        """The MUX copy of the train rows alone (src/KNN_medical_data.cpp:685-689): inputs = rows[n_rows][ncol], allOne."""
        c = Circuit()
        rows = [[c.inputs(self.nb) for _ in range(self.ncol)] for _ in range(n_rows)]
        all_one = c.inputs(self.nb)
        return c, [[copy_through_mux(c, all_one, w) for w in r] for r in rows]

    def phase2(self):
        """DAG on the gathered rows: inputs = rows[ntrain][ncol], dists[ntrain], threshold, allZero, allOne, lsbOne, zero."""
        c = Circuit()
        rows = [[c.inputs(self.nb) for _ in range(self.ncol)] for _ in range(self.ntrain)]
        dists = [c.inputs(self.nb) for _ in range(self.ntrain)]
        thr, all_zero, all_one, lsb_one = (c.inputs(self.nb) for _ in range(4))
        zero = c.inputs(1)[0]
        srows, sdists = sort_with_distance(c, rows, dists, all_zero, all_one, lsb_one, zero)
        count = list(all_zero)
        for j in range(self.ntrain):
            count, _ = full_adder(c, count, srows[j][self.ncol - 1], zero)
        diff = difference(c, thr, count, all_one, lsb_one, zero)
        decision = c.gate(XOR, diff[0], all_zero[0])
        return c, decision, count, sdists, srows


def knn_decision_sharded(ck, plan, test, train, threshold, all_zero, all_one, lsb_one, zero, rank=0, world=1, all_reduce=None, stats=None):
    """Evaluate the KNN decision with the train rows of phase 1 dealt round-robin over `world` ranks (every rank holds the keys).
    test: int32[ncol][nb][words]; train: int32[ntrain][ncol][nb][words]; threshold / all_zero / all_one / lsb_one: int32[nb][words];
    zero: int32[words].  all_reduce(np.ndarray) -> np.ndarray sums an int32 array over the ranks (torch.distributed all_reduce over
    RCCL or gloo; every row is produced by exactly one rank, so the sum IS the gather); None is allowed only for world = 1.
    Returns dict(decision=record, count=records[nb], sorted_dists=records[ntrain][nb], dists=records[ntrain][nb])."""
    nb, ncol, ntrain = plan.nb, plan.ncol, plan.ntrain
    words = ck.words
    test = np.asarray(test, np.int32).reshape(ncol, nb, words)
    train = np.asarray(train, np.int32).reshape(ntrain, ncol, nb, words)
    consts = [np.asarray(v, np.int32).reshape(nb, words) for v in (all_zero, all_one, lsb_one)]
    zero = np.asarray(zero, np.int32).reshape(1, words)
    mine = [j for j in range(ntrain) if j % world == rank]
    gathered = np.zeros((ntrain, ncol + 1, nb, words), np.int32)   # [row][its ncol words, then its distance]
    st1 = {}
    if mine:
        c1, dist, copies = plan.phase1(len(mine))
        in1 = np.concatenate([test.reshape(-1, words), train[mine].reshape(-1, words)] + consts + [zero])
        _t = time.perf_counter()
        v1 = evaluate(ck, c1, in1, st1)
        st1["seconds"] = st1.get("seconds", 0.0) + time.perf_counter() - _t
        for q, j in enumerate(mine):
            for col in range(ncol):
                gathered[j, col] = v1[copies[q][col]]
            gathered[j, ncol] = v1[dist[q]]
    if world > 1:
        if all_reduce is None:
            raise ValueError("knn_decision_sharded: world > 1 needs an all_reduce callable")
        gathered = np.asarray(all_reduce(gathered), np.int32).reshape(gathered.shape)
    c2, decision, count, sdists, _ = plan.phase2()
    thr = np.asarray(threshold, np.int32).reshape(nb, words)
    in2 = np.concatenate([gathered[:, :ncol].reshape(-1, words), gathered[:, ncol].reshape(-1, words), thr] + consts + [zero])
    st2 = {}
    _t = time.perf_counter()
    v2 = evaluate(ck, c2, in2, st2)
    st2["seconds"] = time.perf_counter() - _t
    if stats is not None:
        stats.update(phase1=st1, phase2=st2, my_rows=mine)
    return dict(decision=v2[decision], count=v2[count], sorted_dists=np.stack([v2[w] for w in sdists]), dists=gathered[:, ncol])


def knn_decisions_batched(ck, plan, tests, train, threshold, all_zero, all_one, lsb_one, zero, rank=0, world=1, all_reduce=None, stats=None):
    """The reference's loop over test records (`for i < test_row_size`, src/KNN_medical_data.cpp:676-691) as ONE batched evaluation:
    every test record is an instance of the same two DAGs (KnnPlan.phase1 over all train rows, KnnPlan.phase2) and the instances walk
    the levels side by side (dag_run_batch), so that the 879 levels of the sort / vote chain that hold 1-3 gates per decision hold
    Q-3Q gates per launch.  Sharding is BY QUERY: rank r evaluates the test records q with q % world == r, both phases, with no
    exchange in between; one all_reduce at the end gathers the results (every record is produced by exactly one rank, so the sum is
    the gather).  Gates are deterministic, so record q's result equals knn_decision_sharded on tests[q] bit for bit.
    tests: int32[Q][ncol][nb][words]; the other arguments as in knn_decision_sharded.
    Returns dict(decision=[Q][words], count=[Q][nb][words], sorted_dists=[Q][ntrain][nb][words], dists=[Q][ntrain][nb][words])."""
    nb, ncol, ntrain = plan.nb, plan.ncol, plan.ntrain
    words = ck.words
    tests = np.asarray(tests, np.int32)
    tests = tests.reshape(-1, ncol, nb, words)
    Q = tests.shape[0]
    train = np.asarray(train, np.int32).reshape(ntrain, ncol, nb, words)
    consts = [np.asarray(v, np.int32).reshape(nb, words) for v in (all_zero, all_one, lsb_one)]
    zero = np.asarray(zero, np.int32).reshape(1, words)
    thr = np.asarray(threshold, np.int32).reshape(nb, words)
    mine = [q for q in range(Q) if q % world == rank]
    per = 1 + nb + 2 * ntrain * nb                       # decision | count | sorted distances | distances
    result = np.zeros((Q, per, words), np.int32)
    st1, st2 = {}, {}
    if mine:
        # the MUX copies of the train rows (:685-689) do not depend on the test record and a bootstrapped gate is a deterministic function of its
        # operands: evaluated ONCE per rank, they are the very ciphertexts the reference recomputes for every test record (3.6 % of its rotations)
        cc, copy_w = plan.copies(ntrain)
        stc = {}
        _t = time.perf_counter()
        oc = evaluate_batch(ck, cc, np.concatenate([train.reshape(-1, words), consts[1]])[None], [w for r in copy_w for col in r for w in col], stc)[0]
        c1, dist = plan.phase1_distances(ntrain)
        shared = np.concatenate([train.reshape(-1, words)] + consts + [zero])
        in1 = np.stack([np.concatenate([tests[q].reshape(-1, words), shared]) for q in mine])
        od = evaluate_batch(ck, c1, in1, [w for r in range(ntrain) for w in dist[r]], st1)
        st1["rotations"] = st1.get("rotations", 0) + stc.get("rotations", 0)
        st1["copies_once"] = dict(rotations=stc.get("rotations"), launches=stc.get("launches"))
        st1["seconds"] = time.perf_counter() - _t
        o1 = np.concatenate([np.broadcast_to(oc, (len(mine),) + oc.shape), od], axis=1)   # rows' copies | distances, the layout of phase 2's inputs
        del in1
        c2, decision, count, sdists, _ = plan.phase2()
        tail = np.concatenate([thr] + consts + [zero])
        in2 = np.concatenate([o1, np.broadcast_to(tail, (len(mine),) + tail.shape)], axis=1)   # rows, dists | thr, constants, zero
        sel2 = [decision] + list(count) + [w for d in sdists for w in d]
        _t = time.perf_counter()
        o2 = evaluate_batch(ck, c2, in2, sel2, st2)
        st2["seconds"] = time.perf_counter() - _t
        result[mine, :1 + nb + ntrain * nb] = o2
        result[mine, 1 + nb + ntrain * nb:] = o1[:, ntrain * ncol * nb:]
    if world > 1:
        if all_reduce is None:
            raise ValueError("knn_decisions_batched: world > 1 needs an all_reduce callable")
        result = np.asarray(all_reduce(result), np.int32).reshape(result.shape)
    if stats is not None:
        stats.update(phase1=st1, phase2=st2, my_queries=mine)
    a, b = 1 + nb, 1 + nb + ntrain * nb
    return dict(decision=result[:, 0], count=result[:, 1:a], sorted_dists=result[:, a:b].reshape(Q, ntrain, nb, words),
                dists=result[:, b:].reshape(Q, ntrain, nb, words))


def torch_all_reduce(device=None):
    """all_reduce callable for knn_decision_sharded over torch.distributed (backend nccl = RCCL: pass the rank's cuda device)."""
    import torch
    import torch.distributed as dist

    def f(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t)
        return t.cpu().numpy()
    return f


# ---- the reference's multi-key integer circuits (3gen_mk_gates.jl; bit vectors LSB-first, mk_api.jl:563-576) ----------
def mk_add_3gen(cir, a, b, cin):
    """mk_add_3gen / mk_add_3gen_v2, 3gen_mk_gates.jl:183-220."""
    out = []
    for i in range(len(a)):
        t1 = cir.gate(XOR, a[i], b[i])
        t2 = cir.gate(AND, a[i], b[i])
        out.append(cir.gate(XOR, t1, cin))
        t3 = cir.gate(AND, t1, cin)
        cin = cir.gate(OR, t2, t3)
    return out


def mk_inv_3gen(cir, a, one):
    """:223-233."""
    return [cir.gate(XOR, x, one) for x in a]


def mk_sub_3gen(cir, a, b, one):
    """a - b = a + ~b + 1, :236-244."""
    return mk_add_3gen(cir, a, mk_inv_3gen(cir, b, one), one)


def mk_less_3gen(cir, a, b, one):
    """sign bit of a - b, :247-255."""
    return mk_sub_3gen(cir, a, b, one)[-1]


def mk_grt_3gen(cir, a, b, one):
    """a > b: sign bit of b - a, copied, :258-266."""
    return cir.gate(COPY, mk_sub_3gen(cir, b, a, one)[-1])


def mk_leq_3gen(cir, a, b, one):
    """:269-277."""
    return cir.gate(XOR, mk_grt_3gen(cir, a, b, one), one)


def mk_geq_3gen(cir, a, b, one):
    """:280-288."""
    return cir.gate(XOR, mk_less_3gen(cir, a, b, one), one)


def mk_int_add_with_carry_3gen(cir, a, b, cin):
    """WIDTH sum bits + the carry out, :291-310."""
    out = []
    for i in range(len(a)):
        t1 = cir.gate(XOR, a[i], b[i])
        t2 = cir.gate(AND, a[i], b[i])
        out.append(cir.gate(XOR, t1, cin))
        t3 = cir.gate(AND, t1, cin)
        cin = cir.gate(OR, t2, t3)
    return out + [cin]


def mk_int_mul_3gen(cir, a, b, zero):
    """Shift-and-add multiplier, low WIDTH bits, :312-362 -- the reference's dataflow verbatim, including its last addition of
    partial-product row `ctr` (= WIDTH-1, not WIDTH) and its mk_copy_3gen refreshes."""
    W = len(a)
    cp = lambda w: cir.gate(COPY, w)
    BArr = [[cir.gate(AND, a[j], b[i]) for j in range(W)] for i in range(W)]
    result = [None] * (2 * W + 1)
    result[0] = cp(BArr[0][0])
    tmp_in = [cp(BArr[0][i + 1]) for i in range(W - 1)] + [cp(zero)]
    ctr = 1
    for i in range(2, W):                      # Julia i = 2 .. WIDTH-1 (1-based row i)
        t = mk_int_add_with_carry_3gen(cir, tmp_in, BArr[i - 1], zero)
        result[i - 1] = cp(t[0])
        tmp_in = [cp(t[j + 1]) for j in range(W)]
        ctr = i
    t = mk_int_add_with_carry_3gen(cir, tmp_in, BArr[ctr - 1], zero)
    for i in range(W + 1):
        result[i + ctr] = cp(t[i])
    return [cp(result[i]) for i in range(W)]


def simulate(cir, input_bits):
    """Plaintext evaluation of the DAG (wiring check): bool[n_inputs] -> bool[n_wires]."""
    from . import ANDNY, ANDYN, NAND, NOR, ORNY, ORYN, XNOR
    v = np.zeros(cir.n_wires(), bool)
    v[:cir.n_inputs] = np.asarray(input_bits, bool)
    f = {NAND: lambda a, b: not (a and b), OR: lambda a, b: a or b, AND: lambda a, b: a and b, XOR: lambda a, b: a != b,
         XNOR: lambda a, b: a == b, NOR: lambda a, b: not (a or b), ANDNY: lambda a, b: (not a) and b,
         ANDYN: lambda a, b: a and (not b), ORNY: lambda a, b: (not a) or b, ORYN: lambda a, b: a or (not b)}
    for gi, (op, a, b, c) in enumerate(cir.gates):
        o = cir.n_inputs + gi
        if op == NOT:
            v[o] = not v[a]
        elif op == COPY:
            v[o] = v[a]
        elif op == MUX:
            v[o] = v[b] if v[a] else v[c]
        else:
            v[o] = f[op](bool(v[a]), bool(v[b]))
    return v


simulate_ext = simulate


def simulate_mk(cir, input_bits):
    """simulate() plus the 3-gen three-input AND exactly as the reference defines it (3gen_mk_gates.jl:55-64): bootstrap of
    -1/4 + x + y + z.  Three false operands give the phase -5/8 = +3/8 (mod 1), so the reference's gate answers TRUE there -- it is a
    correct AND only when at least one operand is true.  The simulation mirrors the gate, not the name."""
    from . import AND3
    v = np.zeros(cir.n_wires(), bool)
    v[:cir.n_inputs] = np.asarray(input_bits, bool)
    for gi, (op, a, b, c) in enumerate(cir.gates):
        o = cir.n_inputs + gi
        if op == AND3:
            v[o] = (v[a] and v[b] and v[c]) or not (v[a] or v[b] or v[c])
        else:
            one = Circuit()
            one.n_inputs = o
            one.gates = [(op, a, b, c)]
            v[o] = simulate(one, v[:o])[o]
    return v


# ---- evaluator --------------------------------------------------------------------------------------------------------
def evaluate(ck, cir, input_records, stats=None):
    """Run the DAG on the engine.  input_records: int32[n_inputs][n+1].  Returns int32[n_wires][n+1].
    Single-key contexts use the native scheduler / executor (thfhe_dag_run: wires stay in HBM, no host round trip per level);
    multi-key contexts go level by level through thfhe_mk_gates_mixed (evaluate_levels)."""
    if hasattr(ck, "dag_run"):
        vals, st = ck.dag_run(input_records, np.array(cir.gates, np.int32).reshape(-1, 4))
        if stats is not None:
            stats.update(cir.census(), **st)
        return vals
    return evaluate_levels(ck, cir, input_records, stats)


def evaluate_batch(ck, cir, input_records, out_wires=None, stats=None):
    """`instances` evaluations of one DAG side by side.  input_records: int32[instances][n_inputs][words]; out_wires: wire ids to return
    (None: every wire).  Returns int32[instances][len(out_wires) or n_wires][words].  Contexts with the native executor use
    thfhe_dag_run_batch / thfhe_mk_dag_run_batch (wire tables stay in HBM); others are driven level by level from the host, a level's
    call holding the gates of all instances."""
    x = np.ascontiguousarray(input_records, np.int32)
    Q, n_in, words = x.shape
    assert n_in == cir.n_inputs
    if hasattr(ck, "dag_run_batch"):
        sel = None if out_wires is None else np.asarray(out_wires, np.int32)
        out, st = ck.dag_run_batch(x, np.array(cir.gates, np.int32).reshape(-1, 4), sel)
        if stats is not None:
            stats.update(cir.census(), **st)
        return out if out_wires is not None else np.concatenate([x, out], axis=1)
    from . import AND3 as _AND3
    vals = np.zeros((Q, cir.n_wires(), words), np.int32)
    vals[:, :n_in] = x
    gates, base, launches = cir.gates, cir.n_inputs, 0
    flat = lambda a: a.reshape(-1, words)
    for level in cir.levels():
        if gates[level[0]][0] in (NOT, COPY):
            for g in level:
                src = vals[:, gates[g][1]]
                vals[:, base + g] = (0 - src.astype(np.int64)).astype(np.int32) if gates[g][0] == NOT else src
            continue
        for cls in ("two", "mux", "and3"):
            G = [g for g in level if (gates[g][0] == MUX) == (cls == "mux") and (gates[g][0] == _AND3) == (cls == "and3")]
            if not G:
                continue
            a, b = (flat(vals[:, [gates[g][q] for g in G]]) for q in (1, 2))
            if cls == "two":
                r = ck.gates_mixed(np.tile(np.array([gates[g][0] for g in G], np.int32), Q), a, b)
            else:
                r = ck.gates(MUX if cls == "mux" else _AND3, a, b, flat(vals[:, [gates[g][3] for g in G]]))
            vals[:, base + np.array(G)] = r.reshape(Q, len(G), words)
            launches += 1
    if stats is not None:
        stats.update(cir.census(), launches=launches, instances=Q)
    return vals if out_wires is None else vals[:, np.asarray(out_wires, np.int64)]


def evaluate_levels(ck, cir, input_records, stats=None):
    """The same schedule driven from the host: one host-buffer call per level (works for single-key and multi-key contexts)."""
    from . import AND3 as _AND3
    words = ck.words
    vals = np.zeros((cir.n_wires(), words), np.int32)
    vals[:cir.n_inputs] = np.asarray(input_records, np.int32).reshape(cir.n_inputs, words)
    gates = cir.gates
    base = cir.n_inputs
    launches = 0
    for level in cir.levels():
        op0 = gates[level[0]][0]
        if op0 in (NOT, COPY):
            for g in level:   # in gate order: a NOT may read another NOT of the same depth
                src = vals[gates[g][1]]
                vals[base + g] = (0 - src.astype(np.int64)).astype(np.int32) if gates[g][0] == NOT else src
            continue
        two = [g for g in level if gates[g][0] not in (MUX, _AND3)]
        mux = [g for g in level if gates[g][0] == MUX]
        and3 = [g for g in level if gates[g][0] == _AND3]   # 3-gen three-input AND: its own gate class (thfhe_mk_gates)
        if and3:
            a, b, c = (vals[[gates[g][q] for g in and3]] for q in (1, 2, 3))
            vals[base + np.array(and3)] = ck.gates(_AND3, a, b, c)
            launches += 1
        if two:
            ops = np.array([gates[g][0] for g in two], np.int32)
            a = vals[[gates[g][1] for g in two]]
            b = vals[[gates[g][2] for g in two]]
            vals[base + np.array(two)] = ck.gates_mixed(ops, a, b)
            launches += 1
        if mux:
            a = vals[[gates[g][1] for g in mux]]
            b = vals[[gates[g][2] for g in mux]]
            c = vals[[gates[g][3] for g in mux]]
            vals[base + np.array(mux)] = ck.gates(MUX, a, b, c)
            launches += 1
    if stats is not None:
        stats.update(cir.census(), launches=launches)
    return vals
