"""BASELINE.json configs[3]: the reference's whole KNN decision for one test record (src/KNN_medical_data.cpp:676-732) as
levelised gate DAGs on the engine: Manhattan distances to the 5 train rows (12 columns x 32 bit), the MUX copy of the train
rows, sort_with_distance (5 passes of adjacent compare-swaps, records as payload), the vote over the label column of the K = 5
nearest and the decision bit -- ~7.9e4 two-input gates + ~2.3e4 MUX = ~1.26e5 blind rotations.
Data = first 6 records of the reference's test/bootstrap_modules/data1.csv (tests/golden/data1.csv); every decrypted
intermediate (distances, sorted order, vote count, decision) is checked against plaintext.

    python tools/knn_full_bench.py                       # one GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/knn_full_bench.py
Multi-GPU: phase 1 (distances + copies) is sharded over the ranks by train row, as the reference's `#pragma omp parallel for`
(:681) does over threads; the rows are all-gathered (RCCL, 5 x 15 x 32 records of 2.5 KB) and phase 2 (the sequential sort /
vote chain) is evaluated by every rank on its own replica of the keys."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
sys.path.insert(0, ROOT)
import bench  # rank / barrier plumbing shared with the headline benchmark (imports torch first when WORLD_SIZE > 1)
rank, world, barrier, max_reduce, backend = bench.dist_setup(int(os.environ.get("WORLD_SIZE", "1")))
import thfhe
from thfhe import keygen, circuits as Cc

NB = int(os.environ.get("KNN_BITS", "32"))
NCOL = int(os.environ.get("KNN_COLS", "14"))
NTRAIN = int(os.environ.get("KNN_TRAIN", "5"))
rows = []
with open(os.path.join(ROOT, "tests", "golden", "data1.csv")) as f:
    next(f)
    for line in f:
        r = [int(float(w)) for w in line.strip().split(",")]   # the reference parses every field with `ss >> x` into an int
        rows.append(r[:NCOL - 1] + [r[13]])                    # label column last
        if len(rows) == NTRAIN + 1:
            break
mask = (1 << NB) - 1
rows = [[v & mask for v in r] for r in rows]
train, test = rows[:NTRAIN], rows[NTRAIN]
K = NTRAIN
threshold = K // 2
bits = lambda v: [(v >> (NB - 1 - i)) & 1 for i in range(NB)]
from_bits = lambda b: int("".join("1" if x else "0" for x in b), 2)

p = thfhe.make_params("SK-128")
KS = keygen.SecretKeySet(p, seed=0x5EED0001)
ndev = thfhe.lib().thfhe_device_count()
ck = thfhe.CloudKey(p, KS.bk, KS.ksk, device=int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1))

# ---- phase 1: distances + MUX copies of my train rows ---------------------------------------------------------------------
my_rows = [j for j in range(NTRAIN) if j % world == rank]
c1 = Cc.Circuit()
test_w = [c1.inputs(NB) for _ in range(NCOL)]
train_w = [[c1.inputs(NB) for _ in range(NCOL)] for _ in my_rows]
all_zero, all_one, lsb_one = c1.inputs(NB), c1.inputs(NB), c1.inputs(NB)
zero = c1.inputs(1)[0]
d_w = [Cc.distance_bw_data(c1, test_w[:NCOL - 1], tw[:NCOL - 1], all_zero, all_one, lsb_one, zero) for tw in train_w]
cp_w = [[Cc.copy_through_mux(c1, all_one, w) for w in tw] for tw in train_w]
plain = sum((bits(v) for v in test), [])
for j in my_rows:
    plain += sum((bits(v) for v in train[j]), [])
plain += [0] * NB + [1] * NB + bits(1) + [0]
in1 = KS.encrypt(np.array(plain), seed=0x5EED0002 + rank)

# ---- phase 2: sort + vote + decision (every rank, on the gathered rows) -----------------------------------------------------
c2 = Cc.Circuit()
rows_w = [[c2.inputs(NB) for _ in range(NCOL)] for _ in range(NTRAIN)]
dist_w = [c2.inputs(NB) for _ in range(NTRAIN)]
thr_w, z2, o2, l2 = c2.inputs(NB), c2.inputs(NB), c2.inputs(NB), c2.inputs(NB)
zero2 = c2.inputs(1)[0]
srows, sdists = Cc.sort_with_distance(c2, rows_w, dist_w, z2, o2, l2, zero2)
count = list(z2)
for j in range(K):
    count, _ = Cc.full_adder(c2, count, srows[j][NCOL - 1], zero2)
diff = Cc.difference(c2, thr_w, count, o2, l2, zero2)
decision = c2.gate(thfhe.XOR, diff[0], z2[0])
const2 = KS.encrypt(np.array(bits(threshold) + [0] * NB + [1] * NB + bits(1) + [0]), seed=0x5EED0777)   # same on every rank

cs1, cs2 = (c1.census() if my_rows else dict(gates=0, rotations=0, depth=0, mux=0)), c2.census()
print(f"rank {rank}/{world}: rows {my_rows}; phase-1 DAG {cs1}; phase-2 DAG {cs2}", file=sys.stderr, flush=True)

barrier()
t0 = time.time()
st1, st2 = {}, {}
vals1 = Cc.evaluate(ck, c1, in1, st1) if my_rows else None
t1 = time.time()
words = p.n + 1
mine = np.zeros((NTRAIN, NCOL + 1, NB, words), np.int32)      # [row][words..., distance][bit][record]
for q, j in enumerate(my_rows):
    for c in range(NCOL):
        mine[j, c] = vals1[cp_w[q][c]]
    mine[j, NCOL] = vals1[d_w[q]]
if world > 1:   # every row was produced by exactly one rank: a sum all-reduce is the all-gather
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(mine)
    if backend == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    mine = t.cpu().numpy()
in2 = np.concatenate([mine[:, :NCOL].reshape(-1, words), mine[:, NCOL].reshape(-1, words), const2])
vals2 = Cc.evaluate(ck, c2, in2, st2)
barrier()
dt = max_reduce(time.time() - t0)

if rank == 0:
    dec = lambda wires, v: from_bits(KS.decrypt(v[wires]))
    d_plain = [sum(abs(test[c] - r[c]) for c in range(1, NCOL - 1)) & mask for r in train]
    got_d = [from_bits(KS.decrypt(mine[j, NCOL])) for j in range(NTRAIN)]
    got_sorted = [dec(w, vals2) for w in sdists]
    order = sorted(range(NTRAIN), key=lambda j: d_plain[j])
    got_ids = [dec(srows[j][0], vals2) for j in range(NTRAIN)]
    votes = sum(r[NCOL - 1] for r in train)
    got_count = dec(count, vals2)
    got_dec = bool(KS.decrypt(vals2[[decision]])[0])
    ok = got_d == d_plain and got_sorted == sorted(d_plain) and got_count == votes and got_dec == (votes > threshold)
    ids_ok = sorted(got_ids) == sorted(train[j][0] for j in range(NTRAIN))
    rot = cs2["rotations"] + NTRAIN * (cs1["rotations"] // max(len(my_rows), 1) if my_rows else 0)
    print(json.dumps(dict(workload=f"KNN decision, {NTRAIN} train rows x {NCOL} columns x {NB} bit (reference circuit)", n_gpus=world,
                          blind_rotations=rot, seconds=dt, phase1_seconds=t1 - t0, rotations_per_s=rot / dt,
                          levels=dict(phase1=cs1["depth"], phase2=cs2["depth"]), launches=dict(phase1=st1.get("launches"), phase2=st2.get("launches")),
                          distances=got_d, sorted_distances=got_sorted, sorted_ids=got_ids, count=got_count, decision=got_dec,
                          label_of_test_record=test[NCOL - 1], correct=bool(ok and ids_ok))), flush=True)
