"""BASELINE.json configs[3] on one GPU: the distance phase of the reference's KNN application
(src/KNN_medical_data.cpp:681-691: for each of 5 train rows, distance_bw_data(test row, train row) over 12 columns of
32-bit values) as ONE levelised gate DAG.  Data = first 6 records of the reference's test/bootstrap_modules/data1.csv
(tests/golden/data1.csv).  Checks the decrypted Manhattan distances against plaintext."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import thfhe
from thfhe import keygen, circuits as Cc

NB = int(os.environ.get("KNN_BITS", "32"))
rows = []
with open(os.path.join(ROOT, "tests", "golden", "data1.csv")) as f:
    next(f)
    for line in f:
        rows.append([int(float(w)) for w in line.strip().split(",")])   # the reference parses every field with `ss >> x` into an int
        if len(rows) == 6:
            break
train, test = rows[:5], rows[5]
ncol = 13                                                                # distance_bw_data(..., col_size - 1 = 13, ...): columns 1..12

cir = Cc.Circuit()
test_w = [cir.inputs(NB) for _ in range(ncol)]
train_w = [[cir.inputs(NB) for _ in range(ncol)] for _ in range(5)]
all_zero, all_one, lsb_one = cir.inputs(NB), cir.inputs(NB), cir.inputs(NB)
zero = cir.inputs(1)[0]
outs = [Cc.distance_bw_data(cir, test_w, train_w[j], all_zero, all_one, lsb_one, zero) for j in range(5)]
cs = cir.census()
print("DAG:", cs, "max level width", max(len(l) for l in cir.levels()), flush=True)

p = thfhe.make_params("SK-128")
K = keygen.SecretKeySet(p, seed=0x5EED0001)
ck = thfhe.CloudKey(p, K.bk, K.ksk)
bits = lambda v: [(v >> (NB - 1 - i)) & 1 for i in range(NB)]
plain = sum((bits(test[c]) for c in range(ncol)), [])
for j in range(5):
    plain += sum((bits(train[j][c]) for c in range(ncol)), [])
plain += [0] * NB + [1] * NB + bits(1) + [0]
inputs = K.encrypt(np.array(plain), seed=0x5EED0002)
t0 = time.time()
stats = {}
vals = Cc.evaluate(ck, cir, inputs, stats)
dt = time.time() - t0
ok = True
for j in range(5):
    got = 0
    for b in K.decrypt(vals[outs[j]]):
        got = (got << 1) | int(b)
    exp = sum(abs(test[c] - train[j][c]) for c in range(1, ncol)) % (1 << NB)
    ok &= got == exp
    print(f"train row {j}: distance {got} expected {exp}", flush=True)
print(f"KNN distance phase: {cs['gates']} gates ({cs['rotations']} blind rotations) in {cs['depth']} levels / {stats['launches']} launches: "
      f"{dt:.2f} s -> {cs['gates']/dt:.0f} gates/s, {cs['rotations']/dt:.0f} rotations/s ; correct: {ok}", flush=True)
