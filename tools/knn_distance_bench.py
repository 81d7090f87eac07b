"""BASELINE.json configs[3]: the distance phase of the reference's KNN application
(src/KNN_medical_data.cpp:681-691: for each of 5 train rows, distance_bw_data(test row, train row) over 12 columns of
32-bit values) as levelised gate DAGs.  Data = first 6 records of the reference's test/bootstrap_modules/data1.csv
(tests/golden/data1.csv).  Checks the decrypted Manhattan distances against plaintext.

Multi-GPU: the reference parallelises over the train rows with OpenMP; here the train rows are sharded over the ranks of
`python -m torch.distributed.run --nproc-per-node N tools/knn_distance_bench.py` (rank r takes rows r, r+N, ...; keys are
replicated, there is no data-path collective -- the per-row distances are gathered at the end)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
sys.path.insert(0, ROOT)
import bench  # rank / barrier plumbing shared with the headline benchmark (imports torch first when WORLD_SIZE > 1)
rank, world, barrier, max_reduce, backend = bench.dist_setup(int(os.environ.get("WORLD_SIZE", "1")))
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
my_rows = [j for j in range(5) if j % world == rank]
train_w = [[cir.inputs(NB) for _ in range(ncol)] for _ in my_rows]
all_zero, all_one, lsb_one = cir.inputs(NB), cir.inputs(NB), cir.inputs(NB)
zero = cir.inputs(1)[0]
outs = [Cc.distance_bw_data(cir, test_w, train_w[q], all_zero, all_one, lsb_one, zero) for q in range(len(my_rows))]
cs = cir.census() if my_rows else dict(gates=0, rotations=0, depth=0)
print(f"rank {rank}/{world}: train rows {my_rows}, DAG {cs}", file=sys.stderr, flush=True)

p = thfhe.make_params("SK-128")
K = keygen.SecretKeySet(p, seed=0x5EED0001)
ndev = thfhe.lib().thfhe_device_count()
ck = thfhe.CloudKey(p, K.bk, K.ksk, device=int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1))
bits = lambda v: [(v >> (NB - 1 - i)) & 1 for i in range(NB)]
plain = sum((bits(test[c]) for c in range(ncol)), [])
for j in my_rows:
    plain += sum((bits(train[j][c]) for c in range(ncol)), [])
plain += [0] * NB + [1] * NB + bits(1) + [0]
inputs = K.encrypt(np.array(plain), seed=0x5EED0002 + rank)
barrier()
t0 = time.time()
stats = {}
vals = Cc.evaluate(ck, cir, inputs, stats) if my_rows else None
barrier()
dt = max_reduce(time.time() - t0)
ok = 1.0
for q, j in enumerate(my_rows):
    got = 0
    for b in K.decrypt(vals[outs[q]]):
        got = (got << 1) | int(b)
    exp = sum(abs(test[c] - train[j][c]) for c in range(1, ncol)) % (1 << NB)
    ok = min(ok, float(got == exp))
    print(f"rank {rank}: train row {j}: distance {got} expected {exp}", file=sys.stderr, flush=True)
all_ok = -max_reduce(-ok) == 1.0                      # min over ranks
tot_gates = 5 * 12 * (732 + 159) if NB == 32 else None
if rank == 0:
    import json
    per_row = 12 * (2 * (NB + 2 * (5 * NB - 1)) + NB + (5 * NB - 1))   # distance + accumulating adder per column
    print(json.dumps(dict(workload="KNN distance phase, 5 train rows x 12 columns x %d bit" % NB, n_gpus=world, gates=5 * per_row,
                          seconds=dt, gates_per_s=5 * per_row / dt, correct=bool(all_ok), levels_rank0=cs["depth"])), flush=True)
