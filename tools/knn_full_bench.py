"""BASELINE.json configs[3]: the reference's whole KNN decision for one test record (src/KNN_medical_data.cpp:676-732) through the
product function thfhe.circuits.knn_decision_sharded: Manhattan distances to the 5 train rows (12 columns x 32 bit), the MUX copy of
the train rows, sort_with_distance (5 passes of adjacent compare-swaps, records as payload), the vote over the label column of the
K = 5 nearest and the decision bit -- ~7.9e4 two-input gates + ~2.3e4 MUX = ~1.26e5 blind rotations.
Data = first 6 records of the reference's test/bootstrap_modules/data1.csv (tests/golden/data1.csv); the decrypted distances, sorted
order, vote count and decision are checked against plaintext.

    python tools/knn_full_bench.py                       # one GPU, one test record
    python tools/knn_full_bench.py --queries 64          # the reference's loop over test records (:676) as one batched evaluation
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/knn_full_bench.py
Multi-GPU: phase 1 (distances + copies) is dealt over the ranks by train row, as the reference's `#pragma omp parallel for` (:681) does
over threads; one all-reduce (RCCL) gathers the rows and phase 2 (the sequential sort / vote chain) runs on every rank's key replica.
With --queries Q > 1 the Q test records (rows NTRAIN .. NTRAIN+Q-1 of data1.csv) are instances of the same DAGs walking the levels side by
side (thfhe_dag_run_batch) and the ranks split them BY QUERY (both phases, one all-reduce of the results at the end)."""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=1)
ARGS = ap.parse_args()
Q = ARGS.queries
NB = int(os.environ.get("KNN_BITS", "32"))
NCOL = int(os.environ.get("KNN_COLS", "14"))
NTRAIN = int(os.environ.get("KNN_TRAIN", "5"))
rows = []
with open(os.path.join(ROOT, "tests", "golden", "data1.csv")) as f:
    next(f)
    for line in f:
        r = [int(float(w)) for w in line.strip().split(",")]   # the reference parses every field with `ss >> x` into an int
        rows.append(r[:NCOL - 1] + [r[13]])                    # label column last
        if len(rows) == NTRAIN + Q:
            break
mask = (1 << NB) - 1
rows = [[v & mask for v in r] for r in rows]
train, test, tests = rows[:NTRAIN], rows[NTRAIN], rows[NTRAIN:]
threshold = NTRAIN // 2
bits = lambda v: [(v >> (NB - 1 - i)) & 1 for i in range(NB)]
from_bits = lambda b: int("".join("1" if x else "0" for x in b), 2)

p = thfhe.make_params("SK-128")
KS = keygen.SecretKeySet(p, seed=0x5EED0001)
ndev = thfhe.lib().thfhe_device_count()
local = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
ck = thfhe.CloudKey(p, KS.bk, KS.ksk, device=local)
words = p.n + 1
enc = lambda vals, seed: KS.encrypt(np.array(sum((bits(v) for v in vals), [])), seed=seed).reshape(len(vals), NB, words)   # same on every rank
e_test = enc(test, 0x5EED0100)
e_train = np.stack([enc(r, 0x5EED0200 + j) for j, r in enumerate(train)])
thr, az, ao, lo = (enc([v], 0x5EED0300 + q)[0] for q, v in enumerate((threshold, 0, mask, 1)))
zero = KS.encrypt(np.array([0]), seed=0x5EED0400)[0]
plan = Cc.KnnPlan(NB, NCOL, NTRAIN)
red = None
if world > 1:
    import torch
    red = Cc.torch_all_reduce(torch.device("cuda", local) if backend == "nccl" else None)

if Q > 1:
    e_tests = np.stack([e_test] + [enc(t, 0x5EED0100 + q) for q, t in enumerate(tests) if q > 0])
    barrier()
    t0 = time.time()
    st = {}
    res = Cc.knn_decisions_batched(ck, plan, e_tests, e_train, thr, az, ao, lo, zero, rank, world, red, st)
    barrier()
    dt = max_reduce(time.time() - t0)
    if rank == 0:
        dec = lambda recs: from_bits(KS.decrypt(recs))
        votes = sum(r[NCOL - 1] for r in train)
        ok = True
        for q, t in enumerate(tests):
            d_plain = [sum(abs(t[c] - r[c]) for c in range(1, NCOL - 1)) & mask for r in train]
            ok &= [dec(d) for d in res["dists"][q]] == d_plain and [dec(d) for d in res["sorted_dists"][q]] == sorted(d_plain)
            ok &= dec(res["count"][q]) == votes and bool(KS.decrypt(res["decision"][q][None])[0]) == (votes > threshold)
        s1, s2 = st["phase1"], st["phase2"]
        mine = max(len(st["my_queries"]), 1)
        once = (s1.get("copies_once") or {}).get("rotations") or 0          # the train rows' MUX copies: evaluated once per rank, not once per test record
        per_query = (s1.get("rotations", 0) - once + s2.get("rotations", 0)) // mine
        rot = Q * per_query + world * once                                   # rotations actually evaluated by all ranks
        print(json.dumps(dict(workload=f"{Q} KNN decisions (test records) as instances of one DAG, {NTRAIN} train rows x {NCOL} columns x {NB} bit (reference circuit)",
                              n_gpus=world, queries=Q, blind_rotations=rot, reference_rotations=Q * (per_query + once), seconds=dt, rotations_per_s=rot / dt, seconds_per_decision=dt / Q,
                              levels=dict(phase1=s1.get("levels"), phase2=s2.get("levels")), phase_seconds=dict(phase1=s1.get("seconds"), phase2=s2.get("seconds")),
                              launches=dict(phase1=s1.get("launches"), phase2=s2.get("launches")), sharding="by query", correct=bool(ok))), flush=True)
    sys.exit(0)

barrier()
t0 = time.time()
st = {}
res = Cc.knn_decision_sharded(ck, plan, e_test, e_train, thr, az, ao, lo, zero, rank, world, red, st)
barrier()
dt = max_reduce(time.time() - t0)

if rank == 0:
    dec = lambda recs: from_bits(KS.decrypt(recs))
    d_plain = [sum(abs(test[c] - r[c]) for c in range(1, NCOL - 1)) & mask for r in train]
    got_d = [dec(d) for d in res["dists"]]
    got_sorted = [dec(d) for d in res["sorted_dists"]]
    votes = sum(r[NCOL - 1] for r in train)
    got_count = dec(res["count"])
    got_dec = bool(KS.decrypt(res["decision"][None])[0])
    ok = got_d == d_plain and got_sorted == sorted(d_plain) and got_count == votes and got_dec == (votes > threshold)
    s1, s2 = st["phase1"], st["phase2"]
    rot = s2.get("rotations", 0) + NTRAIN * (s1.get("rotations", 0) // max(len(st["my_rows"]), 1))
    print(json.dumps(dict(workload=f"KNN decision, {NTRAIN} train rows x {NCOL} columns x {NB} bit (reference circuit)", n_gpus=world,
                          blind_rotations=rot, seconds=dt, rotations_per_s=rot / dt,
                          levels=dict(phase1=s1.get("levels"), phase2=s2.get("levels")), phase_seconds=dict(phase1=s1.get("seconds"), phase2=s2.get("seconds")), launches=dict(phase1=s1.get("launches"), phase2=s2.get("launches")),
                          distances=got_d, sorted_distances=got_sorted, count=got_count, decision=got_dec,
                          label_of_test_record=test[NCOL - 1], correct=bool(ok))), flush=True)
