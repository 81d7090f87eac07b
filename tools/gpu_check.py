"""Developer GPU check (run on the MI355X box via gpurun): parity of the HIP path against the CPU oracle on
a handful of gates, then a timing of a 4096-gate NAND batch.  Not part of the product or the test suite."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import thfhe  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--set", default="SK-128")
ap.add_argument("--parity", type=int, default=8)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

t0 = time.time()
p = O.make_params(args.set)
sg = O.SIGMAS[args.set]
K = O.SKKeys(p, 0x5EED0001, sg["bk"], sg["ks"])
print(f"keygen {time.time()-t0:.2f}s", flush=True)
t0 = time.time()
ck = thfhe.CloudKey(thfhe.make_params(args.set), K.bk, K.ksk)
print(f"ctx_create {time.time()-t0:.2f}s", flush=True)
orc = O.Oracle(p, K.bk, K.ksk)

rng = np.random.default_rng(7)
G = args.parity
ba, bb, bc = rng.integers(0, 2, G), rng.integers(0, 2, G), rng.integers(0, 2, G)
ca, cb, cc = K.encrypt_bits(ba, sg["lwe"], 11), K.encrypt_bits(bb, sg["lwe"], 12), K.encrypt_bits(bc, sg["lwe"], 13)

# 1. bootstrap_wo_keyswitch parity
t0 = time.time()
u_gpu = ck.bootstrap_wo_keyswitch(ca[:2])
print(f"first GPU call {time.time()-t0:.2f}s", flush=True)
u_ref = np.stack([orc.bootstrap_wo_keyswitch(ca[i]) for i in range(2)])
print("bootstrap_wo_keyswitch bit-exact:", np.array_equal(u_gpu, u_ref), "mismatches", int((u_gpu != u_ref).sum()), flush=True)
ks_gpu = ck.keyswitch(u_ref)
ks_ref = np.stack([orc.keyswitch(u_ref[i]) for i in range(2)])
print("keyswitch bit-exact:", np.array_equal(ks_gpu, ks_ref), int((ks_gpu != ks_ref).sum()), flush=True)

# 2. all gates
ok_all = True
for op in range(10):
    g = ck.gates(op, ca, cb)
    r = orc.gates(op, ca, cb)
    dec = K.decrypt_bits(g)
    exp = np.array([O.TRUTH[op](bool(x), bool(y)) for x, y in zip(ba, bb)])
    ok = np.array_equal(g, r)
    ok_all &= ok and np.array_equal(dec, exp)
    print(f"gate {op}: bit-exact {ok}, decrypt ok {np.array_equal(dec, exp)}", flush=True)
g = ck.gates(thfhe.MUX, ca, cb, cc)
r = orc.gates(O.MUX, ca, cb, cc)
exp = np.where(ba == 1, bb, bc).astype(bool)
print("MUX bit-exact", np.array_equal(g, r), "decrypt ok", np.array_equal(K.decrypt_bits(g), exp), flush=True)
g = ck.gates(thfhe.NOT, ca)
print("NOT ok", np.array_equal(g, (-ca.astype(np.int64)).astype(np.int32)), flush=True)
print("PARITY", "PASS" if ok_all else "FAIL", flush=True)

# 3. timing
B = args.batch
bits_a, bits_b = rng.integers(0, 2, B), rng.integers(0, 2, B)
xa, xb = K.encrypt_bits(bits_a, sg["lwe"], 21), K.encrypt_bits(bits_b, sg["lwe"], 22)
da, db, do = ck.device_records(B), ck.device_records(B), ck.device_records(B)
da.upload(xa)
db.upload(xb)
ck.reserve(B)
ck.set_profiling(True)
for rep in range(args.reps):
    t0 = time.time()
    ck.gates_dev(thfhe.NAND, da, db, None, do, B)
    ck.sync()
    dt = time.time() - t0
    print(f"rep {rep}: {B} NAND in {dt*1e3:.1f} ms -> {B/dt:.0f} gates/s ; kernels {ck.last_timings()}", flush=True)
out = do.download((B, p.n + 1))
dec = K.decrypt_bits(out)
print("batch decrypt errors:", int((dec != ~(bits_a.astype(bool) & bits_b.astype(bool))).sum()), flush=True)
ph = K.phases(out) / 2.0**32
print("max |phase -+ 1/8| =", float(np.abs(np.abs(ph) - 0.125).max()), flush=True)
chk = orc.gates(O.NAND, xa[:4], xb[:4])
print("batch[0:4] bit-exact vs oracle:", np.array_equal(out[:4], chk), flush=True)
