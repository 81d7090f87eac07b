"""Throughput of the CCS multi-key NAND (the reference's mk_gate_nand, test/runtests.jl:62-102) on one MI355X:
python tools/ccs_bench.py [batch ...]   (host-buffer API: PCIe transfers of the records are inside the timed call)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import thfhe
from thfhe import keygen
name = os.environ.get("CCS_SET", "CCS2")
p = thfhe.make_params(name)
K = keygen.CCSSecretKeySet(p)
ck = thfhe.CCSCloudKey(p, K.bk, K.pk, K.crs, K.ksk, device=0)
for B in [int(x) for x in sys.argv[1:]] or [256]:
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 2, B), rng.integers(0, 2, B)
    xa, xb = K.encrypt(a, 1), K.encrypt(b, 2)
    thfhe.mk_gate_nand(ck, xa[:8], xb[:8])
    t0 = time.perf_counter()
    out = thfhe.mk_gate_nand(ck, xa, xb)
    dt = time.perf_counter() - t0
    ok = bool((K.decrypt(out) == ~(a.astype(bool) & b.astype(bool))).all())
    print(json.dumps(dict(workload=f"{B} mk_gate_nand, CCS scheme {name} (P={p.parties}, n={p.n}, l={p.l}, Bgbit={p.Bgbit})", seconds=dt,
                          gates_per_s=B / dt, decrypt_ok=ok)), flush=True)
