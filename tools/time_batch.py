"""Developer timing: one NAND batch of a given size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import thfhe
from thfhe import keygen
p = thfhe.make_params("SK-128")
K = keygen.SecretKeySet(p, seed=1)
ck = thfhe.CloudKey(p, K.bk, K.ksk)
if os.environ.get("COOP_MAX"):     # kernel-choice thresholds (A/B runs): rotations up to COOP_MAX -> cooperative kernel, up to RING4_MAX -> four-wave ring
    ck.set_coop_threshold(int(os.environ["COOP_MAX"]))
if os.environ.get("RING4_MAX"):
    ck.set_ring4_threshold(int(os.environ["RING4_MAX"]))
for B in [int(x) for x in sys.argv[1:]] or [4096]:
    rng = np.random.default_rng(0)
    xa, xb = K.encrypt(rng.integers(0, 2, B), 1), K.encrypt(rng.integers(0, 2, B), 2)
    da, db, do = ck.device_records(B), ck.device_records(B), ck.device_records(B)
    da.upload(xa); db.upload(xb); ck.reserve(B); ck.set_profiling(True)
    for rep in range(3):
        ck.gates_dev(thfhe.NAND, da, db, None, do, B); ck.sync()
        t = ck.last_timings()
    import hashlib
    digest = hashlib.sha256(do.download((B, p.n + 1)).tobytes()).hexdigest()[:16]   # same inputs -> same bytes, whatever kernel variant ran
    print(f"batch {B} [out sha {digest}]: blind_rotate {t['blind_rotate_ms']:.3f} ms keyswitch {t['keyswitch_ms']:.3f} ms total {t['total_ms']:.3f} ms -> {B/t['total_ms']*1e3:.0f} gates/s", flush=True)
    for d in (da, db, do): d.free()
