"""Developer diagnostic: where the ring kernel's cycles go.  Needs the stamp build (make -C torus-fhe_amd/csrc stamps), loaded through
THFHE_HIP_LIB; prints per-phase cycle totals per CMux averaged over the waves of the first workgroups.
    make -C torus-fhe_amd/csrc stamps && THFHE_HIP_LIB=torus-fhe_amd/lib/libthfhe_hip_stamps.so python tools/ring_stamps.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import thfhe
from thfhe import keygen

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = thfhe.make_params("SK-128")
K = keygen.SecretKeySet(p, seed=1)
ck = thfhe.CloudKey(p, K.bk, K.ksk)
COOP = os.environ.get("STAMP_KERNEL", "ring") == "coop"
ck.set_coop_threshold(1 << 20 if COOP else 0)
ck.set_ring4_threshold(0)
rng = np.random.default_rng(0)
xa, xb = K.encrypt(rng.integers(0, 2, B), 1), K.encrypt(rng.integers(0, 2, B), 2)
da, db, do = ck.device_records(B), ck.device_records(B), ck.device_records(B)
da.upload(xa); db.upload(xb); ck.reserve(B); ck.set_profiling(True)
for _ in range(2):
    ck.gates_dev(thfhe.NAND, da, db, None, do, B); ck.sync()
t = ck.last_timings()
L = thfhe.lib()
nwg = min(B if COOP else (B + 7) // 8, 2048)
buf = np.zeros(nwg * 8 * 8, np.uint64)
L.thfhe_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.thfhe_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
NS = 6 if COOP else 4
st = buf.reshape(nwg, 8, 8)[:, :, :NS].astype(np.float64) / p.n   # cycles per CMux
names = (["F: rotate+decompose+fwd FFT+publish", "wait at barrier 1", "M: spectra reads + MAC", "hand-off / prefetch issue / I: inverse+atomics",
          "wait at barrier 2", "wait at barrier 3"] if COOP else
         ["rotate+decompose+fwd FFT (6 rows)", "chunk barriers (30)", "key reads + MAC (24 chunks)", "4 inverse FFT + acc update"])
print(f"batch {B}: blind rotate {t['blind_rotate_ms']:.3f} ms; s_memtime ticks per CMux per wave (mean over {nwg} workgroups x 8 waves; 100 MHz ticks x clock ratio):")
tot = st.sum(axis=2).mean()
for q, nm in enumerate(names):
    print(f"  {nm:40s} {st[:, :, q].mean():10.1f}  ({100 * st[:, :, q].mean() / tot:5.1f} %)   per-wave min {st[:, :, q].min():9.1f} max {st[:, :, q].max():9.1f}")
print(f"  total {tot:.1f}")
for w in range(8):
    print(f"  wave {w}: " + "  ".join(f"{st[:, w, q].mean():9.1f}" for q in range(NS)))
