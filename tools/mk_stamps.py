"""Developer diagnostic: per-phase cycles of one CMux step of the N = 2048 multi-key kernel (stamp build, see tools/ring_stamps.py)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import thfhe
from thfhe import keygen
name = sys.argv[1] if len(sys.argv) > 1 else "MK4-N2048"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = thfhe.make_params(name)
K = keygen.MKSecretKeySet(p, seed=1, sigma_lwe=2.0**-13.26, sigma_bk=2.0**-30.70)
ck = thfhe.MKCloudKey(p, K.bk, K.ksk)
rng = np.random.default_rng(0)
xa, xb = K.encrypt(rng.integers(0, 2, B), 1), K.encrypt(rng.integers(0, 2, B), 2)
ck.set_profiling(True)
PAIR = len(sys.argv) > 3 and sys.argv[3] == "pair"   # two gates per workgroup (mk_blind_rotate_pair2k_kernel)
if PAIR:
    ck.set_pair_threshold(0)
for _ in range(2):
    out = ck.gates(thfhe.NAND, xa, xb)
t = ck.last_timings()
L = thfhe.lib()
buf = np.zeros(B * 64, np.uint64)
L.thfhe_debug_read_stamps_mk.argtypes = [C.c_void_p, C.c_size_t]
assert L.thfhe_debug_read_stamps_mk(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
steps = p.parties * p.n
st = buf.reshape(B, 8, 8)[:B // 2 if PAIR else B, :, :6].astype(np.float64) / steps
names = ["F: digits + 2 half transforms + publish", "wait barrier 1", "M: 4l chunk multiplies (+ key requests)", "I: 2 half inverses + atomics", "wait barrier 2", "wait barrier 3"]
if PAIR:
    names = ["F: digits + half transform, both half passes", "wait barriers after F", "M: 2 x 2l chunk multiplies into two gates", "I: 4 half inverses + atomics", "wait barriers after M", "wait barrier after I"]
print(f"{name} batch {B}: blind rotate {t['blind_rotate_ms']:.3f} ms; cycles per CMux step and wave")
for q, nm in enumerate(names):
    print(f"  {nm:42s} mean {st[:, :, q].mean():9.1f}  " + " ".join(f"{st[:, w, q].mean():8.0f}" for w in range(8)))
print(f"  total {st.sum(axis=2).mean():.1f}")
