"""Developer check: the ring-4096 rotation piece by piece against the oracle (extracted sample before the key switch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import torch
import oracle_lib as O
import thfhe
from thfhe.party_sharded import HipPartyBackend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = O.make_params("MK64-fft", n=n, parties=P)
s = O.SIGMAS["MK64-fft"]
K = O.MKKeys(p, 91, s["bk"], s["ks"])
orc = O.MKOracle(p, K.bk, K.ksk)
tp = thfhe.make_params(**p.as_dict())
be = HipPartyBackend(tp, (0, P), K.bk, K.ksk, device=0)
a = np.array([0, 1, 1])
ca = K.encrypt_bits(a, s["lwe"], 70)
ta = torch.from_numpy(ca).to("cuda:0")
with be.stream_context():
    bara, barb = be.prologue(-1, 0, ta, None, None)
    acc = be.rotate(bara, barb, thfhe.MU8_64, None)
    u = be.extract(acc)
    ks = be.keyswitch(u)
torch.cuda.synchronize()
print("bara", bara.cpu().numpy()[:, :P * n], "barb", barb.cpu().numpy())
u = u.cpu().numpy()
for g in range(len(a)):
    ref = orc.bootstrap_wo_keyswitch(ca[g])
    d = (u[g].astype(np.int64) - ref.astype(np.int64))
    print("gate", g, "extract equal:", np.array_equal(u[g], ref), "mismatches", int((d != 0).sum()), "of", len(ref), "first", np.nonzero(d)[0][:8], d[np.nonzero(d)[0][:4]])
    refk = orc.keyswitch(ref)
    print("   keyswitch equal:", np.array_equal(ks.cpu().numpy()[g], refk))
