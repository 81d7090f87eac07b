"""KMS scheme (mk_gate_nand_new) timing at the reference's 2-party parameters (mktfhe_parameters_2party_new, mk_api.jl:12-20):
n = 560, N = 2048, Torus64, gsw l = 3 / Bgbit 13, lev 2 / 7, uni 2 / 13.  Prints one JSON line; every output is decrypted and checked."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import thfhe
from thfhe import keygen, kms
name = sys.argv[1] if len(sys.argv) > 1 else "KMS2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p = thfhe.make_kms_params(name, **(dict(n=n) if n else {}))
t0 = time.time()
K = keygen.KMSSecretKeySet(p, seed=1)
t_key = time.time() - t0
ck = kms.KMSCloudKey(p, K.gsw, K.uni, K.pk, K.crs, K.ksk, device=0)
rng = np.random.default_rng(0)
a, b = rng.integers(0, 2, B), rng.integers(0, 2, B)
xa, xb = K.encrypt(a, 1), K.encrypt(b, 2)
kms.mk_gate_nand_new(ck, xa[:4], xb[:4])   # warm-up
bar = kms.modswitch(xa, p.N)
t0 = time.time(); lev = ck.tlev_rotate(0, bar[:, :p.n]); t_rot = time.time() - t0
t0 = time.time()
out = kms.mk_gate_nand_new(ck, xa, xb)
dt = time.time() - t0
ok = bool(np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool))))
kms.mk_gate_nand_new(ck, xa[:4], xb[:4], fast_boot=True)
t0 = time.time()
outf = kms.mk_gate_nand_new(ck, xa, xb, fast_boot=True)      # mk_blind_rotate_new_v2: one RLWE rotation instead of the first party's TLev
dtf = time.time() - t0
okf = bool(np.array_equal(K.decrypt(outf), ~(a.astype(bool) & b.astype(bool))))
print(json.dumps(dict(workload=f"{B} mk_gate_nand_new, {name} (P={p.parties}, n={p.n}, N={p.N}, gsw {p.l_gsw}/{p.bg_gsw}, lev {p.l_lev}/{p.bg_lev}, uni {p.l_uni}/{p.bg_uni})",
                      gates_per_s=B / dt, seconds=dt, fast_boot_gates_per_s=B / dtf, fast_boot_all_decrypt_correct=okf, tlev_rotate_one_party_s=t_rot, host_keygen_s=t_key, all_decrypt_correct=ok)), flush=True)
