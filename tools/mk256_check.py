"""mktfhe_parameters_256party_3gen (3-gen-mk-tfhe/src/mk_api.jl:304-310) at its REAL party count: 256 parties, N = 2048, l = 2, Bgbit = 18, ks 8/2,
with a reduced LWE dimension (default n = 24 of 740: the full key is 185 GB of spectra and 11 s per batch).  Keys are generated on the device,
every output is decrypted and checked; prints one JSON line.   python tools/mk256_check.py [n] [gates]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
import thfhe
from thfhe import keygen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = thfhe.make_params("MK256", n=n)
sig = thfhe.SIGMAS["MK256"]
t0 = time.time()
K = keygen.MKSecretKeySet(p, seed=0x5EED0001, sigma_lwe=sig["lwe"], sigma_bk=sig["bk"], sigma_ks=sig["ks"], device=0)
t_key = time.time() - t0
t0 = time.time()
ck = thfhe.MKCloudKey(p, K.bk, K.ksk, device=0)
t_ctx = time.time() - t0
rng = np.random.default_rng(0)
a, b = rng.integers(0, 2, B), rng.integers(0, 2, B)
xa, xb = K.encrypt(a, 1), K.encrypt(b, 2)
ck.gates(thfhe.NAND, xa[:2], xb[:2])
t0 = time.time()
out = ck.gates(thfhe.NAND, xa, xb)
dt = time.time() - t0
ok = bool(np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool))))
print(json.dumps(dict(workload=f"{B} mk_gate_nand_3gen, MK256 shape (P={p.parties}, n={p.n} of 740, N={p.N}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit})",
                      kernel=ck.rotation_kernel_name(B), cmuxes_per_gate=p.parties * p.n, gates_per_s=B / dt, seconds=dt, keygen_s=t_key, ctx_create_s=t_ctx,
                      all_decrypt_correct=ok)), flush=True)
