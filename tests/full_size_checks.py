"""Round 4: the parameter sets that earlier rounds only extrapolated or checked at reduced size, run at the reference's FULL size on one MI355X
and compared with the CPU oracle where the oracle affords it (verdict r03, "Run what was only extrapolated, and check it").  One JSON line per
check; `python tests/full_size_checks.py <check> [...]` (a script, not collected by pytest: minutes of GPU + host time per check; it lives under
tests/ because it calls the CPU oracle, which is test infrastructure), checks:

  mk32      mktfhe_parameters_32party_3gen (mk_api.jl:225-231 region): P = 32, n = 620, N = 2048 -- a batch timed + decrypted, 2 gates word for word vs the oracle
  mk64 / mk128   the 64- and 128-party sets the same way (1 oracle gate each: 42 k / 86 k sequential CMuxes on one host thread)
  mk256     mktfhe_parameters_256party_3gen (mk_api.jl:304-310): P = 256, n = 740, N = 2048, l = 2, Bgbit = 18 -- the 185 GB of key spectra resident in
            HBM, one batch timed + decrypted, 1 gate vs the oracle at the full size (189 k CMuxes)
  kms2      mktfhe_parameters_2party_new (mk_api.jl:12-20) at n = 560: 8 gates (and 8 fast_boot gates) word for word vs the oracle, OpenMP over gates
  mk64fft / mk512   the sets on the ring of degree 4096 (mk_api.jl:277-283, 316-322): timing + decryption at full size (mk512: as many parties as fit)
  mk64fft-oracle   the 64-party set on the ring of degree 4096 with a VALID key set (from the oracle's key generation): 64 gates decrypted, 1 gate vs the oracle
  ccs{2,4,8,16}-oracle   the CCS sets (mk_api.jl:4-10, 56-62, 111-117, 185-191) at n = 560 with the oracle's keys: 8 / 4 / 2 / 1 gates vs the oracle
  kms4 / kms8   the 4- and 8-party KMS sets (mk_api.jl:64-72, 120-128) at n = 560: 2 / 1 gates vs the oracle
  ccs16     the 16-party CCS set (mk_api.jl:185-191), n = 560: timing + decryption at full size
Keys of the 3-gen sets are generated on the device (thfhe_pm_mac under thfhe/keygen.py) from the host's randomness, so the oracle sees the same key."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))   # oracle_lib
import thfhe
from thfhe import keygen


def emit(**kw):
    print(json.dumps(kw), flush=True)


def note(msg):
    """progress on stderr: the GPU box takes a silent command for a hung one after a few minutes"""
    print(f"[full_size_checks {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def mk_check(name, batch, oracle_gates, parties=None):
    import oracle_lib as O
    over = dict(parties=parties) if parties else {}
    p = thfhe.make_params(name, **over)
    sig = thfhe.SIGMAS[name]
    t0 = time.time()
    K = keygen.MKSecretKeySet(p, seed=0x5EED0001, sigma_lwe=sig["lwe"], sigma_bk=sig["bk"], sigma_ks=sig["ks"], device=0)
    t_key = time.time() - t0
    note(f"{name}: keys generated in {t_key:.0f} s ({K.bk.nbytes / 1e9:.1f} GB of key coefficients)")
    t0 = time.time()
    ck = thfhe.MKCloudKey(p, K.bk, K.ksk, device=0)
    t_ctx = time.time() - t0
    note(f"{name}: key spectra resident after {t_ctx:.0f} s")
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 2, batch), rng.integers(0, 2, batch)
    xa, xb = K.encrypt(a, 1), K.encrypt(b, 2)
    ck.gates(thfhe.NAND, xa[:2], xb[:2])
    t0 = time.time()
    out = ck.gates(thfhe.NAND, xa, xb)
    dt = time.time() - t0
    note(f"{name}: {batch} gates in {dt:.2f} s")
    ok = bool(np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool))))
    rec = dict(check=name, workload=f"{batch} mk_gate_nand_3gen, {name} at full size (P={p.parties}, n={p.n}, N={p.N}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit})",
               kernel=ck.rotation_kernel_name(batch), cmuxes_per_gate=p.parties * p.n, key_coefficients_gb=K.bk.nbytes / 1e9,
               gates_per_s=batch / dt, seconds=dt, keygen_s=t_key, ctx_create_s=t_ctx, all_decrypt_correct=ok)
    if oracle_gates:
        po = O.make_params(name, **over)
        t0 = time.time()
        orc = O.MKOracle(po, K.bk, K.ksk)
        O.lib().oracle_set_threads(min(oracle_gates, O.usable_cpus()))
        note(f"{name}: oracle evaluating {oracle_gates} gate(s) of {p.parties * p.n} CMuxes")
        import threading
        stop = threading.Event()
        threading.Thread(target=lambda: [note("oracle still running") for _ in iter(lambda: stop.wait(120), True)], daemon=True).start()
        ref = orc.gates(O.NAND, xa[:oracle_gates], xb[:oracle_gates])
        stop.set()
        rec.update(oracle_gates=oracle_gates, oracle_seconds=time.time() - t0, words_equal_to_oracle=bool(np.array_equal(out[:oracle_gates], ref)),
                   words_compared=int(ref.size))
    ck.close()
    emit(**rec)


def kms2_check(gates=8, batch=256):
    import oracle_lib as O
    from thfhe import kms
    p = thfhe.make_kms_params("KMS2")
    t0 = time.time()
    K = keygen.KMSSecretKeySet(p, seed=1)
    t_key = time.time() - t0
    ck = kms.KMSCloudKey(p, K.gsw, K.uni, K.pk, K.crs, K.ksk, device=0)
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 2, batch), rng.integers(0, 2, batch)
    xa, xb = K.encrypt(a, 1), K.encrypt(b, 2)
    kms.mk_gate_nand_new(ck, xa[:4], xb[:4])
    t0 = time.time()
    out = kms.mk_gate_nand_new(ck, xa, xb)
    dt = time.time() - t0
    outf = kms.mk_gate_nand_new(ck, xa, xb, fast_boot=True)
    want = ~(a.astype(bool) & b.astype(bool))
    po = O.KmsParams(**{f: getattr(p, f) for f, _ in O.KmsParams._fields_})
    orc = O.KMSOracle(po, K.gsw, K.uni, K.pk, K.crs, K.ksk)
    O.lib().oracle_set_threads(min(gates, O.usable_cpus()))
    t0 = time.time()
    ref = orc.gates(O.NAND, xa[:gates], xb[:gates])
    reff = orc.gates(O.NAND, xa[:gates], xb[:gates], fast_boot=True)
    t_or = time.time() - t0
    emit(check="kms2", workload=f"{batch} mk_gate_nand_new, KMS2 at full size (P={p.parties}, n={p.n}, N={p.N}, gsw {p.l_gsw}/{p.bg_gsw}, lev {p.l_lev}/{p.bg_lev}, uni {p.l_uni}/{p.bg_uni})",
         gates_per_s=batch / dt, seconds=dt, host_keygen_s=t_key, all_decrypt_correct=bool(np.array_equal(K.decrypt(out), want)),
         fast_boot_all_decrypt_correct=bool(np.array_equal(K.decrypt(outf), want)), oracle_gates=gates, oracle_seconds=t_or,
         words_equal_to_oracle=bool(np.array_equal(out[:gates], ref)), fast_boot_words_equal_to_oracle=bool(np.array_equal(outf[:gates], reff)),
         words_compared=int(ref.size) * 2)
    ck.close()


def ccs_oracle_check(name, batch, oracle_gates):
    """A CCS set (mk_gate_nand, J/mk_gates.jl:7-13) at its full size, key material from the oracle's key generation: the batch on the GPU, every
    output decrypted, `oracle_gates` gates word for word against the oracle."""
    import threading
    import oracle_lib as O
    p = O.make_params(name)
    s = O.SIGMAS[name]
    t0 = time.time()
    K = O.CCSKeys(p, 0x5EED0001, s["bk"], s["ks"])
    t_key = time.time() - t0
    note(f"{name}: keys from the oracle in {t_key:.0f} s")
    ck = thfhe.CCSCloudKey(thfhe.make_params(**p.as_dict()), K.bk, K.pk, K.crs, K.ksk, device=0)
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 2, batch), rng.integers(0, 2, batch)
    xa, xb = K.encrypt_bits(a, s["lwe"], 1), K.encrypt_bits(b, s["lwe"], 2)
    thfhe.mk_gate_nand(ck, xa[:2], xb[:2])
    t0 = time.time()
    out = thfhe.mk_gate_nand(ck, xa, xb)
    dt = time.time() - t0
    note(f"{name}: {batch} gates in {dt:.2f} s")
    orc = O.CCSOracle(p, K)
    O.lib().oracle_set_threads(min(oracle_gates, O.usable_cpus()))
    stop = threading.Event()
    threading.Thread(target=lambda: [note(f"{name}: oracle gate(s) still running") for _ in iter(lambda: stop.wait(120), True)], daemon=True).start()
    t0 = time.time()
    ref = orc.gates(O.NAND, xa[:oracle_gates], xb[:oracle_gates])
    t_or = time.time() - t0
    stop.set()
    emit(check=name, workload=f"{batch} mk_gate_nand (CCS), {name} at full size (P={p.parties}, n={p.n}, N={p.N}, l={p.l}, Bgbit={p.Bgbit}), keys from the oracle's key generation",
         gates_per_s=batch / dt, seconds=dt, oracle_keygen_s=t_key, all_decrypt_correct=bool(np.array_equal(K.decrypt_bits(out), ~(a.astype(bool) & b.astype(bool)))),
         oracle_gates=oracle_gates, oracle_seconds=t_or, words_equal_to_oracle=bool(np.array_equal(out[:oracle_gates], ref)), words_compared=int(ref.size))
    ck.close()


def kms_check(name, batch, oracle_gates):
    """A KMS set (mk_gate_nand_new) at its full size: batch on the GPU, decrypted; `oracle_gates` gates word for word against the oracle."""
    import threading
    import oracle_lib as O
    from thfhe import kms
    p = thfhe.make_kms_params(name)
    t0 = time.time()
    K = keygen.KMSSecretKeySet(p, seed=1)
    t_key = time.time() - t0
    note(f"{name}: host key generation {t_key:.0f} s")
    ck = kms.KMSCloudKey(p, K.gsw, K.uni, K.pk, K.crs, K.ksk, device=0)
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 2, batch), rng.integers(0, 2, batch)
    xa, xb = K.encrypt(a, 1), K.encrypt(b, 2)
    kms.mk_gate_nand_new(ck, xa[:2], xb[:2])
    t0 = time.time()
    out = kms.mk_gate_nand_new(ck, xa, xb)
    dt = time.time() - t0
    note(f"{name}: {batch} gates in {dt:.2f} s")
    po = O.KmsParams(**{f: getattr(p, f) for f, _ in O.KmsParams._fields_})
    orc = O.KMSOracle(po, K.gsw, K.uni, K.pk, K.crs, K.ksk)
    O.lib().oracle_set_threads(min(oracle_gates, O.usable_cpus()))
    stop = threading.Event()
    threading.Thread(target=lambda: [note(f"{name}: oracle gate(s) still running") for _ in iter(lambda: stop.wait(120), True)], daemon=True).start()
    t0 = time.time()
    ref = orc.gates(O.NAND, xa[:oracle_gates], xb[:oracle_gates])
    t_or = time.time() - t0
    stop.set()
    emit(check=name, workload=f"{batch} mk_gate_nand_new, {name} at full size (P={p.parties}, n={p.n}, N={p.N}, gsw {p.l_gsw}/{p.bg_gsw}, lev {p.l_lev}/{p.bg_lev}, uni {p.l_uni}/{p.bg_uni})",
         gates_per_s=batch / dt, seconds=dt, host_keygen_s=t_key, all_decrypt_correct=bool(np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))),
         oracle_gates=oracle_gates, oracle_seconds=t_or, words_equal_to_oracle=bool(np.array_equal(out[:oracle_gates], ref)), words_compared=int(ref.size))
    ck.close()


def ccs16_check(batch=64):
    p = thfhe.make_params("CCS16")
    t0 = time.time()
    K = keygen.CCSSecretKeySet(p)
    t_key = time.time() - t0
    ck = thfhe.CCSCloudKey(p, K.bk, K.pk, K.crs, K.ksk, device=0)
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 2, batch), rng.integers(0, 2, batch)
    xa, xb = K.encrypt(a, 1), K.encrypt(b, 2)
    thfhe.mk_gate_nand(ck, xa[:2], xb[:2])
    t0 = time.time()
    out = thfhe.mk_gate_nand(ck, xa, xb)
    dt = time.time() - t0
    emit(check="ccs16", workload=f"{batch} mk_gate_nand (CCS), CCS16 at full size (P={p.parties}, n={p.n}, N={p.N}, l={p.l}, Bgbit={p.Bgbit})", gates_per_s=batch / dt, seconds=dt,
         host_keygen_s=t_key, all_decrypt_correct=bool(np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))))
    ck.close()


def mk_oracle_keys_check(name, batch, oracle_gates):
    """A set whose keys the device key generation cannot make (ring of degree 4096): key material from the ORACLE's key generation (single-threaded C,
    minutes at full size), the batch on the GPU, every output decrypted, `oracle_gates` gates word for word against the oracle."""
    import threading
    import oracle_lib as O
    p = O.make_params(name)
    s = O.SIGMAS[name]
    stop = threading.Event()
    threading.Thread(target=lambda: [note(f"{name}: oracle key generation still running") for _ in iter(lambda: stop.wait(120), True)], daemon=True).start()
    t0 = time.time()
    K = O.MKKeys(p, 0x5EED0001, s["bk"], s["ks"])
    t_key = time.time() - t0
    stop.set()
    note(f"{name}: keys generated by the oracle in {t_key:.0f} s ({K.bk.nbytes / 1e9:.1f} GB + {K.ksk.nbytes / 1e9:.1f} GB)")
    t0 = time.time()
    ck = thfhe.MKCloudKey(thfhe.make_params(name), K.bk, K.ksk, device=0)
    t_ctx = time.time() - t0
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 2, batch), rng.integers(0, 2, batch)
    xa, xb = K.encrypt_bits(a, s["lwe"], 1), K.encrypt_bits(b, s["lwe"], 2)
    ck.gates(thfhe.NAND, xa[:2], xb[:2])
    t0 = time.time()
    out = ck.gates(thfhe.NAND, xa, xb)
    dt = time.time() - t0
    note(f"{name}: {batch} gates in {dt:.2f} s")
    ok = bool(np.array_equal(K.decrypt_bits(out), ~(a.astype(bool) & b.astype(bool))))
    orc = O.MKOracle(p, K.bk, K.ksk)
    O.lib().oracle_set_threads(min(oracle_gates, O.usable_cpus()))
    stop = threading.Event()
    threading.Thread(target=lambda: [note(f"{name}: oracle gate(s) still running") for _ in iter(lambda: stop.wait(120), True)], daemon=True).start()
    t0 = time.time()
    ref = orc.gates(O.NAND, xa[:oracle_gates], xb[:oracle_gates])
    t_or = time.time() - t0
    stop.set()
    emit(check=name, workload=f"{batch} mk_gate_nand_3gen, {name} at full size (P={p.parties}, n={p.n}, N={p.N}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit}), keys from the oracle's key generation",
         kernel=ck.rotation_kernel_name(batch), cmuxes_per_gate=p.parties * p.n, key_coefficients_gb=K.bk.nbytes / 1e9, gates_per_s=batch / dt, seconds=dt,
         oracle_keygen_s=t_key, ctx_create_s=t_ctx, all_decrypt_correct=ok, oracle_gates=oracle_gates, oracle_seconds=t_or,
         words_equal_to_oracle=bool(np.array_equal(out[:oracle_gates], ref)), words_compared=int(ref.size))
    ck.close()


def mk_timing_synthetic(name, batch, parties=None):
    """Timing of a set whose key generation is not available at full size in this tree (the device key-generation products stop at N = 2048 and
    the host path needs hours at N = 4096): key tables of the right SHAPE filled with random words / zeros.  Kernel time does not depend on key
    values; outputs are not checked here (the kernel is compared with the oracle at reduced size in tests/test_gpu_parity_mk.py)."""
    over = dict(parties=parties) if parties else {}
    p = thfhe.make_params(name, **over)
    rng = np.random.default_rng(1)
    t0 = time.time()
    bk = rng.integers(-2**63, 2**63, size=(p.parties, p.n, 4, p.l, p.N), dtype=np.int64)
    ksk = np.zeros((p.parties, p.N, p.ks_t, (1 << p.ks_basebit) - 1, p.n + 1), np.int32)
    t_key = time.time() - t0
    t0 = time.time()
    ck = thfhe.MKCloudKey(p, bk, ksk, device=0)
    t_ctx = time.time() - t0
    x = rng.integers(-2**31, 2**31, size=(batch, p.parties * p.n + 1), dtype=np.int64).astype(np.int32)
    ck.gates(thfhe.NAND, x[:2], x[:2])
    ck.set_profiling(True)
    t0 = time.time()
    ck.gates(thfhe.NAND, x, x)
    dt = time.time() - t0
    tm = ck.last_timings()
    emit(check=name + "-timing", workload=f"{batch} mk_gate_nand_3gen, {name} shape at full size (P={p.parties}, n={p.n}, N={p.N}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit}), "
         "SYNTHETIC key tables (random words): timing only", kernel=ck.rotation_kernel_name(batch), cmuxes_per_gate=p.parties * p.n,
         key_coefficients_gb=bk.nbytes / 1e9, ksk_gb=ksk.nbytes / 1e9, gates_per_s=batch / dt, seconds=dt, blind_rotate_ms=tm["blind_rotate_ms"],
         keyswitch_ms=tm["keyswitch_ms"], cmux_per_s=batch * p.parties * p.n / (tm["blind_rotate_ms"] * 1e-3), table_fill_s=t_key, ctx_create_s=t_ctx)
    ck.close()


CHECKS = {
    "mk32": lambda: mk_check("MK32", 512, 2),
    "mk64": lambda: mk_check("MK64", 512, 1),
    "mk128": lambda: mk_check("MK128", 512, 1),
    "mk256": lambda: mk_check("MK256", 512, 1),
    "mk256-nooracle": lambda: mk_check("MK256", 512, 0),
    "kms2": kms2_check,
    "mk64fft": lambda: mk_timing_synthetic("MK64-fft", 256),
    "mk512": lambda: mk_timing_synthetic("MK512", 256, parties=int(os.environ.get("MK512_PARTIES", "128"))),
    "ccs16": ccs16_check,
    "mk64fft-oracle": lambda: mk_oracle_keys_check("MK64-fft", 64, 1),
    "ccs2-oracle": lambda: ccs_oracle_check("CCS2", 256, 8),
    "ccs4-oracle": lambda: ccs_oracle_check("CCS4", 128, 4),
    "ccs8-oracle": lambda: ccs_oracle_check("CCS8", 64, 2),
    "ccs16-oracle": lambda: ccs_oracle_check("CCS16", 32, 1),
    "kms4": lambda: kms_check("KMS4", 64, 2),
    "kms8": lambda: kms_check("KMS8", 32, 1),
}

if __name__ == "__main__":
    for name in sys.argv[1:] or ["mk32", "kms2"]:
        CHECKS[name]()
