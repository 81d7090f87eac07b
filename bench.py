#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X gate-bootstrapping engine (BASELINE.json `metric`).

A "step" is one pass of the hot path over one batch: 4096 independent single-key bootsNAND gates
(BASELINE.json configs[1]; SK-128 parameters n=630, N=1024, k=1, l=3, Bgbit=7, ks 8/2 -- the set the
reference's fixtures and KNN application use), inputs and keys already resident in HBM.  With
--gpus N the gate batch is sharded: every rank runs its own 4096-gate batch on its own GPU with
replicated keys (weak scaling, no data-path collective -- SURVEY.md section 8e); rank 0 prints ONE
JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--set SK-128] [--batch 4096] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(p, rotations=1):
    """SURVEY.md section 8(d) / BASELINE.md section 3: bytes one gate has to touch.
    bk: whole transformed bootstrapping key once per rotation at 8 B/coefficient; ksk: N*t rows of (n+1) words;
    io: two input records + one output record."""
    P = p.parties
    rowscols = 4 * p.l if p.torus_bits == 64 else (p.k + 1) * p.l * (p.k + 1)
    bk = rotations * P * p.n * rowscols * p.N * 8
    ksk = P * p.k * p.N * p.ks_t * (p.n + 1) * 4
    io = 3 * (P * p.n + 1) * 4
    return dict(bk=bk, ksk=ksk, io=io, total=bk + ksk + io)


def dist_setup(n_gpus):
    """torch.distributed over RCCL ("nccl") when launched by torch.distributed.run; returns (rank, world, barrier, max_reduce, backend).
    The backend is chosen ONCE from the environment (THFHE_BENCH_BACKEND, default nccl; "gloo" for CPU rehearsals) and a failure
    of it is fatal on every rank -- a rank that quietly switched backends would leave the others hanging in the first collective."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, (lambda: None), (lambda x: x), "none"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch   # before libthfhe_hip.so is loaded: both then share torch's HIP runtime (thfhe.lib() enforces the same order)
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    backend = os.environ.get("THFHE_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"[bench] THFHE_BENCH_BACKEND={backend!r}: expected nccl or gloo")
    # RCCL / gloo print start-up banners on the C-level stdout: keep stdout clean for the single JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world)
            dev = torch.device("cuda", local)
            t = torch.zeros(1, device=dev)
            dist.all_reduce(t)  # forces communicator creation outside the timed region
            torch.cuda.synchronize()
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dev = torch.device("cpu")
            dist.barrier()
    except Exception as e:
        print(f"[bench] rank {rank}: {backend} backend failed to initialise: {e!r}", file=sys.stderr, flush=True)
        os._exit(3)   # non-zero on this rank; the launcher tears the others down
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    def barrier():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        dist.barrier()

    def max_reduce(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    return rank, world, barrier, max_reduce, backend


def cpu_baseline(K, p_name, xa, xb, gpu_out, per_thread):
    """Time the CPU oracle (exact-integer restatement of the reference path, OpenMP over gates = the reference's only parallel
    pattern, src/KNN_medical_data.cpp:681) on the first gates of the same workload, on the CPUs this process may really use
    (affinity mask capped by the cgroup quota); median of 3 runs; also cross-check the GPU output on the sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = O.usable_cpus()
    p = O.make_params(p_name)
    orc = O.Oracle(p, K.bk, K.ksk)
    L = O.lib()
    L.oracle_set_threads(threads)
    sample = min(per_thread * threads, xa.shape[0])
    orc.gates(O.NAND, xa[:threads], xb[:threads])   # warm the NTT tables and every thread's scratch arena
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        ref = orc.gates(O.NAND, xa[:sample], xb[:sample])
        runs.append(time.perf_counter() - t0)
    dt = sorted(runs)[1]
    exact = bool(np.array_equal(ref, gpu_out[:sample]))
    L.oracle_set_threads(1)
    n1 = min(4, sample)
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.gates(O.NAND, xa[:n1], xb[:n1])
        one.append((time.perf_counter() - t0) / n1)
    L.oracle_set_threads(threads)
    single = 1.0 / sorted(one)[1]
    value = sample / dt
    return dict(value=value, unit="gates/s", cores=threads, threads=threads, kind="port",
                per_thread_gates_per_s=value / threads, single_thread_value=single, scaling_efficiency=value / (threads * single),
                affinity_cpus=len(os.sched_getaffinity(0)), runs_s=[round(r, 3) for r in runs],
                sample=f"first {sample} NAND gates of the same batch ({per_thread} per thread), exact-integer oracle (64-bit NTT engine, per-thread "
                       f"scratch, no allocation in the CMux loop; NOT libtfhe's AVX FFT), OpenMP schedule(dynamic) over gates on {threads} threads "
                       f"(cgroup CPU quota; the affinity mask shows {len(os.sched_getaffinity(0))}), median of 3 runs = {dt:.2f} s",
                gpu_bit_exact_on_sample=exact)


def load_counters(param_set, batch, kernel):
    """The committed per-launch PMC counters of this workload (profiles/*_counters.json, written by tools/summarize_profile.py from
    tools/profile_bench.sh runs; bench.py cannot run rocprofv3 on itself).  They are used only if they were measured on exactly the kernel
    sources of this tree (sha256, tools/kernel_hash.py), for this parameter set, batch size and kernel; otherwise -> (None, reason)."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_hash import kernel_source_hash
    want = kernel_source_hash()
    stale = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), key=os.path.getmtime, reverse=True):
        try:
            c = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if c.get("param_set") != param_set or c.get("gates_per_launch") != batch or c.get("kernel") != kernel:
            continue
        if c.get("kernel_source_sha256") != want:
            stale += 1
            continue
        return c, os.path.relpath(path, ROOT)
    return None, (f"{stale} counter file(s) for this workload are stale (kernel sources changed since they were measured): traffic / fractions withheld"
                  if stale else "no committed counter file for this workload")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--set", default="SK-128")
    ap.add_argument("--batch", type=int, default=4096, help="gates per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-per-thread", type=int, default=8, help="gates per host thread in the CPU baseline sample")
    args = ap.parse_args()

    import thfhe
    from thfhe import keygen

    rank, world, barrier, max_reduce, backend = dist_setup(args.gpus)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = thfhe.lib().thfhe_device_count()
    if ndev < 1:
        raise thfhe.ThfheError("no HIP device: bench.py measures the GPU engine and has no CPU fallback")
    device = local % ndev

    p = thfhe.make_params(args.set)
    mk = p.torus_bits == 64
    if mk:   # 3-gen multi-key (BASELINE.json configs[2], [4]); noise per J/mk_api.jl:32-38,84-90
        lwe_sigma = {"MK2": 2.0**-13.52, "MK3": 2.0**-13.26, "MK4": 2.0**-13.26, "MK4-N2048": 2.0**-13.26, "MK16": 2.0**-15.34, "MK32": 2.0**-16.12,
                     "MK64": 2.0**-16.90, "MK128": 2.0**-17.42}.get(args.set, 2.0**-13.52)
        wide = p.Bgbit > 10   # the 16+-party sets (J/mk_api.jl:214-298): one level, 24 .. 26-bit base, RLWE noise 2^-62; keys generated on the device
        K = keygen.MKSecretKeySet(p, seed=0x5EED0001, sigma_lwe=lwe_sigma, sigma_bk=2.0**-62 if wide else 2.0**-30.70, device=device if wide else None)
        ck = thfhe.MKCloudKey(p, K.bk, K.ksk, device=device)
    else:    # SURVEY.md section 8(d) synthetic-input recipe
        sig = dict(lwe=2.0**-15, bk=2.0**-25, ks=2.0**-15)
        K = keygen.SecretKeySet(p, seed=0x5EED0001, sigma_lwe=sig["lwe"], sigma_bk=sig["bk"], sigma_ks=sig["ks"])
        ck = thfhe.CloudKey(p, K.bk, K.ksk, device=device)
    words = p.parties * p.n + 1

    B = args.batch
    rng = np.random.default_rng(0x5EED0002 + rank)
    bits_a, bits_b = rng.integers(0, 2, B), rng.integers(0, 2, B)
    xa, xb = K.encrypt(bits_a, seed=0x5EED0002 + 2 * rank), K.encrypt(bits_b, seed=0x5EED0003 + 2 * rank)
    da, db, do = ck.device_records(B), ck.device_records(B), ck.device_records(B)
    da.upload(xa)
    db.upload(xb)
    ck.reserve(B)
    ck.set_profiling(True)  # HIP events around each kernel, on the context's own stream

    for _ in range(args.warmup):
        ck.gates_dev(thfhe.NAND, da, db, None, do, B)
    ck.sync()
    barrier()
    br_ms, ks_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ck.gates_dev(thfhe.NAND, da, db, None, do, B)
        ck.sync()
        tm = ck.last_timings()
        br_ms.append(tm["blind_rotate_ms"])
        ks_ms.append(tm["keyswitch_ms"])
    ck.sync()
    barrier()
    elapsed = max_reduce(time.perf_counter() - t0)

    out = do.download((B, words))
    errors = int((K.decrypt(out) != ~(bits_a.astype(bool) & bits_b.astype(bool))).sum())
    if errors:
        raise RuntimeError(f"rank {rank}: {errors} of {B} bootstrapped NAND outputs decrypt wrongly")

    if rank != 0:
        return
    ab = algorithmic_bytes(p)
    value = world * B * args.steps / elapsed
    br_avg_ms = float(np.mean(br_ms))
    br_s = br_avg_ms * 1e-3
    br_bytes = B * (ab["bk"] + 2 * words * 4 + (p.N + 1) * 4)  # blind-rotate launch: key stream + records in, extracted out
    kernel = ((("mk_blind_rotate_coop2k_kernel" if p.N == 2048 else
                ("mk_blind_rotate_pair_kernel" if (p.l <= 3 and B > 256) else "mk_blind_rotate_coop_kernel")) + f"<{p.l}>") if mk
              else (f"sk_blind_rotate_ring_kernel<{p.l}>" if B > 1024 else f"sk_blind_rotate_coop_kernel<{p.l}>"))
    # SURVEY.md 8(d)'s HBM model, kept under its own name: it charges every gate a private pass over the transformed key, while the kernels
    # share each key chunk between the gates of a workgroup and all workgroups hit L2 / Infinity Cache -- it can exceed 1 and bounds nothing
    hbm_alg = {"bytes_per_launch": br_bytes, "achieved_gbs": br_bytes / br_s / 1e9, "peak_gbs": HBM_PEAK_GBS,
               "frac": br_bytes / br_s / 1e9 / HBM_PEAK_GBS,
               "whole_gate": {"bytes_per_gate": ab["total"], "achieved_gbs": value / world * ab["total"] / 1e9,
                              "frac": value / world * ab["total"] / 1e9 / HBM_PEAK_GBS},
               "note": "algorithmic bytes = SURVEY.md 8(d) per-gate figure x gates per launch; NOT a bound for these kernels (see hbm_measured_frac)"}
    roof = {"kernel": kernel, "avg_launch_ms": br_avg_ms, "keyswitch_avg_launch_ms": float(np.mean(ks_ms)),
            "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
            "hbm_measured_frac": None, "fp64_issue_frac": None, "lds_busy": None, "valu_busy": None, "hbm_algorithmic": hbm_alg}
    counters, why = load_counters(args.set, B, kernel)
    roof["counters_source"] = why
    if counters is not None:
        from summarize_profile import FP64_PEAK_GINST, FP64_PEAK_TFLOPS, derive
        d = derive(counters, br_s)   # counters of the same kernel sources and workload, combined with the launch time measured in THIS run
        roof.update({k: d.get(k) for k in ("traffic", "hbm_measured_frac", "fp64_issue_frac", "lds_busy", "valu_busy")})
        cand = {"fp64_valu_issue": d.get("fp64_issue_frac"), "lds": d.get("lds_busy"), "hbm": d.get("hbm_measured_frac")}
        cand = {k: v for k, v in cand.items() if v is not None}
        if cand:
            roof["bound"] = max(cand, key=cand.get)
        if roof["bound"] == "fp64_valu_issue":
            # instruction roofline of the binding pipe: wave64 FP64 instructions issued per second against 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles
            roof.update(achieved=d["fp64_ginst_per_s"], peak=FP64_PEAK_GINST, unit="Ginst/s (wave64 FP64 VALU instructions)", frac=d["fp64_issue_frac"],
                        fp64_tflops={"achieved": d.get("fp64_tflops"), "peak": FP64_PEAK_TFLOPS, "frac": d.get("fp64_flop_frac"),
                                     "note": "flop view of the same pipe (FMA = 2 flop, add / mul = 1): lower than the issue fraction because the "
                                             "butterflies of the transform are unfused adds"})
        elif roof["bound"] == "lds":
            roof.update(achieved=d["lds_busy"], peak=1.0, unit="LDS-array busy fraction", frac=d["lds_busy"])
        elif roof["bound"] == "hbm":
            roof.update(achieved=d["traffic"] / br_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=d["hbm_measured_frac"])
        roof["effective_clock_ghz_at_profile"] = d.get("effective_clock_ghz")
    res = {
        "metric": "bootstrapped gates/sec (NAND, N=1024)", "value": value, "unit": "gates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} independent bootsNAND per GPU, {'3-gen multi-key' if mk else 'single-key'} {args.set} "
                               f"(P={p.parties}, n={p.n}, N={p.N}, k={p.k}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit}), keys+ciphertexts resident in HBM",
                   "gates_per_gpu_per_step": B, "param_set": args.set, "parallelism": f"gate-batch sharding x{world}, replicated keys",
                   "timing_backend": backend},
        "roofline": roof,
        "bit_exact_decrypt_errors": errors,
    }
    if world == 1 and not args.no_cpu_baseline and not mk:
        res["cpu_baseline"] = cpu_baseline(K, args.set, xa, xb, out, args.cpu_per_thread)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
