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
    """torch.distributed over RCCL ("nccl") when launched by torch.distributed.run; returns (rank, world, barrier, max_reduce)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, (lambda: None), (lambda x: x), "none"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    backend = os.environ.get("THFHE_BENCH_BACKEND", "nccl")
    dev = None
    # RCCL / gloo print start-up banners on the C-level stdout: keep stdout clean for the single JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if backend == "nccl":
        try:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world)
            dev = torch.device("cuda", local)
            t = torch.zeros(1, device=dev)
            dist.all_reduce(t)  # forces communicator creation outside the timed region
            torch.cuda.synchronize()
        except Exception as e:  # RCCL unavailable (e.g. CPU rehearsal): the timing barrier falls back to gloo
            print(f"[bench] rank {rank}: nccl backend unavailable ({e!r}); using gloo for the timing barrier", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            backend = "gloo"
    if backend == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cpu")
        dist.barrier()
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


def cpu_baseline(K, p_name, xa, xb, gpu_out, sample):
    """Time the CPU oracle (exact-integer restatement of the reference path, OpenMP over gates) on the first
    `sample` gates of the same workload, on this host's cores; also cross-check the GPU output on them."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = len(os.sched_getaffinity(0))
    os.environ["OMP_NUM_THREADS"] = str(cores)  # one oracle thread per core this process may run on (set before libgomp loads)
    import oracle_lib as O
    p = O.make_params(p_name)
    orc = O.Oracle(p, K.bk, K.ksk)
    threads = O.lib().oracle_max_threads()
    orc.gates(O.NAND, xa[:1], xb[:1])  # warm the NTT tables
    t0 = time.perf_counter()
    ref = orc.gates(O.NAND, xa[:sample], xb[:sample])
    dt = time.perf_counter() - t0
    exact = bool(np.array_equal(ref, gpu_out[:sample]))
    t1 = time.perf_counter()
    orc.gates(O.NAND, xa[:1], xb[:1])           # one gate = one OpenMP task: the single-thread figure SURVEY.md 8(d) asks for
    dt1 = time.perf_counter() - t1
    return dict(value=sample / dt, unit="gates/s", cores=threads, kind="port", single_thread_value=1.0 / dt1,
                sample=f"first {sample} NAND gates of the same batch, exact-integer oracle (64-bit NTT path, not libtfhe's AVX FFT), "
                       f"OpenMP schedule(dynamic) over gates on {threads} threads, {dt:.2f} s wall",
                gpu_bit_exact_on_sample=exact)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--set", default="SK-128")
    ap.add_argument("--batch", type=int, default=4096, help="gates per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="gates in the CPU baseline sample (0 = 8 per host thread)")
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
        lwe_sigma = {"MK2": 2.0**-13.52, "MK3": 2.0**-13.26, "MK4": 2.0**-13.26, "MK4-N2048": 2.0**-13.26}.get(args.set, 2.0**-13.52)
        K = keygen.MKSecretKeySet(p, seed=0x5EED0001, sigma_lwe=lwe_sigma, sigma_bk=2.0**-30.70)
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
    br_bytes = B * (ab["bk"] + 2 * words * 4 + (p.N + 1) * 4)  # blind-rotate launch: key stream + records in, extracted out
    br_achieved = br_bytes / (br_avg_ms * 1e-3) / 1e9
    res = {
        "metric": "bootstrapped gates/sec (NAND, N=1024)", "value": value, "unit": "gates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} independent bootsNAND per GPU, {'3-gen multi-key' if mk else 'single-key'} {args.set} "
                               f"(P={p.parties}, n={p.n}, N={p.N}, k={p.k}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit}), keys+ciphertexts resident in HBM",
                   "gates_per_gpu_per_step": B, "param_set": args.set, "parallelism": f"gate-batch sharding x{world}, replicated keys",
                   "timing_backend": backend},
        "roofline": {"bound": "hbm", "kernel": (("mk_blind_rotate_coop2k_kernel" if p.N == 2048 else
                                 ("mk_blind_rotate_pair_kernel" if (p.l <= 3 and B > 256) else "mk_blind_rotate_coop_kernel")) + f"<{p.l}>") if mk
                               else (f"sk_blind_rotate_ring_kernel<{p.l}>" if B > 1024 else f"sk_blind_rotate_coop_kernel<{p.l}>"),
                     "achieved": br_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": br_achieved / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_launch": br_bytes, "avg_launch_ms": br_avg_ms,
                     "note": "algorithmic bytes count the whole transformed key once per gate (SURVEY.md 8d); the kernels share every key chunk between "
                             "the gates of a workgroup (8 single-key / 2 multi-key) and all workgroups hit L2/Infinity Cache, so frac is not bounded by 1 "
                             "(measured HBM traffic: profiles/)",
                     "keyswitch_avg_launch_ms": float(np.mean(ks_ms)),
                     "whole_gate": {"bytes_per_gate": ab["total"],
                                    "achieved": value / world * ab["total"] / 1e9,
                                    "frac": value / world * ab["total"] / 1e9 / HBM_PEAK_GBS}},
        "bit_exact_decrypt_errors": errors,
    }
    tfile = os.path.join(ROOT, "profiles", "traffic_sk128.json")
    if not mk and args.set == "SK-128" and B == 4096 and os.path.exists(tfile):
        # HBM bytes per blind-rotate launch from the committed PMC passes of this same workload (bench.py cannot run rocprofv3 on itself)
        res["roofline"]["traffic"] = json.load(open(tfile))["traffic_bytes_per_launch"]
        res["roofline"]["traffic_source"] = "profiles/traffic_sk128.json"
    if world == 1 and not args.no_cpu_baseline and not mk:
        threads = len(os.sched_getaffinity(0))
        sample = args.cpu_sample or max(8, 8 * min(threads, 64))
        res["cpu_baseline"] = cpu_baseline(K, args.set, xa, xb, out, min(sample, B))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
