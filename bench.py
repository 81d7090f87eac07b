#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X gate-bootstrapping engine (BASELINE.json `metric`).

A "step" is one pass of the hot path over one batch: 4096 independent single-key bootsNAND gates
(BASELINE.json configs[1]; SK-128 parameters n=630, N=1024, k=1, l=3, Bgbit=7, ks 8/2 -- the set the
reference's fixtures and KNN application use), inputs and keys already resident in HBM.  With
--gpus N the gate batch is sharded: every rank runs its own 4096-gate batch on its own GPU with
replicated keys (weak scaling, no data-path collective -- SURVEY.md section 8e); rank 0 prints ONE
JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--set SK-128] [--batch 4096] [--mode replicated|party] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset): this process starts the N ranks ITSELF, as child processes and before
anything touches the GPU, relays rank 0's JSON line and exits non-zero if any rank does.  Under a launcher WORLD_SIZE must equal N.

`--mode party` (BASELINE.json configs[4], "RCCL combine"; 3-gen sets only): the parties' keys are dealt over pipeline groups of
min(N, P) GPUs (thfhe/party_sharded.py): the accumulator travels rank to rank (send/recv), the extracted sample is broadcast, the
key-switched parts are all-gathered -- 3gen_mk_internals.jl:78-84 + mk_internals.jl:730-744.  Every group evaluates its own batch of
`--batch` x group-size gates per step, so a GPU's share of CMuxes is the same at every N (weak scaling).
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


def spawn_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one (which has not imported torch
    or loaded the HIP library: a process that has initialised the GPU must never be replaced or forked into ranks), one per GPU, with
    the torch.distributed environment, on a free port of 127.0.0.1.  Rank 0's stdout is relayed (the ONE JSON line), the other
    ranks' stdout goes to stderr.  Returns the exit code: non-zero as soon as any rank fails (the rest are then terminated by PID)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), THFHE_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_gpus) // n_gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc = 0
    pending = set(range(n_gpus))
    out0 = b""
    while pending:
        for r in sorted(pending):
            pr = procs[r]
            try:
                if r == 0:
                    o, _ = pr.communicate(timeout=0.5)
                    out0 += o or b""
                else:
                    pr.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if pr.returncode != 0 and rc == 0:
                rc = pr.returncode if pr.returncode > 0 else 1
                print(f"[bench] rank {r} exited with code {pr.returncode}: stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending:
                    procs[q].terminate()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


_gather_objects = lambda obj: [obj]   # replaced by dist_setup when world > 1: list of every rank's object, in rank order


def gather_objects(obj):
    return _gather_objects(obj)


def dist_setup(n_gpus):
    """torch.distributed over RCCL ("nccl"); returns (rank, world, barrier, max_reduce, backend).  The world comes from the launcher's
    environment (torch.distributed.run, or spawn_ranks above) and MUST equal --gpus: a mismatch exits non-zero instead of measuring a
    different job than the one asked for.  The backend is chosen ONCE from the environment (THFHE_BENCH_BACKEND, default nccl; "gloo"
    for CPU rehearsals) and a failure of it is fatal on every rank -- a rank that quietly switched backends would leave the others
    hanging in the first collective."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != n_gpus:
        print(f"[bench] --gpus {n_gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {n_gpus}` (spawns its own ranks) or "
              f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {n_gpus} --master-addr 127.0.0.1 bench.py --gpus {n_gpus}`",
              file=sys.stderr, flush=True)
        raise SystemExit(2)
    if world == 1:
        return 0, 1, (lambda: None), (lambda x: x), "none"
    global _gather_objects
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch   # before libthfhe_hip.so is loaded: both then share torch's HIP runtime (thfhe.lib() enforces the same order)
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    backend = os.environ.get("THFHE_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"[bench] THFHE_BENCH_BACKEND={backend!r}: expected nccl or gloo")
    if backend == "nccl" and 1 < torch.cuda.device_count() < world:
        # one rank per GPU or nothing: N ranks sharing fewer devices would print an N-GPU line measured on fewer GPUs.  (A launcher that shows every
        # rank exactly ONE device -- per-rank isolation -- is fine: local % 1 = 0; two ranks that really share a device are refused by RCCL itself,
        # "duplicate GPU detected", and the line's `distinct_devices` shows what ran where.)
        print(f"[bench] rank {rank}: --gpus {world} over RCCL needs {world} visible GPUs, this node shows {torch.cuda.device_count()}", file=sys.stderr, flush=True)
        raise SystemExit(4)
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

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    _gather_objects = gather
    return rank, world, barrier, max_reduce, backend


def rank_record(rank, device, gates, seconds):
    """What one rank contributes to the line's `ranks` list: the device it REALLY ran on (index and PCI bus id from the HIP runtime of the
    engine's library) and its own rate, so that a multi-GPU record shows N ranks on N different devices."""
    import ctypes
    import socket
    import thfhe
    buf = ctypes.create_string_buffer(64)
    ok = thfhe.lib().thfhe_device_pci_bus_id(int(device), buf, 64) == 0
    return {"rank": rank, "device": int(device), "pci_bus_id": buf.value.decode() if ok else None, "host": socket.gethostname(),
            "pid": os.getpid(), "gates_per_s": gates / seconds, "seconds": seconds}


def cpu_baseline(K, p_name, mk, xa, xb, gpu_out, per_thread):
    """Time the CPU oracle on the first gates of the same workload, OpenMP over gates (the reference's only parallel pattern,
    src/KNN_medical_data.cpp:681) on the CPUs this process may really use (affinity mask capped by the cgroup quota); median of 3 runs.
    Single-key: the timed engine is the oracle's restatement of what the reference's CPU path computes -- tgsw_extern_mul on the folded
    Complex{Float64} transform (J/tgsw.jl:146-150, J/polynomials.jl:208-247; same algorithm class as libtfhe's FFT) -- and the exact-integer
    engine (64-bit NTT) is timed next to it on a smaller sample and is the one the GPU output is compared with bit for bit.
    3-gen multi-key (J/3gen_mk_internals.jl:59-116 + J/mk_internals.jl:730-744): the exact engine only (the reference's Float64 transform of
    Torus64 words is lossy by construction)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = O.usable_cpus()
    p = O.make_params(p_name)
    orc = (O.MKOracle if mk else O.Oracle)(p, K.bk, K.ksk)
    L = O.lib()
    L.oracle_set_threads(threads)

    def timed(n_gates, engine, reps):
        runs, ref = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            ref = orc.gates(O.NAND, xa[:n_gates], xb[:n_gates], schoolbook=engine)
            runs.append(time.perf_counter() - t0)
        return ref, sorted(runs)[len(runs) // 2], runs

    exact_n = min((1 if mk else 8) * threads if per_thread is None else per_thread * threads, xa.shape[0])
    orc.gates(O.NAND, xa[:threads], xb[:threads])   # warm the NTT tables and every thread's scratch arena
    ref, dt_exact, runs_exact = timed(exact_n, 0, 3)
    exact = bool(np.array_equal(ref, gpu_out[:exact_n]))
    exact_obj = dict(value=exact_n / dt_exact, sample_gates=exact_n, runs_s=[round(r, 3) for r in runs_exact], gpu_bit_exact_on_sample=exact,
                     engine="exact-integer oracle (64-bit NTT mod 2^64 - 2^32 + 1, centred lifting): the checker of every parity test")
    aff = len(os.sched_getaffinity(0))
    if mk:
        L.oracle_set_threads(1)
        t0 = time.perf_counter()
        orc.gates(O.NAND, xa[:1], xb[:1])
        single = 1.0 / (time.perf_counter() - t0)
        L.oracle_set_threads(threads)
        value = exact_obj["value"]
        return dict(value=value, unit="gates/s", cores=threads, threads=threads, kind="port", per_thread_gates_per_s=value / threads,
                    single_thread_value=single, scaling_efficiency=value / (threads * single), affinity_cpus=aff, runs_s=exact_obj["runs_s"],
                    sample=f"first {exact_n} NAND gates of the same batch, exact-integer MK oracle (64-bit NTT), OpenMP schedule(dynamic) over gates on "
                           f"{threads} threads (cgroup CPU quota; the affinity mask shows {aff}), median of 3 runs = {dt_exact:.2f} s",
                    gpu_bit_exact_on_sample=exact)
    fft_n = min((64 if per_thread is None else 8 * per_thread) * threads, xa.shape[0])
    orc.gates(O.NAND, xa[:threads], xb[:threads], schoolbook=2)   # builds the transformed key (J/bootstrap.jl:11-12) once
    ref_fft, dt, runs = timed(fft_n, 2, 3)
    same_bits = bool(np.array_equal(K.decrypt(ref_fft), K.decrypt(gpu_out[:fft_n])))
    L.oracle_set_threads(1)
    n1 = min(4, fft_n)
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.gates(O.NAND, xa[:n1], xb[:n1], schoolbook=2)
        one.append((time.perf_counter() - t0) / n1)
    L.oracle_set_threads(threads)
    single = 1.0 / sorted(one)[1]
    value = fft_n / dt
    return dict(value=value, unit="gates/s", cores=threads, threads=threads, kind="port",
                per_thread_gates_per_s=value / threads, single_thread_value=single, scaling_efficiency=value / (threads * single),
                affinity_cpus=aff, runs_s=[round(r, 3) for r in runs],
                sample=f"first {fft_n} NAND gates of the same batch with the reference's own arithmetic restated in C: tgsw_extern_mul on the folded "
                       f"Complex{{Float64}} transform (J/tgsw.jl:146-150, J/polynomials.jl:208-247; plain radix-2 FFT, no AVX assembly as libtfhe's spqlios), "
                       f"OpenMP schedule(dynamic) over gates on {threads} threads (cgroup CPU quota; the affinity mask shows {aff}), median of 3 runs = {dt:.2f} s",
                decrypts_like_gpu_on_sample=same_bits, words_equal_to_gpu_frac=float((ref_fft == gpu_out[:fft_n]).mean()),
                gpu_bit_exact_on_sample=exact, exact_oracle=exact_obj)


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


def make_keys(args, p, mk, device):
    """Synthetic key material of SURVEY.md section 8(d): deterministic seed, the parameter set's own noise levels (thfhe.SIGMAS, KeyError
    for a set without an entry).  The 16+-party sets generate their bootstrapping key on the device."""
    import thfhe
    from thfhe import keygen
    sig = thfhe.SIGMAS[args.set]
    if mk:
        wide = p.Bgbit > 10
        return keygen.MKSecretKeySet(p, seed=0x5EED0001, sigma_lwe=sig["lwe"], sigma_bk=sig["bk"], sigma_ks=sig["ks"], device=device if wide else None)
    return keygen.SecretKeySet(p, seed=0x5EED0001, sigma_lwe=sig["lwe"], sigma_bk=sig["bk"], sigma_ks=sig["ks"])


def dry_topology(args):
    """--dry-topology: the rank start-up, rendezvous, barrier and max-over-ranks path of a real run without touching a GPU or the HIP
    library (CPU rehearsal of `--gpus N`; tests/test_sharding_gloo.py)."""
    rank, world, barrier, max_reduce, backend = dist_setup(args.gpus)
    barrier()
    slowest = max_reduce(1.0 + rank)
    res = {"dry_topology": True, "n_gpus": world, "requested_gpus": args.gpus, "timing_backend": backend, "mode": args.mode,
           "slowest_rank_time": slowest, "spawned_by_bench": os.environ.get("THFHE_BENCH_SPAWNED") == "1"}
    if args.mode == "party":
        sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
        import thfhe
        from thfhe.party_sharded import party_topology
        P = thfhe.make_params(args.set).parties
        res["party_topology"] = [party_topology(world, P, r) for r in range(world)]
    barrier()
    if rank == 0:
        print(json.dumps(res), flush=True)


def run_party(args, p, rank, world, barrier, max_reduce, backend, device):
    """--mode party: one step = every pipeline group evaluates `batch x group_size` NAND gates through PartyShardedEvaluator."""
    import torch
    import torch.distributed as dist
    import thfhe
    from thfhe.party_sharded import HipPartyBackend, PartyShardedEvaluator, party_topology
    topo = party_topology(world, p.parties, rank)
    group = None
    if world > 1 and topo["groups"] > 1:   # every rank takes part in creating every group
        for g in range(topo["groups"]):
            ranks = list(range(g * topo["group_size"], (g + 1) * topo["group_size"]))
            h = dist.new_group(ranks)
            if g == topo["group"]:
                group = h
    K = make_keys(args, p, True, device)
    first, last = topo["parties"]
    be = HipPartyBackend(p, (first, last), K.bk[first:last], K.ksk[first:last], device=device)
    B = args.batch * topo["group_size"]
    # a pipeline of W ranks and C slices runs at C / (C + W - 1) of its steady rate, and a slice should still fill the chip (>= 256 gates:
    # one workgroup per CU); with one rank per group there is no pipeline and the batch goes down in one launch
    # slices of a group's batch in flight along the party pipeline: 512 gates each -- launches above 256 gates run two gates per workgroup on both ring
    # degrees (the pair kernels take the accumulator in and hand it out since round 4)
    slice_gates = 512
    chunks = args.pipeline_chunks if args.pipeline_chunks > 0 else (1 if topo["group_size"] == 1 else max(1, B // slice_gates))
    ev = PartyShardedEvaluator(p, be, group=group, pipeline_chunks=chunks)
    gseed = 0x5EED0002 + 2 * topo["group"]      # every rank of a group sees the same ciphertexts (mk_internals.jl:23-37)
    rng = np.random.default_rng(gseed)
    bits_a, bits_b = rng.integers(0, 2, B), rng.integers(0, 2, B)
    xa, xb = K.encrypt(bits_a, seed=gseed), K.encrypt(bits_b, seed=gseed + 1)
    ta, tb = torch.from_numpy(xa).to(be.device), torch.from_numpy(xb).to(be.device)
    be.timing = []
    for _ in range(args.warmup):
        out = ev.gates(thfhe.NAND, ta, tb)
    torch.cuda.synchronize(be.device)
    be.timing = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = ev.gates(thfhe.NAND, ta, tb)
        torch.cuda.synchronize(be.device)
    mine = time.perf_counter() - t0
    barrier()
    elapsed = max_reduce(time.perf_counter() - t0)
    ranks = gather_objects(rank_record(rank, device, B * args.steps / topo["group_size"], mine))   # a rank's share of its group's gates
    rot_ms = [a.elapsed_time(b) for a, b in be.timing]
    got = out.cpu().numpy()
    errors = int((K.decrypt(got) != ~(bits_a.astype(bool) & bits_b.astype(bool))).sum())
    if errors:
        raise RuntimeError(f"rank {rank}: {errors} of {B} bootstrapped NAND outputs decrypt wrongly")
    if rank != 0:
        return None
    value = topo["groups"] * B * args.steps / elapsed
    launches_per_step = max(1, len(rot_ms) // max(1, args.steps))
    jobs_per_launch = B / launches_per_step
    ab = algorithmic_bytes(p)
    per_rot = jobs_per_launch * (ab["bk"] * (last - first) // p.parties + 2 * 2 * p.N * 8)   # this rank's share of the key stream + accumulator in / out
    avg_ms = float(np.mean(rot_ms)) if rot_ms else None
    # the piece launches take the accumulator in / out: the one- or two-gate kernel by launch size, on both ring degrees
    roof = {"kernel": be.ck.rotation_kernel_name(int(jobs_per_launch)),
            "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step, "gates_per_launch": jobs_per_launch,
            "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
            "counters_source": "no PMC pass for the party-sharded piece launches (the kernel is the replicated mode's; see that mode's line)",
            "hbm_algorithmic": {"bytes_per_launch": per_rot, "achieved_gbs": per_rot / (avg_ms * 1e-3) / 1e9 if avg_ms else None, "peak_gbs": HBM_PEAK_GBS,
                                "frac": per_rot / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms else None,
                                "note": "SURVEY.md 8(d) model (this rank's parties' key once per gate); not a bound for this kernel"}}
    comm = {"accumulator_bytes_per_gate_per_hop": 2 * p.N * 8, "extracted_sample_bytes_per_gate": (p.N + 1) * 4,
            "keyswitched_part_bytes_per_gate_per_rank": ((last - first) * p.n + 1) * 4, "hops": topo["group_size"] - 1,
            "collectives": "send/recv (accumulator pipeline), broadcast (extracted LWE), all_gather (key-switched parts)" if topo["group_size"] > 1
                           else "none (one rank holds every party: the same piece kernels, no transport)"}
    return {
        "metric": f"bootstrapped gates/sec (NAND, N={p.N})", "value": value, "unit": "gates/s",
        "n_gpus": world, "requested_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} bootsNAND per pipeline group per step ({args.batch} x {topo['group_size']} ranks), 3-gen multi-key {args.set} "
                               f"(P={p.parties}, n={p.n}, N={p.N}, k={p.k}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit}), party-sharded keys, "
                               "ciphertexts resident in HBM",
                   "gates_per_group_per_step": B, "param_set": args.set,
                   "parallelism": f"party pipeline: {topo['groups']} group(s) x {topo['group_size']} rank(s), {last - first} parties per rank, "
                                  f"{chunks} pipeline chunk(s), {'RCCL' if backend == 'nccl' else backend} combine",
                   "mode": "party", "timing_backend": backend},
        "ranks": ranks, "rccl_world": world if backend == "nccl" else None, "distinct_devices": len({(r["host"], r["pci_bus_id"]) for r in ranks}),
        "roofline": roof, "party_comm": comm, "bit_exact_decrypt_errors": errors,
    }, K, xa, xb, got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--set", default="SK-128")
    ap.add_argument("--batch", type=int, default=4096, help="gates per GPU per step")
    ap.add_argument("--mode", choices=("replicated", "party"), default="replicated",
                    help="replicated: every GPU holds all keys and runs its own batch; party: the parties' keys are dealt over the GPUs (3-gen sets)")
    ap.add_argument("--pipeline-chunks", type=int, default=0,
                    help="--mode party: slices of a group's batch in flight along the party pipeline (0 = auto: one per 256 gates -- 512 on the ring of degree 2048 --, 1 when a group is one rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-per-thread", type=int, default=None, help="gates per host thread in the exact-oracle sample of the CPU baseline (default 8; 1 for multi-key sets; the FFT-engine sample is 8x that)")
    ap.add_argument("--dry-topology", action="store_true", help="rank start-up + rendezvous + barrier only (no GPU): rehearsal of --gpus N")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("[bench] --gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))   # nothing above has touched torch, HIP or the GPU
    if args.dry_topology:
        return dry_topology(args)

    if args.mode == "party":
        os.environ["THFHE_TORCH_FIRST"] = "1"   # torch before libthfhe_hip.so: the piece API works on torch tensors
    import thfhe

    rank, world, barrier, max_reduce, backend = dist_setup(args.gpus)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = thfhe.lib().thfhe_device_count()
    if ndev < 1:
        raise thfhe.ThfheError("no HIP device: bench.py measures the GPU engine and has no CPU fallback")
    device = local % ndev

    p = thfhe.make_params(args.set)
    mk = p.torus_bits == 64
    per_thread = args.cpu_per_thread
    cpu_ok = world == 1 and not args.no_cpu_baseline and p.parties * p.n * (p.N // 1024) <= 4500   # bounded sample: <= ~30 s of host work

    if args.mode == "party":
        if not mk:
            raise SystemExit("[bench] --mode party needs a 3-gen multi-key set (MK2, MK4, MK4-N2048, ...)")
        r = run_party(args, p, rank, world, barrier, max_reduce, backend, device)
        if rank != 0:
            return
        res, K, xa, xb, got = r
        if cpu_ok:
            res["cpu_baseline"] = cpu_baseline(K, args.set, True, xa, xb, got, per_thread)
        print(json.dumps(res), flush=True)
        return

    K = make_keys(args, p, mk, device)
    ck = (thfhe.MKCloudKey if mk else thfhe.CloudKey)(p, K.bk, K.ksk, device=device)
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
    mine = time.perf_counter() - t0      # this rank's own clock, before the barrier
    barrier()
    elapsed = max_reduce(time.perf_counter() - t0)
    ranks = gather_objects(rank_record(rank, device, B * args.steps, mine))

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
    kernel = ck.rotation_kernel_name(B)
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
        "metric": f"bootstrapped gates/sec (NAND, N={p.N})", "value": value, "unit": "gates/s",
        "n_gpus": world, "requested_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} independent bootsNAND per GPU, {'3-gen multi-key' if mk else 'single-key'} {args.set} "
                               f"(P={p.parties}, n={p.n}, N={p.N}, k={p.k}, l={p.l}, Bgbit={p.Bgbit}, ks {p.ks_t}/{p.ks_basebit}), keys+ciphertexts resident in HBM",
                   "gates_per_gpu_per_step": B, "param_set": args.set, "parallelism": f"gate-batch sharding x{world}, replicated keys",
                   "mode": "replicated", "timing_backend": backend},
        "ranks": ranks, "rccl_world": world if backend == "nccl" else None, "distinct_devices": len({(r["host"], r["pci_bus_id"]) for r in ranks}),
        "roofline": roof,
        "bit_exact_decrypt_errors": errors,
    }
    if cpu_ok:
        res["cpu_baseline"] = cpu_baseline(K, args.set, mk, xa, xb, out, per_thread)
    elif world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = None
        res["cpu_baseline_note"] = "skipped: one gate of this set is minutes of work for the exact CPU oracle (P x n CMuxes on the N = 2048 ring)"
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
