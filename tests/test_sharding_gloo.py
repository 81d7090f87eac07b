"""N > 1 path of bench.py rehearsed on CPU: two gloo ranks run the rank-sharding / barrier / max-over-ranks logic
(bench.dist_setup) and the per-rank input derivation; no GPU and no oracle involved."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_barrier_and_max_reduce(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys, time
        sys.path.insert(0, {ROOT!r})
        os.environ["THFHE_BENCH_BACKEND"] = "gloo"
        import bench
        rank, world, barrier, max_reduce, backend = bench.dist_setup(2)   # --gpus 2 == WORLD_SIZE
        barrier()
        t = max_reduce(1.0 + rank)          # slowest rank defines the step time
        import numpy as np
        rng = np.random.default_rng(0x5EED0002 + rank)   # per-rank synthetic inputs are disjoint streams
        first = int(rng.integers(0, 2**31))
        barrier()
        print(json.dumps(dict(rank=rank, world=world, t=t, backend=backend, first=first)), flush=True)
    """))
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        so, se = pr.communicate(timeout=240)
        assert pr.returncode == 0, se
        outs.append(json.loads(so.strip().splitlines()[-1]))
    assert sorted(o["rank"] for o in outs) == [0, 1]
    assert all(o["world"] == 2 and o["backend"] == "gloo" and o["t"] == 2.0 for o in outs)
    assert outs[0]["first"] != outs[1]["first"]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


def test_bench_gpus_n_spawns_its_own_ranks():
    # `python bench.py --gpus 2` with NO launcher and NO environment prepared by the caller: bench.py itself starts two ranks (children, before
    # anything touches a GPU), they rendezvous, and rank 0's single JSON line reports the world that really ran
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-topology"], env=_clean_env(THFHE_BENCH_BACKEND="gloo"),
                        capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["requested_gpus"] == 2 and res["spawned_by_bench"] and res["timing_backend"] == "gloo"
    assert res["slowest_rank_time"] == 2.0


def test_bench_party_mode_topology_over_spawned_ranks():
    # BASELINE.json configs[4] on 2 ranks: the 4-party set's keys in two blocks of two parties, one pipeline group
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-topology", "--mode", "party", "--set", "MK4-N2048"],
                        env=_clean_env(THFHE_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-3000:]
    res = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and [t["parties"] for t in res["party_topology"]] == [[0, 2], [2, 4]]
    assert all(t["groups"] == 1 and t["group_size"] == 2 for t in res["party_topology"])


def test_bench_refuses_a_world_that_differs_from_gpus():
    # under a launcher (WORLD_SIZE set) a mismatch with --gpus is an error, not a silently different job
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-topology"],
                        env=_clean_env(THFHE_BENCH_BACKEND="gloo", WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port())),
                        capture_output=True, text=True, timeout=120)
    assert pr.returncode != 0 and "WORLD_SIZE=2" in pr.stderr


def test_bench_spawn_propagates_a_failing_rank():
    # an unknown parameter set makes every rank fail before the rendezvous: the parent must exit non-zero and print no JSON line
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-topology", "--mode", "party", "--set", "NO-SUCH-SET"],
                        env=_clean_env(THFHE_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert pr.returncode != 0
    assert not [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]


def test_algorithmic_bytes_match_baseline_md():
    # BASELINE.md section 3 table
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
    import bench
    import thfhe
    exp = {"SK-128": (61931520, 20676608, 82615700), "SK-80": (32768000, 16416768, 49190780),
           "SK-lib": (100663296, 33587200, 134262796), "MK2": (68157440, 12804096, 80974028),
           "MK4": (200540160, 41861120, 242425772)}
    for name, (bk, ksk, total) in exp.items():
        ab = bench.algorithmic_bytes(thfhe.make_params(name))
        assert (ab["bk"], ab["ksk"], ab["total"]) == (bk, ksk, total), name


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("mode,pset,batch", [("replicated", "SK-128", 256), ("party", "MK2", 128)])
def test_bench_two_spawned_ranks_on_the_one_gpu(mode, pset, batch):
    # The whole N > 1 path of bench.py on real hardware, as far as a one-GPU box allows: `python bench.py --gpus 2` spawns two ranks that SHARE the
    # box's MI355X (device = LOCAL_RANK mod device count), rendezvous over gloo (RCCL refuses two ranks on one device), run their steps between
    # barriers and print ONE line.  Replicated mode: each rank its own gate batch.  Party mode (BASELINE configs[4]'s shape on the 2-party set): one
    # pipeline group of two ranks, one party each -- accumulator send/recv, broadcast of the extracted sample, all-gather of the key-switched parts,
    # every output decrypted by bench.py itself (it raises on a wrong bit).
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--set", pset, "--batch", str(batch), "--mode", mode, "--steps", "2",
                         "--warmup", "1", "--no-cpu-baseline"], env=_clean_env(THFHE_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["requested_gpus"] == 2 and res["config"]["mode"] == mode and res["bit_exact_decrypt_errors"] == 0
    assert res["value"] > 0 and res["config"]["timing_backend"] == "gloo"
    if mode == "party":
        assert "1 group(s) x 2 rank(s), 1 parties per rank" in res["config"]["parallelism"]


def test_bench_eight_spawned_ranks_party_topology_of_configs4():
    # BASELINE.json configs[4] as the driver would ask for it on an 8-GPU node: `python bench.py --gpus 8 --mode party --set MK4-N2048` -- eight
    # ranks spawned by bench.py itself, two pipeline groups of four ranks, one party per rank (dry run: rendezvous + barriers, no GPU)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-topology", "--mode", "party", "--set", "MK4-N2048"],
                        env=_clean_env(THFHE_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-3000:]
    res = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 8 and res["slowest_rank_time"] == 8.0
    topo = res["party_topology"]
    assert [t["group"] for t in topo] == [0, 0, 0, 0, 1, 1, 1, 1] and [t["parties"] for t in topo] == [[0, 1], [1, 2], [2, 3], [3, 4]] * 2
