"""Party-sharded 3-gen MK bootstrap (thfhe.party_sharded).

CPU: two gloo ranks rehearse the accumulator pipeline / broadcast / all-gather schedule with the CPU oracle playing each
party's kernels; the result must equal the monolithic oracle bit for bit.
GPU (-m gpu): two ranks share the one MI355X of the box, each with ONLY its party's keys in a parties=1 HIP context
(gloo transport, staged through host memory); the result must equal the oracle and the replicated-key HIP path."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


RANK_SCRIPT = """
    import json, os, sys
    import numpy as np
    sys.path.insert(0, {tests!r}); sys.path.insert(0, {pkg!r})
    import torch, torch.distributed as dist
    import oracle_lib as O
    import thfhe
    from thfhe.party_sharded import PartyShardedEvaluator, party_topology
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank = dist.get_rank()
    mode = {mode!r}
    p = O.make_params({pset!r}, **{over!r})
    s = O.SIGMAS[{pset!r}]
    K = O.MKKeys(p, 0x5EED0001, s["bk"], s["ks"])
    tp = thfhe.make_params(**p.as_dict())
    first, last = party_topology(dist.get_world_size(), p.parties, rank)["parties"]   # this rank's block of parties
    if mode == "oracle":
        from party_oracle_backend import OraclePartyBackend
        be = OraclePartyBackend(p, (first, last), K.bk[first:last], K.ksk[first:last])
        dev = "cpu"
    else:
        from thfhe.party_sharded import HipPartyBackend
        be = HipPartyBackend(tp, (first, last), K.bk[first:last], K.ksk[first:last], device=0)
        if os.environ.get("THFHE_TEST_PAIR0"):   # two gates per workgroup even for this handful of gates
            be.ck.set_pair_threshold(0)
        dev = "cuda:0"
    ev = PartyShardedEvaluator(tp, be, pipeline_chunks={chunks})
    G = {gates}
    rng = np.random.default_rng(3)
    a, b, c = (rng.integers(0, 2, G) for _ in range(3))
    xa, xb, xc = (K.encrypt_bits(v, s["lwe"], 900 + q) for q, v in enumerate((a, b, c)))
    ta, tb, tc = (torch.from_numpy(v).to(dev) for v in (xa, xb, xc))
    orc = O.MKOracle(p, K.bk, K.ksk)
    res = dict(rank=rank)
    for name, op, args, targs in (("nand", O.NAND, (xa, xb), (ta, tb)), ("xor", O.XOR, (xa, xb), (ta, tb)),
                                  ("and3", O.AND3, (xa, xb, xc), (ta, tb, tc)), ("mux", O.MUX, (xa, xb, xc), (ta, tb, tc)),
                                  ("not", O.NOT, (xa,), (ta,))):
        got = ev.gates(op, *targs).cpu().numpy()
        res[name] = bool(np.array_equal(got, orc.gates(op, *args)))
    got = ev.bootstrap(ta).cpu().numpy()
    res["bootstrap"] = bool(np.array_equal(got, np.stack([orc.keyswitch(orc.bootstrap_wo_keyswitch(r)) for r in xa])))
    res["decrypt"] = bool(np.array_equal(K.decrypt_bits(ev.gates(O.NAND, ta, tb).cpu().numpy()), ~(a.astype(bool) & b.astype(bool))))
    if mode == "hip":   # the replicated-key kernel (all parties in one context) must agree too
        ck = thfhe.MKCloudKey(tp, K.bk, K.ksk, device=0)
        res["replica"] = bool(np.array_equal(ck.gates(O.NAND, xa, xb), ev.gates(O.NAND, ta, tb).cpu().numpy()))
        ck.close()
    dist.barrier()
    print(json.dumps(res), flush=True)
    dist.destroy_process_group()
"""


def run_two_ranks(tmp_path, mode, over, gates, chunks, timeout, pset="MK2", world=2, extra_env=None):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(RANK_SCRIPT.format(tests=os.path.join(ROOT, "tests"), pkg=os.path.join(ROOT, "torus-fhe_amd"),
                                                         mode=mode, over=over, gates=gates, chunks=chunks, pset=pset)))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        so, se = pr.communicate(timeout=timeout)
        assert pr.returncode == 0, se[-3000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    return outs


def test_party_pipeline_two_gloo_ranks_vs_oracle(tmp_path):
    # reduced LWE dimension keeps the CPU oracle fast; ring degree, decomposition and key-switch shape are MK2's
    outs = run_two_ranks(tmp_path, "oracle", dict(n=12), gates=5, chunks=2, timeout=600)
    assert sorted(o["rank"] for o in outs) == [0, 1]
    for o in outs:
        assert all(o[k] for k in ("nand", "xor", "and3", "mux", "not", "bootstrap", "decrypt")), o


def test_party_pipeline_four_gloo_ranks_mk4_shape_vs_oracle(tmp_path):
    # BASELINE.json configs[4] shape: 4 parties, l = 3, Bgbit = 6, ks 5/2 (mk_api.jl:84-90) with a reduced LWE dimension; one rank per
    # party, three pipeline hand-offs of the accumulator, broadcast from the last rank, 4-way all-gather of the key-switched parts
    outs = run_two_ranks(tmp_path, "oracle", dict(n=8), gates=4, chunks=2, timeout=900, pset="MK4", world=4)
    assert sorted(o["rank"] for o in outs) == [0, 1, 2, 3]
    for o in outs:
        assert all(o[k] for k in ("nand", "xor", "and3", "mux", "not", "bootstrap", "decrypt")), o


def test_party_blocks_two_gloo_ranks_hold_two_parties_each(tmp_path):
    # bench.py --mode party at 2 GPUs with the 4-party set: rank r holds parties 2r, 2r+1 (party_topology), ONE accumulator hand-off
    outs = run_two_ranks(tmp_path, "oracle", dict(n=6), gates=3, chunks=2, timeout=900, pset="MK4", world=2)
    for o in outs:
        assert all(o[k] for k in ("nand", "xor", "and3", "mux", "not", "bootstrap", "decrypt")), o


def test_party_topology():
    sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
    from thfhe.party_sharded import party_topology
    assert party_topology(1, 4)["parties"] == (0, 4) and party_topology(1, 4)["groups"] == 1
    assert [party_topology(2, 4, r)["parties"] for r in range(2)] == [(0, 2), (2, 4)]
    assert [party_topology(4, 4, r)["parties"] for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    t = [party_topology(8, 4, r) for r in range(8)]     # BASELINE configs[4] at 8 GPUs: two pipelines of four ranks
    assert all(x["groups"] == 2 and x["group_size"] == 4 for x in t)
    assert [x["group"] for x in t] == [0, 0, 0, 0, 1, 1, 1, 1] and t[5]["group_ranks"] == [4, 5, 6, 7] and t[6]["parties"] == (2, 3)
    with pytest.raises(ValueError):
        party_topology(3, 4)
    with pytest.raises(ValueError):
        party_topology(6, 4)


@pytest.mark.gpu
def test_party_sharded_two_ranks_one_gpu_bit_exact(tmp_path):
    outs = run_two_ranks(tmp_path, "hip", dict(), gates=6, chunks=3, timeout=900)
    for o in outs:
        assert all(o[k] for k in ("nand", "xor", "and3", "mux", "not", "bootstrap", "decrypt", "replica")), o


@pytest.mark.gpu
def test_party_sharded_n2048_two_gates_per_workgroup(tmp_path):
    # the accumulator hand-over (acc_in / acc_out of thfhe_mk_rotate_partial_dev) through mk_blind_rotate_pair2k_kernel: ring of degree 2048,
    # l = 3, five gates (a lone last gate), two ranks with one party each on the one GPU
    outs = run_two_ranks(tmp_path, "hip", dict(n=24, parties=2), gates=5, chunks=1, timeout=900, pset="MK4-N2048", extra_env=dict(THFHE_TEST_PAIR0="1"))
    for o in outs:
        assert all(o[k] for k in ("nand", "xor", "and3", "mux", "not", "bootstrap", "decrypt", "replica")), o


@pytest.mark.gpu
def test_party_sharded_n1024_two_gates_per_workgroup(tmp_path):
    # round 4: mk_blind_rotate_pair_kernel takes the accumulator in and hands it out (acc_in / acc_out of thfhe_mk_rotate_partial_dev), so a
    # party-sharded slice above 256 gates keeps the two-gates-per-workgroup kernel on the ring of degree 1024.  A 300-gate slice (no forced
    # threshold: 300 > 256 selects the pair kernel by itself), MK2 shape with a reduced LWE dimension, two ranks with one party each on the one GPU;
    # then the l = 3 shape (MK4's gadget) on five gates with a lone last gate.
    outs = run_two_ranks(tmp_path, "hip", dict(n=16), gates=300, chunks=1, timeout=1500)
    for o in outs:
        assert all(o[k] for k in ("nand", "xor", "and3", "mux", "not", "bootstrap", "decrypt", "replica")), o
    outs = run_two_ranks(tmp_path, "hip", dict(n=24, parties=2), gates=5, chunks=1, timeout=900, pset="MK4", extra_env=dict(THFHE_TEST_PAIR0="1"))
    for o in outs:
        assert all(o[k] for k in ("nand", "xor", "and3", "mux", "not", "bootstrap", "decrypt", "replica")), o
