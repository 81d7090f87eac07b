"""KMS scheme across ranks (thfhe/kms_sharded.py): the per-party TLev rotations (new_mk_internals.jl:241-252) are dealt over the ranks, one
all-gather, then the sequential relinearisation on every rank.  Two gloo ranks with the CPU oracle playing the engine must produce the very
ciphertexts of the oracle's own mk_gate_nand_new; on the GPU the one-rank composition of the pieces must equal the fused thfhe_kms_gates."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from test_knn_sharded import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
    import hashlib, json, os, sys
    import numpy as np
    sys.path.insert(0, {tests!r}); sys.path.insert(0, {pkg!r})
    import oracle_lib as O
    import thfhe
    from thfhe import keygen, kms_sharded
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    ag = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ag = kms_sharded.torch_all_gather()
    p = thfhe.make_kms_params("KMS4", n=4, N=1024, parties=3)
    K = keygen.KMSSecretKeySet(p, seed=5)
    orc = O.KMSOracle(p, K.gsw, K.uni, K.pk, K.crs, K.ksk)

    class OracleKey:   # the engine of this rehearsal: the call surface of thfhe.kms.KMSCloudKey's pieces
        params = p
        def tlev_rotate(self, party, bara): return np.stack([orc.tlev_rotate(party, b) for b in bara])
        def lev_rlwe_mul(self, party, accum, lev): return np.stack([orc.lev_rlwe_mul(party, a, l) for a, l in zip(accum, lev)])
        def keyswitch(self, u): return np.stack([orc.keyswitch(r) for r in u])

    a, b = np.array([0, 1, 1]), np.array([1, 0, 1])
    xa, xb = K.encrypt(a, 11), K.encrypt(b, 12)
    out = kms_sharded.gates_party_sharded(OracleKey(), O.NAND, xa, xb, rank, world, ag)
    ref = orc.gates(O.NAND, xa, xb)
    print(json.dumps(dict(rank=rank, equal=bool(np.array_equal(out, ref)), mine=kms_sharded.my_parties(p.parties, rank, world),
                          ok=bool(np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))),
                          sha=hashlib.sha256(out.tobytes()).hexdigest())), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
"""


def run(tmp_path, world):
    script = tmp_path / f"kms_rank_w{world}.py"
    script.write_text(textwrap.dedent(SCRIPT.format(tests=os.path.join(ROOT, "tests"), pkg=os.path.join(ROOT, "torus-fhe_amd"))))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="3")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        so, se = pr.communicate(timeout=900)
        assert pr.returncode == 0, se[-3000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    return outs


def test_kms_two_gloo_ranks_equal_the_oracle_gate(tmp_path):
    outs = run(tmp_path, 2)
    assert sorted(o["rank"] for o in outs) == [0, 1]
    assert sorted(sum((o["mine"] for o in outs), [])) == [0, 1, 2]          # every party rotated exactly once
    assert all(o["equal"] and o["ok"] for o in outs)                        # bit for bit the oracle's mk_gate_nand_new, on both ranks
    assert outs[0]["sha"] == outs[1]["sha"]


@pytest.mark.gpu
def test_kms_piecewise_composition_equals_fused_gate_on_gpu(O):
    import thfhe
    from thfhe import keygen, kms, kms_sharded
    p = thfhe.make_kms_params("KMS2", n=24)
    K = keygen.KMSSecretKeySet(p, seed=3)
    ck = kms.KMSCloudKey(p, K.gsw, K.uni, K.pk, K.crs, K.ksk, device=0)
    a, b = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1])
    xa, xb = K.encrypt(a, 41), K.encrypt(b, 42)
    fused = kms.mk_gate_nand_new(ck, xa, xb)
    assert np.array_equal(kms_sharded.gates_party_sharded(ck, thfhe.NAND, xa, xb), fused)
    assert np.array_equal(K.decrypt(fused), ~(a.astype(bool) & b.astype(bool)))
    ck.close()


DEV_SCRIPT = """
    import hashlib, json, os, sys
    import numpy as np
    sys.path.insert(0, {tests!r}); sys.path.insert(0, {pkg!r})
    import torch, torch.distributed as dist
    import oracle_lib as O
    import thfhe
    from thfhe import keygen, kms, kms_sharded
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    p = thfhe.make_kms_params("KMS4", n=10)          # all four parties: two per rank at world = 2
    K = keygen.KMSSecretKeySet(p, seed=7)
    ck = kms.KMSCloudKey(p, K.gsw, K.uni, K.pk, K.crs, K.ksk, device=0)
    rng = np.random.default_rng(2)
    a, b = rng.integers(0, 2, 6), rng.integers(0, 2, 6)
    xa, xb = K.encrypt(a, 11), K.encrypt(b, 12)
    fused = kms.mk_gate_nand_new(ck, xa, xb)              # the one-call path (thfhe_kms_gates) on the same context, before the stream moves
    fused_x = ck.gates(thfhe.XOR, xa, xb)
    boot = kms.mk_bootstrap_new(ck, 1 << 61, xa[:2])
    ev = kms_sharded.KmsShardedEvaluator(ck, device=0)
    ta, tb = torch.from_numpy(xa).to("cuda:0"), torch.from_numpy(xb).to("cuda:0")
    out = ev.gates(thfhe.NAND, ta, tb).cpu().numpy()
    outx = ev.gates(thfhe.XOR, ta, tb).cpu().numpy()
    outb = ev.gates(-1, ta[:2].contiguous()).cpu().numpy()
    orc = O.KMSOracle(p, K.gsw, K.uni, K.pk, K.crs, K.ksk)
    res = dict(rank=rank, world=world, fused=bool(np.array_equal(out, fused)), fused_xor=bool(np.array_equal(outx, fused_x)), boot=bool(np.array_equal(outb, boot)),
               oracle=bool(np.array_equal(out[:2], orc.gates(O.NAND, xa[:2], xb[:2]))),
               ok=bool(np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool)))), sha=hashlib.sha256(out.tobytes()).hexdigest())
    ev.close(); ck.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    print(json.dumps(res), flush=True)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_kms_device_resident_sharded_evaluator_on_gpu(tmp_path, world):
    # thfhe_kms_rotate_parties_dev / thfhe_kms_finish_dev behind thfhe.kms_sharded.KmsShardedEvaluator: nothing but torch device tensors between the phases.
    # world = 1: the composition on one rank; world = 2: two ranks share the box's one MI355X, each rotates two of the four parties, gloo all-gather
    # (staged through host memory; with backend nccl the same call moves the device tensors).  Must equal the fused thfhe_kms_gates and the oracle.
    script = tmp_path / f"kms_dev_w{world}.py"
    script.write_text(textwrap.dedent(DEV_SCRIPT.format(tests=os.path.join(ROOT, "tests"), pkg=os.path.join(ROOT, "torus-fhe_amd"))))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4",
                   THFHE_TORCH_FIRST="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        so, se = pr.communicate(timeout=900)
        assert pr.returncode == 0, se[-3000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    assert all(o["fused"] and o["fused_xor"] and o["boot"] and o["oracle"] and o["ok"] for o in outs), outs
    assert len({o["sha"] for o in outs}) == 1
