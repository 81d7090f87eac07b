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
