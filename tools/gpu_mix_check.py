"""Developer check: can torch (its bundled ROCm 7.0 runtime) and libthfhe_hip.so (hipcc 7.2) share a process?
usage: gpu_mix_check.py torch_first|lib_first   -- prints OK/FAIL lines."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torus-fhe_amd"))
order = sys.argv[1]
import numpy as np


def load_lib():
    import thfhe
    L = thfhe.lib()
    print("lib devices", L.thfhe_device_count(), flush=True)
    return thfhe


def load_torch():
    import torch
    print("torch", torch.__version__, "cuda", torch.cuda.is_available(), flush=True)
    return torch


if order == "torch_first":
    torch = load_torch()
    x = torch.ones(4, device="cuda") * 2
    torch.cuda.synchronize()
    thfhe = load_lib()
else:
    thfhe = load_lib()
    torch = load_torch()
    x = torch.ones(4, device="cuda") * 2
    torch.cuda.synchronize()

from thfhe import keygen
p = thfhe.make_params("SK-128", n=32)
K = keygen.SecretKeySet(p, seed=3)
ck = thfhe.CloudKey(p, K.bk, K.ksk, device=0)
a = np.array([0, 1, 1, 0]); b = np.array([1, 1, 0, 0])
out = ck.gates(thfhe.NAND, K.encrypt(a, 1), K.encrypt(b, 2))
print("gate decrypt ok", np.array_equal(K.decrypt(out), ~(a.astype(bool) & b.astype(bool))), flush=True)
# single-rank RCCL process group
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
try:
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    print("nccl single-rank all_reduce OK", float(t.item()), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("nccl FAIL", repr(e), flush=True)
print(order, "DONE", x.sum().item(), flush=True)
