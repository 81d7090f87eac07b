"""KMS scheme across ranks: the per-party TLev rotations of mk_blind_rotate_new are independent of the accumulator and of each other
(3-gen-mk-tfhe/src/new_mk_internals.jl:241-252: levkey[i] = mk_ith_blind_rotate(..., gsw_key[:, i], bara[:, i]) reads nothing that
mk_lev_rlwe_mul writes), so -- unlike the 3-gen scheme, whose accumulator must travel from party to party (thfhe/party_sharded.py) -- they
shard over ranks with no pipeline (side note):

    rank r      levkey_p = thfhe_kms_tlev_rotate(party p, bara[:, p])          for the parties p = r, r + world, ...   (> 99 % of the work)
    all ranks   one all-gather of the TLev accumulators (l_lev x 2 x N x 8 B = 64 KiB per gate and party)
    all ranks   accum = mk_lev_rlwe_mul(accum, levkey_p, ...) for p = 1 .. P (sequential by construction), extraction, key switch

Every rank ends with the full output records, bit for bit those of the one-GPU call (thfhe_kms_gates): the pieces are the same kernels.
`ck` is a thfhe.kms.KMSCloudKey (or any object with tlev_rotate / lev_rlwe_mul / keyswitch and .params -- the CPU tests put the oracle
there); `all_gather(list_of_(party, array)) -> dict party -> array` moves the rotated accumulators (torch_all_gather below: RCCL on
device tensors with backend "nccl", host staging with "gloo")."""
import numpy as np

from . import MU8_64, ThfheError
from .kms import modswitch, t64tot32

E8, E4 = 1 << 29, 1 << 30
_LIN = {0: (E8, -1, -1), 1: (E8, 1, 1), 2: (-E8, 1, 1), 3: (E4, 2, 2), 4: (-E4, -2, -2), 5: (-E8, -1, -1),
        6: (-E8, -1, 1), 7: (-E8, 1, -1), 8: (E8, -1, 1), 9: (E8, 1, -1)}   # gates.jl:15-161: (constant, cx, cy) by opcode NAND .. ORYN


def my_parties(parties, rank, world):
    return list(range(rank, parties, world))


def bootstrap_party_sharded(ck, x, mu=MU8_64, rank=0, world=1, all_gather=None):
    """mk_bootstrap_new with the TLev rotations dealt over `world` ranks.  x int32[count][P n + 1] (the same on every rank)."""
    p = ck.params
    P, N, n = p.parties, p.N, p.n
    x = np.ascontiguousarray(x, np.int32).reshape(-1, P * n + 1)
    G = x.shape[0]
    bar = modswitch(x, N)
    mine = [(q, ck.tlev_rotate(q, bar[:, q * n:(q + 1) * n])) for q in my_parties(P, rank, world)]
    if world > 1:
        if all_gather is None:
            raise ThfheError("world > 1 needs an all_gather callable")
        lev = all_gather(mine, P, (G, p.l_lev, 2, N))
    else:
        lev = dict(mine)
    # X^{-barb} (mu, ..., mu) as a trivial multi-key RLWE sample (new_mk_internals.jl:271-276)
    accum = np.zeros((G, P + 1, N), np.int64)
    k = (np.arange(N)[None, :] + bar[:, -1:].astype(np.int64)) % (2 * N)
    accum[:, P] = np.where(k >= N, -np.int64(mu), np.int64(mu))
    for q in range(P):
        accum = ck.lev_rlwe_mul(q, accum, lev[q])
    # mk_rlwe_extract_sample_64 + t64tot32 (:294-299), then mk_keyswitch
    u = np.empty((G, P * N + 1), np.int32)
    a = accum[:, :P].view(np.uint64)
    rev = np.concatenate([a[:, :, :1], (np.uint64(0) - a[:, :, :0:-1])], axis=2).view(np.int64)
    u[:, :P * N] = t64tot32(rev).reshape(G, P * N)
    u[:, P * N] = t64tot32(accum[:, P, 0])
    return ck.keyswitch(u)


def gates_party_sharded(ck, op, x, y, rank=0, world=1, all_gather=None):
    """mk_gate_nand_new (new_mk_gates.jl:1-7) and the other two-input gates of gates.jl, TLev rotations sharded by party."""
    if op not in _LIN:
        raise ThfheError("the KMS scheme evaluates two-input bootstrapped gates (opcodes NAND .. ORYN)")
    cb, cx, cy = _LIN[op]
    x, y = np.ascontiguousarray(x, np.int32), np.ascontiguousarray(y, np.int32)
    if x.shape != y.shape:
        raise ValueError("x and y must hold the same number of records")
    t = cx * x.astype(np.int64) + cy * y.astype(np.int64)
    t[:, -1] += cb
    return bootstrap_party_sharded(ck, t.astype(np.uint32).view(np.int32), MU8_64, rank, world, all_gather)


class KmsShardedEvaluator:
    """The same schedule with everything resident in HBM (C ABI: thfhe_kms_rotate_parties_dev / thfhe_kms_finish_dev): rank r rotates the block
    of parties [r P/W, (r+1) P/W) into a device tensor, ONE all-gather of the TLev accumulators (RCCL on device tensors with backend "nccl";
    staged through host memory with "gloo"), then every rank finishes replicated on its own GPU.  Nothing but the records the caller hands in
    and gets back exists outside device memory; kernels, tensor plumbing and the collective are ordered on one side stream.
    `ck` is a thfhe.kms.KMSCloudKey (all parties' keys: the relinearisation chain needs them; only the rotations are sharded)."""

    def __init__(self, ck, device=0, group=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _check, lib
        if not torch.cuda.is_available():
            raise ThfheError("KmsShardedEvaluator needs a HIP device (there is no CPU fallback)")
        self.ck, self.group, self.torch, self.dist, self.C = ck, group, torch, dist, C
        self.solo = group is None and not (dist.is_available() and dist.is_initialized())
        self.rank, self.world = (0, 1) if self.solo else (dist.get_rank(group), dist.get_world_size(group))
        P = ck.params.parties
        if P % self.world:
            raise ValueError(f"the rank count must divide the parties (world {self.world}, parties {P})")
        self.per_rank = P // self.world
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(self.device)
        _check(lib().thfhe_kms_set_stream(ck.h, C.c_void_p(self.stream.cuda_stream)))
        self.host_staged = (not self.solo) and dist.get_backend(group) != "nccl"

    def close(self):
        from . import _check, lib
        if getattr(self.ck, "h", None):
            _check(lib().thfhe_kms_set_stream(self.ck.h, None))

    def gates(self, op, x, y=None):
        """mk_gate_nand_new (op = NAND) and the other two-input gates; op = -1: mk_bootstrap_new of x.  x, y: int32 device tensors
        [count][P n + 1] (the same on every rank); returns the output records as a device tensor on every rank."""
        from . import _check, lib
        torch, dist, C, p = self.torch, self.dist, self.C, self.ck.params
        G, m = x.shape[0], self.per_rank
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            lev_mine = torch.empty((m, G, p.l_lev, 2, p.N), dtype=torch.int64, device=self.device)
            _check(lib().thfhe_kms_rotate_parties_dev(self.ck.h, op, ptr(x), ptr(y), self.rank * m, m, ptr(lev_mine), G))
            if self.world == 1:
                lev_all = lev_mine
            elif self.host_staged:
                parts = [torch.empty(lev_mine.shape, dtype=torch.int64) for _ in range(self.world)]
                dist.all_gather(parts, lev_mine.cpu(), group=self.group)
                lev_all = torch.cat(parts).to(self.device)
            else:
                lev_all = torch.empty((p.parties, G, p.l_lev, 2, p.N), dtype=torch.int64, device=self.device)
                dist.all_gather_into_tensor(lev_all, lev_mine, group=self.group)
            out = torch.empty((G, p.parties * p.n + 1), dtype=torch.int32, device=self.device)
            _check(lib().thfhe_kms_finish_dev(self.ck.h, op, ptr(x), ptr(y), ptr(lev_all), ptr(out), G))
        cur.wait_stream(self.stream)
        return out


def torch_all_gather(device=None):
    """all_gather over torch.distributed: every party's rotated accumulator travels once (sum of zero-initialised int64 tensors, an
    all-reduce: exact, and each party has exactly one owner).  device = a torch cuda device with backend "nccl" (RCCL), None with "gloo"."""
    import torch
    import torch.distributed as dist

    def gather(mine, parties, shape):
        buf = torch.zeros((parties,) + tuple(shape), dtype=torch.int64, device=device)
        for q, arr in mine:
            buf[q] = torch.from_numpy(np.ascontiguousarray(arr)).to(buf.device)
        dist.all_reduce(buf)
        out = buf.cpu().numpy()
        return {q: out[q] for q in range(parties)}

    return gather
