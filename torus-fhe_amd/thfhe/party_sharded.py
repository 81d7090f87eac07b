"""Party-sharded 3-gen multi-key bootstrap (the north-star's "RCCL accumulator combine").

The parties' keys are dealt over the ranks of a pipeline group: with W ranks and P parties (W divides P) rank r holds ONLY
the TransformedBootstrapKeyPart_3gen and KeyswitchKey of parties [r P/W, (r+1) P/W) (a `parties = P/W` thfhe_mk_ctx; W = P
is one rank per party).  The reference's blind rotation is party-major on one accumulator (3gen_mk_internals.jl:78-84), so
the accumulator travels down the ranks as a pipeline (shown for W = P):

    rank 0   acc = X^{-barb} mu ; n CMuxes with party 0's key       --send-->  rank 1   n CMuxes with party 1's key  --> ...
    rank P-1 rlwe_extract_sample_64 + t64tot32 (rlwe.jl:70-74)       --broadcast u (LWE of dimension N)-->  every rank
    rank p   part_p = keyswitch(ks[p], (u.a, 0))  (mk_internals.jl:738-741; rank 0 keeps u.b)
    all-gather of the parts: out.a[:, p] = part_p.a ; out.b = u.b + sum_p part_p.b  (mk_internals.jl:742-743)

The batch is cut into `pipeline_chunks` slices so that rank p works on slice c+1 while rank p+1 works on slice c.
Collectives go through torch.distributed: backend "nccl" (= RCCL over xGMI) moves the device tensors directly and orders
with the HIP stream the kernels are enqueued on (thfhe_mk_set_stream); backend "gloo" stages through host memory (used to
rehearse the schedule on CPU and to run two ranks on one GPU in the tests).

The per-party kernels are the `backend` object; the product backend is HipPartyBackend (libthfhe_hip.so, no CPU fallback).
"""
import contextlib
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import AND, AND3, MU8_64, MUX, NAND, NOT, COPY, OR, XOR, MKCloudKey, ThfheError, _check, lib, make_params

E8 = 1 << 29


def party_topology(world, parties, rank=0):
    """How `world` ranks share the party pipeline of a P-party key set: pipeline groups of W = min(world, P) ranks (W must divide
    P and world), every group evaluates its own gate batch; rank r of a group holds parties [r P/W, (r+1) P/W).
    Returns dict(group_size, groups, group, group_rank, parties=(first, last+1), group_ranks=[global ranks of this rank's group])."""
    W = min(world, parties)
    if world < 1 or parties % W or world % W:
        raise ValueError(f"party-sharded mode: {world} rank(s) cannot share {parties} parties (need min(world, P) | P and | world)")
    g, r, m = rank // W, rank % W, parties // W
    return dict(group_size=W, groups=world // W, group=g, group_rank=r, parties=(r * m, (r + 1) * m),
                group_ranks=list(range(g * W, (g + 1) * W)))


class HipPartyBackend:
    """The kernels of a contiguous block of parties on one MI355X: thfhe_mk_{prologue,rotate_partial,extract,keyswitch}_dev on
    torch device tensors."""

    def __init__(self, params, party, bk_part, ksk_part, device=0):
        """params: the FULL parameter set (parties = P).  party: an index p (then bk_part int64[n][4][l][N], ksk_part
        int32[N][t][base-1][n+1]) or a range (first, last+1) (then bk_part / ksk_part carry a leading axis over those parties)."""
        if not torch.cuda.is_available():
            raise ThfheError("HipPartyBackend needs a HIP device (there is no CPU fallback)")
        if isinstance(party, int):
            party, bk_part, ksk_part = (party, party + 1), np.asarray(bk_part)[None], np.asarray(ksk_part)[None]
        self.params, self.first, self.count = params, int(party[0]), int(party[1] - party[0])
        self.party = self.first
        if not (0 <= self.first and self.count >= 1 and self.first + self.count <= params.parties) or len(bk_part) != self.count or len(ksk_part) != self.count:
            raise ValueError("party range outside the parameter set, or key parts that do not match it")
        self.device = torch.device("cuda", device)
        d = params.as_dict()
        d["parties"] = self.count
        self.ck = MKCloudKey(make_params(**d), np.asarray(bk_part), np.asarray(ksk_part), device)
        # kernels, torch's tensor plumbing and the RCCL collectives are all ordered on ONE side stream owned by this backend
        # (torch's default stream has the null handle, which thfhe_mk_set_stream reads as "the context's own stream")
        self.stream = torch.cuda.Stream(self.device)
        _check(lib().thfhe_mk_set_stream(self.ck.h, C.c_void_p(self.stream.cuda_stream)))
        self.rec_words = params.parties * params.n + 1
        self.timing = None   # a list: (start, end) torch events around every rotation launch, on this backend's stream (bench.py)

    def stream_context(self):
        return torch.cuda.stream(self.stream)

    def close(self):
        self.ck.close()

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def prologue(self, op, which, x, y, z):
        p = self.params
        count = x.shape[0]
        bara, barb = self.empty((count, self.count * p.n), torch.int32), self.empty((count,), torch.int32)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _check(lib().thfhe_mk_prologue_dev(self.ck.h, op, which, ptr(x), ptr(y), ptr(z), self.rec_words, self.first * p.n,
                                           ptr(bara), ptr(barb), count))
        return bara, barb

    def rotate(self, bara, barb, mu, acc_in):
        count = bara.shape[0]
        acc = self.empty((count, 2, self.params.N), torch.int64)
        if self.timing is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(self.stream)
        _check(lib().thfhe_mk_rotate_partial_dev(self.ck.h, C.c_void_p(bara.data_ptr()), C.c_void_p(barb.data_ptr()), mu,
                                                 C.c_void_p(acc_in.data_ptr()) if acc_in is not None else None,
                                                 C.c_void_p(acc.data_ptr()), count))
        if self.timing is not None:
            ev1.record(self.stream)
            self.timing.append((ev0, ev1))
        return acc

    def extract(self, acc):
        u = self.empty((acc.shape[0], self.params.N + 1), torch.int32)
        _check(lib().thfhe_mk_extract_dev(self.ck.h, C.c_void_p(acc.data_ptr()), C.c_void_p(u.data_ptr()), acc.shape[0]))
        return u

    def keyswitch(self, u):
        out = self.empty((u.shape[0], self.count * self.params.n + 1), torch.int32)
        _check(lib().thfhe_mk_keyswitch_dev(self.ck.h, C.c_void_p(u.data_ptr()), C.c_void_p(out.data_ptr()), u.shape[0]))
        return out


class PartyShardedEvaluator:
    """mk_bootstrap_3gen / mk_gate_*_3gen with the parties' keys sharded over the W ranks of `group` (W divides P; rank r holds
    parties [r P/W, (r+1) P/W), see party_topology).  W = 1 (no process group at all) runs the same pieces on one GPU.

    Inputs are full MK records int32[count][P*n+1] (every party sees the ciphertexts, as in the reference); they must live
    on the backend's device.  Every rank returns the full output records.
    """

    def __init__(self, params, backend, group=None, pipeline_chunks=4):
        self.params, self.be, self.group = params, backend, group
        self.solo = group is None and not (dist.is_available() and dist.is_initialized())
        self.rank, self.world = (0, 1) if self.solo else (dist.get_rank(group), dist.get_world_size(group))
        if params.parties % self.world:
            raise ValueError(f"party-sharded mode needs a rank count that divides the parties (world {self.world}, parties {params.parties})")
        self.per_rank = params.parties // self.world
        if getattr(backend, "count", 1) != self.per_rank or getattr(backend, "first", self.rank) != self.rank * self.per_rank:
            raise ValueError("the backend does not hold this rank's block of parties")
        self.chunks = max(1, int(pipeline_chunks))
        self.host_staged = (not self.solo) and dist.get_backend(group) != "nccl"

    # -- transport (RCCL moves device tensors; gloo goes through host memory) --------------------------------------------
    def _send(self, t, dst):
        dist.send(t.cpu() if self.host_staged else t, dist.get_global_rank(self.group, dst) if self.group else dst, group=self.group)

    def _recv(self, like_shape, dtype, src):
        buf = torch.empty(like_shape, dtype=dtype) if self.host_staged else self.be.empty(like_shape, dtype)
        dist.recv(buf, dist.get_global_rank(self.group, src) if self.group else src, group=self.group)
        return buf.to(self.be.device) if self.host_staged else buf

    def _broadcast(self, t, src):
        if self.world == 1:
            return t
        buf = t.cpu() if self.host_staged else t
        dist.broadcast(buf, dist.get_global_rank(self.group, src) if self.group else src, group=self.group)
        return buf.to(self.be.device) if self.host_staged else buf

    def _all_gather(self, t):
        if self.world == 1:
            return [t]
        t = t.cpu().contiguous() if self.host_staged else t.contiguous()
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return [q.to(self.be.device) for q in parts] if self.host_staged else parts

    # -- the pipeline ----------------------------------------------------------------------------------------------------
    def _bootstrap_jobs(self, mu, prologues):
        """prologues: list of (op, which, x, y, z); every entry contributes x.shape[0] jobs, in order."""
        p, be, P = self.params, self.be, self.world
        pieces = []
        for op, which, x, y, z in prologues:
            pieces.append(be.prologue(op, which, x, y, z))
        bara = torch.cat([q[0] for q in pieces]) if len(pieces) > 1 else pieces[0][0]
        barb = torch.cat([q[1] for q in pieces]) if len(pieces) > 1 else pieces[0][1]
        jobs = bara.shape[0]
        bounds = [jobs * c // self.chunks for c in range(self.chunks + 1)]
        us = []
        for c in range(self.chunks):
            lo, hi = bounds[c], bounds[c + 1]
            if hi == lo:
                continue
            acc_in = self._recv((hi - lo, 2, p.N), torch.int64, self.rank - 1) if self.rank > 0 else None
            acc = be.rotate(bara[lo:hi].contiguous(), barb[lo:hi].contiguous(), mu, acc_in)
            if self.rank < P - 1:
                self._send(acc, self.rank + 1)
            else:
                us.append(be.extract(acc))
        if self.rank == P - 1:
            u = torch.cat(us) if len(us) > 1 else us[0]
        else:
            u = be.empty((jobs, p.N + 1), torch.int32)
        u = self._broadcast(u, P - 1)
        if self.rank != 0:  # keyswitch(ks[p], (a, 0)): only one rank carries u.b into the sum
            u = u.clone()
            u[:, p.N] = 0
        parts = self._all_gather(be.keyswitch(u))
        m = self.per_rank * p.n   # mask words per rank: out.a[:, parties of rank q] = part_q.a ; out.b = u.b + sum part.b
        out = be.empty((jobs, p.parties * p.n + 1), torch.int32)
        b = torch.zeros((jobs,), dtype=torch.int64, device=out.device)
        for q, part in enumerate(parts):
            out[:, q * m:(q + 1) * m] = part[:, :m]
            b += part[:, m].to(torch.int64)
        out[:, p.parties * p.n] = b.to(torch.int32)  # wraps mod 2^32
        return out

    def _on_stream(self):
        ctx = getattr(self.be, "stream_context", None)
        return ctx() if ctx else contextlib.nullcontext()

    def _finish(self, out):
        if self.be.device.type == "cuda":  # hand the result back to the caller's stream
            torch.cuda.current_stream(self.be.device).wait_stream(self.be.stream)
        return out

    def bootstrap(self, x, mu=MU8_64):
        """mk_bootstrap_3gen(bk, ks, mu, x), 3gen_mk_internals.jl:112-116."""
        with self._on_stream():
            self._enter(x)
            out = self._bootstrap_jobs(mu, [(-1, 0, x, None, None)])
        return self._finish(out)

    def _enter(self, *tensors):
        if self.be.device.type == "cuda":  # inputs were produced on the caller's stream
            self.be.stream.wait_stream(torch.cuda.default_stream(self.be.device))

    def gates(self, op, x, y=None, z=None):
        """mk_gate_{nand,or,and,xor,3and,mux,not}_3gen, 3gen_mk_gates.jl:8-150."""
        with self._on_stream():
            self._enter(x, y, z)
            out = self._gates(op, x, y, z)
        return self._finish(out)

    def _gates(self, op, x, y=None, z=None):
        if op == NOT:
            return -x
        if op == COPY:
            return x.clone()
        if op in (NAND, OR, AND, XOR):
            return self._bootstrap_jobs(MU8_64, [(op, 0, x, y, None)])
        if op == AND3:
            return self._bootstrap_jobs(MU8_64, [(op, 0, x, y, z)])
        if op == MUX:  # two full ANDs, then (0, 1/8) + t1 + t2 without bootstrapping (3gen_mk_gates.jl:133-150)
            t = self._bootstrap_jobs(MU8_64, [(MUX, 0, x, y, z), (MUX, 1, x, y, z)])
            g = x.shape[0]
            out = t[:g] + t[g:]
            out[:, -1] += E8
            return out
        raise ValueError("gate not defined for the 3-gen multi-key scheme")
