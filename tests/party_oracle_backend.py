"""TEST INFRASTRUCTURE: the per-party kernels of thfhe.party_sharded played by the CPU oracle, so that the pipeline /
collective schedule of PartyShardedEvaluator can be rehearsed on CPU (gloo) and checked against the monolithic oracle.
Never imported by the product."""
import numpy as np
import torch

import oracle_lib as O

E8, E4 = 1 << 29, 1 << 30
LIN = {O.NAND: (E8, -1, -1, 0), O.OR: (E8, 1, 1, 0), O.AND: (-E8, 1, 1, 0), O.XOR: (E4, 2, 2, 0), O.AND3: (-E4, 1, 1, 1),
       -1: (0, 1, 0, 0)}


class OraclePartyBackend:
    def __init__(self, params, party, bk_part, ksk_part):
        """party: an index (key parts without a party axis) or a range (first, last+1) with key parts [parties of the range]..."""
        if isinstance(party, int):
            party, bk_part, ksk_part = (party, party + 1), np.asarray(bk_part)[None], np.asarray(ksk_part)[None]
        self.params, self.first, self.count = params, party[0], party[1] - party[0]
        self.party = self.first
        self.device = torch.device("cpu")
        d = params.as_dict()
        d["parties"] = self.count
        self.p1 = O.make_params(**d)
        self.orc = O.MKOracle(self.p1, np.asarray(bk_part), np.asarray(ksk_part))

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def prologue(self, op, which, x, y, z):
        p = self.params
        cb, cx, cy, cz = LIN[op] if op != O.MUX else ((-E8, 1, 1, 0) if which == 0 else (-E8, -1, 0, 1))
        v = cx * x.numpy().astype(np.int64)
        if cy:
            v = v + cy * y.numpy().astype(np.int64)
        if cz:
            v = v + cz * z.numpy().astype(np.int64)
        v[:, -1] += cb
        v = v.astype(np.uint32).view(np.int32)
        ms = np.vectorize(lambda w: O.lib().oracle_modswitch(int(w), p.N), otypes=[np.int32])
        return (torch.from_numpy(ms(v[:, self.first * p.n:(self.first + self.count) * p.n])), torch.from_numpy(ms(v[:, -1])))

    def rotate(self, bara, barb, mu, acc_in):
        p = self.params
        out = np.zeros((bara.shape[0], 2, p.N), np.int64)
        for g in range(bara.shape[0]):
            if acc_in is None:
                tv = np.full(p.N, mu, np.int64)
                acc = np.zeros((2, p.N), np.int64)
                O.lib().oracle_mul_by_monomial64(O.p64(tv), -int(barb[g]), p.N, O.p64(acc[1]))
            else:
                acc = acc_in[g].numpy().copy()
            for q in range(self.count):      # party-major, 3gen_mk_internals.jl:78-84
                for i in range(p.n):
                    a = int(bara[g, q * p.n + i])
                    if a != 0:
                        acc = self.orc.mux_rotate(q, i, a, acc)
            out[g] = acc.reshape(2, p.N)
        return torch.from_numpy(out)

    def extract(self, acc):
        p = self.params
        a = acc.numpy()
        t = np.vectorize(lambda w: O.lib().oracle_t64tot32(int(w)), otypes=[np.int32])
        u = np.zeros((a.shape[0], p.N + 1), np.int32)
        u[:, 0] = t(a[:, 0, 0])
        u[:, 1:p.N] = t(-a[:, 0, :0:-1])  # a'_j = -mask_{N-j}, wrapping
        u[:, p.N] = t(a[:, 1, 0])
        return torch.from_numpy(u)

    def keyswitch(self, u):
        return torch.from_numpy(np.stack([self.orc.keyswitch(r) for r in u.numpy()]))
