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
        self.params, self.party = params, party
        self.device = torch.device("cpu")
        d = params.as_dict()
        d["parties"] = 1
        self.p1 = O.make_params(**d)
        self.orc = O.MKOracle(self.p1, np.asarray(bk_part)[None], np.asarray(ksk_part)[None])

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
        return (torch.from_numpy(ms(v[:, self.party * p.n:(self.party + 1) * p.n])), torch.from_numpy(ms(v[:, -1])))

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
            for i in range(p.n):
                a = int(bara[g, i])
                if a != 0:
                    acc = self.orc.mux_rotate(0, i, a, acc)
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
