"""Worker of tests/test_partition_gloo.py: one rank of a world_size-N gloo job on CPU.

Runs the SAME partition + HaloExchange code the GPU bench runs (cortex.jl_amd/partition.py), with the CPU checker
(oracle/bp_flood.c) standing in for the device sweeper, and writes this rank's messages and marginals to a file."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cortex.jl_amd import partition  # noqa: E402
from tests.helpers import flood_oracle_from_model  # noqa: E402


def _residual(sw):
    """max |change| of the factor→variable messages (mean and variance) since the last call"""
    cur = np.concatenate([sw.g.f2v_m, sw.g.f2v_v])
    prev = getattr(sw, "_prev", None)
    sw._prev = cur.copy()
    if prev is None:
        return float("inf")
    both = ~np.isnan(cur) & ~np.isnan(prev)
    return float(np.max(np.abs(cur[both] - prev[both]))) if both.any() else float("inf")


class OracleSweeper:
    def __init__(self, part, seed_variance):
        self.g = g = flood_oracle_from_model(part.model, seed_variance)
        self.send_e = g.edge_index(part.send_var, part.send_fac) if len(part.send_var) else np.zeros(0, np.int64)
        self.recv_e = g.edge_index(part.recv_var, part.recv_fac) if len(part.recv_var) else np.zeros(0, np.int64)
        g.fixed_v2f[self.recv_e] = 1
        self.send = torch.zeros((max(len(self.send_e), 1), 2), dtype=torch.float64)
        self.recv = torch.zeros((max(len(self.recv_e), 1), 2), dtype=torch.float64)

    def sweep_begin(self):
        self.g.sweep(1, phases=1)
        n = len(self.send_e)
        self.send[:n, 0] = torch.from_numpy(self.g.v2f_m[self.send_e])
        self.send[:n, 1] = torch.from_numpy(self.g.v2f_v[self.send_e])

    def sweep_main(self):
        pass

    def sweep_end(self):
        n = len(self.recv_e)
        self.g.v2f_m[self.recv_e] = self.recv[:n, 0].numpy()
        self.g.v2f_v[self.recv_e] = self.recv[:n, 1].numpy()
        self.g.sweep(1, phases=2)

    def residual(self):
        return _residual(self)


class OracleStateSweeper:
    """Deep halo with the CPU checker as the sweeper: the exchanged state is the factor→variable messages."""

    def __init__(self, part, seed_variance):
        self.g = g = flood_oracle_from_model(part.model, seed_variance)
        self.send_e = g.edge_index(part.send_var, part.send_fac) if len(part.send_var) else np.zeros(0, np.int64)
        self.recv_e = g.edge_index(part.recv_var, part.recv_fac) if len(part.recv_var) else np.zeros(0, np.int64)
        self.send = torch.zeros((max(len(self.send_e), 1), 2), dtype=torch.float64)
        self.recv = torch.zeros((max(len(self.recv_e), 1), 2), dtype=torch.float64)

    def pack(self):
        n = len(self.send_e)
        self.send[:n, 0] = torch.from_numpy(self.g.f2v_m[self.send_e])
        self.send[:n, 1] = torch.from_numpy(self.g.f2v_v[self.send_e])

    def unpack(self):
        n = len(self.recv_e)
        self.g.f2v_m[self.recv_e] = self.recv[:n, 0].numpy()
        self.g.f2v_v[self.recv_e] = self.recv[:n, 1].numpy()

    def sweep(self, n=1):
        self.g.sweep(n)

    def residual(self):
        return _residual(self)


class OracleMvStateSweeper:
    """Deep halo for d-dimensional messages with the C flooding checker (oracle/mv_flood.c) as the sweeper: the exchanged state
    is every factor→variable message of the redundant variables, as (mean, covariance, defined) rows."""

    def __init__(self, part):
        from oracle.mv import MvFloodC

        self.o = o = MvFloodC(part.model)
        g, d = o.g, o.d
        self.send_e = g.edge_index(part.send_var, part.send_fac) if len(part.send_var) else np.zeros(0, np.int64)
        self.recv_e = g.edge_index(part.recv_var, part.recv_fac) if len(part.recv_var) else np.zeros(0, np.int64)
        w = d + d * d + 1
        self.send = torch.zeros((max(len(self.send_e), 1), w), dtype=torch.float64)
        self.recv = torch.zeros((max(len(self.recv_e), 1), w), dtype=torch.float64)

    def pack(self):
        o, n, d = self.o, len(self.send_e), self.o.d
        self.send[:n, :d] = torch.from_numpy(o.f2v_m[self.send_e])
        self.send[:n, d:d + d * d] = torch.from_numpy(o.f2v_S[self.send_e].reshape(n, d * d))
        self.send[:n, -1] = torch.from_numpy(o.f2v_def[self.send_e].astype(np.float64))

    def unpack(self):
        o, n, d = self.o, len(self.recv_e), self.o.d
        r = self.recv[:n].numpy()
        o.f2v_m[self.recv_e] = r[:, :d]
        o.f2v_S[self.recv_e] = r[:, d:d + d * d].reshape(n, d, d)
        o.f2v_def[self.recv_e] = r[:, -1].astype(np.uint8)

    def sweep(self, n=1):
        self.o.sweep(n)


class OracleChainBlock:
    """CPU stand-in for a chain-scan DeviceGraph holding ONE time block (what partition.ChainScanExchange drives): the same
    three entry points — chain_block_maps, set_messages, sweep — in numpy, natural form, with the map algebra of
    csrc/cx_chain.hip."""

    def __init__(self, part):
        m = part.model
        self.x = np.sort(np.asarray(m.x_ids))
        fv = dict(zip(np.asarray(m.factor_ids).tolist(), np.asarray(m.factor_var).tolist()))
        ev, ef = np.asarray(m.edge_var), np.asarray(m.edge_fac)
        fac_of = {}
        for v, f in zip(ev.tolist(), ef.tolist()):
            fac_of.setdefault(f, []).append(v)
        own = set(self.x.tolist())
        pos = {int(v): i for i, v in enumerate(self.x)}
        n = len(self.x)
        self.q = np.zeros(max(n - 1, 0))
        self.side = np.zeros((n, 2))
        y = dict(zip(np.asarray(m.data_var).tolist(), np.asarray(m.data_y).tolist()))
        for f, vs in fac_of.items():
            a, b = sorted(vs)
            if a in own and b in own:
                self.q[pos[a]] = fv[f]                       # transition x_a -> x_b, consecutive in the block
            elif (a in y) != (b in y):                       # likelihood: datum on one side
                xv, yv = (b, a) if a in y else (a, b)
                self.side[pos[xv]] += [y[yv] / fv[f], 1.0 / fv[f]]
        self.boundary = {}
        self.ends = {}                                       # cut factor -> (own end variable, variance)
        for f, vs in fac_of.items():
            a, b = sorted(vs)
            if (a in own) != (b in own) and a not in y and b not in y:
                self.ends[f] = (a if a in own else b, fv[f])

    @staticmethod
    def _link(u, q):
        D = 1.0 + q * u[1]
        return np.array([1.0 / D, 0.0, u[0] / D, 1.0 / D, u[1] / D, q / D])

    @staticmethod
    def _compose(first, second):
        e1, f1, g1, A1, B1, C1 = first
        e2, f2, g2, A2, B2, C2 = second
        inv = 1.0 / (C2 * B1 + 1.0)
        return np.array([e2 * e1, e2 * f1 + f2 * A1 + g2 * C1, e2 * g1 + f2 * B1 + g2, A2 * A1 + B2 * C1, A2 * B1 + B2, C2 * A1 + C1]) * inv

    def chain_block_maps(self):
        n = len(self.x)
        ident = np.array([1.0, 0.0, 0.0, 1.0, 0.0, 0.0])
        F, B = ident.copy(), ident.copy()
        for l in range(n - 1):
            F = self._link(self.side[l], self.q[l]) if l == 0 else self._compose(F, self._link(self.side[l], self.q[l]))
        for l in range(n - 2, -1, -1):
            Lk = self._link(self.side[l + 1], self.q[l])
            B = Lk if l == n - 2 else self._compose(B, Lk)
        return F, B, self.side[0].copy(), self.side[-1].copy(), int(self.x[0]), int(self.x[-1]), n - 1

    def set_messages(self, var, fac, direction, form, payload):
        from cortex.jl_amd import _lib as L

        if direction == L.TO_FACTOR:                         # a stand-in's message into its cut factor: the block's boundary input
            self.boundary[int(fac[0])] = np.asarray(payload, dtype=float)

    def sweep(self, n=1):
        from cortex.jl_amd.partition import _lin_apply, _rule_additive

        nx = len(self.x)
        alpha, beta = np.zeros((nx, 2)), np.zeros((nx, 2))
        for f, (end, q) in self.ends.items():
            b = self.boundary.get(f)
            if b is None or np.isnan(b[1]):
                continue
            if end == self.x[0] and not (nx == 1 and f == max(self.ends)):
                alpha[0] = _rule_additive(q, b)
            else:
                beta[-1] = _rule_additive(q, b)
        if nx == 1 and len(self.ends) == 2:                  # a one-variable block: lower cut factor id is the left one
            fl, fr = sorted(self.ends)
            alpha[0] = _rule_additive(self.ends[fl][1], self.boundary[fl]); beta[0] = _rule_additive(self.ends[fr][1], self.boundary[fr])
        for l in range(nx - 1):
            alpha[l + 1] = _rule_additive(self.q[l], alpha[l] + self.side[l])
        for l in range(nx - 2, -1, -1):
            beta[l] = _rule_additive(self.q[l], beta[l + 1] + self.side[l + 1])
        nat = alpha + beta + self.side
        self.marg_mean, self.marg_var = nat[:, 0] / nat[:, 1], 1.0 / nat[:, 1]


class OracleMvChainBlock:
    """CPU stand-in for a dim 2..4 chain-scan DeviceGraph holding ONE time block (what partition.ChainScanExchange drives for
    d-dimensional chains): halo_configure, chain_block_maps, set_messages, sweep in numpy — the map algebra of csrc/cx_mvchain.hip
    (f(eta, Lambda) = (c + B (Lambda + P)^-1 (eta + h), C - B (Lambda + P)^-1 B') and its composition) written independently."""

    def __init__(self, part):
        m = part.model
        self.d = d = m.dim
        self.A, self.Q = (np.asarray(z, float) for z in m.psets[0])
        R = np.asarray(m.psets[1][1], float)
        Qi, Ri = np.linalg.inv(self.Q), np.linalg.inv(R)
        A = self.A
        self.tab_f = (A.T @ Qi @ A, Qi @ A, Qi)              # receiver = out (the later state)
        self.tab_b = (Qi, A.T @ Qi, A.T @ Qi @ A)            # receiver = in
        ghosts = set(np.asarray(part.recv_var).tolist())
        self.x = np.array(sorted(v for v in np.asarray(m.x_ids).tolist() if v not in ghosts))
        y = dict(zip(np.asarray(m.data_var).tolist(), np.asarray(m.data_y)))
        pos = {int(v): i for i, v in enumerate(self.x)}
        n = len(self.x)
        self.side = [(np.zeros(d), np.zeros((d, d))) for _ in range(n)]
        fac_of = {}
        for v, f in zip(np.asarray(m.edge_var).tolist(), np.asarray(m.edge_fac).tolist()):
            fac_of.setdefault(f, []).append(v)
        self.cuts = {}                                       # cut factor -> "left" | "right"
        for f, vs in fac_of.items():
            a, b = sorted(vs)
            if a in pos and b in y:
                self.side[pos[a]] = (Ri @ y[b], Ri.copy())
            elif (a in ghosts) != (b in ghosts):
                own = b if a in ghosts else a
                self.cuts[f] = "left" if own == self.x[0] and (n > 1 or a in ghosts) else "right"
        self.boundary, self.after_cut = {}, {}

    def halo_configure(self, *a):
        pass

    @staticmethod
    def _rule(m, tab):
        P, B, C = tab
        W = np.linalg.inv(m[1] + P)
        return B @ W @ m[0], C - B @ W @ B.T

    @staticmethod
    def _compose(m1, m2):
        P1, B1, C1, h1, c1 = m1
        P2, B2, C2, h2, c2 = m2
        S = np.linalg.inv(C1 + P2)
        g = c1 + h2
        return (P1 - B1.T @ S @ B1, B2 @ S @ B1, C2 - B2 @ S @ B2.T, h1 + B1.T @ S @ g, c2 + B2 @ S @ g)

    def _pack(self, m):
        iu = np.triu_indices(self.d)
        P, B, C, h, c = m
        return np.concatenate([P[iu], B.ravel(), C[iu], h, c])

    def chain_block_maps(self):
        n, d = len(self.x), self.d
        link = lambda u, tab: (tab[0] + u[1], tab[1], tab[2], u[0], np.zeros(d))      # noqa: E731
        F = B = None
        for l in range(n - 1):
            e = link(self.side[l], self.tab_f)
            F = e if F is None else self._compose(F, e)
        for l in range(n - 2, -1, -1):
            e = link(self.side[l + 1], self.tab_b)
            B = e if B is None else self._compose(B, e)
        iu = np.triu_indices(d)
        ps = lambda u: np.concatenate([u[0], u[1][iu]])     # noqa: E731
        return self._pack(F), self._pack(B), ps(self.side[0]), ps(self.side[-1]), int(self.x[0]), int(self.x[-1]), n - 1

    def set_messages(self, var, fac, direction, form, payload):
        from cortex.jl_amd import _lib as L

        p = np.asarray(payload, dtype=float)
        if direction == L.TO_FACTOR:                         # a stand-in's message into its cut factor: the block's boundary input
            self.boundary[int(fac[0])] = (p[:self.d], p[self.d:].reshape(self.d, self.d))
        elif int(fac[0]) in self.cuts:                       # dim 64: the cut factor's message into the block's end variable, rule applied by the sender
            self.after_cut[int(fac[0])] = (p[:self.d], p[self.d:].reshape(self.d, self.d))

    def sweep(self, n=1):
        nx, d = len(self.x), self.d
        zero = (np.zeros(d), np.zeros((d, d)))
        add = lambda a, b: (a[0] + b[0], a[1] + b[1])        # noqa: E731
        alpha, beta = [zero] * nx, [zero] * nx
        for f, where in self.cuts.items():
            b, a = self.boundary.get(f), self.after_cut.get(f)
            if a is not None and (b is None or np.isnan(b[1]).any()):
                if where == "left":
                    alpha[0] = a
                else:
                    beta[-1] = a
                continue
            if b is None or np.isnan(b[1]).any():
                continue
            if where == "left":
                alpha[0] = self._rule(b, self.tab_f)
            else:
                beta[-1] = self._rule(b, self.tab_b)
        for l in range(nx - 1):
            alpha[l + 1] = self._rule(add(alpha[l], self.side[l]), self.tab_f)
        for l in range(nx - 2, -1, -1):
            beta[l] = self._rule(add(beta[l + 1], self.side[l + 1]), self.tab_b)
        self.mean, self.cov = np.zeros((nx, d)), np.zeros((nx, d, d))
        for t in range(nx):
            e, lam = add(add(alpha[t], beta[t]), self.side[t])
            self.cov[t] = np.linalg.inv(lam)
            self.mean[t] = self.cov[t] @ e


def main_mvchain():
    """argv: mvchain d T out — time blocks of a d-dimensional chain, ChainScanExchange over gloo (one all-gather of the blocks' maps)"""
    import cortex.jl_amd as cx

    d, T, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    whole = cx.synth.lgssm_chain(T, d=d, seed=12)
    part = partition.contiguous_blocks(whole, rank, world)
    blk = OracleMvChainBlock(part)
    ex = partition.ChainScanExchange(blk, part, dist, torch)
    ex.update()
    np.savez(out + f".rank{rank}.npz", x=blk.x, mean=blk.mean, cov=blk.cov)
    dist.barrier()
    dist.destroy_process_group()


def main_chain():
    """argv: chain T out — time blocks of a scalar state-space chain, ChainScanExchange over gloo"""
    import cortex.jl_amd as cx

    T, out = int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    whole = cx.synth.ssm_chain(T, seed=8, random_variances=True)
    part = partition.contiguous_blocks(whole, rank, world)
    blk = OracleChainBlock(part)
    ex = partition.ChainScanExchange(blk, part, dist, torch)
    ex.update()
    np.savez(out + f".rank{rank}.npz", x=blk.x, mean=blk.marg_mean, var=blk.marg_var)
    dist.barrier()
    dist.destroy_process_group()


def main_mv():
    """argv: mv d T sweeps depth out — time blocks of a d-dimensional chain with a deep halo over gloo"""
    import cortex.jl_amd as cx

    d, T, sweeps, depth, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    whole = cx.synth.lgssm_chain(T, d=d, seed=6)
    part = partition.contiguous_blocks(whole, rank, world, depth=depth)
    sw = OracleMvStateSweeper(part)
    sw.o.seed(0.0, 1e6)
    ex = partition.DeepHaloExchange(sw, part, dist)
    ex.sweep(sweeps)
    m, S, ok = sw.o.marginals()
    li = np.searchsorted(sw.o.g.var_ids, part.owned_x)
    np.savez(out + f".rank{rank}.npz", owned=part.owned_x, mean=m[li], cov=S[li], ok=ok[li])
    dist.barrier()
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "chain":
        return main_chain()
    if sys.argv[1] == "mvchain":
        return main_mvchain()
    if sys.argv[1] == "mv":
        return main_mv()
    rows, cols, sweeps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    depth = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    strong = len(sys.argv) > 6 and sys.argv[6] == "strong"    # `rows` is then the whole grid's row count (uneven blocks)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if depth:
        part = (partition.grid_rows_deep if strong else partition.grid_strip_deep)(rows, cols, rank, world, depth, seed=99)
        sw = OracleStateSweeper(part, 1e6)
        ex = partition.DeepHaloExchange(sw, part, dist)
        if sweeps < 0:        # convergence mode: sweep to a global residual instead of a fixed count
            n_run, res = partition.converge(ex, sw.residual, dist, torch, 1e-12, 4000, 2 * depth)
            np.save(out + f".rank{rank}.conv.npy", np.array([n_run, res]))
        else:
            ex.sweep(sweeps)
    else:
        part = (partition.grid_rows if strong else partition.grid_strip)(rows, cols, rank, world, seed=99)
        sw = OracleSweeper(part, 1e6)
        ex = partition.HaloExchange(sw, part, dist)
        if sweeps < 0:        # the per-sweep message halo under the convergence loop (HaloExchange.sweep(k))
            n_run, res = partition.converge(ex, sw.residual, dist, torch, 1e-12, 4000, 5)
            np.save(out + f".rank{rank}.conv.npy", np.array([n_run, res]))
        else:
            ex.sweep(sweeps)
    # the audit bench.py runs after its timed region: must pass on a correct exchange and fail on a corrupted buffer
    audit_ok = partition.verify_last_exchange(part, sw.send, sw.recv, dist, torch)
    if rank == 0:
        sw.recv[0, 0] += 1.0
    audit_bad = partition.verify_last_exchange(part, sw.send, sw.recv, dist, torch)
    if rank == 0:
        sw.recv[0, 0] -= 1.0
    g = sw.g
    m, v = g.marginals()
    np.savez(out + f".rank{rank}.npz", edge_var=g.edge_var, edge_fac=g.edge_fac, f2v_m=g.f2v_m, f2v_v=g.f2v_v,
             v2f_m=g.v2f_m, v2f_v=g.v2f_v, var_ids=g.var_ids, marg_m=m, marg_v=v, owned=part.model.x_ids if part.owned_x is None else part.owned_x,
             audit_ok=audit_ok, audit_bad=audit_bad)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
