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


def main():
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
