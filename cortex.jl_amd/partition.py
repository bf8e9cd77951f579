"""Graph partitions and the once-per-sweep halo exchange.

The reference has no distributed path at all; this is new functionality built on the same C ABI.
A rank's local graph holds the variables it owns, every factor touching them, and — for factors cut by the
partition — the remote variable as a degree-1 *ghost* whose variable→factor message is imported each sweep.
Per sweep and per neighbouring rank exactly one message per cut factor travels in each direction
(16 bytes, natural form): grid strips exchange n_cols messages per boundary.  No collective is involved:
point-to-point send/recv between partition neighbours (RCCL over xGMI through torch.distributed on GPUs,
gloo in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import synth


@dataclass
class Peer:
    rank: int
    send: slice      # rows of the send buffer that go to this peer
    recv: slice      # rows of the recv buffer filled by this peer


@dataclass
class Partition:
    model: synth.Model
    rank: int
    world: int
    send_var: np.ndarray     # (variable_id, factor_id) of exported variable→factor messages, grouped by peer
    send_fac: np.ndarray
    recv_var: np.ndarray     # (ghost variable_id, factor_id) of imported messages, grouped by peer
    recv_fac: np.ndarray
    peers: List[Peer] = field(default_factory=list)


def grid_strip(rows_per_rank: int, n_cols: int, rank: int, world: int, seed: int = 1234) -> Partition:
    """Row-strip partition of the (rows_per_rank*world) x n_cols Gaussian grid: rank r owns rows
    [r*rows_per_rank, (r+1)*rows_per_rank).  Cut = the vertical factors between neighbouring strips
    (n_cols factors per boundary).  METIS is not available in this image; for a regular grid strips minimise the
    number of neighbours (2) and give perfectly balanced parts."""
    total = rows_per_rank * world
    r0, r1 = rank * rows_per_rank, (rank + 1) * rows_per_rank
    model = synth.gaussian_grid(total, n_cols, seed=seed, row0=r0, row1=r1)
    V, H = total * n_cols, total * (n_cols - 1)
    jj = np.arange(n_cols, dtype=np.int64)
    vid = lambda i: 1 + i * n_cols + jj                  # noqa: E731
    vfac = lambda i: 2 * V + H + 1 + i * n_cols + jj     # noqa: E731  factor between rows i and i+1
    sv, sf, rv, rf, peers = [], [], [], [], []
    pos_s = pos_r = 0
    if rank > 0:       # boundary with the strip above: cut factors between rows r0-1 | r0
        sv.append(vid(r0)); sf.append(vfac(r0 - 1)); rv.append(vid(r0 - 1)); rf.append(vfac(r0 - 1))
        peers.append(Peer(rank - 1, slice(pos_s, pos_s + n_cols), slice(pos_r, pos_r + n_cols)))
        pos_s += n_cols; pos_r += n_cols
    if rank < world - 1:  # boundary with the strip below: cut factors between rows r1-1 | r1
        sv.append(vid(r1 - 1)); sf.append(vfac(r1 - 1)); rv.append(vid(r1)); rf.append(vfac(r1 - 1))
        peers.append(Peer(rank + 1, slice(pos_s, pos_s + n_cols), slice(pos_r, pos_r + n_cols)))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)  # noqa: E731
    return Partition(model=model, rank=rank, world=world, send_var=cat(sv), send_fac=cat(sf), recv_var=cat(rv),
                     recv_fac=cat(rf), peers=peers)


class DeviceSweeper:
    """Adapter: a DeviceGraph whose halo buffers are torch tensors (so torch.distributed can move them)."""

    def __init__(self, dev, part: Partition, torch, device):
        self.dev = dev
        dev.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
        self.send = torch.zeros((max(len(part.send_var), 1), 2), dtype=torch.float64, device=device)
        self.recv = torch.zeros((max(len(part.recv_var), 1), 2), dtype=torch.float64, device=device)
        dev.halo_set_buffers(self.send.data_ptr(), self.recv.data_ptr())

    def sweep_begin(self):
        self.dev.sweep_begin()

    def sweep_main(self):
        self.dev.sweep_main()

    def sweep_end(self):
        self.dev.sweep_end()


class HaloExchange:
    """One partitioned sweep = begin (pack the exported messages) → start send/recv with the partition neighbours →
    main sweep (overlaps the exchange) → wait → end (unpack + push the imported messages through the cut factors).
    `sweeper` provides sweep_begin/main/end and the `send` / `recv` tensors; `dist` is torch.distributed."""

    def __init__(self, sweeper, part: Partition, dist):
        self.sw, self.part, self.dist = sweeper, part, dist

    def sweep(self):
        dist, sw = self.dist, self.sw
        sw.sweep_begin()
        ops = []
        for p in self.part.peers:
            ops.append(dist.P2POp(dist.isend, sw.send[p.send], p.rank))
            ops.append(dist.P2POp(dist.irecv, sw.recv[p.recv], p.rank))
        works = dist.batch_isend_irecv(ops) if ops else []
        sw.sweep_main()
        for w in works:
            w.wait()
        sw.sweep_end()
