#!/usr/bin/env python3
"""tools/bench_strip.py — what ONE rank of the strong-scaling cut of config C4 costs per sweep, measured on one GPU.

The 1415 x 1415 grid cut into `--world` row blocks; this process holds block `--rank` with `--depth` redundant rows per
side (partition.grid_rows_deep).  Sweeps go to the library in batches of `depth` (one cx_sweep call per batch, as
partition.DeepHaloRccl issues them); with --exchange each batch is preceded by a state exchange of a middle rank's volume in
which the rank is its own neighbour (pack kernel, grouped RCCL send/recv to self, unpack kernel: the values written are not
the neighbours', so this mode is for timing only — bitwise equality with the whole grid is tests/test_gpu_partition.py's job).

    python tools/bench_strip.py --depth 8 16 32 [--exchange] [--sweeps 4000]
One JSON line per depth."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402
from cortex.jl_amd import partition  # noqa: E402


def self_exchange(part):
    """the same block as its own neighbour on both sides: every imported list is fed by an exported list of equal length"""
    ps = sorted(part.peers, key=lambda p: p.rank)
    n = [min(p.send.stop - p.send.start, p.recv.stop - p.recv.start) for p in ps]
    peers = [partition.Peer(0, slice(p.send.start, p.send.start + k), slice(p.recv.start, p.recv.start + k)) for p, k in zip(ps, n)]
    return partition.Partition(model=part.model, rank=0, world=1, send_var=part.send_var, send_fac=part.send_fac, recv_var=part.recv_var,
                               recv_fac=part.recv_fac, peers=peers, depth=part.depth, owned_x=part.owned_x, layer_var=part.layer_var, layer=part.layer)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=1415)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--depth", type=int, nargs="+", default=[8])
    ap.add_argument("--sweeps", type=int, default=4000)
    ap.add_argument("--exchange", action="store_true")
    ap.add_argument("--overlap-exchange", action="store_true", help="with --exchange: cx_halo_exchange_sweep (pack, send/recv, unpack on a second stream beside "
                                                                    "the owned part of the batch's first sweep) instead of serially on the compute stream")
    ap.add_argument("--ipc", action="store_true", help="with --exchange: push into the neighbour's IPC receive area + epoch flag (cx_halo_ipc_exchange: "
                                                       "two launches) instead of pack, RCCL send/recv, unpack")
    ap.add_argument("--ipc-form", choices=["two", "one", "overlap", "early"], default="two",
                    help="with --ipc: two launches per exchange (the default wherever neighbours may share a device); ONE launch (what "
                         "ranks on GPUs of their own run: cx_halo_ipc_set_fused; with the rank as its own neighbour this is a timing rig only); "
                         "overlap = cx_halo_ipc_exchange_sweep; early = cx_halo_ipc_batch (the next exchange pushed inside the last sweep of a batch)")
    ap.add_argument("--batch", type=int, default=0, help="sweeps per cx_sweep call (default: depth)")
    ap.add_argument("--no-trim", action="store_true", help="run every redundant row in every sweep (no cx_halo_set_layers)")
    ap.add_argument("--blocks", default="", help="RxC: cut the grid into R x C rectangular blocks (R * C = --world) instead of row strips and hold block "
                                                 "--block r,c (partition.by_assignment_deep on the whole model: no exchange modes, timing of the sweeps only)")
    ap.add_argument("--block", default="0,1")
    a = ap.parse_args()
    torch.cuda.init()
    N = a.grid
    # the whole grid, for the per-sweep figure the strip is compared with
    whole = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    whole.set_stream(torch.cuda.current_stream().cuda_stream)
    cx.synth.load_into_device(cx.synth.gaussian_grid(N, N, seed=1234), whole, seed_variance=1e6)
    whole.sweep(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    whole.sweep(1000)
    torch.cuda.synchronize()
    whole_us = (time.perf_counter() - t0) / 1000 * 1e6
    whole.close()
    for depth in a.depth:
        if a.blocks:
            R_, C_ = (int(x) for x in a.blocks.lower().split("x"))
            br, bc = (int(x) for x in a.block.split(","))
            assert R_ * C_ == a.world and not a.exchange
            rb, cb = np.linspace(0, N, R_ + 1).astype(np.int64), np.linspace(0, N, C_ + 1).astype(np.int64)

            def owner(ids):
                i, j = (np.asarray(ids) - 1) // N, (np.asarray(ids) - 1) % N
                return (np.searchsorted(rb, i, side="right") - 1) * C_ + (np.searchsorted(cb, j, side="right") - 1)
            part = partition.by_assignment_deep(cx.synth.gaussian_grid(N, N, seed=1234), owner, br * C_ + bc, a.world, depth)
        else:
            part = partition.grid_rows_deep(N, N, a.rank, a.world, depth, seed=1234)
        if a.no_trim:
            part.layer_var = part.layer = None
        dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
        dev.set_stream(torch.cuda.current_stream().cuda_stream)
        cx.synth.load_into_device(part.model, dev, seed_variance=1e6)
        st = dev.stats()
        ex = None
        if a.exchange:
            if a.ipc:
                form = "overlap" if a.overlap_exchange else a.ipc_form
                ex = partition.DeepHaloIpc(dev, self_exchange(part), overlap=form == "overlap", early_push=form == "early",
                                           peers_on_other_devices=form == "one")
            else:
                ex = partition.DeepHaloRccl(dev, self_exchange(part), None, torch, torch.device("cuda", 0), overlap=a.overlap_exchange)
        elif not a.no_trim:     # no exchange: the trimming schedule restarts every `depth` sweeps as if one had happened
            dev.halo_configure_state([], [], [], [])
            dev.halo_set_layers(part.layer_var, part.layer, depth)
        batch = a.batch or depth

        def run(n):
            if ex is not None:
                ex.sweep(n)
            else:
                while n > 0:
                    k = min(n, batch)
                    if not a.no_trim:
                        dev.halo_state_unpack()      # empty lists: only restarts the count of sweeps since the last exchange
                    dev.sweep(k)
                    n -= k
        run(400)
        torch.cuda.synchronize()
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            run(a.sweeps)
            e1.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / a.sweeps * 1e6
            devt = e0.elapsed_time(e1) / a.sweeps * 1e3
            best = (wall, devt) if best is None or wall < best[0] else best
        rows_owned = len(part.owned_x) // N
        rows_held = st["n_variables"] // N
        print(json.dumps({"strip": (f"block {a.block} of a {a.blocks} cut, {len(part.owned_x)} owned variables, {st['n_variables']} held incl. stand-ins" if a.blocks else
                                    f"rank {a.rank} of {a.world}, {rows_owned} owned rows + 2 x {depth} redundant ({rows_held} rows held incl. stand-ins)"),
                          "owned_variables": int(len(part.owned_x)), "held_variables": int(st["n_variables"]),
                          "depth": depth, "exchange": (("IPC push, owned part of sweep 1, unpack, rest: one stream" if (a.overlap_exchange or a.ipc_form == "overlap") else "IPC, next exchange pushed inside the last sweep of a batch, unpacked after the owned part of the first" if a.ipc_form == "early" else "IPC push + flag, ONE launch" if a.ipc_form == "one" else "IPC push + flag, two launches") if a.ipc else "overlapped with the owned part of the first sweep" if a.overlap_exchange else "RCCL, serial on the compute stream") if a.exchange else False, "us_per_sweep_wall": best[0], "us_per_sweep_device": best[1],
                          "whole_grid_us_per_sweep": whole_us, "ideal_us": whole_us / a.world,
                          "ratio_to_ideal": best[0] / (whole_us / a.world), "speedup_if_all_ranks_like_this": whole_us / best[0],
                          "slices": st["n_slices"], "halo_messages": int(len(part.recv_var))}), flush=True)
        if ex is not None and a.ipc:
            ex.check()
        dev.close()


if __name__ == "__main__":
    main()
