#!/usr/bin/env python3
"""lab: what ONE exchange of the dim 64 chain scan costs a rank when C5 (T = 1e5) is cut into 8 time blocks: the block's potential
(compose launches + 100 KB to the host), the host's pass over the 8 gathered rows, the local exact sweep.  One GPU, ranks in threads
(the collective is a loopback), rank 3's phases timed on their own afterwards."""
import json
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L, partition  # noqa: E402
from tests.test_gpu_partition import LoopbackDist  # noqa: E402

T, world, d = 100_000, 8, 64
whole = cx.synth.lgssm_chain(T, d=d, seed=91)
ld = LoopbackDist(world, torch)
devs = [None] * world


def run(rank):
    ld.bind(rank)
    part = partition.contiguous_blocks(whole, rank, world)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(part.model, dev)
    ex = partition.ChainScanExchange(dev, part, ld, torch)
    ex.update()
    dev.sync()
    devs[rank] = (dev, ex)


def rounds(fn):
    th = [threading.Thread(target=fn, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]


rounds(run)
dev, ex = devs[3]


def timed(f, n=5):
    f()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e3


out = {"T": T, "world": world, "block_states": T // world}
out["chain_block_maps_ms"] = timed(lambda: dev.chain_block_maps())
out["local_sweep_ms"] = timed(lambda: dev.sweep(1))
out["block_maps_then_sweep_ms"] = timed(lambda: (dev.chain_block_maps(), dev.sweep(1)))


def again(rank):
    ld.bind(rank)
    ld.local.round = again.round
    t0 = time.perf_counter()
    devs[rank][1].update()
    devs[rank][0].sync()
    again.t[rank] = (time.perf_counter() - t0) * 1e3


again.t = [0.0] * world
for r in (1, 2, 3):
    again.round = r
    rounds(again)
out["update_ms_all_ranks_sharing_one_gpu"] = again.t
whole_dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
cx.synth.load_into_device(whole, whole_dev)
whole_dev.sweep(1)
out["whole_chain_sweep_ms"] = timed(lambda: whole_dev.sweep(1), 3)
print(json.dumps(out))
