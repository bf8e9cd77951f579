"""-m gpu: the deep-halo exchange through IPC-mapped receive areas and epoch flags (cx_api_ipc.hip).
Acceptance is the partition's usual one: every message and marginal of an owned variable equals the un-partitioned device sweep
bit for bit — with the rank as its own neighbour, with several handles of one process, and with two PROCESSES on the one GPU that
open each other's memory handles (the form bench.py --halo ipc runs with one process per GPU)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from cortex.jl_amd import partition

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("overlap", [0, 1, 2])
@pytest.mark.parametrize("rows,cols,depth,sweeps", [(30, 64, 4, 11), (120, 300, 8, 27), (64, 256, 16, 33), (354, 1415, 16, 50), (90, 300, 2, 9)])
def test_ipc_exchange_self_neighbour(hip_lib, rows, cols, depth, sweeps, overlap):
    """overlap 1: cx_halo_ipc_exchange_sweep — push, the slices of owned variables only, wait + unpack, the rest of the first sweep
    (two launches of the sweep kernel for that sweep).  overlap 2: cx_halo_ipc_batch — also the NEXT exchange's push inside the last
    sweep of every full batch, between the slices that write the boundary state and the quiet rest."""
    part = partition.deep_self(rows, cols, depth, seed=8)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(part.model, dev, seed_variance=1e6)
    ex = partition.DeepHaloIpc(dev, part, overlap=overlap == 1, early_push=overlap == 2)
    dev.halo_ipc_set_timeout(5.0)
    import torch

    assert ex.audit(None, torch, torch.device("cuda", 0))      # the exchange of the first batch, audited
    dev.profile_enable(1)
    ex.sweep(sweeps)
    _ms, launches = dev.profile_read(L.KERNEL_FUSED)
    dev.profile_enable(0)
    exchanges = -(-sweeps // depth)
    assert ex.check() == exchanges
    # the audited first exchange was made on its own; every later batch of the overlapped form splits its first sweep
    if overlap < 2:
        assert launches == sweeps + (exchanges - 1 if overlap else 0)
    else:
        # full batches after the audited one: first AND last sweep in two parts (where the strip has a quiet run); never fewer launches than sweeps
        assert sweeps <= launches <= sweeps + 2 * exchanges
        if rows >= 300:
            assert launches >= sweeps + 2 * ((sweeps - depth) // depth), "the big strip has a quiet run: the last sweeps were split"
    plain = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(part.model, plain, seed_variance=1e6)
    plain.sweep(sweeps)
    m = part.model
    assert np.array_equal(dev.get_messages(m.edge_var, m.edge_fac, L.TO_VARIABLE, L.FORM_NATURAL),
                          plain.get_messages(m.edge_var, m.edge_fac, L.TO_VARIABLE, L.FORM_NATURAL), equal_nan=True)
    assert np.array_equal(dev.get_marginals(m.x_ids), plain.get_marginals(m.x_ids), equal_nan=True)


@pytest.mark.parametrize("world,rows,cols,depth", [(2, 10, 9, 1), (3, 18, 40, 3), (3, 120, 300, 8)])
def test_ipc_exchange_between_handles_of_one_process(hip_lib, world, rows, cols, depth):
    """Two or three handles (a stream each) of this process, connected by device address.  Launches are asynchronous, so one host
    thread drives all ranks batch by batch; every rank pushes before any rank unpacks (streams of one process may share a hardware
    queue: a waiting unpack in front of the push it waits for would sit there until its time limit)."""
    sweeps = 2 * depth + 3
    whole = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(cx.synth.gaussian_grid(rows, cols, seed=21), whole, seed_variance=1e6)
    whole.sweep(sweeps)
    import torch

    parts = [partition.grid_rows_deep(rows, cols, r, world, depth, seed=21) for r in range(world)]
    devs, exs, streams = [], {}, [torch.cuda.Stream() for _ in range(world)]
    for r in range(world):
        dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
        dev.set_stream(streams[r].cuda_stream)       # a stream per rank: a waiting unpack must not block the neighbour's push
        cx.synth.load_into_device(parts[r].model, dev, seed_variance=1e6)
        devs.append(dev)
        exs[r] = partition.DeepHaloIpc(dev, parts[r], connect=False)
        dev.halo_ipc_set_timeout(5.0)
    for r in range(world):
        exs[r].connect({q: exs[q].info for q in range(world)})
    done = 0
    while done < sweeps:
        run = min(depth, sweeps - done)
        for r in range(world):
            devs[r].halo_ipc_push()
        for r in range(world):
            devs[r].halo_ipc_unpack()
            devs[r].sweep(run)
        done += run
    for r in range(world):
        assert exs[r].check() == -(-sweeps // depth)
        part, m = parts[r], parts[r].model
        own = np.isin(m.edge_var, part.owned_x)
        ev, ef = m.edge_var[own], m.edge_fac[own]
        for direction in (L.TO_VARIABLE, L.TO_FACTOR):
            assert np.array_equal(devs[r].get_messages(ev, ef, direction, L.FORM_NATURAL), whole.get_messages(ev, ef, direction, L.FORM_NATURAL),
                                  equal_nan=True), f"rank {r}"
        assert np.array_equal(devs[r].get_marginals(part.owned_x), whole.get_marginals(part.owned_x), equal_nan=True)


@pytest.mark.parametrize("d,T,world,depth", [(4, 300, 3, 3), (2, 64, 2, 2), (3, 90, 3, 2), (16, 40, 2, 2), (64, 21, 3, 2), (9, 30, 2, 3)])      # (round 6: the matrix-core dims)
def test_ipc_exchange_for_d_dimensional_messages(hip_lib, d, T, world, depth):
    """Time blocks of a d-dimensional chain with a deep halo: a message travels as the 16-byte pairs of its storage form.  Owned
    marginals and messages equal the un-partitioned device sweeps bit for bit (cf. test_gpu_partition.py, same model, caller-owned
    transport)."""
    import torch

    sweeps = 3 * depth + 2
    whole_model = cx.synth.lgssm_chain(T, d=d, seed=9)
    whole = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(whole_model, whole, seed_variance=1e6)
    whole.sweep(sweeps)
    parts = [partition.contiguous_blocks(whole_model, r, world, depth=depth) for r in range(world)]
    devs, exs, streams = [], {}, [torch.cuda.Stream() for _ in range(world)]
    for r in range(world):
        dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
        dev.set_stream(streams[r].cuda_stream)
        cx.synth.load_into_device(parts[r].model, dev, seed_variance=1e6)
        devs.append(dev)
        exs[r] = partition.DeepHaloIpc(dev, parts[r], connect=False)
        dev.halo_ipc_set_timeout(5.0)
    for r in range(world):
        exs[r].connect({q: exs[q].info for q in range(world)})
    done = 0
    while done < sweeps:
        run = min(depth, sweeps - done)
        for r in range(world):
            devs[r].halo_ipc_push()
        for r in range(world):
            devs[r].halo_ipc_unpack()
            devs[r].sweep(run)
        done += run
    total = 0
    for r in range(world):
        assert exs[r].check() == -(-sweeps // depth)
        ids, m = parts[r].owned_x, parts[r].model
        assert np.array_equal(devs[r].get_marginals(ids), whole.get_marginals(ids), equal_nan=True), f"rank {r}"
        own = np.isin(m.edge_var, ids)
        assert np.array_equal(devs[r].get_messages(m.edge_var[own], m.edge_fac[own], L.TO_VARIABLE, L.FORM_NATURAL),
                              whole.get_messages(m.edge_var[own], m.edge_fac[own], L.TO_VARIABLE, L.FORM_NATURAL), equal_nan=True)
        total += len(ids)
    assert total == T


def test_unpack_gives_up_on_a_neighbour_that_never_arrives(hip_lib):
    """The wait is bounded: the grid drains and cx_halo_ipc_status reports the missing neighbour."""
    import torch

    parts = [partition.grid_rows_deep(12, 9, r, 2, 2, seed=3) for r in range(2)]
    devs, exs, streams = [], {}, [torch.cuda.Stream() for _ in range(2)]
    for r in range(2):
        dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
        dev.set_stream(streams[r].cuda_stream)
        cx.synth.load_into_device(parts[r].model, dev, seed_variance=1e6)
        devs.append(dev)
        exs[r] = partition.DeepHaloIpc(dev, parts[r], connect=False)
        dev.halo_ipc_set_timeout(0.2)
    for r in range(2):
        exs[r].connect({q: exs[q].info for q in range(2)})
    exs[0].sweep(2)                      # rank 1 never exchanges
    with pytest.raises(cx.CortexHipError, match="no push is waiting"):
        devs[1].halo_ipc_unpack()
    with pytest.raises(RuntimeError, match="did not arrive"):
        exs[0].check()


def test_ipc_entry_points_refuse_misuse(hip_lib):
    part = partition.deep_self(12, 16, 2, seed=1)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(part.model, dev, seed_variance=1e6)
    with pytest.raises(cx.CortexHipError, match="cx_halo_configure_state first"):
        dev.halo_ipc_alloc()
    ex = partition.DeepHaloIpc(dev, part, connect=False)
    with pytest.raises(cx.CortexHipError, match="not connected"):
        dev.halo_ipc_exchange()
    with pytest.raises(cx.CortexHipError, match="outside the neighbour's receive area"):
        dev.halo_ipc_connect(0, 0, 10**9, ex.area_bytes, same_process_base=ex.base)
    with pytest.raises(cx.CortexHipError, match="no such peer entry"):
        dev.halo_ipc_connect(7, 0, 0, ex.area_bytes, same_process_base=ex.base)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,rows,cols,depth,sweeps,skew,overlap", [(2, 64, 128, 4, 19, 0.0, 0), (2, 354, 1415, 16, 35, 0.0, 0), (3, 96, 128, 4, 27, 0.15, 0),
                                                                      (2, 64, 128, 2, 21, 0.1, 0), (2, 354, 1415, 16, 51, 0.0, 1), (3, 96, 128, 4, 27, 0.15, 1),
                                                                      (2, 354, 1415, 16, 67, 0.0, 2), (3, 300, 256, 4, 31, 0.15, 2), (2, 64, 128, 2, 21, 0.1, 2)])
def test_processes_on_one_gpu(hip_lib, tmp_path, world, rows, cols, depth, sweeps, skew, overlap):
    """Two or three ranks, one PROCESS each, all on cuda:0: each opens its neighbours' hipIpcMemHandles and pushes into them.
    (354 x 1415 at depth 16 is the volume of two neighbouring ranks of bench.py --gpus 8; three ranks give the middle one two
    neighbours.)  skew > 0: the ranks take turns idling between batches, so pushes arrive early and late relative to the reader.
    overlap 1: cx_halo_ipc_exchange_sweep (the owned part of the first sweep between push and unpack); 2: cx_halo_ipc_batch (also the
    next exchange's push inside the last sweep of a batch)."""
    out, port, procs = str(tmp_path / "res"), _free_port(), []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ipc_worker.py"), str(rows), str(cols), str(depth),
                                       str(sweeps), out, str(skew), str(overlap)], env=env, cwd=ROOT))
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r in range(world):
        res = json.load(open(f"{out}.{r}.json"))
        assert res.get("ok") and res.get("audit_start"), res
        assert res["exchanges"] == -(-sweeps // depth)
