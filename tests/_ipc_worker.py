"""One rank of tests/test_gpu_halo_ipc.py::test_two_processes_on_one_gpu: its row block of the grid on cuda:0, the deep halo
exchanged through the neighbour's IPC-mapped receive area (cx_api_ipc.hip), compared bit for bit with the whole grid swept in the
same process.  Rendezvous over gloo (127.0.0.1), used once to carry the memory handles."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rows, cols, depth, sweeps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    skew = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
    mode = int(sys.argv[7]) if len(sys.argv) > 7 else 0          # 0 plain, 1 the exchange around the owned part of sweep 1, 2 + the early push
    import time
    import numpy as np
    import torch
    import torch.distributed as dist

    import cortex.jl_amd as cx
    from cortex.jl_amd import _lib as L
    from cortex.jl_amd import partition

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    res = {"rank": rank, "ok": False}
    try:
        part = partition.grid_rows_deep(rows, cols, rank, world, depth, seed=5)
        dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
        cx.synth.load_into_device(part.model, dev, seed_variance=1e6)
        ex = partition.DeepHaloIpc(dev, part, dist, torch, torch.device("cuda", 0), overlap=mode == 1, early_push=mode == 2)
        dev.halo_ipc_set_timeout(30.0)
        res["audit_start"] = ex.audit(dist, torch, torch.device("cpu"))       # gloo carries the second copy
        if skew > 0:
            # ranks drift apart: every rank idles at batches of its own (rank r before batches r, r + world, ...), the device drained,
            # so that a neighbour's push of the NEXT exchange arrives while this rank still holds the previous one unread or is
            # about to read it — what the two parities of the receive area and the epoch flags are for
            done, batch = 0, 0
            while done < sweeps:
                run = min(depth, sweeps - done)
                if batch % world == rank:
                    dev.sync()
                    time.sleep(skew)
                ex.sweep(run)
                done += run
                batch += 1
        else:
            ex.sweep(sweeps)
        res["exchanges"] = ex.check()
        whole = cx.DeviceGraph(schedule=L.SCHED_FUSED)
        cx.synth.load_into_device(cx.synth.gaussian_grid(rows, cols, seed=5), whole, seed_variance=1e6)
        whole.sweep(sweeps)
        m = part.model
        own = np.isin(m.edge_var, part.owned_x)
        ev, ef = m.edge_var[own], m.edge_fac[own]
        same = all(np.array_equal(dev.get_messages(ev, ef, d, L.FORM_NATURAL), whole.get_messages(ev, ef, d, L.FORM_NATURAL), equal_nan=True)
                   for d in (L.TO_VARIABLE, L.TO_FACTOR))
        same = same and np.array_equal(dev.get_marginals(part.owned_x), whole.get_marginals(part.owned_x), equal_nan=True)
        res["ok"], res["owned"] = bool(same), int(len(part.owned_x))
        dist.barrier()                   # nobody frees its receive area while a neighbour may still push into it
    except Exception as e:               # pragma: no cover
        res["error"] = repr(e)
    with open(f"{out}.{rank}.json", "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
