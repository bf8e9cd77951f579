#!/usr/bin/env python3
"""bench.py — edge-message updates/sec per sweep on the 10M-edge Gaussian grid (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full sum-product sweep (every variable→factor message, every factor→variable message,
every marginal) over the N x N grid held by each rank, i.e. one `update_marginals!` of the reference
(src/inference_engine.jl:559-632) in the device's flooding order.  Weak scaling: every rank owns one
1415 x 1415 strip (10,005,465 bipartite edges) of a (1415*N) x 1415 grid; messages on the cut rows are
exchanged once per sweep (RCCL over xGMI through torch.distributed).  value = message updates of all
ranks / max-over-ranks time.  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel, hipEvent-timed on the library's stream)
and, at N = 1, `cpu_baseline` (the CPU restatement of the reference scheduler on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
BYTES_PER_UPDATE = 32          # read the 16-byte payload once + write it once, f64 scalar Gaussian


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--event-stride", type=int, default=8, help="hipEvent-time every n-th launch of the timed region")
    ap.add_argument("--grid", type=int, default=1415, help="N: each rank holds an N x N grid strip (1415 -> 10,005,465 edges)")
    ap.add_argument("--schedule", choices=["flooding", "fused"], default=os.environ.get("CX_BENCH_SCHEDULE", "fused"))
    ap.add_argument("--materialize", action="store_true", help="also store every variable→factor message each sweep")
    ap.add_argument("--halo", choices=["rccl", "torch"], default=os.environ.get("CX_HALO", "rccl"),
                    help="N > 1: exchange issued by the library on RCCL (default) or by torch.distributed isend/irecv")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank owns an N x N strip of an (N*ranks) x N grid; strong: the ONE N x N grid "
                         "(BASELINE config 4: 10M edges, 8-way cut) is split into row blocks over the ranks")
    ap.add_argument("--halo-depth", type=int, default=int(os.environ.get("CX_HALO_DEPTH", "8")),
                    help="deep halo: each rank keeps this many redundant rows of its neighbours and exchanges their state once "
                         "per that many sweeps (bit-identical to the un-partitioned sweep); 0 = one message halo per sweep")
    ap.add_argument("--cpu-configs", action="store_true", help="CPU baselines of configs C1, C2 and the C4 sample only (no GPU needed)")
    ap.add_argument("--self-halo", action="store_true",
                    help="N = 1 experiment: a cylinder whose wrap-around cut makes rank 0 its own halo neighbour")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-grid", type=int, default=768, help="grid side of the bounded CPU sample")
    ap.add_argument("--seed", type=int, default=1234)
    return ap.parse_args()


def _usable_cores() -> int:
    """cores this process may actually use: the affinity mask, capped by the cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(sample_n: int, seed: int) -> dict:
    """The reference's CPU path, restated (oracle/cortex_ref.c): one `update_marginals!` over all variables of a
    seeded sample grid, single thread (the reference has no threading).  Checker code, timed as a baseline only."""
    cores = _usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))     # read by libgomp when the checker library is first loaded
    from oracle import ref
    import cortex.jl_amd as cx
    from tests.helpers import engine_oracle_from_model

    model = cx.synth.gaussian_grid(sample_n, sample_n, seed=seed)
    E = engine_oracle_from_model(model)
    g = ref.FloodGraph(model.edge_var, model.edge_fac, model.factor_ids, model.factor_var)
    pe = g.partner >= 0
    E.set_messages_to_variable(g.edge_var[pe], g.edge_fac[pe], np.zeros(int(pe.sum())), np.full(int(pe.sum()), 1e6))
    total_upd, total_t, reps = 0, 0.0, 0
    while total_t < 10.0 and reps < 50:
        if reps > 0:  # a reference user re-sets the priors to make them fresh again before the next iteration
            E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
        c0 = E.counters()[0]
        t0 = time.perf_counter()
        E.update_marginals(model.x_ids)
        total_t += time.perf_counter() - t0
        total_upd += E.counters()[0] - c0
        reps += 1
    out = {"value": total_upd / total_t, "unit": "edge-message updates/s", "cores": 1, "kind": "port",
           "sample": f"{reps} update_marginals! sweeps of a {sample_n}x{sample_n} Gaussian grid ({model.n_edges} edges), "
                     f"restated reference scheduler (sequential, readiness bits), {total_t:.1f} s of CPU work"}
    # second figure: the same arithmetic as a flooding sweep over flat arrays on ALL host cores (OpenMP):
    # what a CPU gets once the reference's per-signal bookkeeping is taken away
    from tests.helpers import flood_oracle_from_model
    cores = int(os.environ.get("OMP_NUM_THREADS", cores))
    fg = flood_oracle_from_model(model, 1e6)
    fg.sweep(2, use_omp=True)
    t0, n_upd, sw = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < 4.0:
        n_upd += fg.sweep(4, use_omp=True)
        sw += 4
    dt = time.perf_counter() - t0
    import shutil
    out["reference_julia"] = shutil.which("julia") or "not on this box (BASELINE.md §3.1: the real reference is timed only where Julia is installed)"
    out["flooding_all_cores"] = {"value": n_upd / dt, "unit": "edge-message updates/s", "cores": cores, "kind": "port",
                                 "sample": f"{sw} flooding sweeps of the same grid, flat arrays + OpenMP over {cores} cores, {dt:.1f} s"}
    return out


def cpu_config_table(seed: int):
    """the restated reference scheduler (one core) on configs C1 and C2 and on the C4 sample, one JSON line
    per config.  CPU only; `python bench.py --cpu-configs`."""
    from tests.helpers import engine_oracle_from_model
    import cortex.jl_amd as cx

    rows = []
    for name, T in (("C1", 1_000), ("C2", 250_001)):
        model = cx.synth.ssm_chain(T, seed=seed)
        E = engine_oracle_from_model(model)
        E.set_messages_to_factor(model.data_var, model.data_fac, model.data_y)
        c0 = E.counters()[0]
        t0 = time.perf_counter()
        E.update_marginals(model.x_ids)
        dt = time.perf_counter() - t0
        upd = E.counters()[0] - c0
        rows.append({"config": name, "schedule": "reference update_marginals! (restated, sequential)", "device": "cpu", "cores": 1,
                     "updates_per_sweep": upd, "ms_per_sweep": dt * 1e3, "updates_per_s": upd / dt,
                     "algorithmic_GBps": upd * BYTES_PER_UPDATE / dt / 1e9})
    c4 = cpu_baseline(768, seed)
    rows.append({"config": "C4 sample (768x768)", "schedule": "reference update_marginals! (restated, sequential)", "device": "cpu",
                 "cores": 1, "updates_per_s": c4["value"], "sample": c4["sample"]})
    fa = c4["flooding_all_cores"]
    rows.append({"config": "C4 sample (768x768)", "schedule": "flooding, flat arrays + OpenMP", "device": "cpu", "cores": fa["cores"],
                 "updates_per_s": fa["value"], "sample": fa["sample"]})
    for r in rows:
        print(json.dumps(r), flush=True)


def _watchdog(seconds: float):
    """A multi-rank run that stops making progress (a peer died, a collective never matched) must fail fast instead of
    sitting on the node until the driver's limit."""
    import threading

    def fire():
        sys.stderr.write(f"[bench] no completion after {seconds:.0f} s: aborting\n")
        sys.stderr.flush()
        os._exit(3)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main():
    args = parse()
    if args.cpu_configs:
        cpu_config_table(args.seed)
        return
    import torch
    import cortex.jl_amd as cx
    from cortex.jl_amd import _lib as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP sweep has no CPU fallback")
    # rehearsal knobs (never set by the driver): CX_DIST_BACKEND=gloo + CX_SINGLE_DEVICE=1 run N ranks against ONE GPU with
    # host-staged halos, to exercise the multi-rank control flow on a one-GPU box
    backend = os.environ.get("CX_DIST_BACKEND", "nccl")
    if os.environ.get("CX_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    dog = _watchdog(900.0)
    N = args.grid
    schedule = L.SCHED_FUSED if args.schedule == "fused" else L.SCHED_FLOODING
    dev = cx.DeviceGraph(device=local_rank, schedule=schedule, marginals_in_sweep=True,
                         materialize_messages_to_factor=args.materialize)
    stream = torch.cuda.current_stream()
    dev.set_stream(stream.cuda_stream)

    halo_kind = None
    halo_tensors = None
    if world == 1 and not args.self_halo:
        model = cx.synth.gaussian_grid(N, N, seed=args.seed)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        exchange = None
    else:
        from cortex.jl_amd import partition
        depth = min(args.halo_depth, N if args.scaling == "weak" else max(N // max(world, 1), 1))
        if world == 1:
            part = partition.deep_self(N, N, depth, seed=args.seed) if depth else partition.cylinder_self(N, N, seed=args.seed)[0]
        elif args.scaling == "strong":
            part = partition.contiguous_blocks(cx.synth.gaussian_grid(N, N, seed=args.seed), rank, world, depth=depth)
        elif depth:
            part = partition.grid_strip_deep(N, N, rank, world, depth, seed=args.seed)
        else:
            part = partition.grid_strip(N, N, rank, world, seed=args.seed)
        cx.synth.load_into_device(part.model, dev, seed_variance=1e6)
        exchange = None
        tdev = torch.device("cuda", local_rank)
        if backend != "nccl" and world > 1:
            if depth:
                sweeper = partition.HostStagedStateSweeper(dev, part, torch, tdev)
                exchange = partition.DeepHaloExchange(sweeper, part, dist)
            else:
                sweeper = partition.HostStagedSweeper(dev, part, torch, tdev)
                exchange = partition.HaloExchange(sweeper, part, dist)
            halo_kind = f"REHEARSAL: host-staged over {backend}"
            halo_tensors = (sweeper.send, sweeper.recv)
        elif args.halo == "rccl":
            err = None
            try:
                exchange = (partition.DeepHaloRccl if depth else partition.RcclExchange)(dev, part, dist, torch, tdev)
                halo_kind = "rccl send/recv issued by the library"
                halo_tensors = (exchange.send, exchange.recv)
            except cx.CortexHipError as e:   # e.g. librccl not loadable: fall back to torch.distributed
                err = e
            if dist is not None:             # all ranks take the same path
                okf = torch.tensor([0 if err else 1], dtype=torch.int32, device=red_dev)
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                if okf.item() == 0:
                    exchange = None
            if exchange is None and rank == 0:
                print(f"[bench] RCCL exchange unavailable ({err}); falling back to torch.distributed", file=sys.stderr)
        if exchange is None:
            if depth:
                sweeper = partition.DeviceStateSweeper(dev, part, torch, tdev)
                exchange = partition.DeepHaloExchange(sweeper, part, dist)
            else:
                sweeper = partition.DeviceSweeper(dev, part, torch, tdev)
                exchange = partition.HaloExchange(sweeper, part, dist)
            halo_kind = "torch.distributed isend/irecv"
            halo_tensors = (sweeper.send, sweeper.recv)
        if depth:
            halo_kind = f"deep halo, {depth} redundant rows per side, state exchanged every {depth} sweeps; " + halo_kind
        else:
            halo_kind = "message halo per sweep; " + halo_kind
    st = dev.stats()
    local_updates_per_step = st["n_messages_per_sweep"]      # what one launch computes (redundant rows included)
    updates_per_step = local_updates_per_step
    if exchange is not None and getattr(part, "depth", 0) and world > 1:
        # the metric counts OWNED updates only: 4 directed messages per pairwise factor, a factor belongs to the rank of its
        # lower-id variable: R (C - 1) horizontal + R C vertical factors, the last rank one row fewer
        if args.scaling == "weak":
            updates_per_step = 4 * (N * (N - 1) + (N if rank < world - 1 else N - 1) * N)
        else:   # one N x N grid in total: 8 N (N - 1) directed updates, split evenly for the sum over ranks
            total = 8 * N * (N - 1)
            updates_per_step = total // world + (total % world if rank == 0 else 0)

    def step():
        if exchange is None:
            dev.sweep(1)
        else:
            exchange.sweep()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    dev.residual()   # snapshot: the residual reported below is the change over the timed region
    barrier()
    dev.profile_enable(max(1, args.event_stride))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    dev.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([updates_per_step], dtype=torch.float64, device=red_dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_updates_per_step = float(u.item())
    else:
        total_updates_per_step = float(updates_per_step)

    # dominant kernel: hipEvent durations recorded around every launch of the timed region, on the library's stream
    kern = {}
    for k in (L.KERNEL_FUSED, L.KERNEL_VAR_TO_FACTOR, L.KERNEL_FACTOR_TO_VAR, L.KERNEL_HALO_BEGIN, L.KERNEL_HALO_END):
        ms, n = dev.profile_read(k)
        if n:
            kern[dev.kernel_name(k)] = (ms, n, k)
    res = dev.residual()
    # audit of the last exchange of the run (outside the timed region): what each rank imported == what its neighbour packed
    halo_check = None
    if halo_tensors is not None:
        from cortex.jl_amd import partition
        dev.sync()
        torch.cuda.synchronize()
        halo_check = partition.verify_last_exchange(part, halo_tensors[0], halo_tensors[1], dist, torch)

    if rank == 0:
        value = total_updates_per_step * args.steps / elapsed
        dom = max(kern.items(), key=lambda kv: kv[1][0])
        dom_name, (dom_ms, dom_n, dom_id) = dom
        # algorithmic bytes per launch: §8d's 32 B per directed message update x the updates one launch performs
        if dom_id == L.KERNEL_FUSED:
            upd_per_launch = local_updates_per_step
        else:
            upd_per_launch = local_updates_per_step / 2
        avg_s = dom_ms / dom_n / 1e3
        achieved = upd_per_launch * BYTES_PER_UPDATE / avg_s / 1e9
        out = {
            "metric": "edge-message updates/sec per sweep, 10M-edge Gaussian grid",
            "value": value, "unit": "edge-message updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"C4: {N}x{N} 2-D Gaussian grid loopy BP per GPU" if args.scaling == "weak" else
                                    f"C4: ONE {N}x{N} 2-D Gaussian grid loopy BP cut into {world} row blocks; rank 0 holds") +
                                   f" ({st['n_edges']} bipartite edges, {updates_per_step} directed message updates + "
                                   f"{st['n_variables']} marginals per sweep)",
                       "schedule": args.schedule + ("" if halo_kind is None else f" + {halo_kind}"), "partition": f"{world} row strips",
                       "seed": args.seed},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_kernel_ms": dom_ms / dom_n,
                         "launches": dom_n,
                         "algorithmic_bytes_per_launch": upd_per_launch * BYTES_PER_UPDATE,
                         "all_kernels_ms": {k: v[0] / v[1] for k, v in kern.items()}},
            "hbm_roofline_frac_end_to_end": value * BYTES_PER_UPDATE / 1e9 / (HBM_PEAK_GBS * world),
            "marginals_per_s": (N * N if exchange is not None and getattr(part, "depth", 0) else st["n_variables"]) * world * args.steps / elapsed,   # computed inside the same kernel, not counted in `value`
            "max_message_change_over_run": res,
        }
        if halo_check is not None:
            out["halo_check"] = "ok: last imported halo == neighbours' packed messages, bit for bit" if halo_check else "FAILED"
        traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(traffic_file) and N == 1415 and args.schedule == "fused" and not args.materialize:   # measured for that workload only
            try:
                tr = json.load(open(traffic_file))
                if tr.get("kernel") == dom_name:
                    out["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_grid, args.seed)
        print(json.dumps(out))
    dog.cancel()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
