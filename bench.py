#!/usr/bin/env python3
"""bench.py — edge-message updates/sec per sweep on the 10M-edge Gaussian grid (BASELINE.json, config 4).

    python bench.py --gpus N --steps K --warmup W            # N > 1: this process only spawns and supervises N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full sum-product sweep (every variable→factor message, every factor→variable message, every
marginal), i.e. one `update_marginals!` of the reference (src/inference_engine.jl:559-632) in the device's
flooding order.  The workload is BASELINE config 4: ONE 1415 x 1415 Gaussian grid (10,005,465 bipartite edges,
16,006,480 directed message updates per sweep).  N > 1 cuts that one grid into N row blocks (strong scaling,
the "8-way cut" of the config) with a deep halo: the redundant rows' state travels once per `--halo-depth`
sweeps (RCCL over xGMI, issued by the library); a weak-scaling figure (one 1415 x 1415 strip per rank) is
measured afterwards and printed as the `weak_scaling` field.  value = owned message updates of all ranks /
max-over-ranks time.  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel: device time of the timed region on the
library's stream / launches; counter traffic from profiles/) and, at N = 1, `cpu_baseline` (the CPU restatement
of the reference scheduler on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# dmabuf IPC (the only form this pool's host driver supports) has to be chosen before anything initialises HIP: set here, at import, not
# next to the process group (torch.cuda.is_available() already is a HIP call)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
HBM_COPY_GBS = 6290.0          # what a 16 B-per-lane copy kernel reaches on this part (same guide): the achievable ceiling
BYTES_PER_UPDATE = 32          # read the 16-byte payload once + write it once, f64 scalar Gaussian


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of exactly --steps steps each (barrier + synchronize on both sides); the median is reported")
    ap.add_argument("--event-stride", type=int, default=1,
                    help="per-launch hipEvent pairs in the extra, untimed sampling region (every n-th launch)")
    ap.add_argument("--grid", type=int, default=1415, help="N: the N x N grid (1415 -> 10,005,465 edges)")
    ap.add_argument("--schedule", choices=["flooding", "fused"], default=os.environ.get("CX_BENCH_SCHEDULE", "fused"))
    ap.add_argument("--materialize", action="store_true", help="also store every variable→factor message each sweep")
    ap.add_argument("--halo", choices=["ipc", "rccl", "torch"], default=os.environ.get("CX_HALO", "ipc"),
                    help="N > 1: the exchange pushed by the library into the neighbours' IPC-mapped receive areas (default; audited, "
                         "RCCL as the fallback), issued by the library on RCCL, or by torch.distributed isend/irecv")
    ap.add_argument("--ipc-overlap", action="store_true",
                    help="--halo ipc: the owned part of a batch's first sweep between push and unpack (cx_halo_ipc_exchange_sweep); "
                         "with --halo-depth auto the forms are timed and the fastest is kept")
    ap.add_argument("--ipc-early-push", action="store_true",
                    help="--halo ipc: additionally the NEXT exchange is pushed inside the last sweep of every batch (cx_halo_ipc_batch): "
                         "two partial sweeps of compute between a push and the wait for it")
    ap.add_argument("--ipc-soak", type=int, default=24,
                    help="--halo ipc, N > 1: before anything is timed, this many batches are run with an audit after EVERY exchange (the "
                         "owners' values carried a second time over torch.distributed must equal the redundant rows bit for bit, ranks "
                         "skewed against each other); one mismatch and all ranks fall back to the RCCL exchange")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default, BASELINE config 4): the ONE N x N grid is cut into row blocks over the ranks; "
                         "weak: every rank owns an N x N strip of an (N*ranks) x N grid")
    ap.add_argument("--no-weak-figure", action="store_true", help="N > 1, strong scaling: skip the second (weak-scaling) measurement")
    ap.add_argument("--halo-depth", default=os.environ.get("CX_HALO_DEPTH", "auto"),
                    help="deep halo: each rank keeps this many redundant rows of its neighbours and exchanges their state once "
                         "per that many sweeps (bit-identical to the un-partitioned sweep); 0 = one message halo per sweep; "
                         "auto (default): N > 1 times a few batches at depths 12, 16, 24 and 32 before the warm-up and keeps the fastest "
                         "— what an exchange costs depends on the links between the GPUs, which this program cannot know beforehand "
                         "(on one GPU, the rank as its own neighbour: 16)")
    ap.add_argument("--cpu-configs", action="store_true", help="CPU baselines of configs C1, C2 and the C4 sample only (no GPU needed)")
    ap.add_argument("--self-halo", action="store_true",
                    help="N = 1 experiment: a cylinder whose wrap-around cut makes rank 0 its own halo neighbour")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the other BASELINE configs (C2, C3, C5, variational families)")
    ap.add_argument("--parity-sweeps", type=int, default=8, help="sweeps of the in-run parity check on the CPU sample grid")
    ap.add_argument("--cpu-sample-grid", type=int, default=768, help="grid side of the bounded CPU sample")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous of the spawned ranks only (gloo all-reduce, no GPU): what the CPU test of the launcher runs")
    return ap.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` by itself ---------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv, timeout_s: float = 1500.0) -> int:
    """Spawn one child per rank (this file, same arguments) with the torch.distributed environment set, relay rank 0's
    stdout, wait.  The parent never imports torch nor touches a GPU; nothing is exec'd from a process that has.  A child
    that fails takes the others down and the exit code is non-zero."""
    import signal
    import subprocess

    import tempfile

    port = _free_port()
    procs = []
    # rank 0's stdout goes to a temporary file, relayed after it exits: a pipe that nobody reads while the child runs would block
    # the child once it holds 64 KB (verbose JSON, warnings) — and look like a hung GPU to whoever waits for us
    out_file = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CX_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=out_file if r == 0 else subprocess.DEVNULL, start_new_session=True))
    deadline = time.time() + timeout_s
    rc, out0 = 0, b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write(f"[bench] rank {r} exited with {code}: stopping the other ranks\n")
            if rc != 0 or time.time() > deadline:
                if rc == 0:
                    rc = 3
                    sys.stderr.write(f"[bench] no completion after {timeout_s:.0f} s: stopping the ranks\n")
                break
            if pending:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)   # the child's own session: exactly the processes started here
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    out_file.seek(0)
    out0 = out_file.read()
    out_file.close()
    # stdout carries the ONE JSON line; whatever else a library printed there (gloo's "[Gloo] Rank 0 is connected ..." in the one-GPU
    # rehearsal) goes to stderr
    for line in out0.decode(errors="replace").splitlines(keepends=True):
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    return rc


def launch_check():
    """what a spawned rank does under --launch-check: join the rendezvous, all-reduce the ranks, report (CPU only)"""
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if os.environ.get("CX_BENCH_FAIL_RANK") == str(rank):
        os._exit(7)                          # the launcher test's "a child dies" case
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launcher": "ok", "world": world, "sum_of_ranks_plus_one": int(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


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


def parity_check(sample_n: int, seed: int, sweeps: int, device: int = 0) -> dict:
    """In-run parity (part of the cpu_baseline leg: the only place bench.py touches oracle/): the device and the C flooding checker
    (oracle/bp_flood.c, the reference's rules in the device's sweep order) run the same `sweeps` sweeps from the same seeded state
    of the CPU sample grid; largest relative difference of messages and marginals."""
    import cortex.jl_amd as cx
    from cortex.jl_amd import _lib as L
    from tests.helpers import flood_oracle_from_model

    model = cx.synth.gaussian_grid(sample_n, sample_n, seed=seed)
    dev = cx.DeviceGraph(device=device, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    g = flood_oracle_from_model(model, seed_variance=1e6)
    dev.sweep(sweeps)
    g.sweep(sweeps, use_omp=True)
    got = dev.get_messages(g.edge_var, g.edge_fac, L.TO_VARIABLE)

    def rel(a, b):
        ok = ~np.isnan(b)
        if not np.array_equal(np.isnan(a), ~ok):
            return float("inf")
        scale = max(float(np.median(np.abs(b[ok]))), 1e-300)
        return float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), scale)))

    e_msg = max(rel(got[:, 0], g.f2v_m), rel(got[:, 1], g.f2v_v))
    dev.sweep(1)                                   # the marginals a sweep writes are those of the messages it READ
    marg = dev.get_marginals(model.x_ids)
    m, v = g.marginals()
    e_marg = max(rel(marg[:, 0], m), rel(marg[:, 1], v))
    dev.close()
    return {"max_rel_err_marginals": e_marg, "max_rel_err_messages": e_msg, "tolerance": 1e-6, "ok": bool(e_marg <= 1e-6 and e_msg <= 1e-6),
            "sweeps": sweeps, "sample": f"{sample_n}x{sample_n} grid ({model.n_edges} edges), every factor→variable message and every marginal",
            "checker": "oracle/bp_flood.c: the reference's rules (test/inference_engine_tests.jl:385-432) in moment form, in the device's sweep order"}


def _rel_err(a, b):
    """largest |a - b| relative to the largest |b| (means cross zero: a purely relative test is ill-posed there); inf on a NaN mismatch"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if np.any(np.isnan(a)) or np.any(np.isnan(b)):
        return float("inf")
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def config_parity(device: int = 0) -> dict:
    """Parity of the non-headline configs, in the same run as their timings (part of the cpu_baseline leg: checker code).  Returns
    {"hooks": callbacks that bench_configs' recipes run on their TIMED devices at full size, "standalone": rows computed here on small
    devices}.  What each figure is measured against is named in its `checker`; the reference pins none of these numbers:
    trees are pinned by exact solves, flooding sweeps by the C restatement of the same sweep, the variational families by the array
    form of the reference's update_marginals! (oracle/vmp.py)."""
    import cortex.jl_amd as cx
    from cortex.jl_amd import _lib as L
    from oracle import exact

    def c2(dev, model):          # full size: all T marginals against the Thomas solve of the tridiagonal posterior
        em, ev = exact.ssm_chain_posterior(model.data_y, 1.0, 1.0)
        marg = dev.get_marginals(model.x_ids)
        return {"max_rel_err": max(_rel_err(marg[:, 0], em), _rel_err(marg[:, 1], ev)), "tolerance": 1e-6,
                "checker": "oracle/exact.py: Thomas solve of the chain's tridiagonal posterior", "sample": f"all {len(model.x_ids)} marginals (full size), one sweep"}

    def c3_scan(dev, model):     # full size: all T marginals against the exact block-tridiagonal smoother in C
        d, T = model.dim, len(model.x_ids)
        em, ecov = exact.lgssm_posterior_c(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
        marg = dev.get_marginals(model.x_ids)
        return {"max_rel_err": max(_rel_err(marg[:, :d], em), _rel_err(marg[:, d:].reshape(T, d, d), ecov)), "tolerance": 1e-6,
                "checker": "oracle/blocktri.c: exact smoother by pivoted block elimination", "sample": f"all {T} marginals (full size), one sweep, no seeding"}

    def c5_scan(dev, model):     # full size: eight windows of 40 marginals against the exact smoother of the window + 300 steps either side
        d, T, W, pad = model.dim, len(model.x_ids), 40, 300
        A, Q, R = model.meta["A"], model.meta["Q"], model.meta["R"]
        worst = 0.0
        starts = [0, T - W] + [int(x) for x in np.linspace(T // 7, T - T // 7, 6)]
        for start in starts:
            lo, hi = max(0, start - pad), min(T, start + W + pad)
            em, ecov = exact.lgssm_posterior_c(model.data_y[lo:hi], A, Q, R)
            marg = dev.get_marginals(model.x_ids[start:start + W])
            sl = slice(start - lo, start - lo + W)
            worst = max(worst, _rel_err(marg[:, :d], em[sl]), _rel_err(marg[:, d:].reshape(W, d, d), ecov[sl]))
        return {"max_rel_err": worst, "tolerance": 1e-6, "checker": "oracle/blocktri.c on windows (the model forgets a boundary within ~100 steps: 300 steps of "
                "padding make the window's posterior the chain's to rounding); every one of the 1e5 marginals against the whole-chain solve: "
                "tests/test_gpu_mv64_chain.py::test_config_c5_full_size_one_sweep_every_marginal", "sample": f"{len(starts)} windows of {W} marginals at full size, one sweep, no seeding"}

    rows = {}

    def flooding(d, T, sweeps, lag):      # per-sweep parity of the fused flooding kernels against the C restatement of the same sweep
        from oracle.mv import MvFloodC
        model = cx.synth.lgssm_chain(T, d=d, seed=17)
        dev = cx.DeviceGraph(device=device, dim=d, schedule=L.SCHED_FUSED)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        o = MvFloodC(model)
        o.sweep(lag)
        o.seed(0.0, 1e6)
        g = o.g
        xs = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
        pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if int(np.searchsorted(g.var_ids, g.edge_var[e])) in xs])
        worst = 0.0
        for _ in range(sweeps):
            dev.sweep(1)
            o.sweep(1, use_omp=True)
            got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
            worst = max(worst, _rel_err(got[:, :d], o.f2v_m[pe]), _rel_err(got[:, d:].reshape(-1, d, d), o.f2v_S[pe]))
        dev.close()
        return {"max_rel_err": worst, "tolerance": 1e-6, "checker": "oracle/mv_flood.c: the same flooding sweep in moment form",
                "sample": f"T={T}, every factor→variable message of the latent variables after each of {sweeps} sweeps"}

    def c5_fixed_point():                 # the d = 64 rule to its fixed point on a short chain against the exact smoother
        d, T = 64, 48
        model = cx.synth.lgssm_chain(T, d=d, seed=19)
        dev = cx.DeviceGraph(device=device, dim=d, schedule=L.SCHED_FUSED)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        dev.sweep(T + 2)
        em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
        marg = dev.get_marginals(model.x_ids)
        dev.close()
        return {"max_rel_err": max(_rel_err(marg[:, :d], em), _rel_err(marg[:, d:].reshape(T, d, d), ecov)), "tolerance": 1e-6,
                "checker": "oracle/exact.py: block-tridiagonal posterior", "sample": f"T={T} chain swept to its fixed point ({T + 2} sweeps), all marginals"}

    def vmp_family(name, fam):
        from oracle import vmp
        n = 64
        model = cx.synth.vmp_ssm(n, seed=12)
        dev = cx.DeviceGraph(device=device, family=fam, schedule=L.SCHED_CHAIN_SCAN)
        cx.synth.load_vmp_into_device(model, dev)
        arr = (vmp.StructuredVMP if name == "structured" else vmp.MeanFieldVMP)(model.data_y)
        worst = 0.0
        for _ in range(6):
            for which, ids in ((["x"], L.VMP_ALL_NORMAL), (["ssnoise", "obsnoise"], L.VMP_ALL_PRECISION)):
                dev.update_marginals(ids)
                arr.update(which)
                xs = dev.get_marginals(model.x_ids)
                gm = dev.get_marginals([model.ssnoise, model.obsnoise])
                got = np.concatenate([xs[:, 0], xs[:, 1], gm[0], gm[1]])
                want = np.concatenate([arr.xm, arr.xw, arr.ss, arr.obs])
                worst = max(worst, float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300))))
        dev.close()
        return {"max_rel_err": worst, "tolerance": 1e-6, "checker": "oracle/vmp.py: the array form of the reference's update_marginals! on its variational SSM "
                "(test/inference_engine_tests.jl:593-1147), pinned call by call against the restated engine",
                "sample": f"n={n} states, 6 iterations (states, then both precisions), every marginal after every call"}

    def reference_order(n=320, calls=3):
        """CX_SCHED_REFERENCE against the restated reference engine, call by call: the executions in order, every message of both
        directions, every marginal (the full-size check, two calls at 1415 x 1415: tests/test_gpu_reference_schedule.py)"""
        from oracle import ref
        from tests.helpers import engine_oracle_from_model
        model = cx.synth.gaussian_grid(n, n, seed=77)
        E = engine_oracle_from_model(model, trace=True)
        dev = cx.DeviceGraph(device=device, schedule=L.SCHED_REFERENCE)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        pw = model.factor_kind[np.searchsorted(model.factor_ids, model.edge_fac)] == 1
        E.set_messages_to_variable(model.edge_var[pw], model.edge_fac[pw], np.zeros(pw.sum()), np.full(pw.sum(), 1e6))
        kinds = {ref.VAR_MSG_TO_FACTOR: L.ITEM_MESSAGE_TO_FACTOR, ref.VAR_MSG_TO_VARIABLE: L.ITEM_MESSAGE_TO_VARIABLE, ref.VAR_MARGINAL: L.ITEM_INDIVIDUAL_MARGINAL}
        worst, same_order = 0.0, True
        for call in range(calls):
            if call:
                E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
                dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, np.stack([model.prior_mean, model.prior_variance], axis=1))
            dev.sweep(1)
            E.update_marginals(model.x_ids)
            if call == calls - 1:      # the engine's trace, signal by signal (a Python loop over ~1e6 executions: once)
                want = []
                for _r, _v, s, _b, _a in E.trace():
                    k, v, f, _lo, _hi = E.variant(s)
                    want.append((kinds[k], v, f if k != ref.VAR_MARGINAL else 0, 0, 0))
                same_order = dev.ref_trace() == want
            for to_variable, direction in ((True, L.TO_VARIABLE), (False, L.TO_FACTOR)):
                tags, a, b = E.get_messages(model.edge_var, model.edge_fac, to_variable)
                got = dev.get_messages(model.edge_var, model.edge_fac, direction)
                und = tags == ref.UNDEF
                if not np.array_equal(np.isnan(got[:, 1]), und):
                    worst = float("inf")
                else:
                    worst = max(worst, _rel_err(got[~und, 0], a[~und]), _rel_err(got[~und, 1], b[~und]))
            _t, em, ev = E.get_marginals(model.x_ids)
            marg = dev.get_marginals(model.x_ids)
            worst = max(worst, _rel_err(marg[:, 0], em), _rel_err(marg[:, 1], ev))
        dev.close()
        return {"max_rel_err": worst if same_order else float("inf"), "tolerance": 1e-9, "execution_order_identical": bool(same_order),
                "checker": "oracle/cortex_ref.c: the reference's InferenceEngine restated (Signals, readiness nibbles, default resolver, update_marginals!) with the "
                           "reference's own moment-form rules (test/inference_engine_tests.jl:385-432)",
                "sample": f"{n}x{n} grid, {calls} consecutive calls with the priors re-set in between: the execution trace of the last call, every message of both directions and every marginal after every call"}

    def tree_exact(dev, model):           # the tree schedule's one sweep against a sparse direct solve of the joint Gaussian, on a sample
        ids = model.x_ids[:: max(len(model.x_ids) // 300, 1)]
        _i, em, ev = exact.kary_posterior_sparse(model, ids)
        marg = dev.get_marginals(ids)
        return {"max_rel_err": max(_rel_err(marg[:, 0], em), _rel_err(marg[:, 1], ev)), "tolerance": 1e-9,
                "checker": "oracle/exact.py: sparse LU of the joint precision (every mean; variances by unit-vector solves)",
                "sample": f"{len(ids)} of {len(model.x_ids)} marginals, one sweep"}

    def vmp_wired_check(xs, gm, model, iterations):
        """the wired model's marginals after `iterations` by-class iterations against oracle/vmp.py run the same calls at the same size"""
        from oracle import vmp
        arr = vmp.StructuredVMP(model.data_y)
        for _ in range(iterations):
            for which in (["x"], ["ssnoise"], ["obsnoise"]):
                arr.update(which)
        # the wired handle reports states in moment form (mean, variance); the array form holds (mean, precision)
        got = np.concatenate([xs[:, 0], 1.0 / xs[:, 1], gm[0], gm[1]])
        want = np.concatenate([arr.xm, arr.xw, arr.ss, arr.obs])
        err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-3)))
        return {"max_rel_err": err, "tolerance": 1e-9, "checker": "oracle/vmp.py: the array form of the reference's update_marginals! on its structured variational SSM "
                "(test/inference_engine_tests.jl:807-1147), pinned call by call against the restated engine",
                "sample": f"n={len(xs)} states, {iterations} iterations by class, every state mean and precision and both Gamma marginals"}

    def tiles_fixed_point(d=16, T=48):      # a native-tile dim swept to its fixed point on a short chain against the exact smoother
        model = cx.synth.lgssm_chain(T, d=d, seed=23)
        dev = cx.DeviceGraph(device=device, dim=d, schedule=L.SCHED_FUSED)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        dev.sweep(T + 2)
        em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
        marg = dev.get_marginals(model.x_ids)
        dev.close()
        return {"max_rel_err": max(_rel_err(marg[:, :d], em), _rel_err(marg[:, d:].reshape(T, d, d), ecov)), "tolerance": 1e-6,
                "checker": "oracle/exact.py: block-tridiagonal posterior", "sample": f"d={d}, T={T} chain swept to its fixed point ({T + 2} sweeps), all marginals"}

    for key, fn in (("C3", lambda: flooding(4, 2000, 8, 0)), ("C5", c5_fixed_point), ("d=16 native tiles", tiles_fixed_point), ("C4-reference", reference_order),
                    ("VMP structured", lambda: vmp_family("structured", L.FAMILY_VMP_STRUCTURED)),
                    ("VMP mean_field", lambda: vmp_family("mean_field", L.FAMILY_VMP_MEAN_FIELD))):
        try:
            rows[key] = fn()
        except Exception as e:
            rows[key] = {"error": f"{type(e).__name__}: {e}"}
    return {"hooks": {"C2": c2, "C3-scan": c3_scan, "C5-scan": c5_scan, "tree": tree_exact, "VMP-wired": vmp_wired_check}, "standalone": rows}


def other_configs(parity=None) -> list:
    """The other configs of BASELINE.json on this GPU, a few seconds each (tools/bench_configs.py holds the recipes): C2 (chain scan),
    C3 (d = 4: fused flooding sweep and the exact chain-scan sweep), C5 (d = 64, MFMA: flooding sweep and the exact chain-scan sweep),
    both variational families.  Each row carries ms per sweep, updates per second, its own roofline object (counter traffic from
    profiles/ when the kernel is unchanged) and — with `parity` = config_parity()'s result — a parity object {max_rel_err, tolerance,
    ok, checker, sample} measured in this run."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc

    hooks = (parity or {}).get("hooks", {})
    alone = (parity or {}).get("standalone", {})
    rows = []
    for name, fn in (("C2", lambda: bc.c2(check=hooks.get("C2"))), ("C3", lambda: bc.mv(4, 1_000_000, 30)),
                     ("C3-scan", lambda: bc.mv_scan(4, 1_000_000, 30, check=hooks.get("C3-scan"))[0]),
                     ("C5", lambda: bc.mv(64, 100_000, 12)),
                     # (round 6) d = 16 in its native tile size beside the embedding in 64 x 64 that every d in 5 .. 63 ran in until round 5
                     ("d=16 native tiles", lambda: bc.mv_tiles(16, 100_000, 20, embedded_T=20_000)),
                     ("C5-scan", lambda: bc.mv64_scan(100_000, 8, check=hooks.get("C5-scan"))[0]),
                     ("VMP", lambda: bc.vmp()),
                     # the same structured model as a user wiring under the reference-order schedule (cx_graph_wire): replayed plans per call
                     ("VMP-wired", lambda: bc.vmp_wired(100_000, check=hooks.get("VMP-wired"))),
                     # the headline graph under the reference's OWN order: one cx_sweep = one update_marginals! (sequential, newest values):
                     # stage count, ms per call, calls to the fixed point beside the fused schedule's sweeps
                     ("C4-reference", lambda: bc.reference_order(1415)),
                     # not a BASELINE config: the tree schedule (one sweep = the reference's one-call result on any forest), on a tree small
                     # enough to generate in a second
                     ("tree", lambda: bc.tree(n_factors=30_000, steps=20, check=hooks.get("tree"))),
                     # the same schedule where the level-by-level form would take two launches per level: long paths with side branches and
                     # factors of 2..6 variables (scalar; heavy paths through factors of any arity), and a d = 4 chain with a latent layer
                     ("tree-deep", lambda: bc.tree(n_factors=20_000, steps=20, shape="deep", check=hooks.get("tree"))),
                     ("tree-mv", lambda: bc.tree_mv(d=4, n_spine=50_000))):
        try:
            r = fn()
            rows.extend(r if isinstance(r, list) else [r])
        except Exception as e:      # one config failing must not take the headline line with it
            rows.append({"config": name, "error": f"{type(e).__name__}: {e}"})
    for r in rows:
        key = r.get("config")
        if key == "VMP":
            key = "VMP structured" if "structured" in r.get("workload", "") else "VMP mean_field"
        if "parity" not in r and key in alone:
            r["parity"] = alone[key]
        if "parity" in r and "max_rel_err" in r["parity"]:
            r["parity"]["ok"] = bool(r["parity"]["max_rel_err"] <= r["parity"]["tolerance"])
    return rows


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

HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "details")
HEADLINE_MAX_BYTES = 4096


def _strict(x):
    """JSON has no NaN / Infinity tokens: non-finite floats become null (a failed check keeps its `ok: false`), numpy scalars become
    Python's, long floats are cut to 6 significant digits (the side file keeps them whole)"""
    if isinstance(x, dict):
        return {str(k): _strict(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict(v) for v in x]
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        return float(f"{x:.6g}") if np.isfinite(x) else None
    return x


def headline_line(full: dict, details_path=None) -> str:
    """The ONE line the driver reads: the contract's keys, `roofline`, `cpu_baseline`, `parity` and the path of the side file that
    holds everything else (other configs, every timed region, the long notes).  Strict JSON, under 4 KB, whatever `full` holds."""
    ro = full.get("roofline") or {}
    cb = full.get("cpu_baseline")
    pa = full.get("parity")
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in HEADLINE_KEYS[:12]}
    line["config"] = {k: cfg[k] for k in ("workload", "schedule", "partition", "seed", "halo_depth") if k in cfg}
    line["roofline"] = {k: ro.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "basis", "avg_kernel_ms",
                                                 "payload_bytes_per_launch", "frac_payload", "survey_convention_bytes_per_launch",
                                                 "frac_survey_convention") if k in ro}
    if cb is not None:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
        fa = cb.get("flooding_all_cores")
        if fa:
            line["cpu_baseline"]["flooding_all_cores"] = {k: fa.get(k) for k in ("value", "cores")}
    if pa is not None:
        line["parity"] = {k: pa.get(k) for k in ("ok", "max_rel_err_marginals", "max_rel_err_messages", "tolerance", "sweeps", "sample", "checker") if k in pa}
    oc = full.get("other_configs")
    if oc:
        # a digest only: config name, its time, its fraction, whether its check passed — the rows themselves are in the side file
        line["other_configs_digest"] = [[r.get("config"), r.get("ms_per_sweep", r.get("ms_per_iteration", r.get("ms_per_call"))),
                                         (r.get("roofline") or {}).get("frac"), (r.get("parity") or r.get("self_check") or {}).get("ok")]
                                        if "error" not in r else [r.get("config"), "error"] for r in oc]
    for k in ("weak_scaling", "halo_check"):
        if k in full:
            w = full[k]
            line[k] = {kk: w.get(kk) for kk in ("value", "unit", "ms_per_step")} if isinstance(w, dict) else w
    line["details"] = details_path
    line = _strict(line)

    def dump():
        return json.dumps(line, allow_nan=False, separators=(",", ":"))

    def clip(d, key, n):
        if isinstance(d.get(key), str) and len(d[key]) > n:
            d[key] = d[key][: n - 1] + "…"

    text = dump()
    if len(text.encode()) >= HEADLINE_MAX_BYTES:          # long free-text fields are the only thing that can grow: cut them, in this order
        for d, key, n in ((line["config"], "schedule", 200), (line.get("parity") or {}, "checker", 120), (line.get("cpu_baseline") or {}, "sample", 160),
                          (line["config"], "workload", 240), (line["roofline"], "basis", 120), (line.get("parity") or {}, "sample", 100)):
            clip(d, key, n)
            text = dump()
            if len(text.encode()) < HEADLINE_MAX_BYTES:
                break
    if len(text.encode()) >= HEADLINE_MAX_BYTES:
        line.pop("other_configs_digest", None)
        text = dump()
    assert len(text.encode()) < HEADLINE_MAX_BYTES, len(text.encode())
    return text


def write_details(full: dict, world: int) -> str:
    """everything measured in this run, whole: gpurun_out/bench_details_n<N>.json (relative path returned), and on stderr"""
    rel = os.path.join("gpurun_out", f"bench_details_n{world}.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(_strict_keep(full), f, indent=1)
    except OSError as e:                                  # a read-only checkout must not cost the headline
        print(f"[bench] could not write {rel}: {e}", file=sys.stderr)
        rel = None
    print("[bench] details: " + json.dumps(_strict_keep(full)), file=sys.stderr, flush=True)
    return rel


def _strict_keep(x):
    """as _strict, floats kept whole"""
    if isinstance(x, dict):
        return {str(k): _strict_keep(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict_keep(v) for v in x]
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        return float(x) if np.isfinite(x) else None
    return x


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


class Workload:
    """one DeviceGraph loaded with this rank's part of the grid + the object that runs n sweeps of it"""

    def __init__(self, args, scaling, rank, world, local_rank, backend, dist, torch, cx, L):
        from cortex.jl_amd import partition

        N = args.grid
        self.args, self.scaling, self.world, self.rank = args, scaling, world, rank
        schedule = L.SCHED_FUSED if args.schedule == "fused" else L.SCHED_FLOODING
        self.dev = dev = cx.DeviceGraph(device=local_rank, schedule=schedule, marginals_in_sweep=True,
                                        materialize_messages_to_factor=args.materialize)
        dev.set_stream(torch.cuda.current_stream().cuda_stream)
        self.halo_kind, self.halo_tensors, self.part, self.exchange, self.ipc = None, None, None, None, None
        red_dev = "cuda" if backend == "nccl" else "cpu"
        if world == 1 and not args.self_halo:
            model = cx.synth.gaussian_grid(N, N, seed=args.seed)
            cx.synth.load_into_device(model, dev, seed_variance=1e6)
        else:
            rows_min = N if scaling == "weak" else N // world
            depth = self.depth = max(0, min(args.halo_depth, rows_min))
            if world == 1:
                part = partition.deep_self(N, N, depth, seed=args.seed) if depth else partition.cylinder_self(N, N, seed=args.seed)[0]
            elif scaling == "strong":
                part = (partition.grid_rows_deep(N, N, rank, world, depth, seed=args.seed) if depth
                        else partition.grid_rows(N, N, rank, world, seed=args.seed))
            elif depth:
                part = partition.grid_strip_deep(N, N, rank, world, depth, seed=args.seed)
            else:
                part = partition.grid_strip(N, N, rank, world, seed=args.seed)
            self.part = part
            cx.synth.load_into_device(part.model, dev, seed_variance=1e6)
            exchange = None
            tdev = torch.device("cuda", local_rank)
            err = None
            if args.halo == "ipc" and depth:
                # the library pushes into the neighbours' IPC-mapped receive areas (one launch per exchange, no collective
                # library on the data path).  Audited on this very topology before it is used: the first exchange must put
                # the owners' values into the redundant rows bit for bit (second copy over torch.distributed); anything
                # else — allocation, handle import, a neighbour that does not arrive, a mismatch — and ALL ranks fall back
                # to the RCCL exchange below
                def agreed(e):              # every rank takes the same path: one rank's failure is everybody's
                    if dist is None:
                        return e
                    okf = torch.tensor([0 if e else 1], dtype=torch.int32, device=red_dev)
                    dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                    return e if e or okf.item() == 1 else "another rank failed"
                ex = None
                try:
                    # one launch per exchange only when every rank has a GPU of its own (cx_halo_ipc_set_fused)
                    on_own_gpus = world > 1 and os.environ.get("CX_SINGLE_DEVICE") != "1"
                    ex = partition.DeepHaloIpc(dev, part, dist, torch, tdev, overlap=bool(getattr(args, "ipc_overlap", False)),
                                               early_push=bool(getattr(args, "ipc_early_push", False)), peers_on_other_devices=on_own_gpus)
                    dev.halo_ipc_set_timeout(30.0)
                    if os.environ.get("CX_BENCH_IPC_FAIL") == str(rank):      # rehearsal knob: one rank fails, ALL must fall back
                        raise RuntimeError("injected failure (CX_BENCH_IPC_FAIL)")
                except (cx.CortexHipError, RuntimeError, ValueError) as e:
                    err = e
                err = agreed(err)
                if err is None:             # the audit is collective: entered by all ranks or by none
                    try:
                        if not ex.audit(dist, torch, tdev if backend == "nccl" or dist is None else torch.device("cpu")):
                            err = "audit of the first exchange failed"
                    except (cx.CortexHipError, RuntimeError, ValueError) as e:
                        err = e
                    err = agreed(err)
                soak = int(getattr(args, "ipc_soak", 0)) if world > 1 else 0
                if err is None and soak > 0 and not getattr(args, "_ipc_soaked", False):
                    # The protocol orders a flag behind its data by completion, not by fences (cx_api_ipc.hip) — an argument that only
                    # means something between GPUs.  So it is shown to hold HERE before it is used: `soak` batches, every rank idling
                    # at batches of its own so that pushes arrive early and late, and after EVERY exchange the redundant rows are
                    # compared with the owners' values carried a second time by torch.distributed.  (Once per run: the depth trials
                    # and the timed workload share the verdict.)
                    adev = tdev if backend == "nccl" or dist is None else torch.device("cpu")
                    try:
                        for b in range(soak):
                            if b % world == rank:
                                dev.sync()
                                time.sleep(0.002)
                            ex.sweep(depth)
                            if not ex.audit(dist, torch, adev):
                                err = f"soak: exchange {b + 2} left a redundant row that differs from its owner's"
                                break
                    except (cx.CortexHipError, RuntimeError, ValueError) as e:
                        err = e
                    err = agreed(err)
                    if err is None:
                        args._ipc_soaked = True
                if err is None:
                    exchange = self.ipc = ex
                    self.halo_kind = ("pushed into the neighbours' IPC-mapped receive areas behind an epoch flag" +
                                      (", the next exchange pushed inside the last sweep of a batch, unpacked after the owned part of the first" if getattr(args, "ipc_early_push", False)
                                       else ", the owned part of a batch's first sweep between push and unpack" if getattr(args, "ipc_overlap", False)
                                       else "") + (f" (audited after each of {soak} exchanges before the timing, and at start and end)" if soak else " (audited at start)"))
                elif rank == 0:
                    print(f"[bench] IPC exchange unavailable ({err}); falling back", file=sys.stderr)
                err = None
            if exchange is not None:
                pass
            elif backend != "nccl" and world > 1:
                if depth:
                    sweeper = partition.HostStagedStateSweeper(dev, part, torch, tdev)
                    exchange = partition.DeepHaloExchange(sweeper, part, dist)
                else:
                    sweeper = partition.HostStagedSweeper(dev, part, torch, tdev)
                    exchange = partition.HaloExchange(sweeper, part, dist)
                self.halo_kind = f"REHEARSAL: host-staged over {backend}"
                self.halo_tensors = (sweeper.send, sweeper.recv)
            elif args.halo in ("ipc", "rccl"):
                try:
                    exchange = (partition.DeepHaloRccl if depth else partition.RcclExchange)(dev, part, dist, torch, tdev)
                    self.halo_kind = "rccl send/recv issued by the library"
                    self.halo_tensors = (exchange.send, exchange.recv)
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
                self.halo_kind = "torch.distributed isend/irecv"
                self.halo_tensors = (sweeper.send, sweeper.recv)
            if depth:
                self.halo_kind = f"deep halo, {depth} redundant rows per side, state exchanged every {depth} sweeps; " + self.halo_kind
            else:
                self.halo_kind = "message halo per sweep; " + self.halo_kind
            self.exchange = exchange
        self.st = st = dev.stats()
        self.local_updates_per_step = st["n_messages_per_sweep"]      # what one launch computes (redundant rows included)
        self.updates_per_step = self.local_updates_per_step
        self.owned_variables = st["n_variables"]
        if self.exchange is not None and getattr(self.part, "depth", 0):
            # the metric counts OWNED updates only: 4 directed messages per pairwise factor; a factor belongs to the rank of
            # its lower-id variable: R (C - 1) horizontal + R C vertical factors, the last block one row fewer
            if world == 1:
                rows, last = N, True
            elif scaling == "weak":
                rows, last = N, rank == world - 1
            else:
                b = partition._row_bounds(N, world)
                rows, last = int(b[rank + 1] - b[rank]), rank == world - 1
            self.updates_per_step = 4 * (rows * (N - 1) + (rows - 1 if last else rows) * N)
            self.owned_variables = rows * N

    def run(self, n: int):
        if self.exchange is None:
            self.dev.sweep(n)
        else:
            self.exchange.sweep(n)

    def close(self):
        self.exchange = None
        self.dev.close()


def timed_regions(w: Workload, steps: int, repeats: int, dist, torch, red_dev):
    """`repeats` regions of exactly `steps` steps, each between barrier + synchronize; per region the wall time (max over
    ranks) and the device time between two events on the library's stream.  Returns the lists."""
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    walls, devs = [], []
    for _ in range(max(1, repeats)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        w.run(steps)
        e1.record()
        barrier()
        elapsed = time.perf_counter() - t0
        dev_ms = e0.elapsed_time(e1)
        if dist is not None:
            t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, dev_ms = float(t[0].item()), float(t[1].item())
        walls.append(elapsed)
        devs.append(dev_ms)
    return walls, devs


def run_rank(args):
    import torch
    import cortex.jl_amd as cx
    from cortex.jl_amd import _lib as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    dog = _watchdog(1200.0)
    N = args.grid
    depth_trials = None
    if str(args.halo_depth) == "auto":
        args.halo_depth = 16
        if world > 1:
            # the depth of the halo trades redundant rows (more per sweep) against exchanges (fewer): which wins depends on what an
            # exchange costs between THESE GPUs.  A few batches at each candidate, every rank timing the same region; all ranks see
            # the same maxima and take the same decision.  Not part of the warm-up or of the timed regions.
            rows_min = N if args.scaling == "weak" else N // world
            depth_trials = {}
            def trial(label):
                """one candidate configuration (args as set by the caller): ms per sweep, or None — on EVERY rank — when any rank
                failed; the workload is closed whatever happens (its receive area is mapped by the neighbours)"""
                wt, ms, bad = None, None, 0
                try:
                    wt = Workload(args, args.scaling, rank, world, local_rank, backend, dist, torch, cx, L)
                    wt.run(2 * args.halo_depth)
                    tw, _ = timed_regions(wt, 384, 3, dist, torch, red_dev)       # 384 = 4 x 96 sweeps: whole batches at every candidate
                    ms = min(tw) / 384 * 1e3
                    if wt.ipc is not None:
                        wt.ipc.check()
                except Exception as e:      # a candidate that cannot be built or run is no candidate
                    bad = 1
                    if rank == 0:
                        print(f"[bench] halo trial {label}: {e!r}", file=sys.stderr)
                finally:
                    if wt is not None:
                        try:
                            wt.close()
                        except Exception:
                            pass
                if dist is not None:
                    f = torch.tensor([bad], dtype=torch.int32, device=red_dev)
                    dist.all_reduce(f, op=dist.ReduceOp.MAX)
                    bad = int(f.item())
                return (None if bad else ms), (wt is not None and wt.ipc is not None)

            for cand in (12, 16, 24, 32):
                if cand > rows_min:
                    continue
                args.halo_depth = cand
                ms, _ = trial(f"depth {cand}")
                if ms is not None:
                    depth_trials[cand] = ms
            args.halo_depth = min(depth_trials, key=depth_trials.get) if depth_trials else 16
            # ... and, at that depth, the two forms that put compute between a push and the wait for it: the exchange AROUND the owned
            # part of the first sweep (cx_halo_ipc_exchange_sweep) and, on top of that, the next exchange pushed inside the last sweep of
            # a batch (cx_halo_ipc_batch).  The same results; whether hiding the transfer pays for the extra launches is a property of
            # the links (on one GPU, where nothing travels, it does not)
            if depth_trials and args.halo == "ipc":
                best = depth_trials[args.halo_depth]
                choice = (False, False)
                for label, ov, ep in (("exchange around the owned part of sweep 1", True, False),
                                      ("next exchange pushed inside the last sweep of a batch", False, True)):
                    args.ipc_overlap, args.ipc_early_push = ov, ep
                    ms, was_ipc = trial(label)
                    if ms is not None and was_ipc:
                        depth_trials["%d, %s" % (args.halo_depth, label)] = ms
                        if ms < best:
                            best, choice = ms, (ov, ep)
                args.ipc_overlap, args.ipc_early_push = choice
            if rank == 0:
                print(f"[bench] halo trials (ms per sweep): {depth_trials} -> depth {args.halo_depth}" +
                      (", next exchange pushed inside the last sweep of a batch" if getattr(args, "ipc_early_push", False) else
                       ", exchange around the owned part of sweep 1" if getattr(args, "ipc_overlap", False) else ""), file=sys.stderr)
    else:
        args.halo_depth = int(args.halo_depth)
    w = Workload(args, args.scaling, rank, world, local_rank, backend, dist, torch, cx, L)
    dev = w.dev

    w.run(args.warmup)
    dev.residual()   # snapshot: the residual reported below is the change over the timed regions
    walls, devs = timed_regions(w, args.steps, args.repeats, dist, torch, red_dev)
    order = sorted(range(len(walls)), key=lambda i: walls[i])
    mid = order[len(order) // 2]
    elapsed, dev_ms = walls[mid], devs[mid]
    res = dev.residual()

    # one more, UNTIMED region with a hipEvent pair around every launch (each pair is a barrier packet on the queue, which
    # is why it stays out of the timed regions): per-launch durations of every kernel
    dev.profile_enable(max(1, args.event_stride))
    w.run(min(args.steps, 400))
    dev.sync()
    dev.profile_enable(False)
    kern = {}
    for k in (L.KERNEL_FUSED, L.KERNEL_VAR_TO_FACTOR, L.KERNEL_FACTOR_TO_VAR, L.KERNEL_HALO_BEGIN, L.KERNEL_HALO_END):
        ms, n = dev.profile_read(k)
        if n:
            kern[dev.kernel_name(k)] = (ms, n, k)

    if dist is not None:
        u = torch.tensor([w.updates_per_step, w.owned_variables], dtype=torch.float64, device=red_dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_updates_per_step, total_variables = float(u[0].item()), float(u[1].item())
    else:
        total_updates_per_step, total_variables = float(w.updates_per_step), float(w.owned_variables)

    # audit of the last exchange of the run (outside the timed region): what each rank imported == what its neighbour packed
    halo_check = None
    if w.halo_tensors is not None:
        from cortex.jl_amd import partition
        dev.sync()
        torch.cuda.synchronize()
        halo_check = partition.verify_last_exchange(w.part, w.halo_tensors[0], w.halo_tensors[1], dist, torch)
    elif w.ipc is not None:
        # one more exchange, audited (the run ends between batches only when its sweeps are a multiple of the depth: top up)
        w.ipc.check()
        if w.ipc.k % w.ipc.depth:
            w.run(w.ipc.depth - w.ipc.k % w.ipc.depth)
        halo_check = w.ipc.audit(dist, torch, torch.device("cuda", local_rank) if backend == "nccl" or dist is None else torch.device("cpu"))

    # second figure at N > 1: weak scaling (every rank one N x N strip of an (N * ranks) x N grid)
    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_weak_figure:
        st_strong, halo_kind_strong = dict(w.st), w.halo_kind
        w2 = Workload(args, "weak", rank, world, local_rank, backend, dist, torch, cx, L)
        w2.run(args.warmup)
        wl, _ = timed_regions(w2, args.steps, min(args.repeats, 3), dist, torch, red_dev)
        wl.sort()
        u = torch.tensor([w2.updates_per_step], dtype=torch.float64, device=red_dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        t_w = wl[len(wl) // 2]
        weak = {"value": float(u.item()) * args.steps / t_w, "unit": "edge-message updates/s", "ms_per_step": t_w / args.steps * 1e3,
                "workload": f"every rank one {N}x{N} strip of a {N * world}x{N} grid ({w2.st['n_edges']} bipartite edges per rank)",
                "schedule": args.schedule + f" + {w2.halo_kind}"}
        w2.close()

    if rank == 0:
        st = w.st
        value = total_updates_per_step * args.steps / elapsed
        if kern:
            dom_name, (dom_ms, dom_n, dom_id) = max(kern.items(), key=lambda kv: kv[1][0])
        else:
            dom_name, (dom_ms, dom_n, dom_id) = "k_sweep<fused>", (0.0, 0, L.KERNEL_FUSED)
        # launches per step: the flooding schedule needs two launches per step
        steps_per_launch = 1.0 if args.schedule == "fused" else 0.5
        # algorithmic bytes per launch: §8d's 32 B per directed message update x the updates one launch performs
        upd_per_launch = w.local_updates_per_step * steps_per_launch
        alg_bytes = upd_per_launch * BYTES_PER_UPDATE
        # average launch duration: device time of the median timed region (events on the library's stream around the whole
        # region, launches back to back) / launches.  It contains the inter-launch gaps, so it can only over-state the
        # kernel; the per-launch event pairs of the sampling region are printed beside it.
        region_ms_per_launch = dev_ms / args.steps * steps_per_launch if args.schedule == "fused" else None
        sampled_ms = dom_ms / dom_n if dom_n else None
        avg_ms = region_ms_per_launch if region_ms_per_launch is not None else sampled_ms
        # counter traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections: profiles/): measured on
        # the full 1415 x 1415 launch, carried to other launch sizes per message update
        traffic, traffic_src = None, None
        traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(traffic_file) and args.schedule == "fused" and not args.materialize:
            try:
                tr = json.load(open(traffic_file))
                from cortex.jl_amd.build import sources_sha16
                if tr.get("kernel") == dom_name and tr.get("sources_sha16") == sources_sha16("k_sweep"):
                    per_update = tr["hbm_bytes_per_launch"] / tr.get("updates_per_launch", 16006480)
                    traffic = per_update * upd_per_launch
                    traffic_src = tr.get("source")
                elif tr.get("kernel") == dom_name:
                    traffic_src = "REFUSED: profiles/traffic_latest.json was measured on another version of the kernel's sources (tools/profile_round.sh)"
                    print("[bench] STALE PROFILE: profiles/traffic_latest.json was measured on another version of cx_kernels.hip — the headline's roofline falls back "
                          "to algorithmic bytes until tools/profile_round.sh is re-run and its summary copied into profiles/", file=sys.stderr, flush=True)
            except Exception:
                pass
        achieved_alg = alg_bytes / (avg_ms * 1e-3) / 1e9
        # payload: stored factor→variable messages read (one per edge), those written (one per directed pairwise update pair) and the
        # marginals, 16 B each; the flooding schedule also stores variable→factor messages
        payload_bytes = 16.0 * (st["n_edges"] + w.local_updates_per_step / 2 + st["n_variables"]) * max(steps_per_launch, 1.0)
        if args.schedule != "fused" or args.materialize:
            payload_bytes = 16.0 * (st["n_edges"] + w.local_updates_per_step + st["n_variables"]) * steps_per_launch
        achieved = (traffic / (avg_ms * 1e-3) / 1e9) if traffic else payload_bytes / (avg_ms * 1e-3) / 1e9
        strong = args.scaling == "strong"
        out = {
            "metric": "edge-message updates/sec per sweep, 10M-edge Gaussian grid",
            "value": value, "unit": "edge-message updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ((f"C4: ONE {N}x{N} 2-D Gaussian grid loopy BP" +
                                     (f" cut into {world} row blocks (strong scaling); rank 0 holds" if world > 1 else ";")) if strong else
                                    f"C4 weak scaling: one {N}x{N} strip per GPU of a {N * world}x{N} grid; rank 0 holds") +
                                   f" {st['n_edges']} bipartite edges, {w.updates_per_step} owned directed message updates + "
                                   f"{w.owned_variables} marginals per sweep; whole job {int(total_updates_per_step)} updates per sweep",
                       "schedule": args.schedule +
                                   ("" if w.halo_kind is None else f" + {w.halo_kind}"),
                       "partition": f"{world} row blocks", "seed": args.seed,
                       **({"halo_depth": args.halo_depth, "halo_depth_trials_ms_per_sweep": depth_trials} if depth_trials else {})},
            "timed_regions": {"count": len(walls), "reported": "median", "ms_per_step_each": [x / args.steps * 1e3 for x in walls],
                              "device_ms_per_step_each": [x / args.steps for x in devs]},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # what the number is: bytes that crossed the L2 <-> fabric boundary (FETCH_SIZE x2 + WRITE_SIZE), i.e. HBM traffic
                         # PLUS hits in the 256 MiB Infinity Cache (the index arrays stay resident there) — which is how it can exceed the
                         # 6.29 TB/s a pure HBM copy reaches; `frac` prices it against the 8 TB/s spec, `frac_of_measured_copy` against that copy
                         "bound_detail": "L2-fabric traffic incl. Infinity-Cache hits (memory system, not HBM alone)",
                         "frac_of_measured_copy": achieved / HBM_COPY_GBS, "measured_copy_peak": HBM_COPY_GBS,
                         "basis": ("counter bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/) / avg launch duration (events on the library's stream)"
                                   if traffic else "payload bytes / avg launch duration (no counter traffic on file for this kernel)"),
                         "traffic_source": traffic_src,
                         "avg_kernel_ms": avg_ms,
                         "avg_kernel_basis": "device time of the median timed region (events on the library's stream) / launches",
                         "sampled_kernel_ms": sampled_ms, "sampled_launches": dom_n,
                         # what the fused schedule MUST move per launch: every stored message read once (16 B), every message out of a
                         # pairwise factor and every marginal written once; index and parameter arrays are not payload
                         "payload_bytes_per_launch": payload_bytes, "frac_payload": payload_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # SURVEY §8d's convention (32 B per directed update: payload read + written, variable→factor messages included);
                         # the fused kernel keeps those in registers, so this convention counts bytes that never move and can exceed 1
                         "survey_convention_bytes_per_launch": alg_bytes, "frac_survey_convention": achieved_alg / HBM_PEAK_GBS,
                         "all_kernels_sampled_ms": {k: v[0] / v[1] for k, v in kern.items()}},
            # computed inside the same kernel, not counted in `value`
            "marginals_per_s": total_variables * args.steps / elapsed,
            "max_message_change_over_run": res,
        }
        if halo_check is not None:
            out["halo_check"] = "ok: last imported halo == neighbours' packed messages, bit for bit" if halo_check else "FAILED"
        if weak is not None:
            out["weak_scaling"] = weak
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_grid, args.seed)
            out["parity"] = parity_check(args.cpu_sample_grid, args.seed, args.parity_sweeps, local_rank)
        if world == 1 and not args.no_other_configs and not args.self_halo:
            w.close()                                  # free the headline grid before the other configs allocate theirs
            out["other_configs"] = other_configs(config_parity(local_rank) if not args.no_cpu_baseline else None)
        # ONE compact line on stdout, printed last; everything else to the side file and to stderr
        sys.stderr.flush()
        print(headline_line(out, write_details(out, world)), flush=True)
    dog.cancel()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.cpu_configs:
        cpu_config_table(args.seed)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    if args.launch_check:
        launch_check()
        return 0
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
