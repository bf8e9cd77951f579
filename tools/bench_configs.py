#!/usr/bin/env python3
"""tools/bench_configs.py — timings of the non-headline configs of BASELINE.json (C2 chain scan, C3 d=4, C5 d=64).
Not the driver's bench (bench.py measures C4); prints one JSON line per config for DESIGN.md / profiles/.
    python tools/bench_configs.py [c2] [c3] [c5] [vmp]"""
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


def timed(dev, fn, steps, warmup):
    for _ in range(warmup):
        fn()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    dev.sync()
    return (time.perf_counter() - t0) / steps


def c2():
    T = 250_001
    model = cx.synth.ssm_chain(T, seed=1234)
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev)
    dt = timed(dev, lambda: dev.sweep(1), 50, 5)
    st = dev.stats()
    # one reference update_marginals! on this chain is 5T-4 message computations + T marginals
    # the same chain under the fused flooding schedule: ONE parallel sweep (information moves one step; T sweeps converge)
    fl = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, fl, seed_variance=1e6)
    fl.sweep(3)
    dtf = timed(fl, lambda: fl.sweep(1), 200, 20)
    nf = fl.stats()["n_messages_per_sweep"]
    return {"config": "C2", "workload": f"scalar chain T={T} ({st['n_edges']} edges), chain-scan schedule: exact forward/backward in one sweep",
            "ms_per_sweep": dt * 1e3, "reference_updates_per_sweep": 5 * T - 4, "updates_per_s": (5 * T - 4) / dt,
            "algorithmic_GBps": (5 * T - 4) * 32 / dt / 1e9,
            "flooding": {"ms_per_sweep": dtf * 1e3, "updates_per_sweep": nf, "updates_per_s": nf / dtf, "algorithmic_GBps": nf * 32 / dtf / 1e9}}


def mv(d, T, steps):
    model = cx.synth.lgssm_chain(T, d=d, seed=1234)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(2)
    dev.profile_enable(1)
    dt = timed(dev, lambda: dev.sweep(1), steps, 3)
    ms, n = dev.profile_read(L.KERNEL_FUSED)
    st = dev.stats()
    upd = 2 * (2 * (T - 1))        # per sweep: both directions on every transition edge: v→f and f→v (messages with readers)
    payload = (d + d * d) * 8
    out = {"config": "C3" if d == 4 else "C5", "workload": f"d={d} linear-Gaussian chain T={T} ({st['n_edges']} edges), fused flooding sweep",
           "ms_per_sweep": dt * 1e3, "kernel_ms": ms / max(n, 1), "updates_per_sweep": upd, "updates_per_s": upd / dt,
           "algorithmic_GBps": upd * 2 * payload / dt / 1e9, "payload_bytes": payload}
    if d == 64:
        # MFMA work per factor→variable message: panel 24 + trailing 40 + Yt updates 96 + Yt*W' 64 + Yt Yt' 256 = 480
        # v_mfma_f64_16x16x4_f64, 2*16*16*4 flop each
        nmsg = 2 * (T - 1)
        out["mfma_TFLOPs"] = nmsg * 480 * 2048 / (ms / max(n, 1) / 1e3) / 1e12
        out["f64_matrix_peak_TFLOPs"] = 78.6
    return out


def vmp(n=1_000_000):
    """one variational iteration (all latent states, then both precisions) of the reference's SSM with
    unknown noise precisions, n states: 2n - 1 three-way factors, 6n - 3 edges."""
    model = cx.synth.vmp_ssm(n, seed=1234)
    out = []
    for name, fam in (("structured", L.FAMILY_VMP_STRUCTURED), ("mean_field", L.FAMILY_VMP_MEAN_FIELD)):
        dev = cx.DeviceGraph(family=fam, schedule=L.SCHED_CHAIN_SCAN)
        cx.synth.load_vmp_into_device(model, dev)

        def it():
            dev.update_marginals(L.VMP_ALL_NORMAL)
            dev.update_marginals(L.VMP_ALL_PRECISION)
        dt = timed(dev, it, 50, 10)
        g = dev.get_marginals([model.ssnoise, model.obsnoise])
        st = dev.stats()
        out.append({"config": "VMP", "workload": f"{name} VMP, SSM with unknown precisions, n={n} states ({st['n_edges']} edges, {st['n_factors']} factors)",
                    "ms_per_iteration": dt * 1e3, "messages_per_iteration": st["n_messages_per_sweep"],
                    "messages_per_s": st["n_messages_per_sweep"] / dt, "E_ssnoise": g[0, 0] * g[0, 1], "E_obsnoise": g[1, 0] * g[1, 1]})
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3", "c5"]
    torch.cuda.init()
    for w in which:
        if w == "vmp":
            for r in vmp():
                print(json.dumps(r), flush=True)
            continue
        r = c2() if w == "c2" else (mv(4, 1_000_000, 30) if w == "c3" else mv(64, 100_000, 20))
        print(json.dumps(r), flush=True)
