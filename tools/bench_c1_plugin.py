#!/usr/bin/env python3
"""tools/bench_c1_plugin.py — what the batched boundary costs on config C1 (scalar chain, T = 1,000; BASELINE.json configs[0]).

The reference's scheduler (host mirror of src/inference_engine.jl:559-632) drives the device one `process!` at a time
(per_signal: one cx_update_batch launch + one stream synchronisation per signal) or one wavefront of pending signals at a time
(wavefront).  Prints launches, signals and microseconds per process! / per launch, beside the CPU processor on the same scheduler
and the whole-graph device paths (chain scan: the exact answer in ONE sweep).  One JSON line per mode."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402
from cortex.jl_amd import get_value, get_variable_marginal, update_marginals  # noqa: E402
from tests.test_host_mirror import SSMBeliefPropagationProcessor, make_ssm  # noqa: E402  (the reference's test model builder)


def run(n, processor):
    rng = np.random.default_rng(1234)
    data = [2 * i + rng.standard_normal() for i in range(1, n + 1)]
    engine, x, y, lik, tr = make_ssm(n, processor)
    for i in range(n):
        sig = engine.get_connection_message_to_factor(y[i], lik[i])
        (processor.set_value if isinstance(processor, cx.HipProcessor) else cx.set_value)(sig, data[i])
    t0 = time.perf_counter()
    update_marginals(engine, x)
    if isinstance(processor, cx.HipProcessor):
        processor.dev.sync()                 # small batches return when their launch is queued: the clock stops when the device has
    dt = time.perf_counter() - t0
    m = [get_value(get_variable_marginal(engine.get_variable(v))).mean for v in x[:3]]
    return dt, m


def grid_wavefront(N=256):
    """a loopy N x N Gaussian grid (the C4 model at 1/30 of its size) through the wavefront mode: after seeding, one
    update_marginals! is three wavefronts — every variable→factor message, every factor→variable message, the marginals"""
    from cortex.jl_amd.signal import set_value as host_set

    model = cx.synth.gaussian_grid(N, N, seed=1234)
    proc = cx.HipProcessor(mode="wavefront")
    graph = cx.BipartiteFactorGraph()
    t0 = time.perf_counter()
    # ids in the order synth.gaussian_grid hands them out: variables first, then the factors in factor_ids order
    for _ in range(N * N):
        graph.add_variable(cx.Variable(name="x"))
    kinds, var = model.factor_kind, model.factor_var
    for k, q in zip(kinds, var):
        graph.add_factor(cx.Factor(functional_form=cx.GaussianAdditive(float(q)) if k == L.FACTOR_GAUSS_ADDITIVE else "prior"))
    for v, f in zip(model.edge_var, model.edge_fac):
        graph.add_edge(int(v), int(f), cx.Connection(label="out"))
    engine = cx.InferenceEngine(model_engine=graph, inference_request_processor=proc)
    build_s = time.perf_counter() - t0
    # priors and seeds: the host signals one by one (readiness bits), the device in two calls
    prior_of = dict(zip(model.prior_fac.tolist(), zip(model.prior_mean.tolist(), model.prior_variance.tolist())))
    proc.dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, np.stack([model.prior_mean, model.prior_variance], axis=1))
    proc.dev.seed_messages(L.TO_VARIABLE, 0.0, 1e6)
    for v, f in zip(model.edge_var.tolist(), model.edge_fac.tolist()):
        m, s2 = prior_of.get(f, (0.0, 1e6))
        host_set(engine.get_connection_message_to_variable(v, f), cx.NormalMeanVariance(m, s2))
    ids = [int(v) for v in model.x_ids]
    out = []
    for it in range(3):
        l0 = proc.launches
        t0 = time.perf_counter()
        update_marginals(engine, ids)
        dt = time.perf_counter() - t0
        out.append((dt, proc.launches - l0))
    st = proc.dev.stats()
    print(json.dumps({"config": "grid", "N": N, "path": "HipProcessor(mode='wavefront') behind the host scheduler, loopy grid",
                      "edges": st["n_edges"], "signals_per_update_marginals": st["n_messages_per_sweep"] + N * N, "engine_build_s": build_s,
                      "launches_per_update_marginals": [o[1] for o in out], "ms_per_update_marginals": [o[0] * 1e3 for o in out]}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "grid":
        return grid_wavefront(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    signals = 5 * n - 4 + n
    run(50, cx.HipProcessor(mode="wavefront"))          # warm the library
    for mode in ("per_signal", "wavefront"):
        proc = cx.HipProcessor(mode=mode)
        dt, _ = run(n, proc)
        print(json.dumps({"config": "C1", "T": n, "path": f"HipProcessor(mode={mode!r}) behind the host scheduler", "signals": signals,
                          "launches": proc.launches, "ms_per_update_marginals": dt * 1e3, "us_per_process": dt / signals * 1e6,
                          "us_per_launch": dt / proc.launches * 1e6}), flush=True)
    # the same launches WITHOUT the Python scheduler between them: the batches one update_marginals! issued, recorded (packed cx_item
    # records) and replayed through cx_update_batch_async in a bare loop — what the boundary itself costs a host whose scheduler is free
    for mode in ("per_signal", "wavefront"):
        proc = cx.HipProcessor(mode=mode)
        batches = []
        raw = proc.dev.update_batch_packed
        proc.dev.update_batch_packed = lambda b, k, raw=raw: (batches.append((b, k)), raw(b, k))[1]
        run(n, proc)
        proc.dev.sync()
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            for b, k in batches:
                raw(b, k)
            proc.dev.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        print(json.dumps({"config": "C1", "T": n, "path": f"replay of the {mode} schedule: the recorded batches through cx_update_batch_async, no scheduler in between",
                          "signals": signals, "launches": len(batches), "ms_per_update_marginals": best * 1e3, "us_per_process": best / signals * 1e6,
                          "us_per_launch": best / len(batches) * 1e6}), flush=True)
    # mode "reference": the whole call as ONE cx_sweep_for under CX_SCHED_REFERENCE — the reference's executions, in its order, replayed from
    # a standing plan; the first call finds the plan (the scheduler runs on the library's shadow of the readiness bits), the later ones
    # (new data, same request) replay it
    proc = cx.HipProcessor(mode="reference")
    rng = np.random.default_rng(1234)
    engine, x, y, lik, tr = make_ssm(n, proc)
    times, set_s = [], []
    for it in range(6):
        data = [2 * i + rng.standard_normal() for i in range(1, n + 1)]
        t0 = time.perf_counter()
        proc.dev.set_messages(y, lik, L.TO_FACTOR, L.FORM_POINT, data)      # the data of one call in ONE cx_set_messages (the plug-in's set_value is per signal)
        set_s.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        update_marginals(engine, x)
        proc.dev.sync()
        times.append(time.perf_counter() - t0)
    st = proc.dev.ref_plan_stats()
    print(json.dumps({"config": "C1", "T": n, "path": "HipProcessor(mode='reference'): update_marginals! = ONE cx_sweep_for under CX_SCHED_REFERENCE (the reference's "
                      "executions in its order, one graph launch)", "signals": signals, "plan": st, "ms_first_call_with_planning": times[0] * 1e3,
                      "ms_per_update_marginals": float(np.median(times[2:])) * 1e3, "ms_setting_the_data_one_call": float(np.median(set_s)) * 1e3,
                      "us_per_process": float(np.median(times[2:])) / signals * 1e6}), flush=True)
    dt, _ = run(n, SSMBeliefPropagationProcessor())
    print(json.dumps({"config": "C1", "T": n, "path": "CPU processor (reference arithmetic) on the same host scheduler (Python mirror)",
                      "signals": signals, "ms_per_update_marginals": dt * 1e3, "us_per_process": dt / signals * 1e6}), flush=True)
    model = cx.synth.ssm_chain(n, seed=1234)
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev)
    dev.sweep(3); dev.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        dev.sweep(1)
    dev.sync()
    dt = (time.perf_counter() - t0) / 200
    print(json.dumps({"config": "C1", "T": n, "path": "cx_sweep, chain-scan schedule (whole-call takeover: exact in one sweep)",
                      "signals": signals, "ms_per_update_marginals": dt * 1e3, "us_per_process": dt / signals * 1e6}), flush=True)


if __name__ == "__main__":
    main()
