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
    dt = time.perf_counter() - t0
    m = [get_value(get_variable_marginal(engine.get_variable(v))).mean for v in x[:3]]
    return dt, m


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    signals = 5 * n - 4 + n
    run(50, cx.HipProcessor(mode="wavefront"))          # warm the library
    for mode in ("per_signal", "wavefront"):
        proc = cx.HipProcessor(mode=mode)
        dt, _ = run(n, proc)
        print(json.dumps({"config": "C1", "T": n, "path": f"HipProcessor(mode={mode!r}) behind the host scheduler", "signals": signals,
                          "launches": proc.launches, "ms_per_update_marginals": dt * 1e3, "us_per_process": dt / signals * 1e6,
                          "us_per_launch": dt / proc.launches * 1e6}), flush=True)
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
