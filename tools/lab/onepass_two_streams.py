"""lab: two handles on two streams of their own sweeping at the same time — two one-launch scans that each need their whole grid resident.
Does either ever wait out its bound (state -1, CX_ERR_DEVICE)?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 250_001
torch.cuda.init()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
devs = []
for k, s in enumerate(streams):
    m = cx.synth.ssm_chain(T, seed=7 + k, random_variances=True)
    d = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    d.set_stream(s.cuda_stream)
    cx.synth.load_into_device(m, d)
    d.sweep(2)
    d.sync()
    devs.append((m, d, d.get_marginals(m.x_ids)))
t0 = time.perf_counter()
err = None
try:
    for rnd in range(200):
        for m, d, ref in devs:
            d.sweep(50)      # asynchronous: both streams hold work at the same time
    for m, d, ref in devs:
        d.sync()
except cx.CortexHipError as e:
    err = e
dt = time.perf_counter() - t0
for m, d, ref in devs:
    print(d.chain_scan_stats(), "marginals as before:", bool(np.array_equal(d.get_marginals(m.x_ids), ref)) if err is None else "-", flush=True)
print(f"20000 sweeps on two streams in {dt:.2f} s; error: {err}")
