"""lab: the one-launch chain scan at C2 — time per sweep both ways, where a workgroup's time goes (CX_CHAIN_ONEPASS_STAMPS=1), and how far
the two forms' results are apart.  python tools/lab/c2_onepass.py [T]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 250_001
model = cx.synth.ssm_chain(T, seed=1234)


def make():
    d = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, d)
    d.sweep(3)
    d.sync()
    return d


def timed(d, n=200):
    d.sync()
    t0 = time.perf_counter()
    d.sweep(n)
    d.sync()
    return (time.perf_counter() - t0) / n * 1e6


a = make()
os.environ["CX_CHAIN_ONEPASS"] = "0"
b = make()
del os.environ["CX_CHAIN_ONEPASS"]
for rep in range(3):
    print(f"one launch {timed(a):.2f} us   two launches {timed(b):.2f} us   {a.chain_scan_stats()} {b.chain_scan_stats()}", flush=True)
ma, mb = a.get_marginals(model.x_ids), b.get_marginals(model.x_ids)
print("largest relative difference of a marginal:", float(np.max(np.abs(ma - mb) / np.abs(mb))), "identical:", bool(np.array_equal(ma, mb)))
if os.environ.get("CX_CHAIN_ONEPASS_STAMPS") == "1":
    a.sweep(1)
    a.sync()
