"""lab: soak of the one-launch chain scan — many sweeps on chains of several sizes (one workgroup to the widest resident grid, ragged last
tiles), two handles alternating; the handles must stay on the one-launch form (state 1: no wait ever timed out) and every sweep must
reproduce the first one's marginals exactly (the scan is exact and deterministic)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [3, 1025, 2049, 70_001, 250_001, 262_145, 400_001, 524_289]
devs = []
for T in sizes:
    m = cx.synth.ssm_chain(T, seed=T, random_variances=True)
    d = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(m, d)
    d.sweep(2)
    devs.append((T, m, d, d.get_marginals(m.x_ids)))
t0 = time.perf_counter()
total = 0
for rnd in range(40):
    for T, m, d, ref in devs:
        n = 500 if T < 100_000 else 200
        d.sweep(n)
        total += n
    for T, m, d, ref in devs:
        got = d.get_marginals(m.x_ids)
        assert np.array_equal(got, ref), f"T = {T}: marginals moved in round {rnd}"
for T, m, d, ref in devs:
    st = d.chain_scan_stats()
    assert st["state"] == 1, (T, st)
    print(f"T = {T}: {st}", flush=True)
print(f"{total} sweeps in {time.perf_counter() - t0:.1f} s, every handle still on one launch, marginals bit-identical throughout")
