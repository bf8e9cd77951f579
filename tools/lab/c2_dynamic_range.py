"""lab: the chain scan with factor variances over twelve decades (T = 70,001): one launch (maps up to scale), two launches (maps normalised
to D = 1) and the float64 tridiagonal solve, each against the same solve in 50-digit decimal arithmetic."""
import os
import sys
from decimal import Decimal, getcontext

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

getcontext().prec = 50
T = int(sys.argv[1]) if len(sys.argv) > 1 else 70001
model = cx.synth.ssm_chain(T, seed=T)
rng = np.random.default_rng(T)
r, q = 10.0 ** rng.uniform(-6, 6, T), 10.0 ** rng.uniform(-6, 6, T - 1)
model.factor_var[:] = np.concatenate([r, q])


def thomas(tp):
    one = tp(1)
    rr, qq, yy = [tp(float(x)) for x in r], [tp(float(x)) for x in q], [tp(float(x)) for x in model.data_y]
    diag = [one / x for x in rr]
    for i in range(T - 1):
        w = one / qq[i]
        diag[i] += w; diag[i + 1] += w
    off = [-one / x for x in qq]
    rhs = [yy[i] / rr[i] for i in range(T)]
    # mean by elimination; variances from the two one-sided recursions (diagonal of the inverse of a tridiagonal matrix)
    d = diag[:]
    b = rhs[:]
    for i in range(1, T):
        m = off[i - 1] / d[i - 1]
        d[i] -= m * off[i - 1]
        b[i] -= m * b[i - 1]
    x = [tp(0)] * T
    x[-1] = b[-1] / d[-1]
    for i in range(T - 2, -1, -1):
        x[i] = (b[i] - off[i] * x[i + 1]) / d[i]
    e = diag[:]
    for i in range(T - 2, -1, -1):
        e[i] -= off[i] * off[i] / e[i + 1]
    var = [one / (d[i] + e[i] - diag[i]) for i in range(T)]
    return np.array([float(v) for v in x]), np.array([float(v) for v in var])


def solve(onepass):
    if not onepass:
        os.environ["CX_CHAIN_ONEPASS"] = "0"
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    os.environ.pop("CX_CHAIN_ONEPASS", None)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    dev.sweep(1)
    m = dev.get_marginals(model.x_ids)
    print(dev.chain_scan_stats())
    return m


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), np.median(np.abs(b)))))


xm, xv = thomas(Decimal)
fm, fv = thomas(float)
a, b = solve(True), solve(False)
print(f"T = {T}; largest relative error (by the larger of the entry and the median entry) against the 50-digit solve")
print(f"  float64 Thomas solve : mean {rel(fm, xm):.2e}  variance {rel(fv, xv):.2e}")
print(f"  one launch           : mean {rel(a[:, 0], xm):.2e}  variance {rel(a[:, 1], xv):.2e}")
print(f"  two launches         : mean {rel(b[:, 0], xm):.2e}  variance {rel(b[:, 1], xv):.2e}")
