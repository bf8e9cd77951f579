#!/usr/bin/env python3
"""tests/lab_conditioning.py — what tolerance the d > 1 kernels actually hold on ill-conditioned models (cond(Q) = 1e6, |A| near 1):
device vs the exact smoother, with the two exact solvers' disagreement (numpy LU vs C LU) beside it as the oracle's own error."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact


def model(d, T, condQ, rho, seed, condR=1.0):
    rng = np.random.default_rng(seed)
    U = np.linalg.qr(rng.standard_normal((d, d)))[0]
    V = np.linalg.qr(rng.standard_normal((d, d)))[0]
    Q = U @ np.diag(np.logspace(-np.log10(condQ), 0, d)) @ U.T; Q = 0.5 * (Q + Q.T)
    R = V @ np.diag(np.logspace(0, np.log10(condR), d)) @ V.T; R = 0.5 * (R + R.T)
    A = rho * np.linalg.qr(rng.standard_normal((d, d)))[0]
    return cx.synth.lgssm_chain(T, d=d, seed=seed, A=A, Q=Q, R=R)


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


for d, T, sched in [(4, 400, "scan"), (4, 400, "flood"), (2, 300, "scan"), (64, 14, "flood")]:
    for condQ, rho in [(1.0, 0.95), (1e3, 0.99), (1e6, 0.99), (1e6, 0.999), (1e8, 0.99)]:
        m = model(d, T, condQ, rho, seed=11)
        A, Q, R = m.meta["A"], m.meta["Q"], m.meta["R"]
        em, ec = exact.lgssm_posterior(m.data_y, A, Q, R)
        em2, ec2 = exact.lgssm_posterior_c(m.data_y, A, Q, R)
        dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN if sched == "scan" else L.SCHED_FUSED)
        cx.synth.load_into_device(m, dev)
        dev.sweep(1 if sched == "scan" else T + 3)
        g = dev.get_marginals(m.x_ids)
        print(json.dumps({"d": d, "T": T, "schedule": sched, "condQ": condQ, "rho": rho,
                          "mean_err": relerr(g[:, :d], em), "cov_err": relerr(g[:, d:].reshape(T, d, d), ec),
                          "oracle_mean_err": relerr(em2, em), "oracle_cov_err": relerr(ec2, ec), "nan": int(np.isnan(g).sum())}), flush=True)
        dev.close()
