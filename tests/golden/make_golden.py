#!/usr/bin/env python3
"""tests/golden/make_golden.py — regenerates the committed golden vectors (tests/golden/*.json).

The reference (Cortex.jl, Julia) cannot run in the authoring container, so no vector here comes from executing it.  Each
file holds seeded INPUTS and the EXPECTED outputs of the path, produced by the checker code under oracle/ — which is pinned
separately against the reference's own known-answer tests (tests/test_oracle_reference_kats.py) — and, where the model has
one, by an independent exact solver:

  chain16.json    SSM of test/inference_engine_tests.jl:436-481, T = 16: data, the restated engine's marginals after one
                  update_marginals!, the tridiagonal (Thomas) posterior
  grid8x8.json    8 x 8 Gaussian grid: priors, factor variances, the flooding checker's messages after 5 sweeps and its
                  marginals at convergence, the dense-solve posterior means
  lgssm_d4.json   d = 4 linear-Gaussian chain, T = 8: A, Q, R, data, block-tridiagonal posterior means and covariances
  lgssm_d8.json   d = 8 linear-Gaussian chain, T = 6 (SURVEY §8c "d in {4, 8} blocks"; on the device a dimension between 4 and 64 runs
                  embedded in the d = 64 path): A, Q, R, data, block-tridiagonal posterior
  lgssm_d64.json  d = 64 linear-Gaussian chain, T = 3 (the MFMA path's rule): data, posterior means, the posterior covariance of the
                  middle state and the diagonals of all three (A is regenerated from the seed by cx.synth.lgssm_chain; its first
                  row is stored as a guard)
  vmp_n8.json     variational SSM of :691-770 / :1032-1120, n = 8: data, posteriors after 5 x (x; ssnoise, obsnoise) for the
                  mean-field and the structured family (array form, oracle/vmp.py)
  tree24.json     a forest of 24 factors with two to six variables each (synth.tree_model, two components, observed leaves): the graph,
                  every parameter, priors and data, and the dense-solve posterior (tests/kary_support.py) — what ONE sweep of the tree
                  schedule has to return; the numpy execution of the schedule's plan (tests/test_tree_plan.py) reproduces it on the CPU
  kats.json       constants of the reference's own known answers: Beta-Bernoulli posterior (:360-376), tracing values 2, 4, 9
                  (:1226-1261)

Run from the repository root:  python tests/golden/make_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import cortex.jl_amd as cx  # noqa: E402  (synthetic model builders only; no device code is touched)
from oracle import exact, vmp  # noqa: E402
from tests.helpers import engine_oracle_from_model, flood_oracle_from_model  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    def enc(o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, (np.floating, np.integer)):
            return o.item()
        raise TypeError(type(o))
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, default=enc, indent=0, separators=(",", ":"))
        f.write("\n")


def chain16():
    m = cx.synth.ssm_chain(16, seed=1234)
    E = engine_oracle_from_model(m)
    E.set_messages_to_factor(m.data_var, m.data_fac, m.data_y)
    E.update_marginals(m.x_ids)
    _t, em, ev = E.get_marginals(m.x_ids)
    xm, xv = exact.ssm_chain_posterior(m.data_y, 1.0, 1.0)
    dump("chain16.json", {"T": 16, "seed": 1234, "q": 1.0, "r": 1.0, "x_ids": m.x_ids, "data_var": m.data_var, "data_fac": m.data_fac,
                          "data_y": m.data_y, "engine_mean": em, "engine_variance": ev, "exact_mean": xm, "exact_variance": xv})


def grid8x8():
    m = cx.synth.gaussian_grid(8, 8, seed=1234)
    g = flood_oracle_from_model(m, 1e6)
    g.sweep(5)
    f2v5 = (g.f2v_m.copy(), g.f2v_v.copy())
    g.sweep(400)
    mm, mv = g.marginals()
    xi = np.searchsorted(g.var_ids, m.x_ids)
    mean = exact.grid_posterior_mean(8, 8, m.meta["y"], m.meta["r"], m.meta["qh"], m.meta["qv"])
    dump("grid8x8.json", {"rows": 8, "cols": 8, "seed": 1234, "seed_variance": 1e6, "edge_var": g.edge_var, "edge_fac": g.edge_fac,
                          "sweeps": 5, "f2v_mean_after_5": f2v5[0], "f2v_variance_after_5": f2v5[1], "x_ids": m.x_ids,
                          "bp_mean_converged": mm[xi], "bp_variance_converged": mv[xi], "exact_mean": mean})


def lgssm_d4():
    m = cx.synth.lgssm_chain(8, d=4, seed=1234)
    em, ecov = exact.lgssm_posterior(m.data_y, m.meta["A"], m.meta["Q"], m.meta["R"])
    dump("lgssm_d4.json", {"T": 8, "d": 4, "seed": 1234, "A": m.meta["A"], "Q": m.meta["Q"], "R": m.meta["R"], "data_y": m.data_y,
                           "x_ids": m.x_ids, "posterior_mean": em, "posterior_covariance": ecov})


def lgssm_d8():
    m = cx.synth.lgssm_chain(6, d=8, seed=1234)
    em, ecov = exact.lgssm_posterior(m.data_y, m.meta["A"], m.meta["Q"], m.meta["R"])
    dump("lgssm_d8.json", {"T": 6, "d": 8, "seed": 1234, "A": m.meta["A"], "Q": m.meta["Q"], "R": m.meta["R"], "data_y": m.data_y,
                           "x_ids": m.x_ids, "posterior_mean": em, "posterior_covariance": ecov})


def lgssm_d64():
    m = cx.synth.lgssm_chain(3, d=64, seed=1234)
    em, ecov = exact.lgssm_posterior(m.data_y, m.meta["A"], m.meta["Q"], m.meta["R"])
    dump("lgssm_d64.json", {"T": 3, "d": 64, "seed": 1234, "A_row0": m.meta["A"][0], "q": 0.1, "r": 1.0, "data_y": m.data_y, "x_ids": m.x_ids,
                            "posterior_mean": em, "posterior_covariance_middle": ecov[1],
                            "posterior_variances": np.stack([np.diag(c) for c in ecov])})


def vmp_n8():
    m = cx.synth.vmp_ssm(8, seed=1234)
    out = {"n": 8, "seed": 1234, "data_y": m.data_y, "iterations": 5, "calls": "5 x (update x; update [ssnoise, obsnoise])"}
    for name, cls in (("mean_field", vmp.MeanFieldVMP), ("structured", vmp.StructuredVMP)):
        a = cls(m.data_y)
        for _ in range(5):
            a.update(["x"]); a.update(["ssnoise", "obsnoise"])
        out[name] = {"x_mean": a.xm, "x_precision": a.xw, "ssnoise_shape_scale": list(a.ss), "obsnoise_shape_scale": list(a.obs)}
    dump("vmp_n8.json", out)


def tree24():
    from tests.kary_support import dense_posterior
    m = cx.synth.tree_model(24, seed=77, shape="random", components=2, observe=0.3)
    ids, em, ev = dense_posterior(m)
    dump("tree24.json", {"n_factors": 24, "seed": 77, "shape": "random", "components": 2, "observe": 0.3,
                         "edge_var": m.edge_var, "edge_fac": m.edge_fac, "edge_role": m.edge_role, "factor_ids": m.factor_ids, "factor_kind": m.factor_kind,
                         "factor_params": m.factor_var, "coef_var": m.meta["coef_var"], "coef_fac": m.meta["coef_fac"], "coef": m.meta["coef"],
                         "prior_var": m.prior_var, "prior_fac": m.prior_fac, "prior_mean": m.prior_mean, "prior_variance": m.prior_variance,
                         "data_var": m.data_var, "data_fac": m.data_fac, "data_y": m.data_y,
                         "x_ids": ids, "posterior_mean": em, "posterior_variance": ev})


def kats():
    dump("kats.json", {"beta_bernoulli": {"source": "test/inference_engine_tests.jl:360-376", "prior": [1.0, 1.0],
                                          "rule": "posterior = Beta(1 + #true, 1 + #false)"},
                       "tracing": {"source": "test/inference_engine_tests.jl:1226-1261", "data": [1.0, 2.0], "prior": 3.0,
                                   "message_values": [2.0, 4.0], "marginal": 9.0, "rounds": 2,
                                   "note": "likelihood1 doubles 1.0, likelihood2 doubles 2.0, the prior message is 3.0: 2 + 4 + 3"}})


if __name__ == "__main__":
    chain16(); grid8x8(); lgssm_d4(); lgssm_d8(); lgssm_d64(); vmp_n8(); tree24(); kats()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".json")))
