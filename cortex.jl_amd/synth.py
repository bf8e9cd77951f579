"""Synthetic models of BASELINE.json's configs, as plain id arrays.

Graph shapes follow the reference's own test models: the SSM chain is built exactly as
test/inference_engine_tests.jl:436-453 (x ids 1..n, y ids n+1..2n, likelihood factors 2n+1..3n,
transition factors 3n+1..4n-1 — BipartiteFactorGraphs hands out one shared id sequence).  The grid
model is the C4 shape of SURVEY.md §8d: unary observation factors (messages set by the caller, as the
prior in test/inference_engine_tests.jl:1224) + pairwise "difference" factors.
All randomness is numpy PCG64 with the stated seed; generators return the arrays both the device
path and the CPU checker consume, so no second RNG has to agree bit-for-bit.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _lib as L


@dataclass
class Model:
    edge_var: np.ndarray
    edge_fac: np.ndarray
    factor_ids: np.ndarray
    factor_kind: np.ndarray
    factor_var: np.ndarray          # additive-noise variance per factor (unused for opaque factors)
    x_ids: np.ndarray               # the latent variables whose marginals are requested
    # data: messages the caller sets before inference
    data_var: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))   # clamped observations
    data_fac: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    data_y: np.ndarray = field(default_factory=lambda: np.zeros(0))
    prior_var: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))  # factor→variable messages set directly
    prior_fac: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    prior_mean: np.ndarray = field(default_factory=lambda: np.zeros(0))
    prior_variance: np.ndarray = field(default_factory=lambda: np.zeros(0))
    meta: dict = field(default_factory=dict)

    @property
    def n_edges(self) -> int:
        return len(self.edge_var)


def ssm_chain(T: int, seed: int = 1234, q: float = 1.0, r: float = 1.0, random_variances: bool = False) -> Model:
    """Scalar-Gaussian state-space chain (Kalman smoother), configs C1/C2: E = 4T-2 edges."""
    rng = np.random.default_rng(seed)
    x = np.arange(1, T + 1, dtype=np.int64)
    y = x + T
    lik = x + 2 * T
    tr = np.arange(3 * T + 1, 4 * T, dtype=np.int64)
    edge_var = np.concatenate([y, x, x[:-1], x[1:]])
    edge_fac = np.concatenate([lik, lik, tr, tr])
    fvar_lik = rng.uniform(0.5, 2.0, T) if random_variances else np.full(T, float(r))
    fvar_tr = rng.uniform(0.5, 2.0, T - 1) if random_variances else np.full(T - 1, float(q))
    data = 2.0 * np.arange(1, T + 1) + rng.standard_normal(T)   # test/inference_engine_tests.jl:477-481
    return Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=np.concatenate([lik, tr]),
                 factor_kind=np.full(2 * T - 1, L.FACTOR_GAUSS_ADDITIVE, dtype=np.int32),
                 factor_var=np.concatenate([fvar_lik, fvar_tr]), x_ids=x, data_var=y, data_fac=lik, data_y=data,
                 meta={"T": T, "r": fvar_lik, "q": fvar_tr, "kind": "ssm_chain"})


def gaussian_grid(n_rows: int, n_cols: int, seed: int = 1234, row_offset: int = 0, total_rows: int | None = None) -> Model:
    """2-D Gaussian grid, config C4: n_rows*n_cols unary edges + 2*(horizontal+vertical pairwise factors) edges.
    N = 1415 gives 10,005,465 bipartite edges.  Variable (i, j) has id 1 + i*n_cols + j; unary factor ids follow
    the variables, then horizontal, then vertical pairwise factors.

    Observation field: smooth surface + N(0,1) noise; r_i, q_ij ~ U(0.5, 2) — the precision matrix is strictly
    diagonally dominant, so Gaussian BP converges (Weiss & Freeman 2001)."""
    rng = np.random.default_rng(seed)
    nv = n_rows * n_cols
    idx = np.arange(nv, dtype=np.int64).reshape(n_rows, n_cols)
    var_id = idx + 1
    unary = var_id + nv
    nh, nvert = n_rows * (n_cols - 1), (n_rows - 1) * n_cols
    hfac = (2 * nv + 1 + np.arange(nh, dtype=np.int64)).reshape(n_rows, max(n_cols - 1, 0))
    vfac = (2 * nv + nh + 1 + np.arange(nvert, dtype=np.int64)).reshape(max(n_rows - 1, 0), n_cols)
    edge_var = np.concatenate([var_id.ravel(), var_id[:, :-1].ravel(), var_id[:, 1:].ravel(), var_id[:-1, :].ravel(),
                               var_id[1:, :].ravel()])
    edge_fac = np.concatenate([unary.ravel(), hfac.ravel(), hfac.ravel(), vfac.ravel(), vfac.ravel()])
    ii, jj = np.meshgrid(np.arange(n_rows) + row_offset, np.arange(n_cols), indexing="ij")
    scale = float(total_rows or n_rows)
    field_ = 3.0 * np.sin(2 * np.pi * ii / scale) * np.cos(2 * np.pi * jj / n_cols) + 0.002 * (ii + jj)
    yobs = field_ + rng.standard_normal((n_rows, n_cols))
    r = rng.uniform(0.5, 2.0, (n_rows, n_cols))
    qh = rng.uniform(0.5, 2.0, (n_rows, max(n_cols - 1, 0)))
    qv = rng.uniform(0.5, 2.0, (max(n_rows - 1, 0), n_cols))
    factor_ids = np.concatenate([unary.ravel(), hfac.ravel(), vfac.ravel()])
    factor_kind = np.concatenate([np.full(nv, L.FACTOR_OPAQUE, np.int32), np.full(nh + nvert, L.FACTOR_GAUSS_ADDITIVE, np.int32)])
    factor_var = np.concatenate([np.ones(nv), qh.ravel(), qv.ravel()])
    return Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=factor_ids, factor_kind=factor_kind, factor_var=factor_var,
                 x_ids=var_id.ravel(), prior_var=var_id.ravel(), prior_fac=unary.ravel(), prior_mean=yobs.ravel(),
                 prior_variance=r.ravel(),
                 meta={"n_rows": n_rows, "n_cols": n_cols, "y": yobs, "r": r, "qh": qh, "qv": qv, "kind": "gaussian_grid"})


def load_into_device(model: Model, dev, seed_variance: float | None = None):
    """graph upload + the data injection a user of the reference does with set_value! before update_marginals!."""
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var)
    if len(model.data_var):
        dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, model.data_y)
    if len(model.prior_var):
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT,
                         np.stack([model.prior_mean, model.prior_variance], axis=1))
    if seed_variance is not None:
        dev.seed_messages(L.TO_VARIABLE, 0.0, seed_variance)
    return dev
