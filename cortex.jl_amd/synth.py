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
    # dim > 1 (linear-Gaussian factors x_out = A x_in + N(0, Q)): factor_var holds the parameter-set index
    dim: int = 1
    edge_role: np.ndarray | None = None
    psets: dict = field(default_factory=dict)      # parameter set -> (A, Q)

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


def ssm_chain_linear(T: int, seed: int = 1234, a: float = 0.9, b: float = 0.3, q: float = 0.5, r: float = 0.7) -> Model:
    """The same chain with LINEAR transitions x_{t+1} = a_t x_t + b_t + N(0, q_t) (CX_FACTOR_GAUSS_LINEAR; a_t, b_t, q_t jittered around
    the arguments) and additive likelihoods of variance r: factor_var holds one (q, a, b) row per factor, edge_role the :in / :out ends."""
    rng = np.random.default_rng(seed)
    x = np.arange(1, T + 1, dtype=np.int64)
    y = x + T
    lik = x + 2 * T
    tr = np.arange(3 * T + 1, 4 * T, dtype=np.int64)
    edge_var = np.concatenate([y, x, x[:-1], x[1:]])
    edge_fac = np.concatenate([lik, lik, tr, tr])
    edge_role = np.concatenate([np.full(2 * T, L.ROLE_OUT), np.full(T - 1, L.ROLE_IN), np.full(T - 1, L.ROLE_OUT)]).astype(np.int32)
    at = a * rng.uniform(0.8, 1.2, T - 1) * rng.choice([1.0, -1.0], T - 1, p=[0.8, 0.2])
    bt = b * rng.standard_normal(T - 1)
    qt = q * rng.uniform(0.5, 2.0, T - 1)
    params = np.zeros((2 * T - 1, 3))
    params[:T, 0] = r
    params[T:, 0], params[T:, 1], params[T:, 2] = qt, at, bt
    state = np.empty(T)
    state[0] = rng.standard_normal()
    for t in range(1, T):
        state[t] = at[t - 1] * state[t - 1] + bt[t - 1] + np.sqrt(qt[t - 1]) * rng.standard_normal()
    data = state + np.sqrt(r) * rng.standard_normal(T)
    kind = np.concatenate([np.full(T, L.FACTOR_GAUSS_ADDITIVE), np.full(T - 1, L.FACTOR_GAUSS_LINEAR)]).astype(np.int32)
    return Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=np.concatenate([lik, tr]), factor_kind=kind, factor_var=params,
                 x_ids=x, data_var=y, data_fac=lik, data_y=data, edge_role=edge_role,
                 meta={"T": T, "r": r, "q": qt, "a": at, "b": bt, "kind": "ssm_chain_linear"})


def _grid_rows(seed: int, total_rows: int, n_cols: int, r0: int, r1: int):
    """Per-row random streams keyed by (seed, kind, global row): any rank regenerates exactly the rows it needs, and
    the union over ranks is the same global grid whatever the partition."""
    nr = r1 - r0
    yobs = np.empty((nr, n_cols)); r = np.empty((nr, n_cols)); qh = np.empty((nr, max(n_cols - 1, 0)))
    jj = np.arange(n_cols)
    for k, i in enumerate(range(r0, r1)):
        rng = np.random.default_rng([seed, 0, i])
        field_ = 3.0 * np.sin(2 * np.pi * i / total_rows) * np.cos(2 * np.pi * jj / n_cols) + 0.002 * (i + jj)
        yobs[k] = field_ + rng.standard_normal(n_cols)
        r[k] = rng.uniform(0.5, 2.0, n_cols)
        qh[k] = rng.uniform(0.5, 2.0, max(n_cols - 1, 0))
    return yobs, r, qh


def _grid_qv(seed: int, n_cols: int, i0: int, i1: int):
    """variances of the vertical factors between global rows i and i+1, for i in [i0, i1)."""
    qv = np.empty((max(i1 - i0, 0), n_cols))
    for k, i in enumerate(range(i0, i1)):
        qv[k] = np.random.default_rng([seed, 1, i]).uniform(0.5, 2.0, n_cols)
    return qv


def gaussian_grid(n_rows: int, n_cols: int, seed: int = 1234, row0: int = 0, row1: int | None = None) -> Model:
    """2-D Gaussian grid, config C4: n_rows*n_cols unary edges + 2*(horizontal+vertical pairwise factors) edges.
    N = 1415 gives 10,005,465 bipartite edges.  Global numbering (n_rows x n_cols is the WHOLE grid): variable (i, j)
    has id 1 + i*n_cols + j; unary factor ids follow the variables, then horizontal, then vertical pairwise factors.

    row0/row1 select the strip of rows [row0, row1) owned by one rank; the returned model then also holds the cut
    vertical factors and, as degree-1 "ghost" variables, the rows row0-1 and row1 they connect to
    (meta["ghost_rows"]).  With the defaults the model is the whole grid.

    Observation field: smooth surface + N(0,1) noise; r_i, q_ij ~ U(0.5, 2) — the precision matrix is strictly
    diagonally dominant, so Gaussian BP converges (Weiss & Freeman 2001)."""
    row1 = n_rows if row1 is None else row1
    V = n_rows * n_cols
    H = n_rows * (n_cols - 1)
    var_id = lambda i, j: 1 + i * n_cols + j                       # noqa: E731
    i_own = np.arange(row0, row1, dtype=np.int64)
    jj = np.arange(n_cols, dtype=np.int64)
    vid = var_id(i_own[:, None], jj[None, :])                       # [nr, nc]
    unary = vid + V
    hfac = 2 * V + 1 + i_own[:, None] * (n_cols - 1) + jj[None, :-1]
    yobs, r, qh = _grid_rows(seed, n_rows, n_cols, row0, row1)
    # vertical factors touching owned rows: between rows i and i+1 for i in [max(row0-1,0), min(row1, n_rows-1))
    v0, v1 = max(row0 - 1, 0), min(row1, n_rows - 1)
    iv = np.arange(v0, v1, dtype=np.int64)
    vfac = 2 * V + H + 1 + iv[:, None] * n_cols + jj[None, :]
    qv = _grid_qv(seed, n_cols, v0, v1)
    up_var = var_id(iv[:, None], jj[None, :])
    dn_var = var_id(iv[:, None] + 1, jj[None, :])
    edge_var = np.concatenate([vid.ravel(), vid[:, :-1].ravel(), vid[:, 1:].ravel(), up_var.ravel(), dn_var.ravel()])
    edge_fac = np.concatenate([unary.ravel(), hfac.ravel(), hfac.ravel(), vfac.ravel(), vfac.ravel()])
    factor_ids = np.concatenate([unary.ravel(), hfac.ravel(), vfac.ravel()])
    factor_kind = np.concatenate([np.full(unary.size, L.FACTOR_OPAQUE, np.int32),
                                  np.full(hfac.size + vfac.size, L.FACTOR_GAUSS_ADDITIVE, np.int32)])
    factor_var = np.concatenate([np.ones(unary.size), qh.ravel(), qv.ravel()])
    ghost_rows = [i for i in (row0 - 1, row1) if 0 <= i < n_rows and not (row0 <= i < row1)]
    return Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=factor_ids, factor_kind=factor_kind, factor_var=factor_var,
                 x_ids=vid.ravel(), prior_var=vid.ravel(), prior_fac=unary.ravel(), prior_mean=yobs.ravel(),
                 prior_variance=r.ravel(),
                 meta={"n_rows": n_rows, "n_cols": n_cols, "row0": row0, "row1": row1, "y": yobs, "r": r, "qh": qh,
                       "qv": qv, "qv_row0": v0, "ghost_rows": ghost_rows, "kind": "gaussian_grid"})


def lgssm_chain(T: int, d: int = 4, seed: int = 1234, A=None, Q=None, R=None) -> Model:
    """d-dimensional linear-Gaussian state-space chain, configs C3 (d = 4, T = 1e6) / C5 (d = 64, T = 1e5) of
    SURVEY.md §8d: x_{t+1} = A x_t + w, w ~ N(0, Q);  y_t = x_t + v, v ~ N(0, R);  x_1 ~ N(0, I) for the data only (the
    graph, like the reference's SSM test, has no prior factor).  d = 4: A = 0.95 * blockdiag(rot(0.1), rot(0.1)),
    Q = 0.1 I, R = I.  Other d: A = 0.95 * (random orthogonal, seeded).  Same id scheme as ssm_chain."""
    rng = np.random.default_rng(seed)
    if d % 2 == 0 and d <= 8:
        c, s_ = np.cos(0.1), np.sin(0.1)
        A_default = np.kron(np.eye(d // 2), np.array([[c, -s_], [s_, c]])) * 0.95
    else:
        A_default = 0.95 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    # A, Q, R given: another model of the same shape (e.g. the slow-mixing A = 0.999 I, Q = 1e-4 I, R = 10 I, or ill-conditioned noise)
    A = A if A is None else np.asarray(A, dtype=np.float64)
    if A is None:
        A = A_default
    Q = 0.1 * np.eye(d) if Q is None else np.asarray(Q, dtype=np.float64)
    R = np.eye(d) if R is None else np.asarray(R, dtype=np.float64)
    x = np.arange(1, T + 1, dtype=np.int64)
    y = x + T
    lik = x + 2 * T
    tr = np.arange(3 * T + 1, 4 * T, dtype=np.int64)
    edge_var = np.concatenate([y, x, x[:-1], x[1:]])
    edge_fac = np.concatenate([lik, lik, tr, tr])
    edge_role = np.concatenate([np.full(T, L.ROLE_OUT), np.full(T, L.ROLE_IN), np.full(T - 1, L.ROLE_IN),
                                np.full(T - 1, L.ROLE_OUT)]).astype(np.int32)
    state = np.empty((T, d))
    state[0] = rng.standard_normal(d)
    Lq, Lr = np.linalg.cholesky(Q), np.linalg.cholesky(R)     # (defaults: sqrt(0.1) I and I — the streams of the round-1 models)
    w = rng.standard_normal((T, d)) @ Lq.T
    for t in range(1, T):
        state[t] = A @ state[t - 1] + w[t]
    data = state + rng.standard_normal((T, d)) @ Lr.T
    return Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=np.concatenate([lik, tr]),
                 factor_kind=np.full(2 * T - 1, L.FACTOR_GAUSS_LINEAR, dtype=np.int32),
                 factor_var=np.concatenate([np.ones(T), np.zeros(T - 1)]),   # parameter sets: 1 = likelihood, 0 = transition
                 x_ids=x, data_var=y, data_fac=lik, data_y=data, dim=d, edge_role=edge_role,
                 psets={0: (A, Q), 1: (np.eye(d), R)}, meta={"T": T, "A": A, "Q": Q, "R": R, "kind": "lgssm_chain"})


def lgssm_comb(n_spine: int, d: int = 4, teeth: int = 1, seed: int = 1234) -> Model:
    """d-dimensional linear-Gaussian TREE: a spine of n_spine states x_{t+1} = A x_t + w and, below every state, a path of `teeth` more
    states of the same dynamics (a latent layer); every state observed, y = x + v.  A = 0.9 * (random orthogonal), Q = 0.2 I, R = I.
    The level schedule of such a tree has ~ n_spine levels; over heavy paths it is two or three light depths."""
    rng = np.random.default_rng(seed)
    A = 0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    Q, R = 0.2 * np.eye(d), np.eye(d)
    spine = np.arange(n_spine, dtype=np.int64)
    par = [spine[:-1]]
    chi = [spine[1:]]
    up, nxt = spine, n_spine
    for _ in range(teeth):
        c = np.arange(nxt, nxt + n_spine, dtype=np.int64)
        par.append(up); chi.append(c)
        up, nxt = c, nxt + n_spine
    par, chi = np.concatenate(par), np.concatenate(chi)
    n = int(nxt)
    x = np.arange(1, n + 1, dtype=np.int64)
    y, lik = x + n, x + 2 * n
    tr = 3 * n + 1 + np.arange(len(par), dtype=np.int64)
    edge_var = np.concatenate([y, x, x[par], x[chi]])
    edge_fac = np.concatenate([lik, lik, tr, tr])
    role = np.concatenate([np.full(n, L.ROLE_OUT), np.full(n, L.ROLE_IN), np.full(len(par), L.ROLE_IN), np.full(len(par), L.ROLE_OUT)]).astype(np.int32)
    data = rng.standard_normal((n, d)) * 1.5            # (synthetic data of the model's scale: the timing does not depend on it)
    return Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=np.concatenate([lik, tr]),
                 factor_kind=np.full(n + len(par), L.FACTOR_GAUSS_LINEAR, dtype=np.int32),
                 factor_var=np.concatenate([np.ones(n), np.zeros(len(par))]), x_ids=x, data_var=y, data_fac=lik, data_y=data,
                 dim=d, edge_role=role, psets={0: (A, Q), 1: (np.eye(d), R)}, meta={"kind": "lgssm_comb", "n_spine": n_spine, "teeth": teeth})


def concat_models(models) -> Model:
    """several dim > 1 models as ONE graph of disjoint components (ids shifted past each other): chains of different lengths,
    isolated variables — what the segmented chain scan has to keep apart"""
    off, parts = 0, []
    for m in models:
        parts.append((m, off))
        off += int(max(m.edge_var.max(), m.edge_fac.max()))
    cat = lambda f: np.concatenate([f(m, o) for m, o in parts])
    first = models[0]
    return Model(edge_var=cat(lambda m, o: m.edge_var + o), edge_fac=cat(lambda m, o: m.edge_fac + o),
                 factor_ids=cat(lambda m, o: m.factor_ids + o), factor_kind=cat(lambda m, o: m.factor_kind),
                 factor_var=cat(lambda m, o: m.factor_var), x_ids=cat(lambda m, o: m.x_ids + o),
                 data_var=cat(lambda m, o: m.data_var + o), data_fac=cat(lambda m, o: m.data_fac + o),
                 data_y=np.concatenate([m.data_y for m in models]), dim=first.dim, edge_role=cat(lambda m, o: m.edge_role),
                 psets=first.psets, meta={"parts": [(len(m.x_ids), o) for m, o in parts], "kind": "concat"})


def kary_model(n_factors: int, seed: int = 1234, tree: bool = True, k_choices=(2, 3, 4, 5, 6), observe: float = 0.0) -> Model:
    """Scalar model with linear-Gaussian factors of MORE than two edges (CX_FACTOR_GAUSS_LINEAR_N): x_out = sum_i a_i x_i + b + N(0, q),
    k in `k_choices` inputs each, plus a unary prior factor on every latent variable (message set by the caller, as the prior of
    test/inference_engine_tests.jl:1224).  tree: every new factor hangs off ONE existing variable and brings its other k variables along
    (a bipartite tree: sum-product is exact); otherwise the variables of a factor are drawn from a pool (loopy).  observe: the share of
    leaf variables that carry a point-mass datum instead of a prior.  meta: the coefficient per edge (`coef_var`, `coef_fac`, `coef`),
    `q`, `b`, `out_var` per factor."""
    rng = np.random.default_rng(seed)
    fac_vars, nvar = [], 1
    for f in range(n_factors):
        k = int(rng.choice(k_choices))
        if tree:
            anchor = int(rng.integers(1, nvar + 1))
            vs = [anchor] + list(range(nvar + 1, nvar + k + 1))
            nvar += k
        else:
            pool = max(nvar, k + 2, int(1.6 * (f + 2)))
            vs = sorted(int(v) for v in rng.choice(np.arange(1, pool + 1), size=k + 1, replace=False))
            nvar = max(nvar, pool)
        fac_vars.append(vs)
    x = np.arange(1, nvar + 1, dtype=np.int64)
    used = np.unique(np.concatenate([np.asarray(v) for v in fac_vars]))
    prior_fac_of = {int(v): nvar + 1 + i for i, v in enumerate(used)}           # one unary factor per variable that occurs
    kf_id0 = nvar + len(used) + 1
    ev, ef, er, coef_var, coef_fac, coef, out_var = [], [], [], [], [], [], []
    q, b = rng.uniform(0.3, 1.5, n_factors), rng.standard_normal(n_factors)
    for f, vs in enumerate(fac_vars):
        fid = kf_id0 + f
        o = int(rng.integers(0, len(vs)))
        out_var.append(vs[o])
        for j, v in enumerate(vs):
            ev.append(v); ef.append(fid); er.append(L.ROLE_OUT if j == o else L.ROLE_IN)
            if j != o:
                coef_var.append(v); coef_fac.append(fid)
                coef.append(float(rng.uniform(0.4, 1.3) * rng.choice([1.0, -1.0], p=[0.75, 0.25])))
    # leaves that are observed carry data on their unary factor's edge instead of a prior message
    deg = np.bincount(np.asarray(ev), minlength=nvar + 1)
    is_obs = np.zeros(nvar + 1, dtype=bool)
    if observe > 0:
        leaves = [int(v) for v in used if deg[v] == 1]
        is_obs[[v for v in leaves if rng.random() < observe]] = True
    pv = np.array([v for v in used if not is_obs[v]], dtype=np.int64)
    for v in used:                                   # every occurring variable has its unary factor (observed ones: the datum's factor is the k-ary one itself)
        if not is_obs[v]:
            ev.append(int(v)); ef.append(prior_fac_of[int(v)]); er.append(L.ROLE_OUT)
    data_var = np.array([v for v in used if is_obs[v]], dtype=np.int64)
    data_fac = np.array([ef[[i for i in range(len(ev)) if ev[i] == int(v)][0]] for v in data_var], dtype=np.int64)
    prior_ids = np.array([prior_fac_of[int(v)] for v in pv], dtype=np.int64)
    kf_ids = kf_id0 + np.arange(n_factors, dtype=np.int64)
    params = np.zeros((len(prior_ids) + n_factors, 2))
    params[len(prior_ids):, 0] = q
    params[len(prior_ids):, 1] = b
    return Model(edge_var=np.asarray(ev, dtype=np.int64), edge_fac=np.asarray(ef, dtype=np.int64), factor_ids=np.concatenate([prior_ids, kf_ids]),
                 factor_kind=np.concatenate([np.full(len(prior_ids), L.FACTOR_OPAQUE), np.full(n_factors, L.FACTOR_GAUSS_LINEAR_N)]).astype(np.int32),
                 factor_var=params, x_ids=pv, data_var=data_var, data_fac=data_fac, data_y=rng.standard_normal(len(data_var)) * 2,
                 prior_var=pv, prior_fac=prior_ids, prior_mean=rng.standard_normal(len(pv)) * 2, prior_variance=rng.uniform(0.5, 2.0, len(pv)),
                 edge_role=np.asarray(er, dtype=np.int32),
                 meta={"kind": "kary", "coef_var": np.asarray(coef_var, dtype=np.int64), "coef_fac": np.asarray(coef_fac, dtype=np.int64),
                       "coef": np.asarray(coef), "q": q, "b": b, "out_var": np.asarray(out_var, dtype=np.int64), "kary_ids": kf_ids,
                       "fac_vars": fac_vars, "used": used})


def tree_model(n_factors: int, seed: int = 1234, k_choices=(1, 1, 2, 3, 5), observe: float = 0.25, shape: str = "random", components: int = 1) -> Model:
    """A scalar Gaussian model whose factor graph is a FOREST with factors of every arity: pairwise linear factors x_out = a x_in + b +
    N(0, q) (CX_FACTOR_GAUSS_LINEAR, k = 1 input) next to factors of k >= 2 inputs (CX_FACTOR_GAUSS_LINEAR_N), a unary prior on every
    latent variable and point-mass data on a share `observe` of the leaves.  shape: where a new factor hangs — "random" (any existing
    variable: bushy, depth ~ log n), "deep" (a recent variable: long paths with side branches), "star" (the first variable: one hub of
    degree n_factors + 1), "comb" (pairwise factors only: a spine of states, a tooth of two variables below each — a state-space model
    with a latent layer; depth ~ n_factors / 3).  components: that many disjoint trees.  meta as synth.kary_model's (coefficients of the k-ary factors only, for
    cx_set_factor_coefficients) plus `all_coef_*`: the input coefficient of EVERY factor edge, for a dense solve."""
    rng = np.random.default_rng(seed)
    fac_vars, nvar, roots = [], components, list(range(1, components + 1))
    comp_vars = [[r] for r in roots]
    spine = [[r] for r in roots]
    for f in range(n_factors):
        k = 1 if shape == "comb" else int(rng.choice(k_choices))
        cv = comp_vars[f % components]
        if shape == "comb":
            # a spine of states, each with a tooth of two more variables (a latent layer below every state): depth ~ n / 3, two light depths
            sp, r = spine[f % components], (f // components) % 3
            anchor = sp[-1] if r < 2 else cv[-1]
            if r == 0:
                sp.append(nvar + 1)
        elif shape == "star":
            anchor = cv[0]
        elif shape == "deep":
            anchor = cv[-1 - int(rng.integers(0, min(3, len(cv))))]
        else:
            anchor = cv[int(rng.integers(0, len(cv)))]
        new = list(range(nvar + 1, nvar + k + 1))
        nvar += k
        cv.extend(new)
        fac_vars.append([anchor] + new)
    used = np.arange(1, nvar + 1, dtype=np.int64)
    prior_fac_of = {int(v): nvar + 1 + i for i, v in enumerate(used)}
    kf_id0 = 2 * nvar + 1
    ev, ef, er, cvr, cfc, cf, acv, acf, ac, out_var = [], [], [], [], [], [], [], [], [], []
    q, b = rng.uniform(0.3, 1.5, n_factors), rng.standard_normal(n_factors)
    kinds, params = [], []
    for f, vs in enumerate(fac_vars):
        fid = kf_id0 + f
        o = int(rng.integers(0, len(vs)))
        out_var.append(vs[o])
        pair_a = 0.0
        for j, v in enumerate(vs):
            ev.append(v); ef.append(fid); er.append(L.ROLE_OUT if j == o else L.ROLE_IN)
            if j != o:
                a = float(rng.uniform(0.4, 1.3) * rng.choice([1.0, -1.0], p=[0.75, 0.25]))
                acv.append(v); acf.append(fid); ac.append(a)
                if len(vs) > 2:
                    cvr.append(v); cfc.append(fid); cf.append(a)
                else:
                    pair_a = a
        if len(vs) > 2:
            kinds.append(L.FACTOR_GAUSS_LINEAR_N); params.append([q[f], b[f], 0.0])
        else:
            kinds.append(L.FACTOR_GAUSS_LINEAR); params.append([q[f], pair_a, b[f]])
    deg = np.bincount(np.asarray(ev), minlength=nvar + 1)
    is_obs = np.zeros(nvar + 1, dtype=bool)
    if observe > 0:
        is_obs[[int(v) for v in used if deg[v] == 1 and v not in roots and rng.random() < observe]] = True
    pv = np.array([v for v in used if not is_obs[v]], dtype=np.int64)
    for v in pv:
        ev.append(int(v)); ef.append(prior_fac_of[int(v)]); er.append(L.ROLE_OUT)
    data_var = np.array([v for v in used if is_obs[v]], dtype=np.int64)
    first_fac = {}
    for v, f in zip(ev, ef):
        first_fac.setdefault(int(v), int(f))
    data_fac = np.array([first_fac[int(v)] for v in data_var], dtype=np.int64)
    prior_ids = np.array([prior_fac_of[int(v)] for v in pv], dtype=np.int64)
    kf_ids = kf_id0 + np.arange(n_factors, dtype=np.int64)
    fp = np.zeros((len(prior_ids) + n_factors, 3))
    fp[len(prior_ids):] = np.asarray(params).reshape(n_factors, 3)
    is_kary = np.asarray(kinds) == L.FACTOR_GAUSS_LINEAR_N
    return Model(edge_var=np.asarray(ev, dtype=np.int64), edge_fac=np.asarray(ef, dtype=np.int64), factor_ids=np.concatenate([prior_ids, kf_ids]),
                 factor_kind=np.concatenate([np.full(len(prior_ids), L.FACTOR_OPAQUE), np.asarray(kinds)]).astype(np.int32),
                 factor_var=fp, x_ids=pv, data_var=data_var, data_fac=data_fac, data_y=rng.standard_normal(len(data_var)) * 2,
                 prior_var=pv, prior_fac=prior_ids, prior_mean=rng.standard_normal(len(pv)) * 2, prior_variance=rng.uniform(0.5, 2.0, len(pv)),
                 edge_role=np.asarray(er, dtype=np.int32),
                 meta={"kind": "kary" if is_kary.any() else "tree", "coef_var": np.asarray(cvr, dtype=np.int64), "coef_fac": np.asarray(cfc, dtype=np.int64),
                       "coef": np.asarray(cf), "all_coef_var": np.asarray(acv, dtype=np.int64), "all_coef_fac": np.asarray(acf, dtype=np.int64),
                       "all_coef": np.asarray(ac), "q": q, "b": b, "out_var": np.asarray(out_var, dtype=np.int64), "kary_ids": kf_ids,
                       "fac_vars": fac_vars, "used": used, "is_kary": is_kary})


def load_into_device(model: Model, dev, seed_variance: float | None = None):
    """graph upload + the data injection a user of the reference does with set_value! before update_marginals!."""
    if model.dim > 1:
        for k, (A, Q) in model.psets.items():
            dev.set_factor_matrices(k, A, Q)
        dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var,
                         edge_role=model.edge_role)
        if len(model.data_var):
            dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, model.data_y)
        if seed_variance is not None:
            dev.seed_messages(L.TO_VARIABLE, 0.0, seed_variance)
        return dev
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, edge_role=model.edge_role)
    if model.meta.get("kind") == "kary":
        dev.set_factor_coefficients(model.meta["coef_var"], model.meta["coef_fac"], model.meta["coef"])
    if len(model.data_var):
        dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, model.data_y)
    if len(model.prior_var):
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT,
                         np.stack([model.prior_mean, model.prior_variance], axis=1))
    if seed_variance is not None:
        dev.seed_messages(L.TO_VARIABLE, 0.0, seed_variance)
    return dev


# ---- variational state-space model ----------------------------------------------------------------
@dataclass
class VmpModel:
    """The graph of test/inference_engine_tests.jl:691-715 / :1032-1056: ssnoise (id 1), obsnoise (2), x_1..x_n,
    y_1..y_n, n likelihood factors (y_i, x_i, obsnoise) and n - 1 transition factors (x_i, x_{i+1}, ssnoise)."""
    n: int
    edge_var: np.ndarray
    edge_fac: np.ndarray
    edge_role: np.ndarray
    factor_ids: np.ndarray
    x_ids: np.ndarray
    y_ids: np.ndarray
    ssnoise: int
    obsnoise: int
    data_y: np.ndarray


def vmp_ssm(n: int, seed: int = 1234, ssnoise_real: float = 100.0, obsnoise_real: float = 100.0) -> VmpModel:
    """Random walk with transition precision `ssnoise_real`, observed with precision `obsnoise_real` (:775-785)."""
    rng = np.random.default_rng(seed)
    steps = rng.standard_normal(n) / np.sqrt(ssnoise_real)
    steps[0] = 0.0
    walk = np.cumsum(steps)
    y = walk + rng.standard_normal(n) / np.sqrt(obsnoise_real)
    ss, obs = 1, 2
    x = np.arange(3, 3 + n, dtype=np.int64)
    yv = np.arange(3 + n, 3 + 2 * n, dtype=np.int64)
    lik = np.arange(3 + 2 * n, 3 + 3 * n, dtype=np.int64)
    tr = np.arange(3 + 3 * n, 3 + 4 * n - 1, dtype=np.int64)
    one = np.ones(n, dtype=np.int64)
    ev = np.concatenate([yv, x, obs * one, x[:-1], x[1:], ss * one[:-1]])
    ef = np.concatenate([lik, lik, lik, tr, tr, tr])
    role = np.concatenate([np.full(n, L.ROLE_OUT), np.full(n, L.ROLE_IN), np.full(n, L.ROLE_PRECISION),
                           np.full(n - 1, L.ROLE_IN), np.full(n - 1, L.ROLE_OUT), np.full(n - 1, L.ROLE_PRECISION)]).astype(np.int32)
    return VmpModel(n=n, edge_var=ev, edge_fac=ef, edge_role=role, factor_ids=np.concatenate([lik, tr]), x_ids=x, y_ids=yv,
                    ssnoise=ss, obsnoise=obs, data_y=y)


def load_vmp_into_device(model: VmpModel, dev):
    """graph upload + the initial marginals and the data of :717-736"""
    nf = len(model.factor_ids)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, np.full(nf, L.FACTOR_NORMAL_PRECISION, dtype=np.int32),
                     np.zeros(nf), edge_role=model.edge_role)
    dev.set_marginals([model.ssnoise, model.obsnoise], L.FORM_GAMMA, [1.0, 1.0, 1.0, 1.0])
    dev.set_marginals(model.x_ids, L.FORM_MEAN_PRECISION, np.tile([0.0, 1.0], model.n))
    dev.set_marginals(model.y_ids, L.FORM_POINT, model.data_y)
    return dev
