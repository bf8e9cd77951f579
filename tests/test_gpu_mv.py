"""-m gpu: d-dimensional linear-Gaussian messages (dim 2, 3, 4) — BASELINE.json config 3 is the d = 4 chain.

The reference has no d-dimensional rule, so parity here is (i) per sweep against the numpy restatement oracle/mv.py in
the same flooding order, (ii) at the fixed point against the exact block-tridiagonal smoother, (iii) at the full C3 size
(T = 1e6, 3,999,998 edges) through a size-independent property: after K sweeps the marginals of a window deep inside
the chain equal the exact smoother of a slightly larger window (the smoother forgets exponentially)."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact
from oracle.mv import MvFlood
from tests.helpers import assert_close as _assert_close


def assert_close(a, b, rtol, what=""):
    # matrices carry numerically-zero entries: measure errors against the largest entry
    return _assert_close(a, b, rtol, what, scale_by="max")

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _dev(model, seed_variance=None):
    dev = cx.DeviceGraph(dim=model.dim, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance)
    return dev


@pytest.mark.parametrize("d", [2, 3, 4])
def test_mv_sweeps_match_numpy_restatement(hip_lib, d):
    T = 9
    model = cx.synth.lgssm_chain(T, d=d, seed=3)
    dev = _dev(model)
    o = MvFlood(model)
    g = o.g
    xs_set = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
    # messages into observed variables have no reader and are not computed (lazy, like the reference)
    pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if int(np.searchsorted(g.var_ids, g.edge_var[e])) in xs_set])
    for sweep in range(T + 2):
        dev.sweep(1)
        o.sweep(1)
        got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
        for row, e in zip(got, pe):
            if o.f2v[e] is None:
                assert np.all(np.isnan(row)), f"sweep {sweep} edge {e}: device defined, restatement undefined"
                continue
            m, S = o.f2v[e]
            if not np.all(np.isfinite(S)) or np.linalg.cond(S) > 1e12:
                continue  # message towards an observed variable before it is proper
            assert_close(row[:d], m, 1e-8, f"sweep {sweep} f2v mean edge {e}")
            assert_close(row[d:].reshape(d, d), S, 1e-8, f"sweep {sweep} f2v covariance edge {e}")
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    assert_close(marg[:, :d], em, RTOL, "marginal mean vs block-tridiagonal solve")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, RTOL, "marginal covariance vs block-tridiagonal solve")
    # variable→factor messages are recomputed on demand from the retained buffer
    xs = np.searchsorted(g.var_ids, model.x_ids)
    for t in (1, T // 2):
        e = [e_ for e_ in range(int(g.var_off[xs[t]]), int(g.var_off[xs[t] + 1]))][-1]
        got = dev.get_messages([g.edge_var[e]], [g.edge_fac[e]], L.TO_FACTOR)[0]
        m, S = o.v2f[e]
        assert_close(got[:d], m, 1e-8, "v2f mean"); assert_close(got[d:].reshape(d, d), S, 1e-8, "v2f covariance")


def test_mv_fixed_point_longer_chain(hip_lib):
    d, T = 4, 120
    model = cx.synth.lgssm_chain(T, d=d, seed=11)
    dev = _dev(model)
    dev.sweep(T + 3)
    marg = dev.get_marginals(model.x_ids)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    assert_close(marg[:, :d], em, RTOL, "marginal mean")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, RTOL, "marginal covariance")
    dev.residual()
    dev.sweep(2)
    assert dev.residual() < 1e-10


def test_config_c3_full_size_window_property(hip_lib):
    """BASELINE.json configs[2]: d = 4 state-space chain, T = 1e6 (3,999,998 edges)."""
    d, T, K, W, pad = 4, 1_000_000, 160, 400, 120
    model = cx.synth.lgssm_chain(T, d=d, seed=1234)
    assert model.n_edges == 3_999_998
    # flooding from data alone defines a message only once the wavefront from a chain end has reached it (T sweeps);
    # like any user of loopy/long-range BP, seed the messages with a vague N(0, 1e6 I) and let them be forgotten
    dev = _dev(model, seed_variance=1e6)
    assert dev.stats()["n_edges"] == 3_999_998
    dev.sweep(K)
    A, Q, R = model.meta["A"], model.meta["Q"], model.meta["R"]
    for start in (0, 123_456, 500_000, T - W):
        lo, hi = max(0, start - pad), min(T, start + W + pad)
        em, ecov = exact.lgssm_posterior(model.data_y[lo:hi], A, Q, R)
        ids = model.x_ids[start:start + W]
        # information from more than `pad` steps away (data, seeds, chain ends) is forgotten to < 1e-12:
        # the closed-loop factor of this model is ≈ 0.75 per step
        marg = dev.get_marginals(ids)
        sl = slice(start - lo, start - lo + W)
        assert_close(marg[:, :d], em[sl], 1e-8, f"C3 window at {start}: mean")
        assert_close(marg[:, d:].reshape(W, d, d), ecov[sl], 1e-8, f"C3 window at {start}: covariance")


def test_mv_errors(hip_lib):
    model = cx.synth.lgssm_chain(5, d=4)
    dev = cx.DeviceGraph(dim=4)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, edge_role=model.edge_role)
    with pytest.raises(cx.CortexHipError) as e:
        dev.sweep(1)      # parameter sets never supplied
    assert e.value.code == L.ERR_STATE
    with pytest.raises(cx.CortexHipError) as e:
        dev.set_factor_matrices(0, np.eye(4), -np.eye(4))
    assert e.value.code == L.ERR_INVALID_ARGUMENT
    with pytest.raises(cx.CortexHipError) as e:
        cx.DeviceGraph(dim=65)          # (5 .. 63 run embedded in the dim 64 path: see below)
    assert e.value.code == L.ERR_UNSUPPORTED
    with pytest.raises(cx.CortexHipError) as e:
        cx.DeviceGraph(dim=0)
    assert e.value.code == L.ERR_UNSUPPORTED


# ------------------------------------------------------------------------------------------------ d = 64 (MFMA path)

def test_mv64_sweeps_match_numpy_restatement(hip_lib):
    d, T = 64, 5
    model = cx.synth.lgssm_chain(T, d=d, seed=3)
    dev = _dev(model)
    o = MvFlood(model)
    g = o.g
    xs = np.searchsorted(g.var_ids, model.x_ids)
    # messages into the latent variables (those into observed variables have no reader and are not computed for d = 64)
    pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if np.searchsorted(g.var_ids, g.edge_var[e]) in set(xs)])
    # the d = 64 path evaluates the (constant) messages out of observed variables when the data is injected, i.e. one
    # sweep before the flooding order does: device after k sweeps == restatement after k + 1 sweeps
    o.sweep(1)
    for sweep in range(T + 2):
        dev.sweep(1)
        o.sweep(1)
        got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
        for row, e in zip(got, pe):
            if o.f2v[e] is None:
                assert np.all(np.isnan(row)), f"sweep {sweep} edge {e}: device defined, restatement undefined"
                continue
            m, S = o.f2v[e]
            assert_close(row[:d], m, 1e-8, f"sweep {sweep} f2v mean edge {e}")
            assert_close(row[d:].reshape(d, d), S, 1e-8, f"sweep {sweep} f2v covariance edge {e}")
    marg = dev.get_marginals(model.x_ids)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    assert_close(marg[:, :d], em, 1e-8, "d=64 marginal mean vs block-tridiagonal solve")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-8, "d=64 marginal covariance vs block-tridiagonal solve")
    e = int(g.var_off[xs[2] + 1]) - 1
    got = dev.get_messages([g.edge_var[e]], [g.edge_fac[e]], L.TO_FACTOR)[0]
    m, S = o.v2f[e]
    assert_close(got[:d], m, 1e-8, "d=64 v2f mean"); assert_close(got[d:].reshape(d, d), S, 1e-8, "d=64 v2f covariance")


def test_mv64_fixed_point_and_user_set_messages(hip_lib):
    d, T = 64, 24
    model = cx.synth.lgssm_chain(T, d=d, seed=5)
    dev = _dev(model)
    dev.sweep(T + 2)
    marg = dev.get_marginals(model.x_ids)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    assert_close(marg[:, :d], em, 1e-8, "marginal mean")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-8, "marginal covariance")
    dev.residual()
    dev.sweep(2)
    assert dev.residual() < 1e-9
    # a moment-form message set by the caller reads back unchanged
    rng = np.random.default_rng(0)
    Sg = rng.standard_normal((d, d)); Sg = Sg @ Sg.T + d * np.eye(d); mg = rng.standard_normal(d)
    tr0 = model.factor_ids[T]
    dev.set_messages([model.x_ids[0]], [tr0], L.TO_VARIABLE, L.FORM_MOMENT, np.concatenate([mg, Sg.ravel()]))
    back = dev.get_messages([model.x_ids[0]], [tr0], L.TO_VARIABLE)[0]
    assert_close(back[:d], mg, 1e-9, "round-trip mean"); assert_close(back[d:].reshape(d, d), Sg, 1e-9, "round-trip covariance")


def test_config_c5_full_size_window_property(hip_lib):
    """BASELINE.json configs[4]: d = 64 linear-Gaussian factors, 1e5 nodes (399,998 edges), MFMA update path."""
    d, T, K, W, pad = 64, 100_000, 96, 40, 70
    model = cx.synth.lgssm_chain(T, d=d, seed=1234)
    assert model.n_edges == 399_998
    dev = _dev(model, seed_variance=1e6)
    dev.sweep(K)
    A, Q, R = model.meta["A"], model.meta["Q"], model.meta["R"]
    for start in (0, 50_000, T - W):
        lo, hi = max(0, start - pad), min(T, start + W + pad)
        em, ecov = exact.lgssm_posterior(model.data_y[lo:hi], A, Q, R)
        marg = dev.get_marginals(model.x_ids[start:start + W])
        sl = slice(start - lo, start - lo + W)
        assert_close(marg[:, :d], em[sl], 1e-7, f"C5 window at {start}: mean")
        assert_close(marg[:, d:].reshape(W, d, d), ecov[sl], 1e-7, f"C5 window at {start}: covariance")


# ------------------------------------------------------------------------------------------------ per-sweep parity at length
# oracle/mv_flood.c (C) instead of oracle/mv.py (Python loops): the same flooding order at chain lengths where the
# wavefront from the chain ends has NOT crossed the chain, seeded with a vague N(0, 1e6 I) as the full-size configs are.

def _per_sweep_parity(model, sweeps, tol, lag, seed_variance=1e6):
    from oracle.mv import MvFloodC

    d = model.dim
    dev = _dev(model, seed_variance=seed_variance)
    o = MvFloodC(model)
    # d = 64 evaluates the constant messages out of observed variables when the data is injected (before any seed or sweep):
    # one un-seeded checker sweep defines exactly those messages; everything still undefined is then seeded, as on the device
    o.sweep(lag)
    o.seed(0.0, seed_variance)
    g = o.g
    xs = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
    pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if int(np.searchsorted(g.var_ids, g.edge_var[e])) in xs])
    for sweep in range(sweeps):
        dev.sweep(1)
        o.sweep(1, use_omp=True)
        got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
        assert np.array_equal(~np.isnan(got[:, 0]), o.f2v_def[pe].astype(bool)), f"sweep {sweep}: definedness differs"
        m, S = o.f2v_m[pe], o.f2v_S[pe]
        assert_close(got[:, :d], m, tol, f"sweep {sweep}: f2v means of {len(pe)} messages")
        assert_close(got[:, d:].reshape(-1, d, d), S, tol, f"sweep {sweep}: f2v covariances")
    dev.sweep(1)
    o.sweep(1, use_omp=True)
    mm, SS, ok = o.marginals()
    xi = np.searchsorted(g.var_ids, model.x_ids)
    marg = dev.get_marginals(model.x_ids)
    assert ok[xi].all()
    # the device's marginal is the product of the messages its last sweep READ (one sweep behind the fresh messages)
    return marg, mm[xi], SS[xi]


@pytest.mark.parametrize("d,T,sweeps", [(4, 2000, 12), (2, 500, 6), (3, 300, 6)])
def test_mv_per_sweep_parity_with_c_checker_long_chain(hip_lib, d, T, sweeps):
    model = cx.synth.lgssm_chain(T, d=d, seed=17)
    _per_sweep_parity(model, sweeps, 1e-8, 0)


def test_mv64_per_sweep_parity_with_c_checker(hip_lib):
    """d = 64 at T = 48 (190 edges, 94 MFMA rule evaluations per sweep), every sweep against oracle/mv_flood.c"""
    model = cx.synth.lgssm_chain(48, d=64, seed=19)
    _per_sweep_parity(model, 6, 1e-7, 1)


@pytest.mark.parametrize("d,T", [(4, 40), (64, 6)])
def test_new_rule_matrices_between_sweeps_refresh_the_cached_leaf_messages(hip_lib, d, T):
    """ADVICE r01 (parameter learning / EM): cx_set_factor_matrices on a live handle.  The messages out of observed
    variables, N(A y, Q), are cached in both Jacobi buffers; after new (A, Q) the handle must converge to the posterior
    of the NEW parameters — the same marginals as a fresh handle built with them."""
    import dataclasses

    model = cx.synth.lgssm_chain(T, d=d, seed=23)
    A1, R1 = model.psets[1]
    R2 = 2.5 * R1
    dev = _dev(model)
    dev.sweep(T + 3)
    dev.set_factor_matrices(1, A1, R2)       # the likelihood's noise changes: every leaf message N(y, R) is stale
    dev.sweep(T + 3)
    fresh = _dev(dataclasses.replace(model, psets={0: model.psets[0], 1: (A1, R2)}))
    fresh.sweep(T + 3)
    got, want = dev.get_marginals(model.x_ids), fresh.get_marginals(model.x_ids)
    assert_close(got, want, 1e-10, "marginals after a parameter change vs a fresh handle")
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], R2)
    assert_close(got[:, :d], em, 1e-8 if d < 64 else 1e-7, "mean vs the exact smoother of the new parameters")


def test_observed_variables_have_no_marginal_for_dim_gt_1(hip_lib):
    """dim > 1 never computes the messages INTO observed variables (nobody reads them: lazy, like the reference), so the
    product of incoming messages — the marginal — of an observed variable stays UndefValue().  (The scalar sweep computes
    every message of the graph and therefore also those marginals.)"""
    model = cx.synth.lgssm_chain(12, d=4, seed=2)
    dev = _dev(model)
    dev.sweep(15)
    assert np.all(np.isnan(dev.get_marginals(model.data_var)))
    assert not np.any(np.isnan(dev.get_marginals(model.x_ids)))


def test_mv64_wave_per_message_form_matches_the_workgroup_form(hip_lib, monkeypatch):
    """The d = 64 rule kernel (csrc/cx_mv64w.hip: one wave per message, matrices resident in registers, upper Cholesky with
    accumulators as MFMA operands; the default since round 2) against round 1's workgroup-per-message form (CX_RULE64=g):
    same messages and marginals to rounding (the factorisation order differs), the same UndefValue() pattern, every sweep."""
    d, T = 64, 9
    model = cx.synth.lgssm_chain(T, d=d, seed=13)
    xe = np.isin(model.edge_var, model.x_ids)
    ev, ef = model.edge_var[xe], model.edge_fac[xe]
    a, b = _dev(model), _dev(model)
    for sweep in range(T + 2):
        monkeypatch.setenv("CX_RULE64", "g")
        a.sweep(1)
        monkeypatch.setenv("CX_RULE64", "w")
        b.sweep(1)
        x = a.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
        y = b.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
        assert np.array_equal(np.isnan(x), np.isnan(y)), f"sweep {sweep}: definedness differs"
        ok = ~np.isnan(x[:, 0])
        if ok.any():
            assert_close(y[ok], x[ok], 1e-9, f"sweep {sweep}: natural-form messages, wave form vs workgroup form")
    monkeypatch.delenv("CX_RULE64", raising=False)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    marg = b.get_marginals(model.x_ids)
    assert_close(marg[:, :d], em, 1e-8, "wave form: marginal mean vs block-tridiagonal solve")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-8, "wave form: marginal covariance vs block-tridiagonal solve")


# ------------------------------------------------------------------------------- degree-4 variables (three sources)

def _multi_sensor_lgssm(n, d, seed, sensors=3, pairs=None):
    """states x_1 .. x_n linked as `pairs` (default: a chain), every state observed by `sensors` independent sensors y = x + N(0, R):
    a chain variable of degree 2 + sensors.  Returns the model and the exact posterior (mean, covariance per state) of the joint solve."""
    rng = np.random.default_rng(seed)
    A = 0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    Q, R = 0.2 * np.eye(d), np.eye(d)
    x = np.arange(1, n + 1, dtype=np.int64)
    if pairs is None:
        pairs = [(i, i + 1) for i in range(n - 1)]
    ys = [x + n * (k + 1) for k in range(sensors)]
    liks = [x + n * (sensors + 1 + k) for k in range(sensors)]
    tr = n * (2 * sensors + 1) + 1 + np.arange(len(pairs), dtype=np.int64)
    par = np.array([x[p] for p, _ in pairs]); chi = np.array([x[c] for _, c in pairs])
    edge_var = np.concatenate(ys + [x] * sensors + [par, chi])
    edge_fac = np.concatenate(liks + liks + [tr, tr])
    role = np.concatenate([np.full(n * sensors, L.ROLE_OUT), np.full(n * sensors, L.ROLE_IN), np.full(len(pairs), L.ROLE_IN), np.full(len(pairs), L.ROLE_OUT)]).astype(np.int32)
    state = np.zeros((n, d)); state[0] = rng.standard_normal(d)
    for p, c in pairs:
        state[c] = A @ state[p] + np.sqrt(0.2) * rng.standard_normal(d)
    data = [state + rng.standard_normal((n, d)) for _ in range(sensors)]
    model = cx.synth.Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=np.concatenate(liks + [tr]),
                           factor_kind=np.full(n * sensors + len(pairs), L.FACTOR_GAUSS_LINEAR, dtype=np.int32),
                           factor_var=np.concatenate([np.ones(n * sensors), np.zeros(len(pairs))]), x_ids=x, data_var=np.concatenate(ys),
                           data_fac=np.concatenate(liks), data_y=np.concatenate(data), dim=d, edge_role=role, psets={0: (A, Q), 1: (np.eye(d), R)},
                           meta={"pairs": pairs})
    Qi, Ri = np.linalg.inv(Q), np.linalg.inv(R)
    J = np.zeros((n * d, n * d)); hvec = np.zeros(n * d)
    for i in range(n):
        for k in range(sensors):
            J[i*d:(i+1)*d, i*d:(i+1)*d] += Ri; hvec[i*d:(i+1)*d] += Ri @ data[k][i]
    for p, c in pairs:
        P, C = slice(p*d, (p+1)*d), slice(c*d, (c+1)*d)
        J[P, P] += A.T @ Qi @ A; J[C, C] += Qi; J[P, C] -= A.T @ Qi; J[C, P] -= Qi @ A
    S = np.linalg.inv(J); mean = S @ hvec
    return model, mean.reshape(n, d), np.stack([S[i*d:(i+1)*d, i*d:(i+1)*d] for i in range(n)])


def _branching_lgssm(n, d, seed, b=2, pairs=None, solve=True):
    """a TREE of states with b children per node (node i has children b i + 1 .. b i + b), every state observed: inner nodes have degree
    b + 2 (parent, children, likelihood); b = 2: degree 4, a message out of them sums THREE incoming ones — the branch a chain never takes.
    pairs: any other tree as (parent, child) index pairs, parents before children.  solve = False: no dense solve (large n)."""
    rng = np.random.default_rng(seed)
    A = 0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    Q, R = 0.2 * np.eye(d), np.eye(d)
    x = np.arange(1, n + 1, dtype=np.int64)
    y, lik = x + n, x + 2 * n
    if pairs is None:
        pairs = [(p, c) for p in range(n) for c in range(b * p + 1, b * p + b + 1) if c < n]
    tr = 3 * n + 1 + np.arange(len(pairs), dtype=np.int64)
    par = np.array([x[p] for p, _ in pairs]); chi = np.array([x[c] for _, c in pairs])
    edge_var = np.concatenate([y, x, par, chi])
    edge_fac = np.concatenate([lik, lik, tr, tr])
    role = np.concatenate([np.full(n, L.ROLE_OUT), np.full(n, L.ROLE_IN), np.full(len(pairs), L.ROLE_IN), np.full(len(pairs), L.ROLE_OUT)]).astype(np.int32)
    state = np.zeros((n, d)); state[0] = rng.standard_normal(d)
    for p, c in pairs:
        state[c] = A @ state[p] + np.sqrt(0.2) * rng.standard_normal(d)
    data = state + rng.standard_normal((n, d))
    model = cx.synth.Model(edge_var=edge_var, edge_fac=edge_fac, factor_ids=np.concatenate([lik, tr]),
                           factor_kind=np.full(n + len(pairs), L.FACTOR_GAUSS_LINEAR, dtype=np.int32),
                           factor_var=np.concatenate([np.ones(n), np.zeros(len(pairs))]), x_ids=x, data_var=y, data_fac=lik, data_y=data,
                           dim=d, edge_role=role, psets={0: (A, Q), 1: (np.eye(d), R)}, meta={"pairs": pairs})
    if not solve:
        return model, None, None
    # exact posterior of the tree: joint information matrix (no prior on the root, like the chain models)
    Qi, Ri = np.linalg.inv(Q), np.linalg.inv(R)
    J = np.zeros((n * d, n * d)); hvec = np.zeros(n * d)
    for i in range(n):
        J[i*d:(i+1)*d, i*d:(i+1)*d] += Ri; hvec[i*d:(i+1)*d] += Ri @ data[i]
    for p, c in pairs:
        P, C = slice(p*d, (p+1)*d), slice(c*d, (c+1)*d)
        J[P, P] += A.T @ Qi @ A; J[C, C] += Qi; J[P, C] -= A.T @ Qi; J[C, P] -= Qi @ A
    S = np.linalg.inv(J); mean = S @ hvec
    return model, mean.reshape(n, d), np.stack([S[i*d:(i+1)*d, i*d:(i+1)*d] for i in range(n)])


@pytest.mark.parametrize("d,form,b", [(3, "", 2), (4, "", 2), (64, "", 2), (64, "g", 2), (64, "", 4), (64, "g", 5)])
def test_mv_tree_with_degree_4_variables(hip_lib, monkeypatch, d, form, b):
    """(b > 2 at dim 64: variables of degree 6 and 7 — more than the three sources a rule sums itself: k_v2f64 sums them first)"""
    if form:
        monkeypatch.setenv("CX_RULE64", form)      # the workgroup-per-message form of the d = 64 rule
    n = 15 if d < 64 else (7 if b == 2 else 11)
    model, emean, ecov = _branching_lgssm(n, d, seed=5, b=b)
    dev = _dev(model)
    o = MvFlood(model)
    g = o.g
    xs_set = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
    pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if int(np.searchsorted(g.var_ids, g.edge_var[e])) in xs_set])
    deg = np.diff(g.var_off)
    assert deg.max() == b + 2
    if d == 64:
        o.sweep(1)        # the d = 64 path evaluates the messages out of observed variables at data injection (see above)
    for sweep in range(10):
        dev.sweep(1)
        o.sweep(1)
        got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
        for row, e in zip(got, pe):
            if o.f2v[e] is None:
                assert np.all(np.isnan(row)), f"sweep {sweep} edge {e}: device defined, restatement undefined"
                continue
            m, S = o.f2v[e]
            if not np.all(np.isfinite(S)) or np.linalg.cond(S) > 1e12:
                continue
            assert_close(row[:d], m, 1e-8, f"sweep {sweep} f2v mean edge {e}")
            assert_close(row[d:].reshape(d, d), S, 1e-8, f"sweep {sweep} f2v covariance edge {e}")
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    assert_close(marg[:, :d], emean, 1e-8, "tree marginal mean vs the joint solve")
    assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-8, "tree marginal covariance vs the joint solve")


@pytest.mark.parametrize("d,b", [(2, 3), (3, 5), (4, 4), (4, 6), (4, 7), (2, 12), (3, 9), (4, 17)])
def test_mv_tree_with_variables_of_degree_up_to_eight(hip_lib, d, b):
    """dim 2..4: more than four edges per variable (the reference's resolver takes any degree, src/dependencies.jl:60-125): the sweep
    keeps up to eight incoming messages in registers (k_sweep_mv<D, 8>); b >= 7 (round 5): inner states of degree b + 2 > 8 live in the
    CSR tail of the slot space and get their messages from k_big_mv.  Per sweep against the numpy restatement, then the joint solve."""
    n = 1 + b + b * b
    model, emean, ecov = _branching_lgssm(n, d, seed=6 + b, b=b)
    dev = _dev(model)
    o = MvFlood(model)
    g = o.g
    assert np.diff(g.var_off).max() == b + 2
    xs_set = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
    pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if int(np.searchsorted(g.var_ids, g.edge_var[e])) in xs_set])
    for sweep in range(6):
        dev.sweep(1)
        o.sweep(1)
        got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
        for row, e in zip(got, pe):
            if o.f2v[e] is None:
                assert np.all(np.isnan(row)), f"sweep {sweep} edge {e}: device defined, restatement undefined"
                continue
            m, S = o.f2v[e]
            if not np.all(np.isfinite(S)) or np.linalg.cond(S) > 1e12:
                continue
            assert_close(row[:d], m, 1e-8, f"sweep {sweep} f2v mean edge {e}")
            assert_close(row[d:].reshape(d, d), S, 1e-8, f"sweep {sweep} f2v covariance edge {e}")
    marg = dev.get_marginals(model.x_ids)
    assert_close(marg[:, :d], emean, 1e-8, "marginal mean vs the joint solve")
    assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-8, "marginal covariance vs the joint solve")
    # variable→factor messages on demand and item by item through the boundary see the same degrees
    vm = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_FACTOR)
    assert not np.all(np.isnan(vm))


# ------------------------------------------------------------------------------- dim 5 .. 63: in 1 x 1, 2 x 2 or 4 x 4 tiles of 16

@pytest.mark.parametrize("d", [5, 6, 16, 20, 32, 33, 63])
def test_dimensions_between_4_and_64_run_embedded_in_the_mfma_path(hip_lib, d):
    """cfg.dim of 5 .. 63 (a constant-velocity model has d = 6, ...): the handle runs on the matrix-core path in the smallest of 16, 32,
    64 that holds d (round 6; the chain-scan schedule: always 64), every rule matrix, message and datum block-diagonal (real block, identity
    block) where d is in between — payloads stay d and d + d*d doubles.  Per sweep against the numpy restatement in dimension d, then the
    block-tridiagonal solve; the chain-scan schedule gives the same marginals in ONE sweep."""
    T = 7
    model = cx.synth.lgssm_chain(T, d=d, seed=30 + d)
    dev = _dev(model)
    # the storage is the tile size's: a message record is eta[nd] | Lambda[nd][nd]
    nd = 16 if d <= 16 else 32 if d <= 32 else 64
    st = dev.stats()
    assert st["device_bytes"] < 3.3 * st["n_slots"] * (nd + nd * nd) * 8 + (2 << 20), (d, nd, st)
    o = MvFlood(model)
    g = o.g
    xs_set = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
    pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if int(np.searchsorted(g.var_ids, g.edge_var[e])) in xs_set])
    o.sweep(1)            # the dim 64 path evaluates the messages out of observed variables at data injection
    for sweep in range(T + 1):
        dev.sweep(1)
        o.sweep(1)
        got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
        assert got.shape == (len(pe), d + d * d)
        for row, e in zip(got, pe):
            if o.f2v[e] is None:
                assert np.all(np.isnan(row)), f"sweep {sweep} edge {e}: device defined, restatement undefined"
                continue
            m, S = o.f2v[e]
            if not np.all(np.isfinite(S)) or np.linalg.cond(S) > 1e10:
                continue
            assert_close(row[:d], m, 1e-7, f"d={d} sweep {sweep} f2v mean edge {e}")
            assert_close(row[d:].reshape(d, d), S, 1e-7, f"d={d} sweep {sweep} f2v covariance edge {e}")
    dev.sweep(2)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    marg = dev.get_marginals(model.x_ids)
    assert marg.shape == (T, d + d * d)
    assert_close(marg[:, :d], em, 1e-8, f"d={d}: marginal mean vs block-tridiagonal solve")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-8, f"d={d}: marginal covariance vs block-tridiagonal solve")
    scan = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, scan)
    scan.sweep(1)
    ms = scan.get_marginals(model.x_ids)
    assert_close(ms[:, :d], em, 1e-8, f"d={d}: chain scan, one sweep: means")
    assert_close(ms[:, d:].reshape(T, d, d), ecov, 1e-8, f"d={d}: chain scan, one sweep: covariances")
    # a message the caller sets in moment form (a prior on the first state through its likelihood edge's partner direction is not
    # needed here: natural-form round trip of a stored message is the identity on the real block)
    e0 = pe[0]
    nat = dev.get_messages([g.edge_var[e0]], [g.edge_fac[e0]], L.TO_VARIABLE, L.FORM_NATURAL)
    dev.set_messages([g.edge_var[e0]], [g.edge_fac[e0]], L.TO_VARIABLE, L.FORM_NATURAL, nat)
    back = dev.get_messages([g.edge_var[e0]], [g.edge_fac[e0]], L.TO_VARIABLE, L.FORM_NATURAL)
    assert np.array_equal(nat, back)


@pytest.mark.parametrize("d,b", [(3, 5), (4, 6), (64, 5), (6, 4)])
def test_product_of_messages_items_for_variables_of_degree_above_five(hip_lib, d, b):
    """the reference's default resolver hangs the marginal of a variable of degree > 5 off a segment tree of ProductOfMessages signals
    (src/dependencies.jl:90-173; inference_signal.jl:62-66): for dim > 1 — degrees up to 8 — cx_update_batch takes those items too and
    cx_get_products reads them back: the product of messages lo..hi of the variable, i.e. the sum of their natural parameters"""
    n = 1 + b + 2
    model, _, _ = _branching_lgssm(n, d, seed=33, b=b, solve=False)
    dev = _dev(model)
    dev.sweep(6)                                       # a few flooding sweeps: every message defined
    root = int(model.x_ids[0])                         # degree b + 1 (children + likelihood)
    ev, ef = model.edge_var, model.edge_fac
    facs = np.sort(ef[ev == root])                     # ascending factor id: the order the ranges are resolved over
    deg = len(facs)
    assert deg == b + 1 >= 5
    ranges = [(1, deg // 2), (deg // 2 + 1, deg), (1, deg), (2, 2)]
    kinds = [L.ITEM_PRODUCT_OF_MESSAGES] * len(ranges)
    dev.update_batch(kinds, [root] * len(ranges), [L.item_range(lo, hi) for lo, hi in ranges])
    nat = dev.get_messages(np.full(deg, root), facs, L.TO_VARIABLE, L.FORM_NATURAL)
    got = dev.get_products([root] * len(ranges), [lo for lo, _ in ranges], [hi for _, hi in ranges], L.FORM_NATURAL)
    for (lo, hi), row in zip(ranges, got):
        assert_close(row, nat[lo - 1:hi].sum(axis=0), 1e-12, f"d={d}: ProductOfMessages {lo}:{hi} == the sum of the natural parameters")
    # moment form, and a node nobody computed reads as UndefValue()
    mom = dev.get_products([root, root], [1, 1], [deg, 1], L.FORM_MOMENT)
    tot = nat.sum(axis=0)
    cov = np.linalg.inv(tot[d:].reshape(d, d))
    assert_close(mom[0, :d], cov @ tot[:d], 1e-9, "moment form: mean")
    assert np.all(np.isnan(mom[1]))
    with pytest.raises(cx.CortexHipError, match="outside 1:"):
        dev.update_batch([L.ITEM_PRODUCT_OF_MESSAGES], [root], [L.item_range(1, deg + 1)])


@pytest.mark.parametrize("d", [8, 16, 24, 32])
def test_native_tiles_equal_the_embedding_in_64(hip_lib, monkeypatch, d):
    """(round 6) the same model in its native tile size (1 x 1 or 2 x 2 tiles of 16) and embedded in 4 x 4 tiles (CX_MFMA_DIM=64, the only
    form until round 5): every message after every sweep and every marginal agree to rounding; the native handle holds (nd + nd^2) /
    (64 + 64^2) of the bytes"""
    T = 12
    model = cx.synth.lgssm_chain(T, d=d, seed=70 + d)
    a = _dev(model)
    monkeypatch.setenv("CX_MFMA_DIM", "64")
    b = _dev(model)
    monkeypatch.delenv("CX_MFMA_DIM")
    nd = 16 if d <= 16 else 32
    sa, sb = a.stats(), b.stats()
    assert sa["n_slots"] == sb["n_slots"] and sa["device_bytes"] < sb["device_bytes"] * 1.35 * (nd + nd * nd) / (64 + 64 * 64) + (2 << 20), (sa, sb)
    xs = set(int(v) for v in model.x_ids)
    keep = np.array([int(v) in xs for v in model.edge_var])
    ev, ef = model.edge_var[keep], model.edge_fac[keep]
    for sweep in range(T + 2):
        a.sweep(1); b.sweep(1)
        ma, mb = a.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), b.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
        assert np.array_equal(np.isnan(ma), np.isnan(mb)), f"sweep {sweep}"
        ok = ~np.isnan(ma)
        assert_close(ma[ok], mb[ok], 1e-9, f"d={d} sweep {sweep}: messages, native tiles vs embedded")
    assert_close(a.get_marginals(model.x_ids), b.get_marginals(model.x_ids), 1e-9, f"d={d}: marginals, native tiles vs embedded")
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    marg = a.get_marginals(model.x_ids)
    assert_close(marg[:, :d], em, 1e-8, f"d={d}: marginal mean vs block-tridiagonal solve")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-8, f"d={d}: marginal covariance vs block-tridiagonal solve")
