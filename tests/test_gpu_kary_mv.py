"""-m gpu: linear-Gaussian factors of three to seven d-dimensional variables (d = 2, 3, 4), x_out = A_1 x_1 + ... + A_k x_k + N(0, Q)
(CX_FACTOR_GAUSS_LINEAR_N for dim > 1; csrc/cx_kary_mv_core.h).  The reference wires every message out of a factor to ALL its other
variables' messages into it (src/dependencies.jl:17-31) and leaves the rule to the user; no d-dimensional rule exists anywhere in
it, so the rule is pinned by mathematics: (i) message by message against the moment-form formulas evaluated in numpy from the same
stored inputs, through cx_update_batch; (ii) the exact posterior of tree-shaped models (dense joint solve) — fused sweeps at their
fixed point, ONE sweep of the tree schedule, and the per-signal plug-in path's items."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


def _kary_tree(n_factors, d, seed, k_choices=(2, 3, 4), observe_out=0.0):
    """a TREE of d-dimensional states grown factor by factor: each factor takes an existing state as one input, new states as its other
    inputs and a new state as its output; every state has a Gaussian prior (a unary factor whose message the caller sets).  Returns
    (model, prior natural parameters, per-factor (out, inputs, set of each input), sets, exact mean, exact covariance)."""
    rng = np.random.default_rng(seed)
    nsets = 3
    sets = {s: (0.8 * np.linalg.qr(rng.standard_normal((d, d)))[0] * rng.uniform(0.5, 1.0), (0.3 + 0.2 * s) * np.eye(d) + 0.05 * np.ones((d, d))) for s in range(nsets)}
    states, facs = [0], []
    for _ in range(n_factors):
        k = int(rng.choice(k_choices))
        ins = [int(rng.choice(states))] + [len(states) + i for i in range(k - 1)]
        out = len(states) + k - 1
        states += list(range(len(states), len(states) + k))
        facs.append((out, ins, [int(rng.integers(0, nsets)) for _ in ins], int(rng.integers(0, nsets))))
    n = len(states)
    x = np.arange(1, n + 1, dtype=np.int64)
    unary = x + n
    fid = 2 * n + 1 + np.arange(len(facs), dtype=np.int64)
    ev, ef, role = list(x), list(unary), [L.ROLE_OUT] * n
    for f, (out, ins, _s, _q) in zip(fid, facs):
        ev.append(x[out]); ef.append(f); role.append(L.ROLE_OUT)
        for i in ins:
            ev.append(x[i]); ef.append(f); role.append(L.ROLE_IN)
    kinds = np.concatenate([np.zeros(n, np.int32), np.full(len(facs), L.FACTOR_GAUSS_LINEAR_N, np.int32)])
    params = np.concatenate([np.zeros(n), np.array([q for *_r, q in facs], dtype=float)])
    model = cx.synth.Model(edge_var=np.array(ev, np.int64), edge_fac=np.array(ef, np.int64), factor_ids=np.concatenate([unary, fid]), factor_kind=kinds,
                           factor_var=params, x_ids=x, dim=d, edge_role=np.array(role, np.int32), psets=sets)
    prior_W = np.stack([np.eye(d) * rng.uniform(0.5, 2.0) + 0.1 * np.ones((d, d)) for _ in range(n)])
    prior_eta = rng.standard_normal((n, d))
    # joint: J = blockdiag(prior W) + sum_f G_f' Q_f^-1 G_f with G_f = [I (out), -A_i (inputs)]
    J = np.zeros((n * d, n * d)); h = np.zeros(n * d)
    for i in range(n):
        J[i*d:(i+1)*d, i*d:(i+1)*d] += prior_W[i]; h[i*d:(i+1)*d] += prior_eta[i]
    for out, ins, ss, q in facs:
        Qi = np.linalg.inv(sets[q][1])
        blocks = [(out, np.eye(d))] + [(i, -sets[s][0]) for i, s in zip(ins, ss)]
        for a, Ga in blocks:
            for b, Gb in blocks:
                J[a*d:(a+1)*d, b*d:(b+1)*d] += Ga.T @ Qi @ Gb
    S = np.linalg.inv(J)
    mean = (S @ h).reshape(n, d)
    cov = np.stack([S[i*d:(i+1)*d, i*d:(i+1)*d] for i in range(n)])
    return model, (prior_eta, prior_W), facs, fid, sets, mean, cov


def _load(model, prior, facs, fid, sets, schedule, seed_variance=None):
    d = model.dim
    dev = cx.DeviceGraph(dim=d, schedule=schedule)
    cx.synth.load_into_device(model, dev)
    n = len(model.x_ids)
    eta, W = prior
    dev.set_messages(model.x_ids, model.x_ids + n, L.TO_VARIABLE, L.FORM_NATURAL, np.concatenate([eta, W.reshape(n, d * d)], axis=1))
    ev = [int(model.x_ids[i]) for (_o, ins, _s, _q) in facs for i in ins]
    ef = [int(f) for f, (_o, ins, _s, _q) in zip(fid, facs) for _ in ins]
    es = [s for (_o, _ins, ss, _q) in facs for s in ss]
    dev.set_factor_edge_sets(ev, ef, es)
    if seed_variance is not None:
        dev.seed_messages(L.TO_VARIABLE, 0.0, seed_variance)
    return dev


@pytest.mark.parametrize("d,n_factors,seed", [(2, 1, 1), (3, 6, 2), (4, 12, 3), (4, 40, 4), (2, 25, 5)])
def test_tree_schedule_one_sweep_is_the_exact_posterior(hip_lib, d, n_factors, seed):
    model, prior, facs, fid, sets, mean, cov = _kary_tree(n_factors, d, seed)
    dev = _load(model, prior, facs, fid, sets, L.SCHED_TREE)
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    n = len(model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], mean, 1e-8, "marginal mean vs the joint solve", scale_by="max")
    assert_close(marg[:, d:].reshape(n, d, d), cov, 1e-8, "marginal covariance vs the joint solve", scale_by="max")
    assert dev.tree_plan_stats()["kary_entries"] > 0
    dev.close()


@pytest.mark.parametrize("d,n_factors,seed", [(2, 3, 11), (4, 10, 12), (3, 20, 13)])
def test_fused_sweeps_reach_the_exact_posterior_and_damping_keeps_it(hip_lib, d, n_factors, seed):
    model, prior, facs, fid, sets, mean, cov = _kary_tree(n_factors, d, seed, k_choices=(2, 3, 5, 6))
    n = len(model.x_ids)
    for lam in (0.0, 0.3):
        dev = _load(model, prior, facs, fid, sets, L.SCHED_FUSED, seed_variance=1e6)
        dev.set_damping(lam)
        dev.sweep(4 * n_factors + 40 if lam == 0.0 else 12 * n_factors + 200)
        marg = dev.get_marginals(model.x_ids)
        assert_close(marg[:, :d], mean, 1e-8, f"damping {lam}: marginal mean vs the joint solve", scale_by="max")
        assert_close(marg[:, d:].reshape(n, d, d), cov, 1e-8, f"damping {lam}: marginal covariance vs the joint solve", scale_by="max")
        dev.close()


def _moment(nat, d):
    W = nat[d:].reshape(d, d)
    V = np.linalg.inv(W)
    return V @ nat[:d], V


def test_messages_one_by_one_against_the_moment_form_formulas(hip_lib):
    """cx_update_batch MessageToVariable items on the edges of a 4-input factor, inputs set by hand (one of them an observed datum):
    each message against the formulas at the top of csrc/cx_kary_mv_core.h evaluated in numpy"""
    d = 3
    rng = np.random.default_rng(8)
    sets = {0: (rng.standard_normal((d, d)), 0.4 * np.eye(d) + 0.1), 1: (rng.standard_normal((d, d)), np.eye(d)), 2: (0.5 * np.eye(d), np.eye(d))}
    x = np.arange(1, 6, dtype=np.int64)            # 1 = out, 2..5 inputs
    f = 6
    model = cx.synth.Model(edge_var=x, edge_fac=np.full(5, f, np.int64), factor_ids=np.array([f]), factor_kind=np.array([L.FACTOR_GAUSS_LINEAR_N], np.int32),
                           factor_var=np.array([0.0]), x_ids=x, dim=d, edge_role=np.array([L.ROLE_OUT] + [L.ROLE_IN] * 4, np.int32), psets=sets)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev)
    edge_set = [1, 2, 1, 0]
    dev.set_factor_edge_sets(x[1:], np.full(4, f), edge_set)
    msgs = {}
    for v in x[:4]:
        W = np.eye(d) * rng.uniform(0.5, 2) + 0.2 * np.ones((d, d))
        m = rng.standard_normal(d)
        msgs[int(v)] = (m, np.linalg.inv(W))
        dev.set_messages([v], [f], L.TO_FACTOR, L.FORM_MOMENT, np.concatenate([m, np.linalg.inv(W).reshape(-1)])[None, :])
    y = rng.standard_normal(d)
    msgs[5] = (y, np.zeros((d, d)))
    dev.set_messages([5], [f], L.TO_FACTOR, L.FORM_POINT, y[None, :])
    dev.update_batch([L.ITEM_MESSAGE_TO_VARIABLE] * 5, x, [f] * 5)
    got = dev.get_messages(x, np.full(5, f), L.TO_VARIABLE, L.FORM_NATURAL)
    A = {int(v): sets[s][0] for v, s in zip(x[1:], edge_set)}
    Q = sets[0][1]
    want_out = (sum(A[i] @ msgs[i][0] for i in A), Q + sum(A[i] @ msgs[i][1] @ A[i].T for i in A))
    Wo = np.linalg.inv(want_out[1])
    assert_close(got[0, :d], Wo @ want_out[0], 1e-10, "to x_out: eta", scale_by="max"); assert_close(got[0, d:].reshape(d, d), Wo, 1e-10, "to x_out: Lambda", scale_by="max")
    for row, j in zip(got[1:4], (2, 3, 4)):
        mu = msgs[1][0] - sum(A[i] @ msgs[i][0] for i in A if i != j)
        S = msgs[1][1] + Q + sum(A[i] @ msgs[i][1] @ A[i].T for i in A if i != j)
        Si = np.linalg.inv(S)
        assert_close(row[:d], A[j].T @ Si @ mu, 1e-10, f"to x_{j}: eta", scale_by="max")
        assert_close(row[d:].reshape(d, d), A[j].T @ Si @ A[j], 1e-10, f"to x_{j}: Lambda", scale_by="max")
    dev.close()


def test_refusals_and_new_matrices(hip_lib):
    d = 2
    model, prior, facs, fid, sets, mean, cov = _kary_tree(4, d, seed=21)
    for dim, sched in ((64, L.SCHED_FUSED), (d, L.SCHED_CHAIN_SCAN)):
        dev = cx.DeviceGraph(dim=dim, schedule=sched)
        with pytest.raises(cx.CortexHipError) as ei:
            big = model if dim == d else None
            if big is None:
                import copy
                big = copy.copy(model); big.dim = 64; big.psets = {k: (np.eye(64), np.eye(64)) for k in sets}
            cx.synth.load_into_device(big, dev)
        assert ei.value.code == L.ERR_UNSUPPORTED
        dev.close()
    dev = _load(model, prior, facs, fid, sets, L.SCHED_TREE)
    dev.sweep(1)
    before = dev.get_marginals(model.x_ids)
    A0, Q0 = sets[0]
    dev.set_factor_matrices(0, 0.5 * A0, 2.0 * Q0)          # new matrices under a standing plan: the next sweep is under them
    dev.sweep(1)
    after = dev.get_marginals(model.x_ids)
    assert np.max(np.abs(after - before)) > 1e-6
    dev.set_factor_matrices(0, A0, Q0)
    dev.sweep(1)
    assert_close(dev.get_marginals(model.x_ids), before, 1e-12, "back under the first matrices", scale_by="max")
    with pytest.raises(cx.CortexHipError):
        dev.set_factor_edge_sets([int(model.x_ids[facs[0][0]])], [int(fid[0])], [0])      # the OUT edge has no A
    dev.close()
