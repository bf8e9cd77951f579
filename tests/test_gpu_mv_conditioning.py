"""-m gpu: d-dimensional kernels on ill-conditioned models.

Both small-d and d = 64 rules factor M = Lambda + P with an UN-PIVOTED Cholesky on v_rsq_f64 + a correction step
(csrc/cx_mv_core.h: chol, csrc/cx_mv64w_core.h).  Round 2 tested them on well-conditioned models only (A = 0.95 * orthogonal,
Q = 0.1 I, R = I).  Here: cond(Q) = 1e6 and |A| = 0.99, against the exact smoother in x86 extended precision
(oracle/exact.py:lgssm_posterior_longdouble) — two f64 solvers of these systems already disagree at ~cond(Q) * 1e-15, so an f64
oracle cannot tell a device error from its own.  What is asserted is what was measured (tests/lab_conditioning.py):
the device is as accurate as numpy's pivoted f64 solve of the same posterior; and non-PD inputs leave the affected results
UNDEFINED or unchanged — never a half-written or NaN-poisoned message that spreads."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact

pytestmark = pytest.mark.gpu


def _model(d, T, cond_q, rho, seed):
    rng = np.random.default_rng(seed)
    U = np.linalg.qr(rng.standard_normal((d, d)))[0]
    Q = U @ np.diag(np.logspace(-np.log10(cond_q), 0, d)) @ U.T
    A = rho * np.linalg.qr(rng.standard_normal((d, d)))[0]
    return cx.synth.lgssm_chain(T, d=d, seed=seed, A=A, Q=0.5 * (Q + Q.T), R=np.eye(d))


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


# measured on MI355X (device error | numpy f64 error, both against the long-double smoother), cond(Q) = 1e6, |A| = 0.99:
#   d = 4  chain scan 6.0e-9 | 4.4e-9     d = 4 flooding 6.1e-9 | 4.4e-9     d = 64 flooding 4.1e-7 | 5.5e-7
@pytest.mark.parametrize("d,T,schedule,bound", [(4, 400, L.SCHED_CHAIN_SCAN, 1e-7), (4, 400, L.SCHED_FUSED, 1e-7), (2, 300, L.SCHED_CHAIN_SCAN, 1e-8),
                                                (64, 14, L.SCHED_FUSED, 2e-6)])
def test_cond_1e6_model_device_is_as_accurate_as_a_pivoted_f64_solve(hip_lib, d, T, schedule, bound):
    model = _model(d, T, 1e6, 0.99, seed=11)
    A, Q, R = model.meta["A"], model.meta["Q"], model.meta["R"]
    assert np.linalg.cond(Q) > 9e5
    ref_m, ref_c = exact.lgssm_posterior_longdouble(model.data_y, A, Q, R)
    f64_m, f64_c = exact.lgssm_posterior(model.data_y, A, Q, R)
    dev = cx.DeviceGraph(dim=d, schedule=schedule)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1 if schedule == L.SCHED_CHAIN_SCAN else T + 3)
    g = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(g))
    dm, dc = _rel(g[:, :d], ref_m), _rel(g[:, d:].reshape(T, d, d), ref_c)
    fm, fc = _rel(f64_m, ref_m), _rel(f64_c, ref_c)
    # the 1e-6 target of BASELINE.json holds for d <= 4 with two digits to spare; d = 64 sits at 4e-7
    assert dm <= bound and dc <= bound, (dm, dc)
    # ... and the error is the problem's, not the un-pivoted factorisation's: within a small factor of LAPACK's on the same system
    assert dm <= 6 * fm + 1e-13 and dc <= 6 * fc + 1e-13, (dm, fm, dc, fc)


@pytest.mark.parametrize("d,T", [(4, 24), (64, 6)])
def test_non_positive_definite_input_leaves_results_undefined_or_unchanged(hip_lib, d, T):
    """A message whose precision is not positive definite (a user error, or a diverged loopy run) makes M = Lambda + P of the rules
    downstream indefinite: the Cholesky produces NaN and the result is NOT stored (the message keeps its previous value; the one
    marginal that multiplies the bad message in reads UndefValue()).  Nothing else moves: every other message and marginal is
    bitwise what it was, and no message is ever half-defined."""
    model = cx.synth.lgssm_chain(T, d=d, seed=29)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev)
    dev.sweep(T + 3)                                   # converged (a tree)
    xe = np.isin(model.edge_var, model.x_ids)
    ev, ef = model.edge_var[xe], model.edge_fac[xe]
    before = dev.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    marg_before = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(before)) and not np.any(np.isnan(marg_before))
    # the likelihood message into the middle state becomes N^-1(0, -1e6 I): wildly indefinite
    t = T // 2
    bad = np.concatenate([np.zeros(d), (-1e6 * np.eye(d)).ravel()])
    lik_t = int(model.data_fac[t])
    dev.set_messages([model.x_ids[t]], [lik_t], L.TO_VARIABLE, L.FORM_NATURAL, bad)
    dev.sweep(3)
    after = dev.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    marg_after = dev.get_marginals(model.x_ids)
    # no half-written message: a row is all-NaN or all-finite
    row_nan = np.isnan(after)
    assert np.all(row_nan.all(axis=1) | ~row_nan.any(axis=1))
    changed = ~np.all((after == before) | (np.isnan(after) & np.isnan(before)), axis=1)
    bad_row = (ev == model.x_ids[t]) & (ef == lik_t)
    assert changed[bad_row].all() and not changed[~bad_row].any(), "a message other than the one the caller overwrote moved"
    # the marginal of x_t is undefined; all others are untouched
    others = np.arange(T) != t
    assert np.all(np.isnan(marg_after[t])) and np.array_equal(marg_after[others], marg_before[others])
    # repairing the input repairs the state
    good = np.concatenate([np.linalg.solve(model.meta["R"], model.data_y[t]), np.linalg.inv(model.meta["R"]).ravel()])
    dev.set_messages([model.x_ids[t]], [lik_t], L.TO_VARIABLE, L.FORM_NATURAL, good)
    dev.sweep(T + 3)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    g = dev.get_marginals(model.x_ids)
    assert _rel(g[:, :d], em) < 1e-8 and _rel(g[:, d:].reshape(T, d, d), ecov) < 1e-8
