"""oracle/blocktri.c (the exact smoother in C, for full-size chains) pinned against oracle/exact.py's numpy statement."""
import time

import numpy as np
import pytest

from oracle import exact


@pytest.mark.parametrize("d,T", [(1, 1), (1, 50), (2, 2), (3, 17), (4, 400), (8, 30)])
def test_c_block_tridiagonal_solver_matches_numpy(d, T):
    rng = np.random.default_rng(d * 100 + T)
    A = 0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0] + 0.05 * rng.standard_normal((d, d))
    Gq = rng.standard_normal((d, d)); Q = 0.1 * (Gq @ Gq.T) + 0.05 * np.eye(d)
    Gr = rng.standard_normal((d, d)); R = Gr @ Gr.T + 0.5 * np.eye(d)
    y = rng.standard_normal((T, d))
    m0, c0 = exact.lgssm_posterior(y, A, Q, R)
    m1, c1 = exact.lgssm_posterior_c(y, A, Q, R)
    assert np.max(np.abs(m1 - m0)) <= 1e-11 * max(1.0, np.max(np.abs(m0)))
    assert np.max(np.abs(c1 - c0)) <= 1e-11 * np.max(np.abs(c0))


def test_c_solver_slow_mixing_model_and_speed():
    """the model the d-dimensional chain scan is tested on: A = 0.999 I, Q = 1e-4 I, R = 10 I (information travels ~1000 steps)"""
    d, T = 4, 3000
    rng = np.random.default_rng(5)
    A, Q, R = 0.999 * np.eye(d), 1e-4 * np.eye(d), 10.0 * np.eye(d)
    y = rng.standard_normal((T, d))
    m0, c0 = exact.lgssm_posterior(y, A, Q, R)
    t0 = time.time()
    m1, c1 = exact.lgssm_posterior_c(y, A, Q, R)
    assert time.time() - t0 < 1.0
    assert np.max(np.abs(m1 - m0)) <= 1e-9 * np.max(np.abs(m0))
    assert np.max(np.abs(c1 - c0)) <= 1e-9 * np.max(np.abs(c0))


def test_longdouble_smoother_agrees_with_the_f64_ones_and_is_the_more_accurate():
    """oracle/exact.py:lgssm_posterior_longdouble — the reference of the conditioning tests.  On a benign model all three agree to
    1e-13; on cond(Q) = 1e6 the two f64 solvers differ from EACH OTHER by ~1e-9 and each from the long-double one by about as much."""
    d, T = 4, 60
    rng = np.random.default_rng(3)
    y = rng.standard_normal((T, d))
    A = 0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    m0, c0 = exact.lgssm_posterior(y, A, 0.1 * np.eye(d), np.eye(d))
    m2, c2 = exact.lgssm_posterior_longdouble(y, A, 0.1 * np.eye(d), np.eye(d))
    assert np.max(np.abs(m2 - m0)) <= 1e-13 * np.max(np.abs(m0)) and np.max(np.abs(c2 - c0)) <= 1e-13 * np.max(np.abs(c0))
    U = np.linalg.qr(rng.standard_normal((d, d)))[0]
    Q = U @ np.diag(np.logspace(-6, 0, d)) @ U.T; Q = 0.5 * (Q + Q.T)
    ms = [f(y, 0.99 / 0.9 * A, Q, np.eye(d))[0] for f in (exact.lgssm_posterior, exact.lgssm_posterior_c, exact.lgssm_posterior_longdouble)]
    e01, e02, e12 = (np.max(np.abs(ms[i] - ms[j])) / np.max(np.abs(ms[2])) for i, j in ((0, 1), (0, 2), (1, 2)))
    assert e02 < 1e-6 and e12 < 1e-6 and e01 < 1e-6
    assert np.finfo(np.longdouble).eps < 1e-18        # x86 extended precision is what this container and the GPU box have
