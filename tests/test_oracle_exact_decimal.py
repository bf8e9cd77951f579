"""the 50-digit chain solve of oracle/exact.py (the checker of the chain scan where the float64 solve is not accurate enough) against the
float64 solve on a benign chain, and against a dense solve of a small ill-scaled one"""
import numpy as np

from oracle import exact


def test_decimal_chain_solve_equals_the_float_solve_on_a_benign_chain():
    rng = np.random.default_rng(0)
    T = 400
    y, r, q = rng.standard_normal(T), rng.uniform(0.5, 2, T), rng.uniform(0.5, 2, T - 1)
    a, b = exact.ssm_chain_posterior(y, r, q), exact.ssm_chain_posterior_decimal(y, r, q)
    np.testing.assert_allclose(b[0], a[0], rtol=0, atol=1e-13)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-13)


def test_decimal_chain_solve_on_a_small_ill_scaled_chain():
    # exact rational arithmetic as the arbiter: T = 6, variances over twelve decades
    from fractions import Fraction

    y = [1.0, -2.0, 0.5, 3.0, 0.25, -1.5]
    r = [1e-6, 1e6, 1.0, 1e3, 1e-3, 1e6]
    q = [1e6, 1e-6, 1e3, 1.0, 1e-6]
    T = len(y)
    A = [[Fraction(0)] * T for _ in range(T)]
    for i in range(T):
        A[i][i] += 1 / Fraction(r[i])
    for i in range(T - 1):
        w = 1 / Fraction(q[i])
        A[i][i] += w; A[i + 1][i + 1] += w; A[i][i + 1] -= w; A[i + 1][i] -= w
    # Gauss-Jordan on [A | I | b]
    M = [row[:] + [Fraction(int(i == j)) for j in range(T)] + [Fraction(y[i]) / Fraction(r[i])] for i, row in enumerate(A)]
    for c in range(T):
        p = M[c][c]
        M[c] = [v / p for v in M[c]]
        for i in range(T):
            if i != c and M[i][c] != 0:
                f = M[i][c]
                M[i] = [a - f * b for a, b in zip(M[i], M[c])]
    mean = np.array([float(M[i][2 * T]) for i in range(T)])
    var = np.array([float(M[i][T + i]) for i in range(T)])
    got = exact.ssm_chain_posterior_decimal(y, r, q)
    np.testing.assert_allclose(got[0], mean, rtol=1e-14)
    np.testing.assert_allclose(got[1], var, rtol=1e-14)
