"""oracle/exact.py — exact Gaussian posteriors by direct linear algebra.  TEST INFRASTRUCTURE ONLY.

The reference pins nothing numeric for Gaussian belief propagation (its SSM test asserts only
signs/monotonicity, test/inference_engine_tests.jl:485-487), so the mathematics does: on a
tree, sum-product marginals equal the marginals of the joint Gaussian whatever the schedule;
on a loopy Gaussian graph the converged BP *means* are exact (variances are not).
These solvers share no code and no formulation with the message-passing paths they check.
"""
from __future__ import annotations

import numpy as np


def tridiag_posterior(diag, off, h):
    """Mean and marginal variances of N^{-1}(h, J) with J symmetric tridiagonal
    (diag[T], off[T-1] = J[t, t+1]).  Schur-complement recursions from both ends:
        L_t = J_tt - off_{t-1}^2 / L_{t-1},   R_t = J_tt - off_t^2 / R_{t+1},
        var_t = 1 / (L_t + R_t - J_tt),  mean by the matching forward/backward substitutions."""
    diag = np.asarray(diag, dtype=np.float64)
    off = np.asarray(off, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    T = len(diag)
    L = np.empty(T); R = np.empty(T); hl = np.empty(T); hr = np.empty(T)
    L[0] = diag[0]; hl[0] = h[0]
    for t in range(1, T):
        g = off[t - 1] / L[t - 1]
        L[t] = diag[t] - g * off[t - 1]
        hl[t] = h[t] - g * hl[t - 1]
    R[T - 1] = diag[T - 1]; hr[T - 1] = h[T - 1]
    for t in range(T - 2, -1, -1):
        g = off[t] / R[t + 1]
        R[t] = diag[t] - g * off[t]
        hr[t] = h[t] - g * hr[t + 1]
    prec = L + R - diag
    var = 1.0 / prec
    mean = (hl + hr - h) * var
    return mean, var


def ssm_chain_posterior(y, r, q):
    """Posterior of the reference's SSM (test/inference_engine_tests.jl:436-453), generalised to
    per-step variances: y_t ~ N(x_t, r_t), x_{t+1} ~ N(x_t, q_t); no prior on x_1."""
    y = np.asarray(y, dtype=np.float64)
    T = len(y)
    r = np.broadcast_to(np.asarray(r, dtype=np.float64), (T,))
    q = np.broadcast_to(np.asarray(q, dtype=np.float64), (max(T - 1, 0),))
    diag = 1.0 / r
    diag = diag.copy()
    diag[:-1] += 1.0 / q
    diag[1:] += 1.0 / q
    off = -1.0 / q
    return tridiag_posterior(diag, off, y / r)


def ssm_chain_posterior_decimal(y, r, q, digits=50):
    """ssm_chain_posterior in `digits`-digit decimal arithmetic: the float64 solve above loses what 1 / r + 1 / q loses when the variances
    span many decades (1.9e-6 of the mean at twelve decades and T = 70,001; tools/lab/c2_dynamic_range.py).  Mean by elimination, variances
    from the two one-sided recursions (the diagonal of the inverse of a tridiagonal matrix)."""
    from decimal import Decimal, getcontext

    getcontext().prec = digits
    T = len(y)
    one = Decimal(1)
    rr = [Decimal(float(x)) for x in np.broadcast_to(np.asarray(r, dtype=np.float64), (T,))]
    qq = [Decimal(float(x)) for x in np.broadcast_to(np.asarray(q, dtype=np.float64), (max(T - 1, 0),))]
    diag = [one / x for x in rr]
    for i in range(T - 1):
        w = one / qq[i]
        diag[i] += w
        diag[i + 1] += w
    off = [-one / x for x in qq]
    b = [Decimal(float(y[i])) / rr[i] for i in range(T)]
    d = diag[:]
    for i in range(1, T):
        m = off[i - 1] / d[i - 1]
        d[i] -= m * off[i - 1]
        b[i] -= m * b[i - 1]
    x = [Decimal(0)] * T
    x[-1] = b[-1] / d[-1]
    for i in range(T - 2, -1, -1):
        x[i] = (b[i] - off[i] * x[i + 1]) / d[i]
    e = diag[:]
    for i in range(T - 2, -1, -1):
        e[i] -= off[i] * off[i] / e[i + 1]
    return np.array([float(v) for v in x]), np.array([float(one / (d[i] + e[i] - diag[i])) for i in range(T)])


def grid_precision(n_rows, n_cols, r, qh, qv):
    """Sparse precision of the grid model: unary y_i ~ N(x_i, r_i) and pairwise
    difference factors x_i - x_j ~ N(0, q_ij).  qh[i, j] couples (i, j)-(i, j+1); qv[i, j]
    couples (i, j)-(i+1, j).  Variable index = i * n_cols + j."""
    import scipy.sparse as sp

    r = np.asarray(r, dtype=np.float64).reshape(n_rows, n_cols)
    idx = np.arange(n_rows * n_cols).reshape(n_rows, n_cols)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [(1.0 / r).ravel()]

    def couple(a, b, q):
        w = 1.0 / np.asarray(q, dtype=np.float64).ravel()
        a, b = a.ravel(), b.ravel()
        rows.extend([a, b, a, b]); cols.extend([a, b, b, a]); vals.extend([w, w, -w, -w])

    if n_cols > 1:
        couple(idx[:, :-1], idx[:, 1:], np.asarray(qh).reshape(n_rows, n_cols - 1))
    if n_rows > 1:
        couple(idx[:-1, :], idx[1:, :], np.asarray(qv).reshape(n_rows - 1, n_cols))
    J = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(n_rows * n_cols,) * 2).tocsc()
    return J


def grid_posterior_mean(n_rows, n_cols, y, r, qh, qv):
    import scipy.sparse.linalg as spla

    J = grid_precision(n_rows, n_cols, r, qh, qv)
    h = (np.asarray(y, dtype=np.float64).ravel() / np.asarray(r, dtype=np.float64).ravel())
    return spla.spsolve(J, h)


def block_tridiag_posterior(Jd, Jo, h):
    """Block-tridiagonal analogue of tridiag_posterior.  Jd[T, d, d] diagonal blocks, Jo[T-1, d, d]
    = J[t, t+1] blocks, h[T, d].  Returns means [T, d] and marginal covariances [T, d, d]."""
    Jd = np.asarray(Jd, dtype=np.float64); Jo = np.asarray(Jo, dtype=np.float64); h = np.asarray(h, dtype=np.float64)
    T, d = h.shape
    L = np.empty_like(Jd); R = np.empty_like(Jd); hl = np.empty_like(h); hr = np.empty_like(h)
    L[0] = Jd[0]; hl[0] = h[0]
    for t in range(1, T):
        G = np.linalg.solve(L[t - 1], Jo[t - 1]).T  # Jo^T L^{-1}
        L[t] = Jd[t] - G @ Jo[t - 1]
        hl[t] = h[t] - G @ hl[t - 1]
    R[T - 1] = Jd[T - 1]; hr[T - 1] = h[T - 1]
    for t in range(T - 2, -1, -1):
        G = np.linalg.solve(R[t + 1].T, Jo[t].T).T  # Jo R^{-1}
        R[t] = Jd[t] - G @ Jo[t].T
        hr[t] = h[t] - G @ hr[t + 1]
    P = L + R - Jd
    cov = np.linalg.inv(P)
    mean = np.einsum("tij,tj->ti", cov, hl + hr - h)
    return mean, cov


def lgssm_posterior(y, A, Q, R, H=None):
    """Posterior of x_{t+1} = A x_t + w, w~N(0,Q);  y_t = H x_t + v, v~N(0,R); no prior on x_1
    (SURVEY §8d C3/C5 shapes).  y[T, m]."""
    y = np.asarray(y, dtype=np.float64)
    T = y.shape[0]
    d = A.shape[0]
    H = np.eye(d) if H is None else H
    Qi = np.linalg.inv(Q); Ri = np.linalg.inv(R)
    obs = H.T @ Ri @ H
    Jd = np.tile(obs, (T, 1, 1))
    Jd[:-1] += A.T @ Qi @ A
    Jd[1:] += Qi
    Jo = np.tile(-(A.T @ Qi), (T - 1, 1, 1))
    h = y @ (Ri @ H)  # rows: H^T R^{-1} y_t
    return block_tridiag_posterior(Jd, Jo, h)


def lgssm_posterior_c(y, A, Q, R):
    """lgssm_posterior (H = I) by oracle/blocktri.c: the same block-tridiagonal statement with pivoted LU solves in C, for the
    full-size chains (T = 1e6) where the numpy loop above would take minutes.  Pinned against it in tests/test_blocktri_checker.py."""
    import ctypes as C

    from .ref import lib

    y = np.ascontiguousarray(y, dtype=np.float64)
    T, d = y.shape
    A = np.ascontiguousarray(A, dtype=np.float64); Q = np.ascontiguousarray(Q, dtype=np.float64); R = np.ascontiguousarray(R, dtype=np.float64)
    mean = np.empty((T, d)); cov = np.empty((T, d, d))
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib().cxo_lgssm_posterior(d, T, p(y), p(A), p(Q), p(R), p(mean), p(cov))
    if rc != 0:
        raise RuntimeError(f"cxo_lgssm_posterior: status {rc}")
    return mean, cov


# ---- extended precision (x86 long double, 64-bit mantissa): the reference for ill-conditioned models, where two f64 solvers of
# ---- the same system already disagree at cond(Q) * 1e-16 -----------------------------------------------------------------------
def _ld_solve(M, B):
    """M X = B by Gaussian elimination with partial pivoting in np.longdouble (numpy.linalg has no long double path)"""
    M = np.array(M, dtype=np.longdouble); B = np.array(B, dtype=np.longdouble)
    d = M.shape[0]
    for c in range(d):
        p = c + int(np.argmax(np.abs(M[c:, c])))
        if p != c:
            M[[c, p]] = M[[p, c]]; B[[c, p]] = B[[p, c]]
        f = M[c + 1:, c] / M[c, c]
        M[c + 1:] -= f[:, None] * M[c]
        B[c + 1:] -= f[:, None] * B[c]
    for r in range(d - 1, -1, -1):
        B[r] = (B[r] - M[r, r + 1:] @ B[r + 1:]) / M[r, r]
    return B


def lgssm_posterior_longdouble(y, A, Q, R):
    """lgssm_posterior (H = I) with every operation in np.longdouble: the same Schur-complement recursions from both ends"""
    ld = np.longdouble
    y = np.asarray(y, dtype=ld); A = np.asarray(A, dtype=ld); Q = np.asarray(Q, dtype=ld); R = np.asarray(R, dtype=ld)
    T, d = y.shape
    I = np.eye(d, dtype=ld)
    Qi, Ri = _ld_solve(Q, I), _ld_solve(R, I)
    AtQi = A.T @ Qi
    AtQiA = AtQi @ A
    Jo = -AtQi
    Jd = [Ri + (AtQiA if t < T - 1 else 0) + (Qi if t > 0 else 0) for t in range(T)]
    h = [Ri @ y[t] for t in range(T)]
    L, hl = [Jd[0]], [h[0]]
    for t in range(1, T):
        W = _ld_solve(L[t - 1], np.concatenate([Jo, hl[t - 1][:, None]], axis=1))      # L^-1 [Jo | hl]
        L.append(Jd[t] - Jo.T @ W[:, :d]); hl.append(h[t] - Jo.T @ W[:, d])
    Rr, hr = [None] * T, [None] * T
    Rr[T - 1], hr[T - 1] = Jd[T - 1], h[T - 1]
    for t in range(T - 2, -1, -1):
        W = _ld_solve(Rr[t + 1], np.concatenate([Jo.T, hr[t + 1][:, None]], axis=1))
        Rr[t] = Jd[t] - Jo @ W[:, :d]; hr[t] = h[t] - Jo @ W[:, d]
    mean, cov = np.empty((T, d)), np.empty((T, d, d))
    for t in range(T):
        P = L[t] + Rr[t] - Jd[t]
        X = _ld_solve(P, np.concatenate([I, (hl[t] + hr[t] - h[t])[:, None]], axis=1))
        cov[t] = X[:, :d].astype(np.float64); mean[t] = X[:, d].astype(np.float64)
    return mean, cov


def kary_posterior_sparse(model, sample_ids=None):
    """Exact posterior of a synth.kary_model / synth.tree_model (linear-Gaussian factors x_out = sum_i a_i x_i + b + N(0, q) of any
    arity, unary priors, point-mass data) at sizes a dense inverse does not reach: the joint precision J = sum of prior precisions +
    sum_f c_f c_f' / q_f assembled sparse (observed variables conditioned on their data), one sparse LU, the mean of every free
    variable and the variance of `sample_ids` (default: all) from one solve per sampled unit vector.  Returns (ids, mean, variance)
    over the sample.  Same mathematics as tests/kary_support.dense_posterior, which pins it on small models."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import splu

    meta = model.meta
    used = np.asarray(meta["used"], dtype=np.int64)
    obs = {int(v): float(y) for v, y in zip(model.data_var, model.data_y)}
    is_obs = np.zeros(int(used.max()) + 2, dtype=bool)
    if obs:
        is_obs[np.fromiter(obs.keys(), dtype=np.int64)] = True
    free = used[~is_obs[used]]
    pos = np.full(int(used.max()) + 2, -1, dtype=np.int64)
    pos[free] = np.arange(len(free))
    n = len(free)
    rows, cols, vals = [pos[np.asarray(model.prior_var)]], [pos[np.asarray(model.prior_var)]], [1.0 / np.asarray(model.prior_variance)]
    h = np.zeros(n)
    np.add.at(h, pos[np.asarray(model.prior_var)], np.asarray(model.prior_mean) / np.asarray(model.prior_variance))
    cv, cf, ca = (meta["all_coef_var"], meta["all_coef_fac"], meta["all_coef"]) if "all_coef" in meta else (meta["coef_var"], meta["coef_fac"], meta["coef"])
    coef = {(int(v), int(f)): float(a) for v, f, a in zip(cv, cf, ca)}
    r_, c_, v_ = [], [], []
    for fi, fid in enumerate(meta["kary_ids"]):
        vs = meta["fac_vars"][fi]
        out = int(meta["out_var"][fi])
        c = {v: (1.0 if v == out else -coef[(v, int(fid))]) for v in vs}
        b, q = float(meta["b"][fi]), float(meta["q"][fi])
        rhs = b - sum(c[v] * obs[v] for v in vs if v in obs)
        fv = [v for v in vs if v not in obs]
        for a in fv:
            h[pos[a]] += c[a] * rhs / q
            for bb in fv:
                r_.append(pos[a]); c_.append(pos[bb]); v_.append(c[a] * c[bb] / q)
    rows.append(np.asarray(r_, dtype=np.int64)); cols.append(np.asarray(c_, dtype=np.int64)); vals.append(np.asarray(v_))
    J = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsc()
    lu = splu(J)
    mean = lu.solve(h)
    ids = free if sample_ids is None else np.asarray(sample_ids, dtype=np.int64)
    p = pos[ids]
    if np.any(p < 0):
        raise ValueError("a sampled variable is observed or does not occur in the model")
    var = np.empty(len(ids))
    for lo in range(0, len(ids), 256):                   # unit vectors in blocks: n x 256 doubles at a time
        blk = p[lo:lo + 256]
        E = np.zeros((n, len(blk)))
        E[blk, np.arange(len(blk))] = 1.0
        var[lo:lo + 256] = lu.solve(E)[blk, np.arange(len(blk))]
    return ids, mean[p], var
