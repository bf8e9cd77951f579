"""oracle/mv.py — numpy restatement of d-dimensional linear-Gaussian sum-product.  TEST INFRASTRUCTURE ONLY.

The reference contains no d-dimensional rule ("parity unpinned" for dim > 1): its nearest relatives are
the scalar rules of test/inference_engine_tests.jl:385-432 and the MvNormalMeanPrecision struct of
test/runtests.jl:70-77.  This file states the d-dimensional analogue in the form those tests use — messages as
(mean, covariance), products through precisions, the factor x_out = A x_in + N(0, Q) forward as (A m, A S A' + Q) — and
applies it in the device's flooding order.  Pure Python loops: small graphs only.  Exactness is pinned separately by
oracle/exact.py:lgssm_posterior (block-tridiagonal solve)."""
from __future__ import annotations

import numpy as np

import ctypes as C

from .ref import FloodGraph, _p, lib


def product(a, b):
    """(m1, S1) x (m2, S2): precision-weighted, like product(::NormalMeanPrecision, ...) (test/runtests.jl:79-85)."""
    (m1, S1), (m2, S2) = a, b
    W1, W2 = np.linalg.inv(S1), np.linalg.inv(S2)
    S = np.linalg.inv(W1 + W2)
    return S @ (W1 @ m1 + W2 @ m2), S


class MvFlood:
    def __init__(self, model):
        self.d = model.dim
        g = FloodGraph(model.edge_var, model.edge_fac, model.factor_ids, np.zeros(len(model.factor_ids)))
        self.g = g
        role = np.asarray(model.edge_role)[g.order]
        fid = np.asarray(model.factor_ids); pset = np.asarray(model.factor_var).astype(int)
        srt = np.argsort(fid)
        self.pset = pset[srt][np.searchsorted(fid[srt], g.edge_fac)]
        self.role = role
        self.psets = model.psets
        self.f2v = [None] * g.ne      # (mean, cov) or None (UndefValue)
        self.v2f = [None] * g.ne
        self.point = {}               # edge -> observed datum
        if len(model.data_var):
            for e, y in zip(g.edge_index(model.data_var, model.data_fac), np.asarray(model.data_y)):
                self.point[int(e)] = np.asarray(y, dtype=float)

    def _rule(self, e):
        """factor→variable on receiving edge e from the message on the partner edge."""
        g, p = self.g, int(self.g.partner[e])
        A, Q = self.psets[int(self.pset[e])]
        forward = self.role[e] == 0      # receiver is the OUT edge
        if p in self.point:
            y = self.point[p]
            if forward:
                return A @ y, Q.copy()
            Qi = np.linalg.inv(Q)
            S = np.linalg.inv(A.T @ Qi @ A)
            return S @ (A.T @ Qi @ y), S
        if self.v2f[p] is None:
            return None
        m, S = self.v2f[p]
        if forward:
            return A @ m, A @ S @ A.T + Q
        # backward: information form of ∫ N(x_out; A x_in, Q) m(x_out) dx_out
        Wq = np.linalg.inv(S + Q)
        W = A.T @ Wq @ A
        Sx = np.linalg.inv(W)
        return Sx @ (A.T @ Wq @ m), Sx

    def sweep(self, n=1):
        g = self.g
        for _ in range(n):
            new_v2f = list(self.v2f)
            for v in range(g.nv):
                s, t = int(g.var_off[v]), int(g.var_off[v + 1])
                for e in range(s, t):
                    if e in self.point or g.partner[e] < 0 or t - s < 2:
                        continue
                    acc, ok = None, True
                    for o in range(s, t):
                        if o == e:
                            continue
                        if self.f2v[o] is None:
                            ok = False
                            break
                        acc = self.f2v[o] if acc is None else product(acc, self.f2v[o])
                    if ok:
                        new_v2f[e] = acc
            self.v2f = new_v2f
            new_f2v = list(self.f2v)
            for e in range(g.ne):
                if g.partner[e] >= 0:
                    r = self._rule(e)
                    if r is not None:
                        new_f2v[e] = r
            self.f2v = new_f2v

    def marginal(self, v_index):
        g = self.g
        acc = None
        for o in range(int(g.var_off[v_index]), int(g.var_off[v_index + 1])):
            if self.f2v[o] is None:
                return None
            acc = self.f2v[o] if acc is None else product(acc, self.f2v[o])
        return acc


class MvFloodC:
    """The same flooding sweeps in C (oracle/mv_flood.c), array state instead of Python lists: per-sweep parity of the
    dim > 1 kernels at chain lengths MvFlood's Python loops cannot reach.  `f2v[e]` / `v2f[e]` / `marginal(v)` read like
    MvFlood's.  Pinned against MvFlood in tests/test_mv_flood_checker.py."""

    class _View:
        def __init__(self, m, S, defined, d):
            self.m, self.S, self.defined, self.d = m, S, defined, d

        def __getitem__(self, e):
            if not self.defined[e]:
                return None
            return self.m[e], self.S[e]

    def __init__(self, model):
        self.d = d = model.dim
        self.g = g = FloodGraph(model.edge_var, model.edge_fac, model.factor_ids, np.zeros(len(model.factor_ids)))
        self.role = np.ascontiguousarray(np.asarray(model.edge_role)[g.order], dtype=np.int32)
        fid = np.asarray(model.factor_ids); pset = np.asarray(model.factor_var).astype(int)
        srt = np.argsort(fid)
        self.pset = np.ascontiguousarray(pset[srt][np.searchsorted(fid[srt], g.edge_fac)], dtype=np.int32)
        nps = max(model.psets) + 1
        self.A, self.Q = np.zeros((nps, d, d)), np.zeros((nps, d, d))
        for k, (A, Q) in model.psets.items():
            self.A[k], self.Q[k] = A, Q
        ne = g.ne
        self.f2v_m, self.f2v_S, self.f2v_def = np.zeros((ne, d)), np.zeros((ne, d, d)), np.zeros(ne, np.uint8)
        self.v2f_m, self.v2f_S, self.v2f_def = np.zeros((ne, d)), np.zeros((ne, d, d)), np.zeros(ne, np.uint8)
        self.is_point, self.point_y = np.zeros(ne, np.uint8), np.zeros((ne, d))
        if len(model.data_var):
            e = g.edge_index(model.data_var, model.data_fac)
            self.is_point[e] = 1
            self.point_y[e] = np.asarray(model.data_y, dtype=float)
        self.f2v = self._View(self.f2v_m, self.f2v_S, self.f2v_def, d)
        self.v2f = self._View(self.v2f_m, self.v2f_S, self.v2f_def, d)

    def seed(self, mean: float, variance: float):
        """what cx_seed_messages(TO_VARIABLE) does: every still-undefined message of a pairwise factor becomes N(mean, variance I)"""
        und = (self.f2v_def == 0) & (self.g.partner >= 0)
        self.f2v_m[und] = mean
        self.f2v_S[und] = variance * np.eye(self.d)
        self.f2v_def[und] = 1

    def sweep(self, n=1, use_omp=False):
        L, g, dbl, u8 = lib(), self.g, C.c_double, C.c_uint8
        total = 0
        for _ in range(n):
            r = L.cxo_mv_flood_sweep(self.d, g.nv, _p(g.var_off, C.c_int64), g.ne, _p(g.partner, C.c_int64), _p(self.pset, C.c_int32),
                                     _p(self.role, C.c_int32), _p(self.A, dbl), _p(self.Q, dbl), _p(self.is_point, u8),
                                     _p(self.point_y, dbl), _p(self.f2v_m, dbl), _p(self.f2v_S, dbl), _p(self.f2v_def, u8),
                                     _p(self.v2f_m, dbl), _p(self.v2f_S, dbl), _p(self.v2f_def, u8), int(use_omp))
            if r < 0:
                raise MemoryError("cxo_mv_flood_sweep")
            total += r
        return total

    def marginals(self):
        """(mean [nv, d], covariance [nv, d, d], defined [nv]) for every variable, in g.var_ids order"""
        L, g, dbl, u8 = lib(), self.g, C.c_double, C.c_uint8
        m, S, ok = np.zeros((g.nv, self.d)), np.zeros((g.nv, self.d, self.d)), np.zeros(g.nv, np.uint8)
        if L.cxo_mv_flood_marginals(self.d, g.nv, _p(g.var_off, C.c_int64), _p(self.f2v_m, dbl), _p(self.f2v_S, dbl),
                                    _p(self.f2v_def, u8), _p(m, dbl), _p(S, dbl), _p(ok, u8)) != 0:
            raise MemoryError("cxo_mv_flood_marginals")
        return m, S, ok.astype(bool)

    def marginal(self, v_index):
        m, S, ok = self.marginals()
        return (m[v_index], S[v_index]) if ok[v_index] else None
