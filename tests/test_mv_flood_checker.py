"""CPU: the C flooding checker for d-dimensional messages (oracle/mv_flood.c) against the numpy restatement
(oracle/mv.py, the form the device tests were written against in round 1), against the committed golden vectors and,
at its fixed point, against the exact block-tridiagonal smoother.  The reference holds no d-dimensional rule
(parity unpinned for dim > 1); these tests pin the checker family against itself and against mathematics."""
import json
import os

import numpy as np
import pytest

import cortex.jl_amd as cx
from oracle import exact
from oracle.mv import MvFlood, MvFloodC

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("d,T", [(2, 7), (4, 9), (8, 5)])
def test_c_checker_equals_numpy_restatement_every_sweep(d, T):
    model = cx.synth.lgssm_chain(T, d=d, seed=3)
    a, b = MvFlood(model), MvFloodC(model)
    for sweep in range(T + 2):
        a.sweep(1); b.sweep(1)
        for e in range(a.g.ne):
            for name in ("f2v", "v2f"):
                x, y = getattr(a, name)[e], getattr(b, name)[e]
                assert (x is None) == (y is None), f"sweep {sweep} edge {e} {name}: definedness differs"
                if x is None:
                    continue
                scale = max(1.0, float(np.max(np.abs(x[1]))))
                if not np.all(np.isfinite(x[1])) or np.linalg.cond(x[1]) > 1e10:
                    continue     # improper message towards an observed variable (nobody reads it)
                np.testing.assert_allclose(y[0], x[0], rtol=0, atol=1e-9 * max(1.0, float(np.max(np.abs(x[0])))))
                np.testing.assert_allclose(y[1], x[1], rtol=0, atol=1e-9 * scale)
    for v in range(a.g.nv):
        x, y = a.marginal(v), b.marginal(v)
        assert (x is None) == (y is None)


def test_c_checker_fixed_point_is_the_exact_smoother():
    d, T = 4, 150
    model = cx.synth.lgssm_chain(T, d=d, seed=11)
    o = MvFloodC(model)
    o.sweep(T + 3, use_omp=True)
    m, S, ok = o.marginals()
    xs = np.searchsorted(o.g.var_ids, model.x_ids)
    assert ok[xs].all()
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    np.testing.assert_allclose(m[xs], em, rtol=0, atol=1e-9 * np.max(np.abs(em)))
    np.testing.assert_allclose(S[xs], ecov, rtol=0, atol=1e-9 * np.max(np.abs(ecov)))


def test_c_checker_reproduces_the_committed_golden_vectors():
    """tests/golden/lgssm_d4.json: seeded d = 4 chain with its exact posterior (generator tests/golden/make_golden.py)"""
    gold = json.load(open(os.path.join(HERE, "golden", "lgssm_d4.json")))
    T = gold["T"]
    model = cx.synth.lgssm_chain(T, d=4, seed=gold["seed"])
    np.testing.assert_array_equal(np.asarray(gold["data_y"]), model.data_y)
    o = MvFloodC(model)
    o.sweep(T + 2)
    m, S, ok = o.marginals()
    xs = np.searchsorted(o.g.var_ids, np.asarray(gold["x_ids"]))
    pm, pc = np.asarray(gold["posterior_mean"]), np.asarray(gold["posterior_covariance"]).reshape(T, 4, 4)
    np.testing.assert_allclose(m[xs], pm, rtol=0, atol=1e-9 * np.max(np.abs(pm)))
    np.testing.assert_allclose(S[xs], pc, rtol=0, atol=1e-9 * np.max(np.abs(pc)))
