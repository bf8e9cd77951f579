"""-m gpu: CX_SCHED_CHAIN_SCAN for d-dimensional messages (dim 2, 3, 4; csrc/cx_mvchain.hip).

What is checked: ONE cx_sweep on a state-space chain is what ONE update_marginals! of the reference computes on such a graph —
the exact forward/backward smoother (src/inference_engine.jl:575-608; the SSM of
test/inference_engine_tests.jl:436-487) — with no seeding and at every time step.  The reference has no d-dimensional rule, so
the numbers are pinned by the exact block-tridiagonal solve (oracle/exact.py; oracle/blocktri.c at full size) and, for the
messages, by the numpy restatement oracle/mv.py run to its fixed point."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact
from oracle.mv import MvFlood
from tests.helpers import assert_close as _assert_close

pytestmark = pytest.mark.gpu


def assert_close(a, b, rtol, what=""):
    return _assert_close(a, b, rtol, what, scale_by="max")


def _dev(model, schedule=L.SCHED_CHAIN_SCAN):
    dev = cx.DeviceGraph(dim=model.dim, schedule=schedule)
    cx.synth.load_into_device(model, dev)
    return dev


def _check_exact(dev, model, tol, what, c_solver=False):
    d, T = model.dim, len(model.x_ids)
    solve = exact.lgssm_posterior_c if c_solver else exact.lgssm_posterior
    em, ecov = solve(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    marg = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(marg)), f"{what}: undefined marginals after one sweep"
    assert_close(marg[:, :d], em, tol, f"{what}: marginal means, all {T} steps")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, tol, f"{what}: marginal covariances, all {T} steps")


@pytest.mark.parametrize("K", [1, 2, 5])
@pytest.mark.parametrize("d,T", [(2, 2), (2, 9), (3, 40), (4, 3), (4, 9), (4, 700)])
def test_one_sweep_is_the_exact_smoother(hip_lib, monkeypatch, d, T, K):
    """no seeding, one cx_sweep, every marginal; K = links per thread of the scan (1: pure scan, 5: ragged last thread)"""
    monkeypatch.setenv("CX_MVC_K", str(K))
    model = cx.synth.lgssm_chain(T, d=d, seed=3 + T)
    dev = _dev(model)
    dev.sweep(1)
    _check_exact(dev, model, 1e-9, f"d={d} T={T} K={K}")
    # a second sweep changes nothing (the scan recomputes the same exact messages)
    before = dev.get_marginals(model.x_ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(model.x_ids))


@pytest.mark.parametrize("d", [2, 4])
def test_messages_equal_the_flooding_fixed_point(hip_lib, d):
    """every factor→variable and variable→factor message of the latent variables after ONE scan sweep == oracle/mv.py after T + 2
    flooding sweeps (its fixed point on a tree) == what the reference's sequential passes leave in the Signals"""
    T = 12
    model = cx.synth.lgssm_chain(T, d=d, seed=7)
    dev = _dev(model)
    dev.sweep(1)
    o = MvFlood(model)
    o.sweep(T + 2)
    g = o.g
    xs = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
    pe = np.array([e for e in np.flatnonzero(g.partner >= 0) if int(np.searchsorted(g.var_ids, g.edge_var[e])) in xs])
    got = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE)
    for row, e in zip(got, pe):
        m, S = o.f2v[e]
        assert_close(row[:d], m, 1e-9, f"f2v mean edge {e}"); assert_close(row[d:].reshape(d, d), S, 1e-9, f"f2v covariance edge {e}")
    # variable→factor messages into the transition factors, recomputed on demand from the stored messages
    xi = np.searchsorted(g.var_ids, model.x_ids)
    for t in (0, 1, T // 2, T - 1):
        for e in range(int(g.var_off[xi[t]]), int(g.var_off[xi[t] + 1])):
            if o.v2f[e] is None or g.partner[e] < 0:
                continue
            deg = int(g.var_off[xi[t] + 1] - g.var_off[xi[t]])
            if deg < 2:
                continue
            got = dev.get_messages([g.edge_var[e]], [g.edge_fac[e]], L.TO_FACTOR)[0]
            m, S = o.v2f[e]
            if not np.all(np.isfinite(S)) or np.linalg.cond(S) > 1e12:
                continue
            assert_close(got[:d], m, 1e-9, "v2f mean"); assert_close(got[d:].reshape(d, d), S, 1e-9, "v2f covariance")


def test_several_tiles_and_the_scan_of_tile_totals(hip_lib, monkeypatch):
    """K = 1, T = 70,000: 274 tiles of 256 links — more than one chunk of the one-workgroup scan of the tile totals"""
    monkeypatch.setenv("CX_MVC_K", "1")
    model = cx.synth.lgssm_chain(70_000, d=2, seed=21)
    dev = _dev(model)
    dev.sweep(1)
    _check_exact(dev, model, 1e-9, "d=2 T=70000 K=1", c_solver=True)


def test_slow_mixing_model_where_flooding_fails(hip_lib):
    """A = 0.999 I, Q = 1e-4 I, R = 10 I, T = 20,000.  Information travels ~1000 steps here: K flooding sweeps
    (one link per sweep) are nowhere near; one scan sweep is exact at every step."""
    d, T = 4, 20_000
    model = cx.synth.lgssm_chain(T, d=d, seed=31, A=0.999 * np.eye(d), Q=1e-4 * np.eye(d), R=10.0 * np.eye(d))
    dev = _dev(model)
    dev.sweep(1)
    _check_exact(dev, model, 1e-8, "slow-mixing model", c_solver=True)
    # the flooding schedule, seeded as the round-2 tests seed it, after 160 sweeps: still far from the posterior in the middle
    flood = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, flood, seed_variance=1e6)
    flood.sweep(160)
    em, _ = exact.lgssm_posterior_c(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    mid = slice(T // 2 - 50, T // 2 + 50)
    fm = flood.get_marginals(model.x_ids[mid])[:, :d]
    assert np.max(np.abs(fm - em[mid])) > 1e-3 * np.max(np.abs(em[mid])), "flooding unexpectedly converged: the model is not slow-mixing"


def test_config_c3_full_size_one_sweep_every_marginal(hip_lib):
    """BASELINE.json configs[2] at full size: d = 4, T = 1e6 (3,999,998 edges).  ONE sweep, no seeding, ALL T marginals against the
    exact smoother (oracle/blocktri.c) — the reference's one update_marginals! on this graph."""
    d, T = 4, 1_000_000
    model = cx.synth.lgssm_chain(T, d=d, seed=1234)
    assert model.n_edges == 3_999_998
    dev = _dev(model)
    dev.sweep(1)
    _check_exact(dev, model, 1e-8, "C3 full size", c_solver=True)


def test_new_data_and_new_rule_matrices_between_sweeps(hip_lib):
    """the side sums are cached between sweeps: new data (set_messages) and new (A, Q) (cx_set_factor_matrices) must refresh them"""
    import dataclasses

    d, T = 4, 300
    model = cx.synth.lgssm_chain(T, d=d, seed=41)
    dev = _dev(model)
    dev.sweep(2)
    y2 = model.data_y + 0.5
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
    dev.sweep(1)
    _check_exact(dev, dataclasses.replace(model, data_y=y2), 1e-9, "after new data")
    R2 = 2.5 * model.meta["R"]
    dev.set_factor_matrices(1, np.eye(d), R2)
    dev.sweep(1)
    m2 = dataclasses.replace(model, data_y=y2, meta={**model.meta, "R": R2})
    _check_exact(dev, m2, 1e-9, "after a new likelihood covariance")
    A2 = 0.9 * model.meta["A"]
    dev.set_factor_matrices(0, A2, model.meta["Q"])
    dev.sweep(1)
    _check_exact(dev, dataclasses.replace(m2, meta={**m2.meta, "A": A2}), 1e-9, "after a new transition matrix")


def test_disjoint_chains_and_isolated_variables(hip_lib, monkeypatch):
    """a graph of several components: chains of 1 (an isolated variable: no link), 2, 300 and 7 states — the segmented scan must not
    carry anything across a path boundary, in either direction, wherever the boundaries fall inside a thread's K links"""
    d = 3
    for K in (1, 2, 4):
        monkeypatch.setenv("CX_MVC_K", str(K))
        A = cx.synth.lgssm_chain(2, d=d, seed=50).meta["A"]          # one parameter set for all components
        parts = [cx.synth.lgssm_chain(T, d=d, seed=50 + T, A=A) for T in (1, 2, 300, 1, 7)]
        model = cx.synth.concat_models(parts)
        dev = _dev(model)
        dev.sweep(1)
        for part, (n, off) in zip(parts, model.meta["parts"]):
            em, ecov = exact.lgssm_posterior(part.data_y, part.meta["A"], part.meta["Q"], part.meta["R"])
            marg = dev.get_marginals(part.x_ids + off)
            assert_close(marg[:, :d], em, 1e-9, f"K={K}: component of {n} states, means")
            assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-9, f"K={K}: component of {n} states, covariances")


def test_unsupported_graphs_are_refused_not_approximated(hip_lib):
    d = 4
    # a latent variable without a datum on its observation: the observation variable is a non-observed reader off the chains
    model = cx.synth.lgssm_chain(6, d=d, seed=2)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
    for k, (A, Q) in model.psets.items():
        dev.set_factor_matrices(k, A, Q)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, edge_role=model.edge_role)
    dev.set_messages(model.data_var[:-1], model.data_fac[:-1], L.TO_FACTOR, L.FORM_POINT, model.data_y[:-1])
    with pytest.raises(cx.CortexHipError) as e:
        dev.sweep(1)
    assert e.value.code == L.ERR_UNSUPPORTED


def test_checkpoint_round_trip_under_the_chain_schedule(hip_lib):
    d, T = 4, 200
    model = cx.synth.lgssm_chain(T, d=d, seed=61)
    dev = _dev(model)
    dev.sweep(1)
    blob = dev.export_state()
    want = dev.get_marginals(model.x_ids)
    other = _dev(model)
    other.import_state(blob)
    assert np.array_equal(other.get_marginals(model.x_ids), want)
    other.sweep(1)
    assert_close(other.get_marginals(model.x_ids), want, 1e-12, "a sweep after the import")


def test_marginals_on_demand_are_the_same_marginals(hip_lib, monkeypatch):
    """compute_marginals_in_sweep = 2: a sweep leaves the forward and backward sums, the marginal pass runs before the first reader.
    Bit for bit the marginals of the default mode — after a sweep, after reads of other state in between (messages, a checkpoint,
    a batch of items), after new data, over several components, for every K"""
    d = 3
    for K in (1, 4, 16):
        monkeypatch.setenv("CX_MVC_K", str(K))
        A = cx.synth.lgssm_chain(2, d=d, seed=70).meta["A"]
        parts = [cx.synth.lgssm_chain(T, d=d, seed=70 + T, A=A) for T in (1, 2, 700, 9)]
        model = cx.synth.concat_models(parts)
        eager = _dev(model)
        lazy = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN, marginals_in_sweep=2)
        cx.synth.load_into_device(model, lazy)
        eager.sweep(1); lazy.sweep(1)
        assert np.array_equal(eager.get_marginals(model.x_ids), lazy.get_marginals(model.x_ids)), f"K={K}: first read"
        assert np.array_equal(eager.get_marginals(model.x_ids[:3]), lazy.get_marginals(model.x_ids[:3])), f"K={K}: second read"
        # two sweeps without a read in between; a FEW marginals first (formed from the walks' sums for just those variables, the pass
        # for all of them still owed: first and last variable of every component, isolated ones, scattered inner ones), then messages
        # (their walks rewrite the same sums), then all marginals
        eager.sweep(2); lazy.sweep(2)
        few = np.unique(np.r_[model.x_ids[:4], model.x_ids[-3:], model.x_ids[[5, 350, 351, 699, 700, 701, 702]]])
        assert len(few) * 8 < 2 * len(model.x_ids)
        assert np.array_equal(eager.get_marginals(few), lazy.get_marginals(few)), f"K={K}: a few marginals on demand"
        assert np.array_equal(eager.get_marginals(few[::-1]), lazy.get_marginals(few[::-1])), f"K={K}: again, another order"
        ev, ef = model.edge_var[:40], model.edge_fac[:40]
        assert np.array_equal(eager.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), lazy.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), equal_nan=True)
        assert np.array_equal(eager.get_marginals(model.x_ids), lazy.get_marginals(model.x_ids)), f"K={K}: after get_messages"
        # a checkpoint taken while the marginals are still owed carries them
        lazy.sweep(1); eager.sweep(1)
        blob = lazy.export_state()
        other = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN, marginals_in_sweep=2)
        cx.synth.load_into_device(model, other)
        other.import_state(blob)
        assert np.array_equal(other.get_marginals(model.x_ids), eager.get_marginals(model.x_ids)), f"K={K}: through a checkpoint"
        # new data, sweep, read
        y2 = model.data_y - 0.25
        for dev in (eager, lazy):
            dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
            dev.sweep(1)
        assert np.array_equal(eager.get_marginals(model.x_ids), lazy.get_marginals(model.x_ids)), f"K={K}: after new data"
        for dev in (eager, lazy, other):
            dev.close()
