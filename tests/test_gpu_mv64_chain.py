"""-m gpu: CX_SCHED_CHAIN_SCAN for dim 64 (csrc/cx_mv64chain.hip; the plan: csrc/cx_chain64_plan.h).

What is checked: ONE cx_sweep on a d = 64 state-space chain is what ONE update_marginals! of the reference computes on such a
graph — the exact forward/backward smoother (src/inference_engine.jl:575-608; the SSM of
test/inference_engine_tests.jl:436-487) — with no seeding and at EVERY time step, BASELINE config C5 (T = 1e5) included.
The reference has no d-dimensional rule (parity unpinned for d > 1), so the numbers are pinned by the exact block-tridiagonal
solve (oracle/exact.py; oracle/blocktri.c at full size) and, for the messages, by the numpy restatement oracle/mv.py run to its
fixed point."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact
from oracle.mv import MvFlood
from tests.helpers import assert_close as _assert_close

pytestmark = pytest.mark.gpu
D = 64


def assert_close(a, b, rtol, what=""):
    return _assert_close(a, b, rtol, what, scale_by="max")


def _dev(model, schedule=L.SCHED_CHAIN_SCAN, **kw):
    dev = cx.DeviceGraph(dim=model.dim, schedule=schedule)
    cx.synth.load_into_device(model, dev, **kw)
    return dev


def _marginals(dev, ids, chunk=4096):
    return np.concatenate([dev.get_marginals(ids[i:i + chunk]) for i in range(0, len(ids), chunk)])


def _check_exact(dev, model, tol, what, c_solver=False):
    d, T = model.dim, len(model.x_ids)
    solve = exact.lgssm_posterior_c if c_solver else exact.lgssm_posterior
    em, ecov = solve(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    marg = _marginals(dev, model.x_ids)
    assert not np.any(np.isnan(marg)), f"{what}: undefined marginals after one sweep"
    assert_close(marg[:, :d], em, tol, f"{what}: marginal means, all {T} steps")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, tol, f"{what}: marginal covariances, all {T} steps")


@pytest.mark.parametrize("T,K,fan", [(2, 0, 4), (3, 1, 2), (6, 2, 2), (9, 4, 4), (40, 3, 3), (131, 2, 4), (300, 0, 4)])
def test_one_sweep_is_the_exact_smoother(hip_lib, monkeypatch, T, K, fan):
    """no seeding, one cx_sweep, every marginal; K = links per level-0 block (0: the default), fan = potentials per group: one block
    (no composition at all), one level, several levels, ragged last blocks and groups"""
    if K:
        monkeypatch.setenv("CX_MVC64_K", str(K))
    monkeypatch.setenv("CX_MVC64_FAN", str(fan))
    model = cx.synth.lgssm_chain(T, d=D, seed=3 + T)
    dev = _dev(model)
    dev.sweep(1)
    _check_exact(dev, model, 1e-9, f"T={T} K={K} fan={fan}")
    st = dev.chain_plan_stats()
    assert st["rules"] >= 2 * (T - 1) and st["links_per_block"] == (K or 4)
    if K and T - 1 > K * fan:
        assert st["levels"] >= 2
    # a second sweep recomputes the same exact messages from the same inputs
    before = dev.get_marginals(model.x_ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(model.x_ids))


def test_messages_equal_the_flooding_fixed_point(hip_lib, monkeypatch):
    """every factor→variable message of the latent variables after ONE scan sweep == oracle/mv.py after T + 2 flooding sweeps (its
    fixed point on a tree) == what the reference's sequential passes leave in the Signals; and == the flooding kernel's own fixed point"""
    monkeypatch.setenv("CX_MVC64_K", "2")
    monkeypatch.setenv("CX_MVC64_FAN", "2")
    T = 11
    model = cx.synth.lgssm_chain(T, d=D, seed=7)
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
        assert_close(row[:D], m, 1e-9, f"f2v mean edge {e}"); assert_close(row[D:].reshape(D, D), S, 1e-9, f"f2v covariance edge {e}")
    flood = _dev(model, schedule=L.SCHED_FUSED, seed_variance=1e6)
    flood.sweep(T + 2)
    ref = flood.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE, form=L.FORM_NATURAL)
    mine = dev.get_messages(g.edge_var[pe], g.edge_fac[pe], L.TO_VARIABLE, form=L.FORM_NATURAL)
    assert_close(mine, ref, 1e-9, "chain scan vs the flooding kernel at its fixed point (natural form)")


def test_slow_mixing_model_where_flooding_fails(hip_lib):
    """A = 0.999 I, Q = 1e-4 I, R = 10 I, T = 3,000.  Information travels ~1000 steps here: 96 flooding sweeps (one
    link per sweep) are nowhere near; one scan sweep is exact at every step."""
    T = 3000
    model = cx.synth.lgssm_chain(T, d=D, seed=31, A=0.999 * np.eye(D), Q=1e-4 * np.eye(D), R=10.0 * np.eye(D))
    dev = _dev(model)
    dev.sweep(1)
    _check_exact(dev, model, 1e-7, "slow-mixing model", c_solver=True)
    flood = _dev(model, schedule=L.SCHED_FUSED, seed_variance=1e6)
    flood.sweep(96)
    em, _ = exact.lgssm_posterior_c(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    mid = slice(T // 2 - 20, T // 2 + 20)
    fm = flood.get_marginals(model.x_ids[mid])[:, :D]
    assert np.max(np.abs(fm - em[mid])) > 1e-3 * np.max(np.abs(em[mid])), "flooding unexpectedly converged: the model is not slow-mixing"


def test_config_c5_full_size_one_sweep_every_marginal(hip_lib):
    """BASELINE.json configs[4] at full size: d = 64, T = 1e5 (399,998 edges).  ONE sweep, no seeding, ALL T marginals against the
    exact smoother (oracle/blocktri.c) — the reference's one update_marginals! on this graph."""
    T = 100_000
    model = cx.synth.lgssm_chain(T, d=D, seed=1234)
    assert model.n_edges == 399_998
    dev = _dev(model)
    dev.sweep(1)
    st = dev.chain_plan_stats()
    assert st["compositions"] > 0 and st["rules"] >= 2 * (T - 1)
    em, ecov = exact.lgssm_posterior_c(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    worst_m = worst_c = 0.0
    sm, sc = np.max(np.abs(em)), np.max(np.abs(ecov))
    for i in range(0, T, 5000):                     # 5000 marginals = 166 MB per read
        marg = dev.get_marginals(model.x_ids[i:i + 5000])
        assert not np.any(np.isnan(marg)), f"undefined marginals in [{i}, {i + 5000})"
        worst_m = max(worst_m, float(np.max(np.abs(marg[:, :D] - em[i:i + 5000])) / sm))
        worst_c = max(worst_c, float(np.max(np.abs(marg[:, D:].reshape(-1, D, D) - ecov[i:i + 5000])) / sc))
    assert worst_m <= 1e-7 and worst_c <= 1e-7, (worst_m, worst_c)


def test_new_data_and_new_rule_matrices_between_sweeps(hip_lib, monkeypatch):
    import dataclasses

    monkeypatch.setenv("CX_MVC64_K", "3")
    T = 50
    model = cx.synth.lgssm_chain(T, d=D, seed=41)
    dev = _dev(model)
    dev.sweep(2)
    y2 = model.data_y + 0.5
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
    dev.sweep(1)
    _check_exact(dev, dataclasses.replace(model, data_y=y2), 1e-9, "after new data")
    R2 = 2.5 * model.meta["R"]
    dev.set_factor_matrices(1, np.eye(D), R2)
    dev.sweep(1)
    m2 = dataclasses.replace(model, data_y=y2, meta={**model.meta, "R": R2})
    _check_exact(dev, m2, 1e-9, "after a new likelihood covariance")
    A2 = 0.9 * model.meta["A"]
    dev.set_factor_matrices(0, A2, model.meta["Q"])
    dev.sweep(1)
    _check_exact(dev, dataclasses.replace(m2, meta={**m2.meta, "A": A2}), 1e-9, "after a new transition matrix")


def test_disjoint_chains_and_isolated_variables(hip_lib, monkeypatch):
    """several components — chains of 1 (an isolated variable: no link), 2, 30 and 7 states: nothing is carried across a path boundary"""
    monkeypatch.setenv("CX_MVC64_K", "2")
    monkeypatch.setenv("CX_MVC64_FAN", "3")
    A = cx.synth.lgssm_chain(2, d=D, seed=50).meta["A"]          # one parameter set for all components
    parts = [cx.synth.lgssm_chain(T, d=D, seed=50 + T, A=A) for T in (1, 2, 30, 1, 7)]
    model = cx.synth.concat_models(parts)
    dev = _dev(model)
    dev.sweep(1)
    for part, (n, off) in zip(parts, model.meta["parts"]):
        em, ecov = exact.lgssm_posterior(part.data_y, part.meta["A"], part.meta["Q"], part.meta["R"])
        marg = dev.get_marginals(part.x_ids + off)
        assert_close(marg[:, :D], em, 1e-9, f"component of {n} states, means")
        assert_close(marg[:, D:].reshape(n, D, D), ecov, 1e-9, f"component of {n} states, covariances")


def test_undefined_inputs_leave_the_messages_undefined_and_unsupported_graphs_are_refused(hip_lib):
    # a latent variable whose observation carries no datum: the observation variable is a non-observed reader off the chains
    model = cx.synth.lgssm_chain(6, d=D, seed=2)
    dev = cx.DeviceGraph(dim=D, schedule=L.SCHED_CHAIN_SCAN)
    for k, (A, Q) in model.psets.items():
        dev.set_factor_matrices(k, A, Q)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, edge_role=model.edge_role)
    dev.set_messages(model.data_var[:-1], model.data_fac[:-1], L.TO_FACTOR, L.FORM_POINT, model.data_y[:-1])
    with pytest.raises(cx.CortexHipError) as e:
        dev.sweep(1)
    assert e.value.code == L.ERR_UNSUPPORTED


def test_block_potential_of_a_whole_chain(hip_lib):
    """cx_chain_block_maps for dim 64 (round 4; what a partition's time blocks exchange: tests/test_gpu_partition.py): on a chain that
    is not cut at all, the forward map applied to the empty message, plus the side information of the last variable, IS that
    variable's marginal — and the backward map gives the first variable's — in natural form."""
    from cortex.jl_amd import partition

    T = 9
    model = cx.synth.lgssm_chain(T, d=D, seed=3)
    dev = _dev(model)
    fwd, bwd, s_first, s_last, v0, v1, nl = dev.chain_block_maps()
    assert (v0, v1, nl) == (int(model.x_ids[0]), int(model.x_ids[-1]), T - 1)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    zero = (np.zeros(D), np.zeros((D, D)))
    for M, side, t in ((fwd, s_last, T - 1), (bwd, s_first, 0)):
        e, lam = partition._mv_map_apply(M, D, *zero)
        e, lam = e + side[:D], lam + partition._mv_sym(side[D:], D)
        cov = np.linalg.inv(lam)
        assert_close(cov, ecov[t], 1e-9, f"covariance of state {t} from the block potential")
        assert_close(cov @ e, em[t], 1e-9, f"mean of state {t} from the block potential")
    dev.sweep(1)          # the sweep after it starts from the potentials that are on the device already
    _check_exact(dev, model, 1e-9, "sweep after cx_chain_block_maps")
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, model.data_y + 1.0)
    dev.chain_block_maps()
    dev.set_messages(model.data_var[:1], model.data_fac[:1], L.TO_FACTOR, L.FORM_POINT, model.data_y[:1])      # ... unless data changed in between
    dev.sweep(1)
    y2 = model.data_y + 1.0
    y2[0] = model.data_y[0]
    em2, _ = exact.lgssm_posterior(y2, model.meta["A"], model.meta["Q"], model.meta["R"])
    assert_close(dev.get_marginals(model.x_ids)[:, :D], em2, 1e-9, "new data between the block maps and the sweep")


def test_checkpoint_round_trip_under_the_chain_schedule(hip_lib):
    T = 20
    model = cx.synth.lgssm_chain(T, d=D, seed=61)
    dev = _dev(model)
    dev.sweep(1)
    blob = dev.export_state()
    want = dev.get_marginals(model.x_ids)
    other = _dev(model)
    other.import_state(blob)
    assert np.array_equal(other.get_marginals(model.x_ids), want)
    other.sweep(1)
    assert_close(other.get_marginals(model.x_ids), want, 1e-12, "a sweep after the import")


@pytest.mark.parametrize("d,sensors,T", [(64, 3, 9), (64, 6, 7), (6, 4, 30), (16, 6, 7), (24, 3, 9), (32, 5, 12)])
def test_chain_variables_of_degree_up_to_eight(hip_lib, d, sensors, T):
    """a chain whose states are observed by several sensors each (degree 2 + sensors, up to 8): a rule or a joint of the plan sums three
    sources — the side information of such a position is summed into one message first (k_side64) — ONE sweep == the joint solve"""
    from tests.test_gpu_mv import _multi_sensor_lgssm

    model, emean, ecov = _multi_sensor_lgssm(T, d, seed=11 + sensors, sensors=sensors)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], emean, 1e-8, "marginal mean vs the joint solve")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-8, "marginal covariance vs the joint solve")
    before = dev.get_marginals(model.x_ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(model.x_ids))
