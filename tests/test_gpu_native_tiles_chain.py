"""-m gpu: the chain-scan and tree schedules of d = 5 .. 32 in their native tile size (round 6: csrc/cx_mv64chain.hip k_compose_nt / k_walk_nt run
the plan of csrc/cx_chain64_plan.h on 1 x 1 or 2 x 2 tiles of 16; until then these dims were embedded in 4 x 4 tiles there).  ONE cx_sweep
on a state-space chain == the exact forward/backward smoother at every time step (the block-tridiagonal solve of oracle/exact.py), and ==
the embedding in 64 (CX_MFMA_DIM=64)."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact
from tests.helpers import assert_close as _assert_close

pytestmark = pytest.mark.gpu


def assert_close(a, b, rtol, what=""):
    return _assert_close(a, b, rtol, what, scale_by="max")


def _solve(model, schedule=L.SCHED_CHAIN_SCAN, sweeps=1):
    dev = cx.DeviceGraph(dim=model.dim, schedule=schedule)
    cx.synth.load_into_device(model, dev)
    dev.sweep(sweeps)
    return dev


def _marginals(dev, ids, chunk=4096):
    return np.concatenate([dev.get_marginals(ids[i:i + chunk]) for i in range(0, len(ids), chunk)])


@pytest.mark.parametrize("d,T,K,fan", [(16, 2, 0, 2), (16, 3, 1, 2), (16, 9, 4, 4), (16, 131, 2, 4), (16, 300, 0, 2), (32, 2, 0, 2), (32, 40, 3, 3), (32, 300, 0, 2),
                                       (9, 57, 2, 2), (5, 300, 0, 2), (20, 131, 3, 2), (31, 64, 0, 4)])
def test_one_sweep_is_the_exact_smoother(hip_lib, monkeypatch, d, T, K, fan):
    """no seeding, one cx_sweep, every marginal; K = links per level-0 block (0: the default), fan = potentials per group; d = 9, 5, 20, 31:
    the user's dim inside the next tile size"""
    if K:
        monkeypatch.setenv("CX_MVC64_K", str(K))
    monkeypatch.setenv("CX_MVC64_FAN", str(fan))
    model = cx.synth.lgssm_chain(T, d=d, seed=3 + T + d)
    dev = _solve(model)
    em, ecov = exact.lgssm_posterior(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    marg = _marginals(dev, model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], em, 1e-9, f"d={d} T={T}: marginal means")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-9, f"d={d} T={T}: marginal covariances")
    st = dev.chain_plan_stats()
    assert st["rules"] >= 2 * (T - 1)
    # a second sweep leaves the exact result where it is
    dev.sweep(1)
    assert_close(_marginals(dev, model.x_ids), marg, 1e-12, "second sweep")


@pytest.mark.parametrize("d", [16, 32, 11])
def test_native_tiles_hold_a_fraction_of_the_bytes_and_equal_the_embedding(hip_lib, monkeypatch, d):
    T = 400
    model = cx.synth.lgssm_chain(T, d=d, seed=77)
    a = _solve(model)
    monkeypatch.setenv("CX_MFMA_DIM", "64")
    b = _solve(model)
    monkeypatch.delenv("CX_MFMA_DIM")
    ma, mb = _marginals(a, model.x_ids), _marginals(b, model.x_ids)
    assert_close(ma, mb, 1e-9, f"d={d}: native tiles vs the embedding in 64")
    tr = model.factor_ids[T:]
    for vs in (model.x_ids[:-1], model.x_ids[1:]):
        assert_close(a.get_messages(vs, tr, L.TO_VARIABLE), b.get_messages(vs, tr, L.TO_VARIABLE), 1e-8, "chain messages")
    nd = 16 if d <= 16 else 32
    assert a.stats()["device_bytes"] < b.stats()["device_bytes"] * 1.6 * (nd + nd * nd) / (64 + 64 * 64)


def test_a_long_chain_at_d16(hip_lib):
    """T = 20,000: several levels of the composition tree at the default block size; every marginal against the C block-tridiagonal solve"""
    d, T = 16, 20_000
    model = cx.synth.lgssm_chain(T, d=d, seed=5)
    dev = _solve(model)
    em, ecov = exact.lgssm_posterior_c(model.data_y, model.meta["A"], model.meta["Q"], model.meta["R"])
    marg = _marginals(dev, model.x_ids)
    assert_close(marg[:, :d], em, 1e-8, "marginal means")
    assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-8, "marginal covariances")
    assert dev.chain_plan_stats()["levels"] >= 2


@pytest.mark.parametrize("d", [16, 8, 32])
def test_the_tree_schedule_over_heavy_paths_at_native_tiles(hip_lib, d):
    """a chain with a latent state below each state (cx.synth.lgssm_comb: heavy path = the spine, light edges = the teeth) under
    CX_SCHED_TREE: ONE sweep == the fused schedule at its fixed point on the same graph (2 x depth sweeps)"""
    n = 60
    model = cx.synth.lgssm_comb(n, d=d, teeth=1, seed=9)
    tree = _solve(model, L.SCHED_TREE)
    st = tree.tree_heavy_path_stats()
    assert st["launches"] > 0, st      # the plan runs over heavy paths (a chain with side branches)
    fused = _solve(model, L.SCHED_FUSED, sweeps=2 * n + 8)
    ids = model.x_ids
    assert_close(_marginals(tree, ids), _marginals(fused, ids), 1e-8, f"d={d}: tree schedule vs the fused fixed point")


@pytest.mark.parametrize("d", [16, 24])
def test_new_data_and_new_rule_matrices_between_sweeps(hip_lib, monkeypatch, d):
    import dataclasses

    monkeypatch.setenv("CX_MVC64_K", "3")
    T = 50
    model = cx.synth.lgssm_chain(T, d=d, seed=41)
    dev = _solve(model, sweeps=2)

    def check(m, what):
        em, ecov = exact.lgssm_posterior(m.data_y, m.meta["A"], m.meta["Q"], m.meta["R"])
        marg = _marginals(dev, m.x_ids)
        assert_close(marg[:, :d], em, 1e-9, what + ", means")
        assert_close(marg[:, d:].reshape(T, d, d), ecov, 1e-9, what + ", covariances")

    y2 = model.data_y + 0.5
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
    dev.sweep(1)
    check(dataclasses.replace(model, data_y=y2), "after new data")
    R2 = 2.5 * model.meta["R"]
    dev.set_factor_matrices(1, np.eye(d), R2)
    dev.sweep(1)
    m2 = dataclasses.replace(model, data_y=y2, meta={**model.meta, "R": R2})
    check(m2, "after a new likelihood covariance")
    A2 = 0.9 * model.meta["A"]
    dev.set_factor_matrices(0, A2, model.meta["Q"])
    dev.sweep(1)
    check(dataclasses.replace(m2, meta={**m2.meta, "A": A2}), "after a new transition matrix")


@pytest.mark.parametrize("d", [16, 32])
def test_disjoint_chains_and_isolated_variables(hip_lib, monkeypatch, d):
    """several components — chains of 1 (an isolated variable: no link), 2, 30 and 7 states: nothing is carried across a path boundary"""
    monkeypatch.setenv("CX_MVC64_K", "2")
    monkeypatch.setenv("CX_MVC64_FAN", "3")
    A = cx.synth.lgssm_chain(2, d=d, seed=50).meta["A"]          # one parameter set for all components
    parts = [cx.synth.lgssm_chain(T, d=d, seed=50 + T, A=A) for T in (1, 2, 30, 1, 7)]
    model = cx.synth.concat_models(parts)
    dev = _solve(model)
    for part, (n, off) in zip(parts, model.meta["parts"]):
        em, ecov = exact.lgssm_posterior(part.data_y, part.meta["A"], part.meta["Q"], part.meta["R"])
        marg = dev.get_marginals(part.x_ids + off)
        assert_close(marg[:, :d], em, 1e-9, f"component of {n} states, means")
        assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-9, f"component of {n} states, covariances")


def test_time_blocks_are_refused_below_64(hip_lib):
    """cx_chain_block_maps exchanges 64 x 64 potentials: a chain of d = 16 has to be embedded by the caller (CX_MFMA_DIM=64)"""
    model = cx.synth.lgssm_chain(12, d=16, seed=1)
    dev = _solve(model)
    with pytest.raises(cx.CortexHipError) as e:
        dev.chain_block_maps()
    assert e.value.code == L.ERR_UNSUPPORTED
