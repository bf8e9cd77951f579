"""-m gpu: CX_SCHED_TREE — ONE cx_sweep on a graph whose non-observed part is a forest is what ONE update_marginals! of the
reference leaves there (src/inference_engine.jl:575-608: the forward and the reverse pass; on a tree every message is then final),
for factors of any arity and variables of any degree.

Pinned by: the dense solve of the joint Gaussian (tests/kary_support.py); the flooding schedule run to its fixed point on the same
device (message by message); the restated reference scheduler on the reference's own state-space model
(test/inference_engine_tests.jl:436-487) — and the plan itself by its numpy execution on the CPU (tests/test_tree_plan.py)."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from tests.helpers import assert_close, engine_oracle_from_model
from tests.kary_support import dense_posterior

pytestmark = pytest.mark.gpu


def _tree_dev(m):
    dev = cx.DeviceGraph(schedule=L.SCHED_TREE)
    cx.synth.load_into_device(m, dev)
    return dev


@pytest.mark.parametrize("shape", ["random", "deep", "star"])
@pytest.mark.parametrize("n_factors,components,seed", [(1, 1, 1), (40, 1, 2), (40, 4, 3), (3000, 2, 4)])
def test_one_sweep_is_the_exact_posterior(hip_lib, shape, n_factors, components, seed):
    m = cx.synth.tree_model(n_factors, seed=seed, shape=shape, components=components, observe=0.3)
    dev = _tree_dev(m)
    dev.sweep(1)
    ids, em, ev = dense_posterior(m)
    marg = dev.get_marginals(ids)
    assert not np.any(np.isnan(marg)), "undefined marginals after one sweep"
    assert_close(marg[:, 0], em, 1e-9, f"{shape}: marginal means vs the dense solve")
    assert_close(marg[:, 1], ev, 1e-9, f"{shape}: marginal variances vs the dense solve")
    st = dev.tree_plan_stats()
    assert st["components"] == components and st["marginals"] == len(ids)
    hp = dev.tree_heavy_path_stats()
    assert st["stages"] == 2 * st["depth"] + 1 or 0 < hp["launches"] < 2 * st["depth"] + 1      # level by level, or over heavy paths where that takes fewer launches
    # a second sweep recomputes the same messages from the same inputs
    before = dev.get_marginals(ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(ids))


@pytest.mark.parametrize("shape", ["random", "deep", "star", "comb"])
@pytest.mark.parametrize("n_factors,components,seed", [(1, 1, 1), (40, 1, 2), (40, 4, 3), (3000, 2, 4)])
def test_one_sweep_over_heavy_paths_is_the_exact_posterior(hip_lib, monkeypatch, shape, n_factors, components, seed):
    """the same sweep over HEAVY PATHS (cx_tree_plan.h: build_hp; forced here, chosen by launch count otherwise): the paths of one
    light depth are one segmented scan per direction (csrc/cx_chain.hip), light edges and factors of more than two edges stay items"""
    monkeypatch.setenv("CX_TREE_HP", "1")
    m = cx.synth.tree_model(n_factors, seed=seed, shape=shape, components=components, observe=0.3)
    dev = _tree_dev(m)
    dev.sweep(1)
    ids, em, ev = dense_posterior(m)
    marg = dev.get_marginals(ids)
    assert not np.any(np.isnan(marg)), "undefined marginals after one sweep"
    assert_close(marg[:, 0], em, 1e-9, f"{shape}: marginal means vs the dense solve")
    assert_close(marg[:, 1], ev, 1e-9, f"{shape}: marginal variances vs the dense solve")
    hp = dev.tree_heavy_path_stats()
    if hp["launches"]:            # (a graph without a single two-edge link between sending variables has no path: level by level then)
        assert hp["paths"] >= 1 and len(ids) >= hp["single_variables"] + 2 * hp["paths"]
    before = dev.get_marginals(ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(ids))
    # message by message against the level schedule on the same device
    monkeypatch.setenv("CX_TREE_HP", "0")
    lv = _tree_dev(m)
    lv.sweep(1)
    assert lv.tree_heavy_path_stats()["launches"] == 0
    obs = set(int(v) for v in m.data_var)
    keep = np.array([int(v) not in obs for v in m.edge_var])
    evv, eff = m.edge_var[keep], m.edge_fac[keep]
    for direction in (L.TO_VARIABLE, L.TO_FACTOR):
        a, b = dev.get_messages(evv, eff, direction, L.FORM_NATURAL), lv.get_messages(evv, eff, direction, L.FORM_NATURAL)
        assert np.array_equal(np.isnan(a), np.isnan(b)), "the same messages are defined"
        assert_close(a[~np.isnan(a)], b[~np.isnan(b)], 1e-9, f"{shape}: messages, direction {direction}")


def test_a_state_space_model_with_a_latent_layer_takes_a_constant_number_of_launches(hip_lib, monkeypatch):
    """a spine of 20,000 states with a tooth of two variables below each (depth ~ 10,000 levels): by default the sweep runs over heavy
    paths — three light depths, a few dozen launches — and leaves what the level schedule leaves after its ~ 20,000 stages"""
    m = cx.synth.tree_model(60_000, seed=5, shape="comb", observe=0.3)
    dev = _tree_dev(m)
    dev.sweep(1)
    st, hp = dev.tree_plan_stats(), dev.tree_heavy_path_stats()
    assert st["depth"] >= 19_000 and 0 < hp["launches"] <= 30 and hp["light_depths"] <= 3
    monkeypatch.setenv("CX_TREE_HP", "0")
    lv = _tree_dev(m)
    lv.sweep(1)
    assert lv.tree_heavy_path_stats()["launches"] == 0 and lv.tree_plan_stats()["stages"] == 2 * st["depth"] + 1
    ids = m.x_ids
    a, b = dev.get_marginals(ids), lv.get_marginals(ids)
    assert not np.any(np.isnan(a))
    assert_close(a, b, 1e-9, "heavy paths vs level by level: marginals")
    # new data: the plan stays, the sweep follows
    rng = np.random.default_rng(2)
    y2 = m.data_y + rng.standard_normal(len(m.data_y))
    for d_ in (dev, lv):
        d_.set_messages(m.data_var, m.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
        d_.sweep(1)
    assert_close(dev.get_marginals(ids), lv.get_marginals(ids), 1e-9, "after new data")


def test_heavy_paths_run_through_factors_of_more_than_two_edges(hip_lib, monkeypatch):
    """long paths with side branches and factors of 2..6 variables (depth ~ 0.7 n levels): a heavy edge also runs through a factor with
    more than two edges — given its light children's messages such a factor is a pairwise rule between the path's two variables
    (csrc/cx_kary.hip: k_kary_link_params) — so the sweep takes a few dozen launches, not two per level, and leaves the level
    schedule's messages and marginals"""
    m = cx.synth.tree_model(20_000, seed=7, shape="deep", observe=0.3)
    dev = _tree_dev(m)
    dev.sweep(1)
    st, hp = dev.tree_plan_stats(), dev.tree_heavy_path_stats()
    assert st["depth"] >= 10_000 and 0 < hp["launches"] <= 80 and hp["light_depths"] <= 6, (st, hp)
    monkeypatch.setenv("CX_TREE_HP", "0")
    lv = _tree_dev(m)
    lv.sweep(1)
    assert lv.tree_heavy_path_stats()["launches"] == 0 and lv.tree_plan_stats()["stages"] == 2 * st["depth"] + 1
    ids = m.x_ids
    a, b = dev.get_marginals(ids), lv.get_marginals(ids)
    assert not np.any(np.isnan(a))
    assert_close(a, b, 1e-9, "heavy paths through k-ary factors vs level by level: marginals")
    obs = set(int(v) for v in m.data_var)
    keep = np.array([int(v) not in obs for v in m.edge_var])
    evv, eff = m.edge_var[keep], m.edge_fac[keep]
    for direction in (L.TO_VARIABLE, L.TO_FACTOR):
        x, y = dev.get_messages(evv, eff, direction, L.FORM_NATURAL), lv.get_messages(evv, eff, direction, L.FORM_NATURAL)
        assert np.array_equal(np.isnan(x), np.isnan(y)), "the same messages are defined"
        assert_close(x[~np.isnan(x)], y[~np.isnan(y)], 1e-9, f"messages, direction {direction}")
    # new coefficients and new data: the plan stays, the links' pairwise parameters follow (they are formed every sweep)
    rng = np.random.default_rng(3)
    coef2 = m.meta["coef"] * rng.uniform(0.8, 1.2, len(m.meta["coef"]))
    y2 = m.data_y + rng.standard_normal(len(m.data_y))
    for d_ in (dev, lv):
        d_.set_factor_coefficients(m.meta["coef_var"], m.meta["coef_fac"], coef2)
        d_.set_messages(m.data_var, m.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
        d_.sweep(1)
    assert_close(dev.get_marginals(ids), lv.get_marginals(ids), 1e-9, "after new coefficients and data")


@pytest.mark.parametrize("n_factors,seed", [(1, 1), (30, 2), (400, 3)])
def test_heavy_paths_on_a_graph_of_k_ary_factors_only(hip_lib, monkeypatch, n_factors, seed):
    """no pairwise factor at all (synth.kary_model: factors of 3..7 variables and unary priors — the handle keeps no per-slot (a, b)):
    every link of every path runs through a k-ary factor; the dense solve's marginals"""
    monkeypatch.setenv("CX_TREE_HP", "1")
    m = cx.synth.kary_model(n_factors, seed=seed, tree=True, observe=0.25)
    dev = _tree_dev(m)
    dev.sweep(1)
    hp = dev.tree_heavy_path_stats()
    assert hp["launches"] > 0 and hp["paths"] >= 1
    ids, em, ev = dense_posterior(m)
    marg = dev.get_marginals(ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, 0], em, 1e-9, "k-ary only: marginal means vs the dense solve")
    assert_close(marg[:, 1], ev, 1e-9, "k-ary only: marginal variances vs the dense solve")
    before = dev.get_marginals(ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(ids))


def test_messages_equal_the_flooding_fixed_point(hip_lib):
    """every factor→variable message into a non-observed variable and every variable→factor message that has a reader: one tree sweep
    == the flooding schedule after diameter-many sweeps on the same device"""
    m = cx.synth.tree_model(120, seed=9, shape="random", components=2, observe=0.25)
    tree = _tree_dev(m)
    tree.sweep(1)
    flood = cx.DeviceGraph(schedule=L.SCHED_FLOODING)
    cx.synth.load_into_device(m, flood)
    flood.sweep(2 * tree.tree_plan_stats()["depth"] + 4)
    obs = set(int(v) for v in m.data_var)
    keep = np.array([int(v) not in obs for v in m.edge_var])
    ev, ef = m.edge_var[keep], m.edge_fac[keep]
    a, b = tree.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), flood.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    assert not np.any(np.isnan(a))
    assert_close(a, b, 1e-9, "factor→variable messages, natural form")
    a, b = tree.get_messages(ev, ef, L.TO_FACTOR, L.FORM_NATURAL), flood.get_messages(ev, ef, L.TO_FACTOR, L.FORM_NATURAL)
    has = ~np.isnan(a[:, 1])          # lazy: no message towards a factor whose other variables are all observed, none out of a degree-1 variable
    assert has.sum() > len(ev) // 3
    assert_close(a[has], b[has], 1e-9, "variable→factor messages that the tree schedule computes")


def test_the_reference_state_space_model(hip_lib):
    """the SSM of test/inference_engine_tests.jl:436-487 (a path is a tree): one sweep == the restated reference scheduler's one
    update_marginals! == the chain-scan schedule"""
    m = cx.synth.ssm_chain(200, seed=5, random_variances=True)
    dev = _tree_dev(m)
    dev.sweep(1)
    eng = engine_oracle_from_model(m)
    eng.update_marginals(m.x_ids)
    _, em, ev = eng.get_marginals(m.x_ids)
    got = dev.get_marginals(m.x_ids)
    assert_close(got[:, 0], em, 1e-9, "means vs the restated scheduler")
    assert_close(got[:, 1], ev, 1e-9, "variances vs the restated scheduler")
    scan = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(m, scan)
    scan.sweep(1)
    assert_close(got, scan.get_marginals(m.x_ids), 1e-10, "tree schedule vs chain scan")
    assert dev.tree_plan_stats()["depth"] <= 200 + 1      # rooted at the middle of the path


def test_a_hub_of_degree_two_thousand(hip_lib):
    """the star: one variable in 2,000 factors (the segment-tree case of src/dependencies.jl:128-173) — two levels"""
    m = cx.synth.tree_model(2000, seed=12, shape="star", k_choices=(1, 2), observe=0.2)
    dev = _tree_dev(m)
    dev.sweep(1)
    ids, em, ev = dense_posterior(m)
    marg = dev.get_marginals(ids)
    assert_close(marg[:, 0], em, 1e-9, "means")
    assert_close(marg[:, 1], ev, 1e-9, "variances")
    assert dev.tree_plan_stats()["depth"] <= 4


def test_new_data_and_newly_observed_variables(hip_lib):
    """data changes keep the plan; observing a latent variable cuts the tree there (the plan is rebuilt)"""
    import dataclasses
    m = cx.synth.tree_model(80, seed=21, shape="deep", observe=0.3)
    dev = _tree_dev(m)
    dev.sweep(1)
    y2 = m.data_y + 1.0
    dev.set_messages(m.data_var, m.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
    dev.sweep(1)
    m2 = dataclasses.replace(m, data_y=y2)
    ids, em, ev = dense_posterior(m2)
    marg = dev.get_marginals(ids)
    assert_close(marg[:, 0], em, 1e-9, "after new data: means")
    assert_close(marg[:, 1], ev, 1e-9, "after new data: variances")
    # observe an inner variable through one of its factors: everything it separates becomes independent
    fac_of = {}
    for v, f in zip(m.edge_var, m.edge_fac):
        if int(f) in set(int(x) for x in m.meta["kary_ids"]):
            fac_of.setdefault(int(v), int(f))
    deg = np.bincount(m.edge_var)
    inner = next(int(v) for v in m.x_ids if deg[v] >= 3 and int(v) in fac_of)
    before = dev.tree_plan_stats()
    inner_facs = m.edge_fac[m.edge_var == inner]          # the datum travels on every edge of the variable (a message is per edge)
    dev.set_messages(np.full(len(inner_facs), inner), inner_facs, L.TO_FACTOR, L.FORM_POINT, np.full(len(inner_facs), 0.7))
    assert dev.tree_plan_stats()["stages"] == 0, "a newly observed variable invalidates the plan"
    dev.sweep(1)
    after = dev.tree_plan_stats()
    assert after["marginals"] == before["marginals"] - 1 and after["components"] >= before["components"]
    # dense solve with the extra datum: the observed variable's prior no longer counts (its messages are never read)
    keep = m.prior_var != inner
    m3 = dataclasses.replace(m2, data_var=np.r_[m.data_var, inner], data_fac=np.r_[m.data_fac, fac_of[inner]], data_y=np.r_[y2, 0.7],
                             prior_var=m.prior_var[keep], prior_fac=m.prior_fac[keep], prior_mean=m.prior_mean[keep], prior_variance=m.prior_variance[keep])
    ids, em, ev = dense_posterior(m3)
    marg = dev.get_marginals(ids)
    assert_close(marg[:, 0], em, 1e-9, "after observing an inner variable: means")
    assert_close(marg[:, 1], ev, 1e-9, "after observing an inner variable: variances")


def test_cycles_are_refused_not_approximated(hip_lib):
    m = cx.synth.gaussian_grid(5, 5, seed=3)
    dev = cx.DeviceGraph(schedule=L.SCHED_TREE)
    cx.synth.load_into_device(m, dev)
    with pytest.raises(cx.CortexHipError) as e:
        dev.sweep(1)
    assert e.value.code == L.ERR_UNSUPPORTED and "cycle" in str(e.value)
    loopy = cx.synth.kary_model(50, seed=4, tree=False)
    dev2 = cx.DeviceGraph(schedule=L.SCHED_TREE)
    cx.synth.load_into_device(loopy, dev2)
    with pytest.raises(cx.CortexHipError) as e:
        dev2.sweep(1)
    assert e.value.code == L.ERR_UNSUPPORTED


def test_a_large_bushy_tree_one_sweep(hip_lib):
    """200,000 factors, ~10^6 edges: the depth stays logarithmic, one sweep is a few dozen launches; checked against the flooding
    schedule at its fixed point on a sample of the marginals"""
    m = cx.synth.tree_model(200_000, seed=31, shape="random", observe=0.2)
    dev = _tree_dev(m)
    dev.sweep(1)
    st = dev.tree_plan_stats()
    assert st["depth"] < 120, st
    flood = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m, flood)
    flood.sweep(2 * st["depth"] + 6)
    ids = m.x_ids[:: max(len(m.x_ids) // 50_000, 1)]
    a, b = dev.get_marginals(ids), flood.get_marginals(ids)
    assert not np.any(np.isnan(a))
    assert_close(a, b, 1e-9, "marginals vs the fused schedule at its fixed point")


@pytest.mark.parametrize("d,b,n", [(2, 2, 31), (3, 4, 85), (4, 3, 121), (4, 6, 259), (4, 11, 133), (2, 30, 31)])
def test_d_dimensional_trees_one_sweep(hip_lib, d, b, n):
    """dim 2..4: a tree of states with b children each (degree b + 2 <= 8), every state observed through a likelihood factor: one sweep of
    the tree schedule == the joint solve == the fused schedule at its fixed point, message by message"""
    from tests.test_gpu_mv import _branching_lgssm

    model, emean, ecov = _branching_lgssm(n, d, seed=40 + b, b=b)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], emean, 1e-8, "marginal mean vs the joint solve", scale_by="max")
    assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-8, "marginal covariance vs the joint solve", scale_by="max")      # (covariances ∝ I here: the median entry is ~ 0)
    st = dev.tree_plan_stats()
    assert st["components"] == 1 and st["marginals"] == n and st["kary_entries"] == 0
    fused = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, fused)
    fused.sweep(2 * st["depth"] + 6)
    xs = set(int(v) for v in model.x_ids)
    keep = np.array([int(v) in xs for v in model.edge_var])
    ev, ef = model.edge_var[keep], model.edge_fac[keep]
    a, bb = dev.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), fused.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    assert not np.any(np.isnan(a))
    assert_close(a, bb, 1e-8, "factor→variable messages vs the fused schedule at its fixed point", scale_by="max")
    before = dev.get_marginals(model.x_ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(model.x_ids))


def _comb_pairs(n_spine, teeth=1):
    """a spine of states 0 .. n_spine - 1, below every state a path of `teeth` more states (a latent layer): (parent, child) pairs, parents first"""
    pairs, nxt = [], n_spine
    for i in range(n_spine):
        if i + 1 < n_spine:
            pairs.append((i, i + 1))
        up = i
        for _ in range(teeth):
            pairs.append((up, nxt)); up = nxt; nxt += 1
    return sorted(pairs), nxt


@pytest.mark.parametrize("d,n_spine,teeth", [(2, 2, 1), (2, 60, 1), (2, 150, 2), (3, 2, 1), (3, 60, 1), (3, 150, 2), (4, 2, 1), (4, 60, 1), (4, 150, 2),
                                             (64, 2, 1), (64, 7, 1), (64, 12, 2), (6, 20, 1), (33, 9, 2)])
def test_d_dimensional_heavy_paths(hip_lib, monkeypatch, d, n_spine, teeth):
    """dim 2..4 over heavy paths (the scans of csrc/cx_mvchain.hip per light depth, light edges as items) and dim 64 — with 5..63 embedded
    in it — (a plan of compositions and walks per light depth, csrc/cx_mv64chain.hip): a chain of states with a latent layer below each —
    the joint solve's marginals, the level schedule's messages"""
    from tests.test_gpu_mv import _branching_lgssm

    pairs, n = _comb_pairs(n_spine, teeth)
    model, emean, ecov = _branching_lgssm(n, d, seed=70 + d, pairs=pairs)
    monkeypatch.setenv("CX_TREE_HP", "1")
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    hp = dev.tree_heavy_path_stats()
    assert hp["launches"] > 0 and hp["paths"] >= 1 and hp["light_depths"] <= 3
    marg = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], emean, 1e-8, "marginal mean vs the joint solve", scale_by="max")
    assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-8, "marginal covariance vs the joint solve", scale_by="max")
    monkeypatch.setenv("CX_TREE_HP", "0")
    lv = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, lv)
    lv.sweep(1)
    assert lv.tree_heavy_path_stats()["launches"] == 0
    xs = set(int(v) for v in model.x_ids)
    keep = np.array([int(v) in xs for v in model.edge_var])
    ev, ef = model.edge_var[keep], model.edge_fac[keep]
    a, b = dev.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), lv.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "the same messages are defined"
    assert_close(a[~np.isnan(a)], b[~np.isnan(b)], 1e-8, "factor→variable messages vs the level schedule", scale_by="max")
    before = dev.get_marginals(model.x_ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(model.x_ids))


@pytest.mark.parametrize("d,b,n", [(2, 2, 40), (4, 3, 60), (3, 6, 80)])
def test_d_dimensional_bushy_trees_over_heavy_paths(hip_lib, monkeypatch, d, b, n):
    """forced over heavy paths where the level schedule would be chosen: many light depths, short paths, variables of degree up to 8"""
    from tests.test_gpu_mv import _branching_lgssm

    model, emean, ecov = _branching_lgssm(n, d, seed=90 + b, b=b)
    monkeypatch.setenv("CX_TREE_HP", "1")
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    assert dev.tree_heavy_path_stats()["launches"] > 0
    marg = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], emean, 1e-8, "marginal mean vs the joint solve", scale_by="max")
    assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-8, "marginal covariance vs the joint solve", scale_by="max")


@pytest.mark.parametrize("d,n_spine", [(4, 90), (2, 40), (64, 10)])
def test_observing_a_state_rebuilds_the_heavy_paths_of_dim_greater_one(hip_lib, monkeypatch, d, n_spine):
    """a standing heavy-path plan of dim 2..4 / 64, then a state in the middle of the spine becomes observed (a point mass on every one
    of its edges): the plan is rebuilt — the spine is two paths now — and the sweep leaves what the level schedule leaves"""
    from tests.test_gpu_mv import _branching_lgssm

    pairs, n = _comb_pairs(n_spine, 1)
    model, _, _ = _branching_lgssm(n, d, seed=81, pairs=pairs, solve=False)
    mid = int(model.x_ids[n_spine // 2])
    mid_facs = model.edge_fac[model.edge_var == mid]
    value = np.tile(np.linspace(-0.5, 0.5, d), (len(mid_facs), 1))
    free = np.array([v for v in model.x_ids if int(v) != mid])
    got = {}
    for hp_env in ("1", "0"):
        monkeypatch.setenv("CX_TREE_HP", hp_env)
        dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
        cx.synth.load_into_device(model, dev)
        dev.sweep(1)
        paths_before = dev.tree_heavy_path_stats()["paths"]
        dev.set_messages(np.full(len(mid_facs), mid), mid_facs, L.TO_FACTOR, L.FORM_POINT, value)
        assert dev.tree_plan_stats()["stages"] == 0, "a newly observed variable invalidates the plan"
        dev.sweep(1)
        if hp_env == "1":
            assert dev.tree_heavy_path_stats()["paths"] >= paths_before and dev.tree_heavy_path_stats()["launches"] > 0
        got[hp_env] = dev.get_marginals(free)
        assert not np.any(np.isnan(got[hp_env]))
    assert_close(got["1"], got["0"], 1e-8, "after observing a state: heavy paths vs level by level", scale_by="max")


def test_a_d_dimensional_state_space_model_with_a_latent_layer(hip_lib, monkeypatch):
    """d = 4, 20,000 states on the spine and a latent state below each (depth ~ 20,000 levels): by default over heavy paths — two light
    depths, a dozen launches — and the level schedule's marginals; new data under the standing plan"""
    from tests.test_gpu_mv import _branching_lgssm

    pairs, n = _comb_pairs(20_000, 1)
    model, _, _ = _branching_lgssm(n, 4, seed=77, pairs=pairs, solve=False)
    dev = cx.DeviceGraph(dim=4, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    st, hp = dev.tree_plan_stats(), dev.tree_heavy_path_stats()
    assert st["depth"] >= 19_000 and 0 < hp["launches"] <= 30 and hp["light_depths"] <= 3, (st, hp)
    monkeypatch.setenv("CX_TREE_HP", "0")
    lv = cx.DeviceGraph(dim=4, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, lv)
    lv.sweep(1)
    a, b = dev.get_marginals(model.x_ids), lv.get_marginals(model.x_ids)
    assert not np.any(np.isnan(a))
    assert_close(a, b, 1e-8, "heavy paths vs level by level: marginals", scale_by="max")
    rng = np.random.default_rng(4)
    y2 = model.data_y + rng.standard_normal(model.data_y.shape)
    for d_ in (dev, lv):
        d_.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
        d_.sweep(1)
    assert_close(dev.get_marginals(model.x_ids), lv.get_marginals(model.x_ids), 1e-8, "after new data", scale_by="max")


def test_d_dimensional_chain_is_a_tree_too(hip_lib):
    m = cx.synth.lgssm_chain(300, d=4, seed=8)
    tree = cx.DeviceGraph(dim=4, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(m, tree)
    tree.sweep(1)
    assert 0 < tree.tree_heavy_path_stats()["launches"] <= 8       # the heavy-path plan roots a chain at one of its ends: ONE path, one final scan
    scan = cx.DeviceGraph(dim=4, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(m, scan)
    scan.sweep(1)
    assert_close(tree.get_marginals(m.x_ids), scan.get_marginals(m.x_ids), 1e-9, "tree schedule vs chain scan on a d = 4 chain", scale_by="max")


@pytest.mark.parametrize("heavy_paths", ["", "1"])
def test_many_small_random_forests(hip_lib, monkeypatch, heavy_paths):
    """forty small forests of every shape (factor arities, hubs, long paths, several components, a random share of observed leaves): one
    sweep == the dense solve, and the plan's bookkeeping adds up — as the library chooses, and forced over heavy paths"""
    if heavy_paths:
        monkeypatch.setenv("CX_TREE_HP", heavy_paths)
    rng = np.random.default_rng(99)
    for trial in range(40):
        n_factors = int(rng.integers(1, 40))
        shape = ["random", "deep", "star"][trial % 3]
        comps = int(rng.integers(1, 4))
        m = cx.synth.tree_model(n_factors, seed=1000 + trial, shape=shape, components=comps, observe=float(rng.uniform(0, 0.6)),
                                k_choices=tuple(int(k) for k in rng.integers(1, 7, size=3)))
        dev = _tree_dev(m)
        dev.sweep(1)
        ids, em, ev = dense_posterior(m)
        marg = dev.get_marginals(ids)
        assert_close(marg[:, 0], em, 1e-9, f"trial {trial} ({shape}, {n_factors} factors, {comps} components): means")
        assert_close(marg[:, 1], ev, 1e-9, f"trial {trial}: variances")
        st = dev.tree_plan_stats()
        assert st["marginals"] == len(ids) and st["messages_up"] == st["messages_down"]
        dev.close()


def test_without_marginals_in_the_sweep_the_last_stage_is_left_out(hip_lib):
    """compute_marginals_in_sweep = 0: the messages of one sweep are the same, the marginals are the caller's to ask for item by item"""
    m = cx.synth.tree_model(60, seed=5, shape="random", observe=0.3)
    a, b = _tree_dev(m), cx.DeviceGraph(schedule=L.SCHED_TREE, marginals_in_sweep=0)
    cx.synth.load_into_device(m, b)
    a.sweep(1); b.sweep(1)
    obs = set(int(v) for v in m.data_var)
    keep = np.array([int(v) not in obs for v in m.edge_var])
    ev, ef = m.edge_var[keep], m.edge_fac[keep]
    assert np.array_equal(a.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), b.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), equal_nan=True)
    ids, em, ev_ = dense_posterior(m)
    assert np.all(np.isnan(b.get_marginals(ids)))
    b.update_batch([L.ITEM_INDIVIDUAL_MARGINAL] * len(ids), ids, [0] * len(ids))
    assert np.array_equal(a.get_marginals(ids), b.get_marginals(ids))


@pytest.mark.parametrize("d,b,n", [(64, 2, 7), (64, 2, 15), (6, 2, 15), (33, 1, 9), (64, 3, 13), (64, 6, 15), (6, 4, 21)])
def test_dim_64_trees_one_sweep(hip_lib, d, b, n):
    """dim 64 (and 5 .. 63 embedded in it): a tree of states with b children each (degree b + 2 <= 8; a dim 64 rule sums at most three
    sources itself: a sender of degree 5 .. 8 has its other messages summed by k_v2f64 first), every state observed: ONE sweep of the tree
    schedule — a stage is one launch of the MFMA rule kernel over its records — == the joint solve; the fused schedule needs depth-many
    sweeps for the same messages"""
    from tests.test_gpu_mv import _branching_lgssm

    model, emean, ecov = _branching_lgssm(n, d, seed=60 + n, b=b)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], emean, 1e-8, f"d={d}: marginal mean vs the joint solve", scale_by="max")
    assert_close(marg[:, d:].reshape(n, d, d), ecov, 1e-8, f"d={d}: marginal covariance vs the joint solve", scale_by="max")
    st = dev.tree_plan_stats()
    fused = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, fused)
    fused.sweep(2 * st["depth"] + 4)
    xs = set(int(v) for v in model.x_ids)
    keep = np.array([int(v) in xs for v in model.edge_var])
    ev, ef = model.edge_var[keep], model.edge_fac[keep]
    a, bb = dev.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL), fused.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    assert not np.any(np.isnan(a))
    assert_close(a, bb, 1e-8, "factor→variable messages vs the fused schedule at its fixed point", scale_by="max")
    before = dev.get_marginals(model.x_ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(model.x_ids))


def test_a_dim_64_state_space_model_with_a_latent_layer(hip_lib, monkeypatch):
    """d = 64, 400 states on the spine and a latent state below each (depth ~ 400 levels, 801 stages level by level): by default over
    heavy paths — fewer launches — and the level schedule's marginals"""
    from tests.test_gpu_mv import _branching_lgssm

    pairs, n = _comb_pairs(400, 1)
    model, _, _ = _branching_lgssm(n, 64, seed=78, pairs=pairs, solve=False)
    dev = cx.DeviceGraph(dim=64, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    st, hp = dev.tree_plan_stats(), dev.tree_heavy_path_stats()
    assert st["depth"] >= 390 and 0 < hp["launches"] < 2 * st["depth"] + 1 and hp["light_depths"] <= 3, (st, hp)
    monkeypatch.setenv("CX_TREE_HP", "0")
    lv = cx.DeviceGraph(dim=64, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, lv)
    lv.sweep(1)
    assert lv.tree_heavy_path_stats()["launches"] == 0
    a, b = dev.get_marginals(model.x_ids), lv.get_marginals(model.x_ids)
    assert not np.any(np.isnan(a))
    assert_close(a, b, 1e-7, "heavy paths vs level by level: marginals", scale_by="max")
    before = dev.get_marginals(model.x_ids)
    dev.sweep(1)
    assert np.array_equal(before, dev.get_marginals(model.x_ids))


def test_dim_64_chain_under_the_tree_schedule_equals_the_chain_scan(hip_lib):
    m = cx.synth.lgssm_chain(40, d=64, seed=9)
    tree = cx.DeviceGraph(dim=64, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(m, tree)
    tree.sweep(1)
    scan = cx.DeviceGraph(dim=64, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(m, scan)
    scan.sweep(1)
    assert_close(tree.get_marginals(m.x_ids), scan.get_marginals(m.x_ids), 1e-8, "tree schedule vs chain scan on a d = 64 chain", scale_by="max")


@pytest.mark.parametrize("d,b,n,heavy", [(4, 3, 40, "0"), (4, 3, 40, "1"), (64, 2, 7, "0"), (64, 2, 9, "1"), (6, 2, 7, "0")])
def test_new_rule_matrices_under_a_standing_tree_plan(hip_lib, monkeypatch, d, b, n, heavy):
    """cx_set_factor_matrices between two sweeps of the tree schedule (the parameter-learning flow): the stages of a sweep are a captured
    HIP graph with the rule tables' addresses baked in, so the tables are rewritten in place and a graph over a table that has to move
    (one more parameter set) is dropped — the second sweep is the exact posterior under the NEW (A, Q), equal to a fresh handle's."""
    import copy

    from tests.test_gpu_mv import _branching_lgssm

    monkeypatch.setenv("CX_TREE_HP", heavy)
    model, emean, ecov = _branching_lgssm(n, d, seed=31, b=b)
    rng = np.random.default_rng(7)
    A_old = 0.7 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    old = copy.copy(model)
    old.psets = {0: (A_old, 0.5 * np.eye(d)), 1: model.psets[1]}
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(old, dev)
    dev.sweep(1)
    stale = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(stale))
    dev.set_factor_matrices(0, *model.psets[0])                     # same size: rewritten in place
    dev.sweep(1)
    fresh = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, fresh)
    fresh.sweep(1)
    want = fresh.get_marginals(model.x_ids)
    got = dev.get_marginals(model.x_ids)
    assert np.max(np.abs(stale - want)) > 1e-3, "the two parameter sets must give different posteriors for this test to mean anything"
    assert_close(got, want, 1e-10, "marginals after new matrices vs a fresh handle", scale_by="max")
    assert_close(got[:, :d], emean, 1e-8, "marginal mean vs the joint solve", scale_by="max")
    assert_close(got[:, d:].reshape(n, d, d), ecov, 1e-8, "marginal covariance vs the joint solve", scale_by="max")
    dev.set_factor_matrices(2, np.eye(d), np.eye(d))                # one more set: the tables move, the captured graph goes
    dev.set_factor_matrices(0, A_old, 0.5 * np.eye(d))
    dev.sweep(1)
    assert_close(dev.get_marginals(model.x_ids), stale, 1e-10, "back to the first parameters after the tables moved", scale_by="max")
    dev.set_factor_matrices(0, *model.psets[0])
    dev.sweep(2)
    assert_close(dev.get_marginals(model.x_ids), want, 1e-10, "and to the second", scale_by="max")
    dev.close(); fresh.close()
