"""The committed golden vectors (tests/golden/*.json, generator: tests/golden/make_golden.py).

CPU half: the checker code still reproduces them (they pin the oracle against drift) and the two independent routes they
hold agree with each other — the restated engine against the Thomas solve, converged loopy BP against the dense solve.
GPU half: the HIP path through the C ABI against the same files."""
import json
import os

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact, vmp
from tests.helpers import engine_oracle_from_model, flood_oracle_from_model

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return {k: (np.asarray(v) if isinstance(v, list) else v) for k, v in json.load(f).items()}


# ---------------------------------------------------------------------------------------------------------------- CPU
def test_chain_vector_engine_and_exact_solver_agree():
    d = load("chain16.json")
    m = cx.synth.ssm_chain(16, seed=1234)
    assert np.array_equal(m.data_y, d["data_y"])                         # the seeded generator has not drifted
    np.testing.assert_allclose(d["engine_mean"], d["exact_mean"], rtol=1e-12)
    np.testing.assert_allclose(d["engine_variance"], d["exact_variance"], rtol=1e-12)
    E = engine_oracle_from_model(m)
    E.set_messages_to_factor(m.data_var, m.data_fac, m.data_y)
    E.update_marginals(m.x_ids)
    _t, em, ev = E.get_marginals(m.x_ids)
    assert np.array_equal(em, d["engine_mean"]) and np.array_equal(ev, d["engine_variance"])
    # the reference test's own assertions on its chain (test/inference_engine_tests.jl:485-487)
    assert np.all(np.diff(d["engine_mean"]) >= 0) and np.all(d["engine_variance"] >= 0)


def test_grid_vector_flooding_checker_and_dense_solve_agree():
    d = load("grid8x8.json")
    m = cx.synth.gaussian_grid(8, 8, seed=1234)
    g = flood_oracle_from_model(m, 1e6)
    assert np.array_equal(g.edge_var, d["edge_var"]) and np.array_equal(g.edge_fac, d["edge_fac"])
    g.sweep(5)
    assert np.array_equal(g.f2v_m, d["f2v_mean_after_5"], equal_nan=True) and np.array_equal(g.f2v_v, d["f2v_variance_after_5"], equal_nan=True)
    np.testing.assert_allclose(d["bp_mean_converged"], d["exact_mean"], rtol=1e-10)   # loopy Gaussian BP means are exact


def test_block_chain_and_variational_vectors_reproduce():
    d = load("lgssm_d4.json")
    em, ecov = exact.lgssm_posterior(d["data_y"], d["A"], d["Q"], d["R"])
    np.testing.assert_allclose(em, d["posterior_mean"], rtol=1e-13)
    np.testing.assert_allclose(ecov, d["posterior_covariance"], rtol=1e-13)
    d64 = load("lgssm_d64.json")
    m64 = cx.synth.lgssm_chain(3, d=64, seed=1234)                 # A comes from the seed: its first row guards the generator
    np.testing.assert_allclose(m64.meta["A"][0], d64["A_row0"], rtol=0, atol=0)
    assert np.array_equal(m64.data_y, d64["data_y"])
    em, ecov = exact.lgssm_posterior(m64.data_y, m64.meta["A"], m64.meta["Q"], m64.meta["R"])
    np.testing.assert_allclose(em, d64["posterior_mean"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ecov[1], d64["posterior_covariance_middle"], rtol=1e-12, atol=1e-14)
    # the numpy flooding restatement (what the d = 64 kernel is checked against sweep by sweep) reaches the same posterior
    from oracle.mv import MvFlood
    o = MvFlood(m64)
    o.sweep(6)
    for t, xi in enumerate(np.searchsorted(o.g.var_ids, m64.x_ids)):
        mu, S = o.marginal(int(xi))
        np.testing.assert_allclose(mu, d64["posterior_mean"][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.diag(S), d64["posterior_variances"][t], rtol=1e-9)
    v = load("vmp_n8.json")
    for name, cls in (("mean_field", vmp.MeanFieldVMP), ("structured", vmp.StructuredVMP)):
        a = cls(v["data_y"])
        for _ in range(5):
            a.update(["x"]); a.update(["ssnoise", "obsnoise"])
        np.testing.assert_allclose(a.xm, v[name]["x_mean"], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(list(a.ss) + list(a.obs), v[name]["ssnoise_shape_scale"] + v[name]["obsnoise_shape_scale"], rtol=1e-13)


def test_reference_known_answer_constants():
    k = load("kats.json")
    t = k["tracing"]
    assert [2 * x for x in t["data"]] == t["message_values"] and sum(t["message_values"]) + t["prior"] == t["marginal"]
    assert k["beta_bernoulli"]["prior"] == [1.0, 1.0]


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_chain_against_the_golden_vector(hip_lib):
    d = load("chain16.json")
    for schedule, sweeps in ((L.SCHED_CHAIN_SCAN, 1), (L.SCHED_FUSED, 18), (L.SCHED_FLOODING, 18)):
        dev = cx.DeviceGraph(schedule=schedule)
        cx.synth.load_into_device(cx.synth.ssm_chain(16, seed=1234), dev)
        dev.sweep(sweeps)
        if schedule != L.SCHED_CHAIN_SCAN:
            dev.update_batch([L.ITEM_INDIVIDUAL_MARGINAL] * 16, d["x_ids"], [0] * 16)
        m = dev.get_marginals(d["x_ids"])
        np.testing.assert_allclose(m[:, 0], d["engine_mean"], rtol=1e-10)
        np.testing.assert_allclose(m[:, 1], d["engine_variance"], rtol=1e-10)


@pytest.mark.gpu
def test_device_grid_against_the_golden_vector(hip_lib):
    d = load("grid8x8.json")
    model = cx.synth.gaussian_grid(8, 8, seed=1234)
    for schedule in (L.SCHED_FUSED, L.SCHED_FLOODING):
        dev = cx.DeviceGraph(schedule=schedule)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        dev.sweep(5)
        got = dev.get_messages(d["edge_var"], d["edge_fac"], L.TO_VARIABLE)
        keep = ~np.isnan(d["f2v_variance_after_5"])
        np.testing.assert_allclose(got[keep, 0], d["f2v_mean_after_5"][keep], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(got[keep, 1], d["f2v_variance_after_5"][keep], rtol=1e-9)
        dev.sweep(400)
        dev.update_batch([L.ITEM_INDIVIDUAL_MARGINAL] * 64, d["x_ids"], [0] * 64)
        m = dev.get_marginals(d["x_ids"])
        np.testing.assert_allclose(m[:, 0], d["exact_mean"], rtol=1e-9)
        np.testing.assert_allclose(m[:, 1], d["bp_variance_converged"], rtol=1e-9)


@pytest.mark.gpu
def test_device_block_chain_and_vmp_against_the_golden_vectors(hip_lib):
    d = load("lgssm_d4.json")
    model = cx.synth.lgssm_chain(8, d=4, seed=1234)
    assert np.array_equal(model.data_y, d["data_y"])
    dev = cx.DeviceGraph(dim=4, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev)
    dev.sweep(10)
    marg = dev.get_marginals(d["x_ids"])
    np.testing.assert_allclose(marg[:, :4], d["posterior_mean"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(marg[:, 4:].reshape(8, 4, 4), d["posterior_covariance"], rtol=1e-9, atol=1e-12)
    d64 = load("lgssm_d64.json")
    m64 = cx.synth.lgssm_chain(3, d=64, seed=1234)
    h64 = cx.DeviceGraph(dim=64, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m64, h64)
    h64.sweep(6)
    q64 = h64.get_marginals(d64["x_ids"])
    np.testing.assert_allclose(q64[:, :64], d64["posterior_mean"], rtol=1e-8, atol=1e-11)
    cov = q64[:, 64:].reshape(3, 64, 64)
    np.testing.assert_allclose(cov[1], d64["posterior_covariance_middle"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(np.stack([np.diag(c) for c in cov]), d64["posterior_variances"], rtol=1e-8)
    v = load("vmp_n8.json")
    vm = cx.synth.vmp_ssm(8, seed=1234)
    for name, fam in (("mean_field", L.FAMILY_VMP_MEAN_FIELD), ("structured", L.FAMILY_VMP_STRUCTURED)):
        h = cx.DeviceGraph(family=fam, schedule=L.SCHED_CHAIN_SCAN)
        cx.synth.load_vmp_into_device(vm, h)
        for _ in range(5):
            h.update_marginals(vm.x_ids); h.update_marginals([vm.ssnoise, vm.obsnoise])
        q = h.get_marginals(vm.x_ids)
        g = h.get_marginals([vm.ssnoise, vm.obsnoise])
        np.testing.assert_allclose(q[:, 0], v[name]["x_mean"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(q[:, 1], v[name]["x_precision"], rtol=1e-9)
        np.testing.assert_allclose(g.ravel(), v[name]["ssnoise_shape_scale"] + v[name]["obsnoise_shape_scale"], rtol=1e-9)


# ---------------------------------------------------------------------------------------------------------------- the tree schedule
def _tree24_model(d):
    m = cx.synth.tree_model(int(d["n_factors"]), seed=int(d["seed"]), shape=d["shape"], components=int(d["components"]), observe=float(d["observe"]))
    # the seeded generator has not drifted: graph, parameters, priors and data are the file's
    for key, have in (("edge_var", m.edge_var), ("edge_fac", m.edge_fac), ("edge_role", m.edge_role), ("factor_ids", m.factor_ids), ("factor_kind", m.factor_kind),
                      ("factor_params", m.factor_var), ("coef", m.meta["coef"]), ("prior_mean", m.prior_mean), ("prior_variance", m.prior_variance),
                      ("data_var", m.data_var), ("data_y", m.data_y)):
        assert np.array_equal(np.asarray(have), d[key]), key
    return m


def test_tree_vector_plan_execution_and_dense_solve_agree():
    """CPU: the plan of CX_SCHED_TREE (the product's cx_tree_plan.h, compiled without HIP) executed in numpy on the file's model == the
    file's dense-solve posterior"""
    from tests.hostlogic import FlatGraph
    from tests.test_tree_plan import PlanRun, flat_of

    d = load("tree24.json")
    m = _tree24_model(d)
    g = flat_of(m)
    rc, err = g.tree()
    assert rc == L.OK, err
    marg = PlanRun(g, m).run()
    got = np.array([marg[int(i)] for i in d["x_ids"]])
    np.testing.assert_allclose(got[:, 0], d["posterior_mean"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(got[:, 1], d["posterior_variance"], rtol=1e-10)


def test_tree_vector_heavy_path_plan_execution_agrees_too():
    """CPU: the same file under the heavy-path plan (cx_tree_plan.h: build_hp — scans along heavy paths, through the file's factors of
    three to six variables where a path runs through them, light edges as items)"""
    from tests.test_tree_plan import HpRun, flat_of

    d = load("tree24.json")
    m = _tree24_model(d)
    g = flat_of(m)
    rc, err = g.tree_hp()
    assert rc == L.OK, err
    marg = HpRun(g, m).run()
    got = np.array([marg[int(i)] for i in d["x_ids"]])
    np.testing.assert_allclose(got[:, 0], d["posterior_mean"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(got[:, 1], d["posterior_variance"], rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("heavy_paths", ["0", "1"])
def test_device_tree_against_the_golden_vector(hip_lib, monkeypatch, heavy_paths):
    monkeypatch.setenv("CX_TREE_HP", heavy_paths)      # level by level / over heavy paths
    d = load("tree24.json")
    m = _tree24_model(d)
    dev = cx.DeviceGraph(schedule=L.SCHED_TREE)
    cx.synth.load_into_device(m, dev)
    dev.sweep(1)
    assert (dev.tree_heavy_path_stats()["launches"] > 0) == (heavy_paths == "1")
    got = dev.get_marginals(d["x_ids"])
    np.testing.assert_allclose(got[:, 0], d["posterior_mean"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(got[:, 1], d["posterior_variance"], rtol=1e-10)


# ---------------------------------------------------------------------------------------------------------------- d = 8
def test_d8_vector_is_what_the_block_solver_returns():
    d = load("lgssm_d8.json")
    m = cx.synth.lgssm_chain(6, d=8, seed=1234)
    assert np.array_equal(m.data_y, d["data_y"]) and np.array_equal(m.meta["A"], d["A"])
    em, ecov = exact.lgssm_posterior(m.data_y, m.meta["A"], m.meta["Q"], m.meta["R"])
    assert np.array_equal(em, d["posterior_mean"]) and np.array_equal(ecov, d["posterior_covariance"])


@pytest.mark.gpu
def test_device_d8_chain_against_the_golden_vector(hip_lib):
    """a dimension between 4 and 64 (embedded in the dim 64 path): flooding to its fixed point and ONE chain-scan sweep"""
    d = load("lgssm_d8.json")
    m = cx.synth.lgssm_chain(6, d=8, seed=1234)
    for schedule, sweeps in ((L.SCHED_FUSED, 10), (L.SCHED_CHAIN_SCAN, 1)):
        dev = cx.DeviceGraph(dim=8, schedule=schedule)
        cx.synth.load_into_device(m, dev)
        dev.sweep(sweeps)
        marg = dev.get_marginals(d["x_ids"])
        np.testing.assert_allclose(marg[:, :8], d["posterior_mean"], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(marg[:, 8:].reshape(6, 8, 8), d["posterior_covariance"], rtol=1e-8, atol=1e-11)
