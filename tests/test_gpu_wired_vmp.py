"""-m gpu: cx_graph_wire with messages that depend on marginals on the device.

The reference's two variational test models, re-expressed as WIRINGS under CX_SCHED_REFERENCE (the transcribed resolvers record their
add_dependency! / link_signal_to_variable! calls as triples: tests/wired_vmp_support.py), and a third model — a tree of latent means with
grouped unknown precisions and priors — run call by call against the restated engine (oracle/cortex_ref.c driven by the transcribed rules):
the same executions in the same order (cx_ref_trace) and the same marginals after EVERY update_marginals! call, the structured experiment's
mixed last request included.  At scale the wired handle is compared with the fused family handles (cx_update_marginals), and the tree model
with dense coordinate ascent.  The GPU-free half (scheduler + numpy items) is tests/test_wired_vmp.py."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from tests import vmp_support as vs
from tests import wired_vmp_support as ws
from tests.test_wired_vmp import _compare, _run, _run_tree

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [4, 12, 60])
def test_mean_field_wiring_on_the_device(hip_lib, n):
    data = vs.dataset(n, seed=7)
    _, want = _run(ws.TracedOracleBackend(vs.mean_field_rule), "mean_field", data, 3)
    be = ws.DeviceBackend()
    _, got = _run(be, "mean_field", data, 3)
    _compare(got, want, 1e-10, f"mean field n={n}")
    st = be.dev.ref_plan_stats()
    assert st["hits"] > 0, st      # the calls of the second and third iteration replay standing plans


@pytest.mark.parametrize("n", [4, 5, 7, 12, 33, 100])
def test_structured_wiring_on_the_device_with_the_mixed_request(hip_lib, n):
    iters = 3 if n < 100 else 6
    data = vs.dataset(n, seed=11)
    _, want = _run(ws.TracedOracleBackend(vs.structured_rule), "structured", data, iters)
    be = ws.DeviceBackend()
    _, got = _run(be, "structured", data, iters)
    _compare(got, want, 1e-9, f"structured n={n}")
    jm, jc = be.dev.get_joint_marginals(np.arange(3 * n + 3, 4 * n + 2))      # the transition factors (ids after 2 + 2 n variables and n likelihoods)
    assert jm.shape == (n - 1, 2) and not np.any(np.isnan(jm)) and not np.any(np.isnan(jc))
    # the joint marginals of neighbouring states agree with the states' own marginals where the two were computed from the same messages:
    # the last call of an iteration names everything, the states last
    xs = np.array([be.get_marginal(v)[1] for v in be._ids[0]])
    np.testing.assert_allclose(jm[:, 0], xs[:-1, 0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(jm[:, 1], xs[1:, 0], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("n,chain_min", [(12, 3), (33, 3), (100, 8), (300, None)])
def test_structured_wiring_with_its_chains_as_scan_steps(hip_lib, monkeypatch, n, chain_min):
    """(round 6) the states' call of the structured wiring is the reference's forward / backward pass: two chains of pairs whose follower is
    the structured rule N(mean m, 1 / (var m + 1 / E[precision])).  They run as prefix scans (cx_planscan.hip) inside the reference-order
    plan — the executions the library reports and every marginal after every call stay the restated engine's, the mixed request included;
    the plan of the states' call is a handful of stages instead of n."""
    if chain_min is not None:
        monkeypatch.setenv("CX_REF_CHAIN_MIN", str(chain_min))      # (read when a plan is levelled; the default is 128 pairs)
    iters = 3
    data = vs.dataset(n, seed=21)
    _, want = _run(ws.TracedOracleBackend(vs.structured_rule), "structured", data, iters)
    be = ws.DeviceBackend()
    _, got = _run(be, "structured", data, iters)
    _compare(got, want, 1e-9, f"structured n={n}, chains as scans")
    st = be.dev.ref_plan_stats()      # the last call of the experiment: the states together with both precisions
    assert st["executions"] >= 4 * (n - 1) and st["stages"] <= 40, st


@pytest.mark.parametrize("K,seed", [(6, 0), (25, 1), (200, 2)])
def test_a_tree_of_means_with_grouped_precisions_on_the_device(hip_lib, K, seed):
    _, want = _run_tree(ws.TracedOracleBackend(vs.structured_rule), K, seed, 4)
    _, got = _run_tree(ws.DeviceBackend(factor_kinds={"prior": ws.FACTOR_OPAQUE}), K, seed, 4)
    _compare(got, want, 1e-9, f"tree K={K}")


def test_the_tree_model_on_the_device_is_dense_coordinate_ascent(hip_lib):
    K = 300
    be = ws.DeviceBackend(factor_kinds={"prior": ws.FACTOR_OPAQUE})
    m = ws.make_tree_model(be, K, seed=9)
    dense = ws.DenseTreeVMP(m)
    for it in range(10):
        ws.set_priors(be, m); be.update_marginals(list(m.x)); dense.update_x()
        got = be.dev.get_marginals(m.x)
        np.testing.assert_allclose(got[:, 0], dense.mu, rtol=1e-8, atol=1e-11, err_msg=f"iteration {it}: state means")
        np.testing.assert_allclose(got[:, 1], np.diag(dense.Sigma), rtol=1e-8, err_msg=f"iteration {it}: state variances")
        ws.set_priors(be, m); be.update_marginals(m.tp + m.op); dense.update_precisions()
        np.testing.assert_allclose(be.dev.get_marginals(m.tp + m.op), np.array(dense.tp + dense.op), rtol=1e-8, err_msg=f"iteration {it}: precisions")
    st = be.dev.ref_plan_stats()
    assert st["hits"] >= 14 and st["misses"] <= 6, st      # two standing plans in the steady state


@pytest.mark.parametrize("kind,n", [("mean_field", 12), ("structured", 5), ("structured", 12)])
def test_a_users_resolver_through_the_plugin(hip_lib, kind, n):
    """the drop-in boundary for a USER resolver: the model engine, the transcribed resolver and HipProcessor(mode = "reference") go into
    InferenceEngine as they would in the reference; the resolver wires the HOST engine's signals, the processor reads that wiring back
    and hands it to the device.  Every update_marginals! of the reference's experiment equals the restated engine: executions and marginals."""
    data = vs.dataset(n, seed=13)
    rule = vs.mean_field_rule if kind == "mean_field" else vs.structured_rule
    _, want = _run(ws.TracedOracleBackend(rule), kind, data, 3)
    _, got = _run(ws.PluginBackend(), kind, data, 3)
    _compare(got, want, 1e-9, f"plug-in, {kind} n={n}")


def test_the_tree_model_through_the_plugin(hip_lib):
    _, want = _run_tree(ws.TracedOracleBackend(vs.structured_rule), 25, 1, 3)
    _, got = _run_tree(ws.PluginBackend(), 25, 1, 3)
    _compare(got, want, 1e-9, "plug-in, tree K=25")


def _wired_ssm(model, kind):
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    nf = len(model.factor_ids)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, np.full(nf, L.FACTOR_NORMAL_PRECISION, dtype=np.int32), np.zeros(nf), edge_role=model.edge_role)
    if kind == "mean_field":
        t = cx.wiring.mean_field(model.edge_var, model.edge_fac, model.edge_role)
    else:
        t = cx.wiring.structured(model.edge_var, model.edge_fac, model.edge_role, clustered_factors=model.factor_ids[model.n:])
    dev.graph_wire(t.signals, t.dependencies, t.flags)
    dev.set_marginals([model.ssnoise, model.obsnoise], L.FORM_GAMMA, [1.0, 1.0, 1.0, 1.0])
    dev.set_marginals(model.x_ids, L.FORM_MEAN_PRECISION, np.tile([0.0, 1.0], model.n))
    dev.set_marginals(model.y_ids, L.FORM_POINT, model.data_y)
    return dev


@pytest.mark.parametrize("kind,family,n", [("mean_field", L.FAMILY_VMP_MEAN_FIELD, 3000), ("structured", L.FAMILY_VMP_STRUCTURED, 3000),
                                           ("structured", L.FAMILY_VMP_STRUCTURED, 50_000)])
def test_the_wired_models_equal_the_fused_families_at_scale(hip_lib, kind, family, n):
    """the same model as a wiring (vectorised triples, cortex.jl_amd.wiring) and as a fused family handle: equal marginals after every
    by-class call.  (Mean field at 3,000: a precision's marginal is ONE product over 3,000 messages there — a flat dependency list is a
    serial sum on the device; the structured wiring's segment trees are the parallel form.)"""
    model = cx.synth.vmp_ssm(n, seed=21)
    wired = _wired_ssm(model, kind)
    fused = cx.DeviceGraph(family=family, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_vmp_into_device(model, fused)
    seq = [model.x_ids, [model.ssnoise], [model.obsnoise], [model.obsnoise, model.ssnoise], model.x_ids, [model.ssnoise, model.obsnoise]]
    for it in range(3):
        for ids in seq:
            wired.sweep_for(ids); fused.update_marginals(ids)
            a, b = wired.get_marginals(model.x_ids), fused.get_marginals(model.x_ids)
            np.testing.assert_allclose(a[:, 0], b[:, 0], rtol=1e-8, atol=1e-10, err_msg=f"{kind} n={n} it={it}: state means")
            np.testing.assert_allclose(1.0 / a[:, 1], b[:, 1], rtol=1e-8, err_msg=f"{kind} n={n} it={it}: state precisions")
            np.testing.assert_allclose(wired.get_marginals([model.ssnoise, model.obsnoise]), fused.get_marginals([model.ssnoise, model.obsnoise]), rtol=1e-8,
                                       err_msg=f"{kind} n={n} it={it}: precisions")
    st = wired.ref_plan_stats()
    assert st["hits"] > 0, st


def test_a_wired_handle_round_trips_through_a_checkpoint(hip_lib):
    """messages, marginals, segment-tree nodes, joint marginals and the readiness shadow travel: a handle restored in the middle of an
    iteration continues exactly as the original"""
    model = cx.synth.vmp_ssm(40, seed=2)
    a = _wired_ssm(model, "structured")
    seq = [model.x_ids, [model.ssnoise], [model.obsnoise], [model.ssnoise, model.obsnoise] + list(model.x_ids)]
    for ids in seq[:2]:
        a.sweep_for(ids)
    blob = a.export_state()
    b = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    nf = len(model.factor_ids)
    b.graph_create(model.edge_var, model.edge_fac, model.factor_ids, np.full(nf, L.FACTOR_NORMAL_PRECISION, dtype=np.int32), np.zeros(nf), edge_role=model.edge_role)
    t = cx.wiring.structured(model.edge_var, model.edge_fac, model.edge_role, clustered_factors=model.factor_ids[model.n:])
    b.graph_wire(t.signals, t.dependencies, t.flags)
    b.import_state(blob)
    for ids in seq[2:] + seq:
        a.sweep_for(ids); b.sweep_for(ids)
        assert a.ref_trace() == b.ref_trace()
        assert np.array_equal(a.get_marginals(model.x_ids), b.get_marginals(model.x_ids))
        assert np.array_equal(a.get_marginals([1, 2]), b.get_marginals([1, 2]))
    for p, q in zip(a.get_joint_marginals(model.factor_ids[model.n:]), b.get_joint_marginals(model.factor_ids[model.n:])):
        assert np.array_equal(p, q)


def test_an_imported_readiness_state_is_state_and_keeps_the_wirings_own_flags(hip_lib):
    """ a checkpoint's readiness section carries the dynamic Computed / Fresh bits only: the Intermediate / Weak bits are the
    wiring's, so a blob written under another wiring is refused instead of changing which dependencies count as weak; and an import counts
    as "a value was set": a cx_graph_wire after it is refused instead of silently throwing the imported readiness away."""
    model = cx.synth.vmp_ssm(12, seed=4)
    a = _wired_ssm(model, "structured")
    a.sweep_for(model.x_ids)
    blob = a.export_state()
    nf = len(model.factor_ids)

    def fresh():
        d = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        d.graph_create(model.edge_var, model.edge_fac, model.factor_ids, np.full(nf, L.FACTOR_NORMAL_PRECISION, dtype=np.int32), np.zeros(nf), edge_role=model.edge_role)
        return d
    t = cx.wiring.structured(model.edge_var, model.edge_fac, model.edge_role, clustered_factors=model.factor_ids[model.n:])
    b = fresh()
    b.graph_wire(t.signals, t.dependencies, t.flags)
    b.import_state(blob)
    with pytest.raises(cx.CortexHipError, match="wiring is fixed"):
        b.graph_wire(t.signals, t.dependencies, t.flags)
    # the same triples with every weak dependency made strong: the same sizes, other static bits
    c = fresh()
    strong = np.asarray(t.flags).copy() & ~np.int32(L.WIRE_WEAK)
    assert np.any(strong != np.asarray(t.flags))
    try:
        c.graph_wire(t.signals, t.dependencies, strong)
    except cx.CortexHipError:
        return                  # (a wiring no rule serves is refused at wiring time: nothing to import into)
    with pytest.raises(cx.CortexHipError, match="does not fit this handle's wiring"):
        c.import_state(blob)
    # ... and another resolver's wiring (other dependency lists)
    d = fresh()
    m = cx.wiring.mean_field(model.edge_var, model.edge_fac, model.edge_role)
    d.graph_wire(m.signals, m.dependencies, m.flags)
    with pytest.raises(cx.CortexHipError):
        d.import_state(blob)


def test_refusals_on_the_device(hip_lib):
    model = cx.synth.vmp_ssm(6, seed=1)
    nf = len(model.factor_ids)
    kinds = np.full(nf, L.FACTOR_NORMAL_PRECISION, dtype=np.int32)
    for sched in (L.SCHED_FUSED, L.SCHED_FLOODING, L.SCHED_TREE):
        dev = cx.DeviceGraph(schedule=sched)
        with pytest.raises(cx.CortexHipError, match="variational rules only"):
            dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, kinds, np.zeros(nf), edge_role=model.edge_role)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, kinds, np.zeros(nf), edge_role=model.edge_role)
    # under the DEFAULT wiring the factor's messages would depend on messages: no rule — the call says so when one becomes pending
    dev.set_marginals(model.y_ids, L.FORM_POINT, model.data_y)
    dev.set_messages(model.y_ids, model.factor_ids[:model.n], L.TO_FACTOR, L.FORM_POINT, model.data_y)
    dev.set_messages(np.full(model.n, model.obsnoise), model.factor_ids[:model.n], L.TO_FACTOR, L.FORM_NATURAL, np.tile([0.0, 1.0], model.n))
    with pytest.raises(cx.CortexHipError, match="wire its variational dependencies"):
        dev.sweep_for(model.x_ids[:1])
    # marginal forms by variable kind
    with pytest.raises(cx.CortexHipError, match="is a precision"):
        dev.set_marginals([model.ssnoise], L.FORM_MOMENT, [1.0, 1.0])
    with pytest.raises(cx.CortexHipError, match="is a Normal variable"):
        dev.set_marginals(model.x_ids[:1], L.FORM_GAMMA, [1.0, 1.0])
    # a wiring after a value was set
    t = cx.wiring.mean_field(model.edge_var, model.edge_fac, model.edge_role)
    with pytest.raises(cx.CortexHipError, match="wiring is fixed"):
        dev.graph_wire(t.signals, t.dependencies, t.flags)
