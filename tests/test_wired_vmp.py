"""cx_graph_wire with messages that depend on marginals, GPU-free part: the reference's two variational test models
(test/inference_engine_tests.jl:593-805 "Mean Field", :807-1147 "Structured") expressed as WIRINGS — the transcribed resolvers of
tests/vmp_support.py record their add_dependency! / link_signal_to_variable! calls as cx_graph_wire triples — and run by the product's
host logic (cx_refsched.h: the shadow scheduler, the levelling) with a numpy restatement of the device items.  Pinned against the restated
engine (oracle/cortex_ref.c) driven by the same resolvers and the transcribed rules: per update_marginals! call the SAME executions in the
SAME order, and the same marginals — including the structured experiment's last request, which names the states together with the
precisions and is evaluated in an order that emerges from the readiness flags (the fused families refuse it outside class-by-class order)."""
import numpy as np
import pytest

from cortex.jl_amd import _lib as L
from tests import vmp_support as vs
from tests import wired_vmp_support as ws


def _run(be, kind, data, iterations, calls_of=None):
    log = []

    def on_call(it, ids):
        x, y, obsnoise, ssnoise = be._ids
        log.append((list(np.atleast_1d(ids)), be.trace_rows(), [be.get_marginal(v) for v in list(x) + [ssnoise, obsnoise]]))

    # (run_experiment hands the ids back only at the end: the model is built inside it, so the callback reads them from the back-end)
    orig = vs.make_ssm_model

    def make(be_, n, fr, vr):
        x, y, obsnoise, ssnoise = orig(be_, n, fr, vr)
        be_._ids = (x, y, obsnoise, ssnoise)
        return x, y, obsnoise, ssnoise

    vs.make_ssm_model = make
    try:
        out = vs.run_experiment(be, kind, data, iterations, on_call=on_call, calls_of=calls_of)
    finally:
        vs.make_ssm_model = orig
    return out, log


def _compare(log_a, log_b, rtol, what):
    assert len(log_a) == len(log_b)
    for c, ((ids_a, rows_a, marg_a), (ids_b, rows_b, marg_b)) in enumerate(zip(log_a, log_b)):
        assert ids_a == ids_b
        assert rows_a == rows_b, f"{what}, call {c} (request of {len(ids_a)} ids): executions differ; first difference at " \
                                 f"{next((i, a, b) for i, (a, b) in enumerate(zip(rows_a + [None], rows_b + [None])) if a != b)}"
        for i, (a, b) in enumerate(zip(marg_a, marg_b)):
            ws.assert_same_value(a, b, rtol, f"{what}, call {c}, marginal {i}")


@pytest.mark.parametrize("n", [4, 12, 40])
def test_mean_field_wiring_call_by_call(n):
    """MeanFieldResolver (:599-621): every marginal a flat product of all its incoming messages (intermediate), every message weakly
    dependent on the two other marginals of its factor.  n = 40: a precision's marginal lists 39 / 40 dependencies."""
    data = vs.dataset(n, seed=7)
    _, want = _run(ws.TracedOracleBackend(vs.mean_field_rule), "mean_field", data, 3)
    _, got = _run(ws.ShadowBackend(), "mean_field", data, 3)
    _compare(got, want, 1e-11, f"mean field n={n}")
    assert sum(len(r) for _i, r, _m in got) > 0


def test_long_dependency_lists_leave_the_thread_per_item_launch(monkeypatch):
    """a list item of more sources than cx_refsched.h: kWideList is summed by a workgroup of its own (k_wide_sum): here with the threshold
    lowered to 8, so that the flat marginals of the two precisions (19 / 20 messages) take that path — same executions, same values"""
    monkeypatch.setenv("CXH_REF_WIDE_LIST", "8")
    data = vs.dataset(20, seed=7)
    _, want = _run(ws.TracedOracleBackend(vs.mean_field_rule), "mean_field", data, 2)
    be = ws.ShadowBackend()
    _, got = _run(be, "mean_field", data, 2)
    _compare(got, want, 1e-11, "mean field n=20, wide lists")
    assert be.last_wide > 0


@pytest.mark.parametrize("n", [4, 5, 7, 12, 33])
def test_structured_wiring_call_by_call_with_the_mixed_request(n):
    """StructuredResolver (:816-897): the default variable wiring (segment trees above degree 5: n - 1 > 5), mean-field likelihoods, joint
    marginals of neighbouring states linked to both, chain messages with strong dependencies.  All 13 calls per iteration, the mixed last one
    (states + both precisions in one request) included, in every degree regime."""
    data = vs.dataset(n, seed=11)
    _, want = _run(ws.TracedOracleBackend(vs.structured_rule), "structured", data, 3)
    _, got = _run(ws.ShadowBackend(), "structured", data, 3)
    _compare(got, want, 1e-10, f"structured n={n}")
    kinds = {r[0] for _i, rows, _m in got for r in rows}
    assert ws.KJOINT in kinds and (n - 1 <= 5 or ws.KPROD in kinds)


@pytest.mark.parametrize("kind,n", [("mean_field", 9), ("structured", 5), ("structured", 14)])
def test_the_vectorised_wirings_are_the_resolvers_calls(kind, n):
    """cortex.jl_amd.wiring builds the two resolvers' triples with array operations: per signal the same dependencies with the same flags in the
    same order, per variable the same linked signals in the same order — and, run on the shadow, the same executions as the restated engine"""
    data = vs.dataset(n, seed=3)
    by_calls, by_arrays = ws.ShadowBackend(), ws.ShadowBackend()
    by_arrays.vectorised = kind
    _, want = _run(by_calls, kind, data, 2)
    _, got = _run(by_arrays, kind, data, 2)

    def per_signal(be):
        deps, links = {}, {}
        for s, d, fl in zip(be.sig, be.dep, be.flags):
            s, d = tuple(int(x) for x in s), tuple(int(x) for x in d)
            if fl & L.WIRE_LINK:
                links.setdefault(d, []).append(s)
            elif fl & L.WIRE_DEFAULT_VARIABLE:
                deps.setdefault(s, []).append(("default", 0))
            else:
                deps.setdefault(s, []).append((d, int(fl)))
        return deps, links

    assert per_signal(by_arrays) == per_signal(by_calls)
    _compare(got, want, 1e-13, f"{kind} n={n}: array wiring vs resolver calls")
    _, oracle = _run(ws.TracedOracleBackend(vs.mean_field_rule if kind == "mean_field" else vs.structured_rule), kind, data, 2)
    _compare(got, oracle, 1e-10, f"{kind} n={n}: array wiring vs the restated engine")


@pytest.mark.parametrize("kind,n", [("mean_field", 9), ("structured", 5), ("structured", 14)])
def test_a_wiring_read_back_from_the_host_engine(kind, n):
    """cortex.jl_amd.wiring.from_engine: the resolvers run on the host mirror of the reference's API (as a user's resolver would), the
    triples are read back from the signals' dependency lists, nibbles, listen masks and linked signals — and drive the shadow scheduler to
    the restated engine's executions and values"""
    data = vs.dataset(n, seed=5)
    be = ws.ShadowBackend()
    be.vectorised = "from_engine"
    _, got = _run(be, kind, data, 2)
    _, oracle = _run(ws.TracedOracleBackend(vs.mean_field_rule if kind == "mean_field" else vs.structured_rule), kind, data, 2)
    _compare(got, oracle, 1e-10, f"{kind} n={n}: wiring read back from the host engine")


def test_structured_wiring_reaches_the_references_own_assertions():
    """the facts the reference asserts (:1142-1143): both precisions recovered above 90 after 100 iterations on 100 points with true
    precisions 100 — here 40 iterations of the by-class calls on the shadow + numpy items (values only; the order is pinned above)"""
    data = vs.dataset(100, seed=1234)
    be = ws.ShadowBackend()
    out = vs.run_experiment(be, "structured", data, 40, calls_of=vs.structured_calls_by_class)
    assert vs.mean(out["ssnoise"]) > 50 and vs.mean(out["obsnoise"]) > 50


def test_wirings_the_device_has_no_rule_for_are_refused():
    be = ws.ShadowBackend()
    a, b, g = be.add_variable("x"), be.add_variable("x"), be.add_variable("ssnoise")
    f = be.add_factor("transition")
    for v in (a, b, g):
        be.add_edge(v, f)
    be._index()
    from tests.hostlogic import FlatGraph
    ev, ef, role, fids, kinds = be.graph_arrays()
    gph = FlatGraph(ev, ef, fids, kinds, np.zeros(len(fids)), edge_role=role, schedule=L.SCHED_REFERENCE)
    assert gph.status == 0, gph.error
    # a message to a Normal variable that depends on ONE marginal only
    rc, err = gph.ref_wire([be.message_to_variable(a, f)], [be.marginal(g)], [L.WIRE_WEAK])
    assert rc != 0 and "precision's marginal" in err
    # a message to the precision that depends on a message
    rc, err = gph.ref_wire([be.message_to_variable(g, f)], [be.message_to_factor(a, f)], [0])
    assert rc != 0 and "JointMarginal" in err
    # a joint marginal with two of its three inputs
    rc, err = gph.ref_wire([(ws.KJOINT, 0, f)] * 2, [be.message_to_factor(a, f), be.message_to_factor(b, f)], [L.WIRE_WEAK] * 2)
    assert rc != 0 and "JointMarginal" in err
    # a linked signal that is not a joint marginal
    rc, err = gph.ref_wire([be.message_to_factor(a, f)], [be.marginal(a)], [L.WIRE_LINK])
    assert rc != 0 and "CX_WIRE_LINK" in err
    # the same factor outside the reference-order schedule, and a precision that is also a Normal variable elsewhere
    g2 = FlatGraph(ev, ef, fids, kinds, np.zeros(len(fids)), edge_role=role, schedule=L.SCHED_FUSED)
    assert g2.status != 0 and "variational rules only" in g2.error
    ev2, ef2, role2 = np.append(ev, g), np.append(ef, f + 1), np.append(role, 0)
    ev2, ef2, role2 = np.append(ev2, a), np.append(ef2, f + 1), np.append(role2, 1)
    g3 = FlatGraph(ev2, ef2, np.append(fids, f + 1), np.append(kinds, 1), np.zeros(len(fids) + 1), edge_role=role2, schedule=L.SCHED_REFERENCE)
    assert g3.status != 0 and "Gamma-distributed" in g3.error


def _run_tree(be, K, seed, iterations):
    m = ws.make_tree_model(be, K, seed)
    log = []
    for it in range(1, iterations + 1):
        for c, ids in enumerate(ws.tree_calls(m, it)):
            if c != 2:      # (one call per iteration finds the priors as the previous call left them)
                ws.set_priors(be, m)
            be.update_marginals(ids)
            log.append((list(ids), be.trace_rows(), [be.get_marginal(v) for v in m.x + m.tp + m.op]))
    return m, log


@pytest.mark.parametrize("K,seed", [(6, 0), (25, 1), (60, 2)])
def test_a_tree_of_means_with_grouped_precisions_call_by_call(K, seed):
    """a third model under the same machinery: states on a random tree (degrees above 5: segment trees), two transition and two
    observation precisions, Normal / Gamma priors as opaque unary factors; structured wiring.  Every call — by class, one precision alone,
    reversed request order, states and precisions in one request — equals the restated engine: executions and marginals."""
    _, want = _run_tree(ws.TracedOracleBackend(vs.structured_rule), K, seed, 4)
    _, got = _run_tree(ws.ShadowBackend(factor_kinds={"prior": ws.FACTOR_OPAQUE}), K, seed, 4)
    _compare(got, want, 1e-9, f"tree K={K}")
    assert all(len(rows) > 0 for _ids, rows, _m in got[:5]), "every call of the first iteration computes something"
    assert {ws.KJOINT, ws.KPROD} <= {r[0] for _i, rows, _m in got for r in rows} or K < 10


def test_the_tree_model_by_class_is_dense_coordinate_ascent():
    """[states], [precisions], ... on the wired model is coordinate ascent on q(x) q(precisions): the states' marginals are those of the
    K-variate Gaussian with the expected precisions (belief propagation is exact on the tree), a precision's Gamma collects the expected
    squared differences under the pairwise marginals — checked against dense linear algebra after every pair of calls"""
    K = 40
    be = ws.ShadowBackend(factor_kinds={"prior": ws.FACTOR_OPAQUE})
    m = ws.make_tree_model(be, K, seed=5)
    dense = ws.DenseTreeVMP(m)
    for it in range(12):
        ws.set_priors(be, m); be.update_marginals(list(m.x)); dense.update_x()
        got = np.array([be.get_marginal(v)[1] for v in m.x])
        np.testing.assert_allclose(got[:, 0], dense.mu, rtol=1e-9, atol=1e-12, err_msg=f"iteration {it}: state means")
        np.testing.assert_allclose(1.0 / got[:, 1], np.diag(dense.Sigma), rtol=1e-9, err_msg=f"iteration {it}: state variances")
        ws.set_priors(be, m); be.update_marginals(m.tp + m.op); dense.update_precisions()
        got = np.array([be.get_marginal(v)[1] for v in m.tp + m.op])
        np.testing.assert_allclose(got, np.array(dense.tp + dense.op), rtol=1e-9, err_msg=f"iteration {it}: precisions")
    est = [s * c for s, c in dense.tp + dense.op]
    assert all(e > 5.0 for e in est), est      # (the estimates move towards the generating precisions 25, 100, 50, 200)
