"""The product's host-side mirror of the Cortex API (cortex.jl_amd/{signal,inference_signal,model_engine,
dependencies,inference_engine}.py) against (i) the reference's engine-level known answers, written the way the
reference's tests are written, and (ii) the C restatement of the oracle: identical wiring and identical execution
order on random graphs ("schedule/indexing bit-exact")."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import InferenceSignalVariants as V
from cortex.jl_amd import (BipartiteFactorGraph, Connection, Factor, InferenceEngine, Variable, get_value,
                           get_variable_marginal, request_inference_for, scan_inference_request, set_value,
                           update_marginals)
from oracle import ref


class NMV:
    def __init__(self, mean, variance):
        self.mean, self.variance = mean, variance


def product(l, r):  # test/runtests.jl:40-46, same operation order
    xi = l.mean / l.variance + r.mean / r.variance
    w = 1 / l.variance + 1 / r.variance
    variance = 1 / w
    return NMV(variance * xi, variance)


class SSMBeliefPropagationProcessor(cx.AbstractInferenceRequestProcessor):
    """test/inference_engine_tests.jl:383-432"""

    def _fold(self, deps):
        acc = get_value(deps[0])
        for d in deps[1:]:
            acc = product(acc, get_value(d))
        return acc

    def compute_individual_marginal(self, engine, variant, signal, dependencies):
        return self._fold(dependencies)

    compute_product_of_messages = compute_individual_marginal
    compute_message_to_factor = compute_individual_marginal

    def compute_message_to_variable(self, engine, variant, signal, dependencies):
        assert len(dependencies) == 1
        q = engine.get_factor(variant.factor_id).functional_form.variance
        x = get_value(dependencies[0])
        if isinstance(x, (int, float)):
            return NMV(x, q)
        return NMV(x.mean, x.variance + q)


def make_ssm(n, processor, trace=False):
    """make_ssm_model, test/inference_engine_tests.jl:436-462"""
    graph = BipartiteFactorGraph()
    x = [graph.add_variable(Variable(name="x", index=(i,))) for i in range(1, n + 1)]
    y = [graph.add_variable(Variable(name="y", index=(i,))) for i in range(1, n + 1)]
    likelihood = [graph.add_factor(Factor(functional_form=cx.GaussianAdditive(1.0))) for _ in range(n)]
    transition = [graph.add_factor(Factor(functional_form=cx.GaussianAdditive(1.0))) for _ in range(n - 1)]
    for i in range(n):
        graph.add_edge(y[i], likelihood[i], Connection(label="out"))
        graph.add_edge(x[i], likelihood[i], Connection(label="out"))
    for i in range(n - 1):
        graph.add_edge(x[i], transition[i], Connection(label="out"))
        graph.add_edge(x[i + 1], transition[i], Connection(label="in"))
    engine = InferenceEngine(model_engine=graph, dependency_resolver=cx.DefaultDependencyResolver(),
                             inference_request_processor=processor, trace=trace)
    return engine, x, y, likelihood, transition


def test_warning_for_isolated_variable():
    graph = BipartiteFactorGraph()
    v = graph.add_variable(Variable(name="v"))
    engine = InferenceEngine(model_engine=graph)
    assert len(engine.get_warnings()) == 1
    assert engine.get_warnings()[0].description == "Variable has no connected factors"
    assert engine.get_warnings()[0].context == v


def test_unsupported_model_engine():
    """test/model_engine_tests.jl:114-169"""
    with pytest.raises(cx.UnsupportedModelEngineError):
        InferenceEngine(model_engine=object())


def test_scan_known_answers():
    """test/inference_engine_tests.jl:93-239"""
    def small():
        graph = BipartiteFactorGraph()
        f1 = graph.add_factor(Factor(functional_form="left"))
        f2 = graph.add_factor(Factor(functional_form="right"))
        vc = graph.add_variable(Variable(name="center"))
        graph.add_edge(vc, f1, Connection(label="param"))
        graph.add_edge(vc, f2, Connection(label="param"))
        return graph, f1, f2, vc

    graph, f1, f2, vc = small()
    assert scan_inference_request(request_inference_for(InferenceEngine(model_engine=graph), vc)) == []
    for which, expect in (("l", 1), ("r", 1), ("lr", 2)):
        graph, f1, f2, vc = small()
        engine = InferenceEngine(model_engine=graph, resolve_dependencies=False)
        vm = get_variable_marginal(engine.get_variable(vc))
        left, right = cx.create_inference_signal(), cx.create_inference_signal()
        cx.add_dependency(engine.get_connection_message_to_variable(vc, f1), left)
        cx.add_dependency(engine.get_connection_message_to_variable(vc, f2), right)
        cx.add_dependency(vm, engine.get_connection_message_to_variable(vc, f1))
        cx.add_dependency(vm, engine.get_connection_message_to_variable(vc, f2))
        if "l" in which:
            set_value(left, 1.0)
        if "r" in which:
            set_value(right, 1.0)
        steps = scan_inference_request(request_inference_for(engine, vc))
        assert len(steps) == expect
        want = [engine.get_connection_message_to_variable(vc, f) for f, c in ((f1, "l"), (f2, "r")) if c in which]
        assert all(a is b for a, b in zip(steps, want))


def test_ssm_bp_like_the_reference_test():
    """test/inference_engine_tests.jl:464-488 (+ exact smoother, + 5n−4 executions)."""
    from oracle import exact
    n = 100
    rng = np.random.default_rng(1234)
    dataset = [2 * i + rng.standard_normal() for i in range(1, n + 1)]
    engine, x, y, likelihood, transition = make_ssm(n, SSMBeliefPropagationProcessor(), trace=True)
    for i in range(n):
        set_value(engine.get_connection_message_to_factor(y[i], likelihood[i]), dataset[i])
    update_marginals(engine, x)
    answer = [get_value(get_variable_marginal(engine.get_variable(v))) for v in x]
    means = np.array([a.mean for a in answer]); variances = np.array([a.variance for a in answer])
    assert np.all(means >= 0) and np.all(np.diff(means) >= 0) and np.all(variances >= 0)
    em, ev = exact.ssm_chain_posterior(dataset, 1.0, 1.0)
    np.testing.assert_allclose(means, em, rtol=1e-12)
    np.testing.assert_allclose(variances, ev, rtol=1e-12)
    rounds = engine.get_trace().inference_requests[0].rounds
    assert len(rounds) == 3 and sum(len(r.executions) for r in rounds) == 5 * n - 4 + n


def test_tracing_known_answer():
    """test/inference_engine_tests.jl:1149-1261"""
    class P(cx.AbstractInferenceRequestProcessor):
        def compute_message_to_variable(self, engine, variant, signal, dependencies):
            ff = engine.get_factor(variant.factor_id).functional_form
            assert ff in ("likelihood1", "likelihood2")
            return 2 * get_value(dependencies[0])

        def compute_individual_marginal(self, engine, variant, signal, dependencies):
            return sum(get_value(d) for d in dependencies)

    graph = BipartiteFactorGraph()
    p = graph.add_variable(Variable(name="p"))
    o1 = graph.add_variable(Variable(name="y1")); o2 = graph.add_variable(Variable(name="y2"))
    fp = graph.add_factor(Factor(functional_form="prior"))
    f1 = graph.add_factor(Factor(functional_form="likelihood1")); f2 = graph.add_factor(Factor(functional_form="likelihood2"))
    graph.add_edge(p, fp, Connection(label="out")); graph.add_edge(p, f1, Connection(label="in")); graph.add_edge(p, f2, Connection(label="in"))
    graph.add_edge(o1, f1, Connection(label="out")); graph.add_edge(o2, f2, Connection(label="out"))
    engine = InferenceEngine(model_engine=graph, inference_request_processor=P(), trace=True)
    set_value(engine.get_connection_message_to_factor(o1, f1), 1)
    set_value(engine.get_connection_message_to_factor(o2, f2), 2)
    set_value(engine.get_connection_message_to_variable(p, fp), 3)
    update_marginals(engine, p)
    assert get_value(get_variable_marginal(engine.get_variable(p))) == 9
    req = engine.get_trace().inference_requests[0]
    assert len(req.rounds) == 2
    ex = req.rounds[0].executions
    assert [e.signal.variant for e in ex] == [V.MessageToVariable(p, f1), V.MessageToVariable(p, f2)]
    assert all(isinstance(e.value_before_execution, cx.UndefValue) for e in ex)
    assert [e.value_after_execution for e in ex] == [2, 4]
    (m,) = req.rounds[1].executions
    assert m.signal.variant == V.IndividualMarginal(p) and m.value_after_execution == 9


def test_unimplemented_rule_and_nonpending_compute_errors():
    """inference_engine.jl:358 (error(...) for a missing rule), signal.jl:399-405 (ArgumentError)."""
    graph = BipartiteFactorGraph()
    v = graph.add_variable(Variable(name="v")); o = graph.add_variable(Variable(name="o"))
    f = graph.add_factor(Factor(functional_form="f"))
    graph.add_edge(v, f, Connection(label="out")); graph.add_edge(o, f, Connection(label="out"))

    class Empty(cx.AbstractInferenceRequestProcessor):
        pass

    engine = InferenceEngine(model_engine=graph, inference_request_processor=Empty())
    set_value(engine.get_connection_message_to_factor(o, f), 1.0)
    with pytest.raises(NotImplementedError, match="compute_message_to_variable!"):
        update_marginals(engine, v)
    s = cx.Signal()
    with pytest.raises(ValueError, match="not pending"):
        cx.compute(lambda sig, deps: 1, s)
    cx.compute(lambda sig, deps: 1, s, force=True)
    assert get_value(s) == 1


def _random_graph(rng, n_var, n_fac, loopy):
    """random bipartite graph with pairwise + unary factors; returns (edges, factor arity)."""
    edges = []
    if not loopy:  # random tree over variables, each tree edge becomes a pairwise factor
        pairs = [(int(rng.integers(0, i)), i) for i in range(1, n_var)]
    else:
        pairs = set()
        while len(pairs) < n_fac:
            a, b = rng.integers(0, n_var, 2)
            if a != b:
                pairs.add((int(min(a, b)), int(max(a, b))))
        pairs = sorted(pairs)
    return pairs


@pytest.mark.parametrize("seed,loopy", [(0, False), (1, False), (2, False), (3, True), (4, True)])
def test_wiring_and_execution_order_equal_the_c_restatement(seed, loopy):
    rng = np.random.default_rng(seed)
    n_var = int(rng.integers(4, 30))
    pairs = _random_graph(rng, n_var, n_var + 5, loopy)
    # python mirror
    graph = BipartiteFactorGraph()
    E = ref.Engine(ref.P_SSM_BP, trace=True)
    xs = [graph.add_variable(Variable(name="x", index=(i,))) for i in range(n_var)]
    assert xs == [E.add_variable() for _ in range(n_var)]
    obs = [graph.add_variable(Variable(name="y", index=(i,))) for i in range(n_var)]
    assert obs == [E.add_variable() for _ in range(n_var)]
    liks, facs = [], []
    for i in range(n_var):
        q = float(rng.uniform(0.5, 2.0))
        f = graph.add_factor(Factor(functional_form=cx.GaussianAdditive(q)))
        assert f == E.add_factor(ref.F_GAUSS_ADD, q)
        liks.append(f)
    for _ in pairs:
        q = float(rng.uniform(0.5, 2.0))
        f = graph.add_factor(Factor(functional_form=cx.GaussianAdditive(q)))
        assert f == E.add_factor(ref.F_GAUSS_ADD, q)
        facs.append(f)
    for i in range(n_var):
        graph.add_edge(obs[i], liks[i], Connection(label="out")); E.add_edge(obs[i], liks[i])
        graph.add_edge(xs[i], liks[i], Connection(label="out")); E.add_edge(xs[i], liks[i])
    for (a, b), f in zip(pairs, facs):
        graph.add_edge(xs[a], f, Connection(label="out")); E.add_edge(xs[a], f)
        graph.add_edge(xs[b], f, Connection(label="in")); E.add_edge(xs[b], f)
    engine = InferenceEngine(model_engine=graph, inference_request_processor=SSMBeliefPropagationProcessor(), trace=True)
    E.finalize()

    def tag(sig):
        v = sig.variant
        if isinstance(v, V.MessageToFactor): return (ref.VAR_MSG_TO_FACTOR, v.variable_id, v.factor_id)
        if isinstance(v, V.MessageToVariable): return (ref.VAR_MSG_TO_VARIABLE, v.variable_id, v.factor_id)
        if isinstance(v, V.IndividualMarginal): return (ref.VAR_MARGINAL, v.variable_id, 0)
        if isinstance(v, V.ProductOfMessages): return (ref.VAR_PRODUCT, v.variable_id, 0)
        raise AssertionError(v)

    # identical wiring: dependency lists, in order, of every message and marginal
    for vid in xs:
        for fid in graph.get_connected_factor_ids(vid):
            for get_py, get_c in ((engine.get_connection_message_to_variable, E.message_to_variable),
                                  (engine.get_connection_message_to_factor, E.message_to_factor)):
                py = [tag(d) for d in get_py(vid, fid).dependencies]
                c = [E.variant(d)[:3] if E.variant(d)[0] != ref.VAR_MARGINAL else (ref.VAR_MARGINAL, E.variant(d)[1], 0)
                     for d in E.dependencies(get_c(vid, fid))]
                c = [(k, v, f if k not in (ref.VAR_PRODUCT,) else 0) for k, v, f in c]
                assert py == c
    data = rng.standard_normal(n_var)
    for i in range(n_var):
        set_value(engine.get_connection_message_to_factor(obs[i], liks[i]), float(data[i]))
        E.set_value(E.message_to_factor(obs[i], liks[i]), float(data[i]))
    if loopy:  # seed every pairwise factor→variable message so that the loops can start
        for (a, b), f in zip(pairs, facs):
            for v in (xs[a], xs[b]):
                set_value(engine.get_connection_message_to_variable(v, f), NMV(0.0, 100.0))
                E.set_value(E.message_to_variable(v, f), (ref.NORMAL, 0.0, 100.0))
    for _call in range(3 if loopy else 1):
        update_marginals(engine, xs)
        E.update_marginals(xs)
        py_trace = [(ri, tag(e.signal)) for ri, r in enumerate(engine.get_trace().inference_requests[-1].rounds) for e in r.executions]
        c_trace = []
        for r, _v, s, _b, _a in E.trace():
            k, v, f, _, _ = E.variant(s)
            c_trace.append((r, (k, v, f if k != ref.VAR_MARGINAL else 0)))
        assert py_trace == c_trace           # schedule parity: same signals, same order, same rounds
        for vid in xs:
            m = get_value(get_variable_marginal(engine.get_variable(vid)))
            t, a, b = E.get_value(E.marginal(vid))
            if isinstance(m, cx.UndefValue):
                assert t == ref.UNDEF
            else:
                assert (m.mean, m.variance) == (a, b)   # bit-identical: same arithmetic, same order


# ---- test/model_engine_tests.jl:1-112 and test/ext/bipartite_factor_graphs_ext_tests.jl: the graph data structures and the
#      accessors HipProcessor.attach ingests a model through — assertions transcribed, on the host mirror
def test_variable_factor_connection_data_structures():
    from cortex.jl_amd.model_engine import (add_local_marginal_to_factor, get_connection_message_to_factor,
                                           get_connection_message_to_variable, get_factor_functional_form, get_factor_local_marginals,
                                           get_variable_linked_signals, link_signal_to_variable)
    for name in ("v", "v1", "v2", "v3"):                                  # model_engine_tests.jl:1-21
        v = Variable(name=name)
        assert v.name == name and v.index is None
        assert isinstance(get_variable_marginal(v), cx.Signal) and isinstance(get_variable_linked_signals(v), list)
    for index in (1, 2, 3):
        v = Variable(name="v", index=index)
        assert v.name == "v" and v.index == index and isinstance(get_variable_marginal(v), cx.Signal)
    assert get_variable_linked_signals(Variable(name="v")) == []          # :23-29
    external = cx.create_inference_signal()                               # :31-39
    assert get_variable_marginal(Variable(name="v", marginal=external)) is external
    v1, other = Variable(name="v1"), cx.create_inference_signal()         # :41-52
    link_signal_to_variable(v1, other)
    assert any(s is other for s in get_variable_linked_signals(v1))
    for form in ("f", "g", "h"):                                          # :54-63
        f = Factor(functional_form=form)
        assert get_factor_functional_form(f) == form and isinstance(get_factor_local_marginals(f), list)
    f = Factor(functional_form="f")                                       # :65-84
    assert get_factor_local_marginals(f) == []
    lm = cx.create_inference_signal()
    add_local_marginal_to_factor(f, lm)
    assert any(s is lm for s in get_factor_local_marginals(f))
    for label in ("c", "d", "e"):                                         # :86-104
        for index in (1, 2, 3):
            c = Connection(label=label, index=index)
            assert c.label == label and c.index == index
            assert isinstance(get_connection_message_to_variable(c), cx.Signal) and isinstance(get_connection_message_to_factor(c), cx.Signal)
    assert Connection(label="c").index == 0                               # :106-112


def test_bipartite_factor_graph_backend_accessors():
    """test/ext/bipartite_factor_graphs_ext_tests.jl:1-93"""
    from cortex.jl_amd.model_engine import get_connection_message_to_factor, get_connection_message_to_variable, get_factor_functional_form
    for kw in ({}, {"resolve_dependencies": False}, {"prepare_signals_metadata": False}):
        assert isinstance(InferenceEngine(model_engine=BipartiteFactorGraph(), **kw), InferenceEngine)
    graph = BipartiteFactorGraph()
    a = graph.add_variable(Variable(name="a")); b = graph.add_variable(Variable(name="b", index=(1,))); c = graph.add_variable(Variable(name="c", index=(2, 3)))
    f1 = graph.add_factor(Factor(functional_form="f1")); f2 = graph.add_factor(Factor(functional_form="f2"))
    graph.add_edge(a, f1, Connection(label="out")); graph.add_edge(b, f2, Connection(label="theta"))
    engine = InferenceEngine(model_engine=graph)
    assert [engine.get_variable(v).name for v in (a, b, c)] == ["a", "b", "c"]
    assert [engine.get_variable(v).index for v in (a, b, c)] == [None, (1,), (2, 3)]
    assert all(isinstance(get_variable_marginal(engine.get_variable(v)), cx.Signal) for v in (a, b, c))
    assert get_factor_functional_form(engine.get_factor(f1)) == "f1" and get_factor_functional_form(engine.get_factor(f2)) == "f2"
    for v, f, label in ((a, f1, "out"), (b, f2, "theta")):
        conn = engine.get_connection(v, f)
        assert isinstance(conn, Connection) and conn.label == label
        assert engine.get_connection_message_to_variable(v, f) is get_connection_message_to_variable(conn)
        assert engine.get_connection_message_to_factor(v, f) is get_connection_message_to_factor(conn)
    for v, f in ((a, f2), (b, f1)):                                       # :84-86: no such connection
        with pytest.raises(Exception):
            engine.get_connection(v, f)
    assert set(engine.get_variable_ids()) == {a, b, c} and set(engine.get_factor_ids()) == {f1, f2}
    assert set(engine.get_connected_variable_ids(f1)) == {a} and set(engine.get_connected_variable_ids(f2)) == {b}
    assert set(engine.get_connected_factor_ids(a)) == {f1} and set(engine.get_connected_factor_ids(b)) == {f2} and set(engine.get_connected_factor_ids(c)) == set()


def test_isa_variant_with_inference_signal_variants():
    """test/inference_engine_tests.jl:10-31 (the JET / allocation lines are Julia's)"""
    V = cx.InferenceSignalVariants
    c = cx.create_inference_signal()
    assert cx.isa_variant(c, V.Unspecified)
    cx.set_variant(c, V.MessageToVariable(1, 2))
    assert cx.isa_variant(c, V.MessageToVariable) and not cx.isa_variant(c, V.MessageToFactor)
    cx.set_variant(c, V.MessageToFactor(1, 2))
    assert cx.isa_variant(c, V.MessageToFactor) and not cx.isa_variant(c, V.MessageToVariable)


def test_resolve_dependencies_visits_every_variable_and_factor():
    """test/dependencies_tests.jl:1-37: a custom resolver sees all of them"""
    from cortex.jl_amd.dependencies import resolve_dependencies

    class Recording(cx.AbstractDependencyResolver):
        def __init__(self):
            self.variables, self.factors = set(), set()

        def resolve_variable_dependencies(self, engine, variable_id):
            self.variables.add(variable_id)

        def resolve_factor_dependencies(self, engine, factor_id):
            self.factors.add(factor_id)

    graph = BipartiteFactorGraph()
    x, y, z = (graph.add_variable(Variable(name=n)) for n in "xyz")
    f1, f2 = graph.add_factor(Factor(functional_form="f1")), graph.add_factor(Factor(functional_form="f2"))
    engine = InferenceEngine(model_engine=graph)
    r = Recording()
    resolve_dependencies(r, engine)
    assert r.variables == {x, y, z} and r.factors == {f1, f2}


def test_signal_variant_defaults_and_updates():
    """test/signal_tests.jl:23-52 "Signal Variant" (default values, custom variant; the typed-Signal testset is Julia's type system)"""
    s = cx.Signal(42)
    assert isinstance(cx.get_variant(s), cx.UndefVariant) and isinstance(cx.get_variant(cx.Signal()), cx.UndefVariant)
    cx.set_variant(s, 1)
    assert cx.get_variant(s) == 1 and cx.isa_variant(s, int) and not cx.isa_variant(s, str)
    cx.set_variant(s, "2")
    assert cx.get_variant(s) == "2" and cx.isa_variant(s, str) and not cx.isa_variant(s, int)
    s = cx.Signal(42, variant=1)
    assert cx.get_variant(s) == 1
    cx.set_variant(s, 2); assert cx.get_variant(s) == 2
    cx.set_variant(s, "3"); assert cx.get_variant(s) == "3"
