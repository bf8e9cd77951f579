"""-m gpu: the drop-in itself — the reference's SSM test (test/inference_engine_tests.jl:379-488) with the processor
swapped for HipProcessor, in its three modes.  The host keeps Signals/readiness/scheduler (cortex.jl_amd mirror of
src/), the arithmetic runs on the MI355X through the C ABI."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import InferenceSignalVariants as V
from cortex.jl_amd import _lib as L
from cortex.jl_amd import get_value, get_variable_marginal, update_marginals
from oracle import exact, ref
from tests.helpers import assert_close
from tests.test_host_mirror import SSMBeliefPropagationProcessor, make_ssm

pytestmark = pytest.mark.gpu


def _dataset(n, seed=1234):
    rng = np.random.default_rng(seed)
    return [2 * i + rng.standard_normal() for i in range(1, n + 1)]


def _run(n, processor):
    dataset = _dataset(n)
    engine, x, y, likelihood, transition = make_ssm(n, processor, trace=True)
    for i in range(n):
        sig = engine.get_connection_message_to_factor(y[i], likelihood[i])
        if isinstance(processor, cx.HipProcessor):
            processor.set_value(sig, dataset[i])
        else:
            cx.set_value(sig, dataset[i])
    update_marginals(engine, x)
    answer = [get_value(get_variable_marginal(engine.get_variable(v))) for v in x]
    return engine, x, np.array([a.mean for a in answer]), np.array([a.variance for a in answer]), dataset


def test_per_signal_mode_is_the_reference_schedule(hip_lib):
    n = 60
    proc = cx.HipProcessor(mode="per_signal")
    engine, x, mean, var, dataset = _run(n, proc)
    assert np.all(mean >= 0) and np.all(np.diff(mean) >= 0) and np.all(var >= 0)   # the reference's own assertions
    em, ev = exact.ssm_chain_posterior(dataset, 1.0, 1.0)
    assert_close(mean, em, 1e-9, "marginal mean vs exact smoother")
    assert_close(var, ev, 1e-9, "marginal variance vs exact smoother")
    # same execution order as the CPU processor driving the same scheduler, one launch per signal
    engine_cpu, _, mean_cpu, var_cpu, _ = _run(n, SSMBeliefPropagationProcessor())
    order_cpu = [e.signal.variant for r in engine_cpu.get_trace().inference_requests[0].rounds for e in r.executions]
    assert proc.execution_log == order_cpu and proc.launches == 5 * n - 4 + n
    assert_close(mean, mean_cpu, 1e-9, "device vs reference-arithmetic processor")
    assert_close(var, var_cpu, 1e-9, "device vs reference-arithmetic processor")


def test_wavefront_mode_batches_independent_signals(hip_lib):
    n = 300
    proc = cx.HipProcessor(mode="wavefront")
    engine, x, mean, var, dataset = _run(n, proc)
    em, ev = exact.ssm_chain_posterior(dataset, 1.0, 1.0)
    assert_close(mean, em, 1e-9, "wavefront marginal mean")
    assert_close(var, ev, 1e-9, "wavefront marginal variance")
    assert sorted(map(repr, proc.execution_log)) == sorted(
        map(repr, [e.signal.variant for r in _run(n, SSMBeliefPropagationProcessor())[0].get_trace().inference_requests[0].rounds
                   for e in r.executions]))      # the same set of signals, each computed exactly once
    assert proc.launches <= 2 * n + 4            # O(depth) launches instead of 6n


def test_sweep_mode_reaches_the_same_marginals(hip_lib):
    n = 40
    proc = cx.HipProcessor(mode="sweep", n_sweeps=n + 2)
    engine, x, mean, var, dataset = _run(n, proc)
    em, ev = exact.ssm_chain_posterior(dataset, 1.0, 1.0)
    assert_close(mean, em, 1e-9, "sweep marginal mean")
    assert_close(var, ev, 1e-9, "sweep marginal variance")
    # messages are readable through the same accessors a reference user would use
    m = proc.read(V.MessageToVariable(x[3], engine.get_connected_factor_ids(x[3])[-1]))
    assert np.isfinite(m.mean) and m.variance > 0
    # the engine was built with trace = true: a whole-call takeover leaves one request with one round holding the
    # requested marginals in request order, undefined before and device-held after
    tr = engine.get_trace().inference_requests
    assert len(tr) == 1 and len(tr[0].rounds) == 1
    ex = tr[0].rounds[0].executions
    assert [e.variable_id for e in ex] == list(x)
    assert all(isinstance(e.value_before_execution, cx.UndefValue) for e in ex)
    assert all(isinstance(e.value_after_execution, cx.HipValue) for e in ex)
    assert ex[7].value_after_execution.mean == pytest.approx(em[7], rel=1e-9)


@pytest.mark.parametrize("schedule", ["tree", "chain_scan"])
def test_sweep_mode_with_an_exact_schedule_needs_one_sweep(hip_lib, schedule):
    """update_marginals! taken over whole: with CX_SCHED_TREE (any forest) or CX_SCHED_CHAIN_SCAN (paths) ONE device sweep is the
    reference's result — the fused schedule above needs n + 2"""
    n = 300
    proc = cx.HipProcessor(mode="sweep", n_sweeps=1, schedule=cx._lib.SCHED_TREE if schedule == "tree" else cx._lib.SCHED_CHAIN_SCAN)
    engine, x, mean, var, dataset = _run(n, proc)
    em, ev = exact.ssm_chain_posterior(dataset, 1.0, 1.0)
    assert_close(mean, em, 1e-9, f"{schedule}: marginal mean after one sweep")
    assert_close(var, ev, 1e-9, f"{schedule}: marginal variance after one sweep")


def test_linear_gaussian_factor_rule(hip_lib):
    """x_{t+1} = a x_t + b + N(0, q): the device's GAUSS_LINEAR rule against the exact posterior."""
    n, a, b, q, r = 30, 0.9, 0.3, 0.5, 0.7
    rng = np.random.default_rng(5)
    ys = rng.standard_normal(n) * 2
    graph = cx.BipartiteFactorGraph()
    x = [graph.add_variable(cx.Variable(name="x", index=(i,))) for i in range(n)]
    y = [graph.add_variable(cx.Variable(name="y", index=(i,))) for i in range(n)]
    lik = [graph.add_factor(cx.Factor(functional_form=cx.GaussianAdditive(r))) for _ in range(n)]
    tr = [graph.add_factor(cx.Factor(functional_form=cx.GaussianLinear(a, b, q))) for _ in range(n - 1)]
    for i in range(n):
        graph.add_edge(y[i], lik[i], cx.Connection(label="out")); graph.add_edge(x[i], lik[i], cx.Connection(label="out"))
    for i in range(n - 1):
        graph.add_edge(x[i], tr[i], cx.Connection(label="in")); graph.add_edge(x[i + 1], tr[i], cx.Connection(label="out"))
    proc = cx.HipProcessor(mode="wavefront")
    engine = cx.InferenceEngine(model_engine=graph, inference_request_processor=proc)
    for i in range(n):
        proc.set_value(engine.get_connection_message_to_factor(y[i], lik[i]), float(ys[i]))
    update_marginals(engine, x)
    got = np.array([[get_value(get_variable_marginal(engine.get_variable(v))).mean,
                     get_value(get_variable_marginal(engine.get_variable(v))).variance] for v in x])
    # exact: precision J, potential h of the joint
    J = np.zeros((n, n)); h = np.zeros(n)
    for i in range(n):
        J[i, i] += 1 / r; h[i] += ys[i] / r
    for i in range(n - 1):   # (x_{i+1} - a x_i - b)^2 / q
        J[i, i] += a * a / q; J[i + 1, i + 1] += 1 / q; J[i, i + 1] -= a / q; J[i + 1, i] -= a / q
        h[i] -= a * b / q; h[i + 1] += b / q
    S = np.linalg.inv(J)
    assert_close(got[:, 0], S @ h, 1e-9, "linear-Gaussian marginal mean")
    assert_close(got[:, 1], np.diag(S), 1e-9, "linear-Gaussian marginal variance")


def test_unsupported_variant_is_an_error_not_a_crash(hip_lib):
    proc = cx.HipProcessor(mode="per_signal")
    with pytest.raises(NotImplementedError):
        proc._item(V.Unspecified())


# ------------------------------------------------------------------------------------------------ a15: ProductOfMessages
class BetaBernoulliProcessor(cx.AbstractInferenceRequestProcessor):
    """test/inference_engine_tests.jl:245-309, on the host (the CPU arm of the comparison)"""

    def compute_message_to_variable(self, engine, variant, signal, dependencies):
        assert engine.get_factor(variant.factor_id).functional_form == "bernoulli"
        r = float(bool(get_value(dependencies[0])))
        return cx.Beta(1.0 + r, 2.0 - r)

    def _fold(self, engine, variant, signal, dependencies):
        acc = get_value(dependencies[0])
        for d in dependencies[1:]:
            nxt = get_value(d)
            acc = cx.Beta(acc.a + nxt.a - 1, acc.b + nxt.b - 1)
        return acc

    compute_individual_marginal = _fold
    compute_product_of_messages = _fold


def make_beta_bernoulli(n, processor, trace=True):
    """make_beta_bernoulli_model, test/inference_engine_tests.jl:311-337"""
    graph = cx.BipartiteFactorGraph()
    p = graph.add_variable(cx.Variable(name="p"))
    o, f = [], []
    for i in range(1, n + 1):
        oi = graph.add_variable(cx.Variable(name="o", index=(i,)))
        fi = graph.add_factor(cx.Factor(functional_form="bernoulli"))
        o.append(oi); f.append(fi)
        graph.add_edge(p, fi, cx.Connection(label="out"))
        graph.add_edge(oi, fi, cx.Connection(label="out"))
    engine = cx.InferenceEngine(model_engine=graph, dependency_resolver=cx.DefaultDependencyResolver(),
                                inference_request_processor=processor, trace=trace)
    return engine, p, o, f


def _beta_experiment(n, processor, data):
    engine, p, o, f = make_beta_bernoulli(n, processor)
    for i in range(n):
        sig = engine.get_connection_message_to_factor(o[i], f[i])
        if isinstance(processor, cx.HipProcessor):
            processor.set_value(sig, bool(data[i]))
        else:
            cx.set_value(sig, bool(data[i]))
    update_marginals(engine, [p])
    return engine, p, get_value(get_variable_marginal(engine.get_variable(p)))


@pytest.mark.parametrize("mode", ["per_signal", "wavefront"])
@pytest.mark.parametrize("n", [100, 6, 13])
def test_beta_bernoulli_model_through_the_plugin(hip_lib, mode, n):
    """The reference's Beta-Bernoulli test (test/inference_engine_tests.jl:241-377, n = 100) under DefaultDependencyResolver:
    the variable p has degree n > 5, so its marginal hangs off the segment tree of ProductOfMessages signals
    (dependencies.jl:90-173).  Every process! — MessageToVariable through the :bernoulli rule, ProductOfMessages,
    IndividualMarginal — is a device computation (CX_ITEM_*); the posterior is the exact Beta(1 + Σ, 1 + n − Σ)."""
    rng = np.random.default_rng(1234 + n)
    data = rng.random(n) < 0.5
    proc = cx.HipProcessor(mode=mode, family="beta", schedule=cx._lib.SCHED_FLOODING)
    engine, p, answer = _beta_experiment(n, proc, data)
    assert (answer.a, answer.b) == (1.0 + data.sum(), 1.0 + n - data.sum())          # known answer, exact (:360-376)
    engine_cpu, _, answer_cpu = _beta_experiment(n, BetaBernoulliProcessor(), data)
    assert (answer_cpu.a, answer_cpu.b) == (answer.a, answer.b)
    order_cpu = [e.signal.variant for r in engine_cpu.get_trace().inference_requests[0].rounds for e in r.executions]
    kinds = {type(v).__name__ for v in order_cpu}
    assert kinds == {"MessageToVariable", "ProductOfMessages", "IndividualMarginal"}
    if mode == "per_signal":
        assert proc.execution_log == order_cpu and proc.launches == len(order_cpu)     # the reference's execution order
    else:
        assert sorted(map(repr, proc.execution_log)) == sorted(map(repr, order_cpu))   # same signals, each exactly once
        assert proc.launches <= 2 + int(np.ceil(np.log2(n))) + 2                       # O(tree depth) launches
    # every segment-tree intermediate holds on the device what the CPU processor computed for the same signal
    prods_cpu = {(e.signal.variant.range): get_value(e.signal) for r in engine_cpu.get_trace().inference_requests[0].rounds
                 for e in r.executions if isinstance(e.signal.variant, V.ProductOfMessages)}
    assert len(prods_cpu) >= 2
    for (lo, hi), want in prods_cpu.items():
        got = proc.read(V.ProductOfMessages(p, (lo, hi), ()))
        assert (got.a, got.b) == (want.a, want.b), f"ProductOfMessages {lo}:{hi}"
    nat = proc.dev.get_products([p], [1], [n], cx._lib.FORM_NATURAL)      # a node nobody computed is UndefValue()
    assert np.all(np.isnan(nat))


def test_product_of_messages_and_joint_marginal_items_gaussian(hip_lib):
    """CX_ITEM_PRODUCT_OF_MESSAGES / CX_ITEM_JOINT_MARGINAL straight through the C ABI on a Gaussian model: a hub variable
    observed through 12 noisy factors (degree 13 with its chain link: the segment-tree case) on a short chain.
    Products = precision-weighted products of the named message ranges; the joint marginal of a transition factor's two
    variables = the 2 x 2 block of the exact posterior."""
    L = cx._lib
    T, m = 6, 12
    rng = np.random.default_rng(3)
    x = np.arange(1, T + 1)                       # chain x_1..x_T
    tr = 100 + np.arange(T - 1)                   # transition factors
    yv = 200 + np.arange(T)                       # one observation per state ...
    lk = 300 + np.arange(T)
    hub_obs = 400 + np.arange(m)                  # ... and 12 more of x_3
    hub_lk = 500 + np.arange(m)
    ev = np.concatenate([x[:-1], x[1:], yv, x, hub_obs, np.full(m, x[2])])
    ef = np.concatenate([tr, tr, lk, lk, hub_lk, hub_lk])
    fids = np.concatenate([tr, lk, hub_lk])
    q = np.concatenate([rng.uniform(0.5, 2, T - 1), rng.uniform(0.5, 2, T), rng.uniform(0.5, 2, m)])
    dev = cx.DeviceGraph(schedule=L.SCHED_FLOODING)
    dev.graph_create(ev, ef, fids, np.full(len(fids), L.FACTOR_GAUSS_ADDITIVE, np.int32), q)
    ys, hub_y = rng.standard_normal(T) * 2, rng.standard_normal(m) + 1.0
    dev.set_messages(np.concatenate([yv, hub_obs]), np.concatenate([lk, hub_lk]), L.TO_FACTOR, L.FORM_POINT, np.concatenate([ys, hub_y]))
    dev.sweep(T + 2)                               # a tree: converged
    hub = int(x[2])
    facs = np.sort(ef[ev == hub])                  # neighbour order = ascending factor id
    assert len(facs) == m + 3
    msgs = dev.get_messages(np.full(len(facs), hub), facs, L.TO_VARIABLE)
    ranges = [(1, 7), (8, 15), (1, 3), (4, 7), (5, 5), (1, 15)]
    dev.update_batch([L.ITEM_PRODUCT_OF_MESSAGES] * len(ranges), [hub] * len(ranges), [L.item_range(a, b) for a, b in ranges])
    got = dev.get_products([hub] * len(ranges), [a for a, _ in ranges], [b for _, b in ranges])
    for (a, b), g in zip(ranges, got):
        w = (1.0 / msgs[a - 1:b, 1]).sum()
        xi = (msgs[a - 1:b, 0] / msgs[a - 1:b, 1]).sum()
        assert g[0] == pytest.approx(xi / w, rel=1e-12) and g[1] == pytest.approx(1.0 / w, rel=1e-12), (a, b)
    marg = dev.get_marginals([hub])[0]
    assert got[-1][0] == pytest.approx(marg[0], rel=1e-12) and got[-1][1] == pytest.approx(marg[1], rel=1e-12)   # 1:deg == marginal
    with pytest.raises(cx.CortexHipError):
        dev.update_batch([L.ITEM_PRODUCT_OF_MESSAGES], [hub], [L.item_range(3, 16)])       # range outside 1:15
    # joint marginals of every transition factor vs the exact posterior covariance blocks
    dev.update_batch([L.ITEM_JOINT_MARGINAL] * (T - 1), [0] * (T - 1), tr)
    jm, jc = dev.get_joint_marginals(tr)
    J = np.zeros((T, T)); h = np.zeros(T)
    for i in range(T):
        J[i, i] += 1 / q[T - 1 + i]; h[i] += ys[i] / q[T - 1 + i]
    for k in range(m):
        J[2, 2] += 1 / q[2 * T - 1 + k]; h[2] += hub_y[k] / q[2 * T - 1 + k]
    for i in range(T - 1):
        J[i, i] += 1 / q[i]; J[i + 1, i + 1] += 1 / q[i]; J[i, i + 1] -= 1 / q[i]; J[i + 1, i] -= 1 / q[i]
    S = np.linalg.inv(J); mu = S @ h
    for i in range(T - 1):
        np.testing.assert_allclose(jm[i], mu[i:i + 2], rtol=1e-10)
        np.testing.assert_allclose(jc[i], S[i:i + 2, i:i + 2], rtol=1e-10)
    # a likelihood factor: one variable observed -> degenerate joint (datum, 0 variance) x the latent's posterior given it
    dev.update_batch([L.ITEM_JOINT_MARGINAL], [0], [int(lk[0])])
    jm1, jc1 = dev.get_joint_marginals([int(lk[0])])
    assert jm1[0][1] == ys[0] and jc1[0][1, 1] == 0.0 and jc1[0][0, 1] == 0.0            # x_1 (id 1) first, y (id 200) second
    assert jm1[0][0] == pytest.approx(mu[0], rel=1e-10) and jc1[0][0, 0] == pytest.approx(S[0, 0], rel=1e-10)
    assert np.all(np.isnan(dev.get_joint_marginals([int(lk[1])])[0]))                      # never computed: UndefValue()


def test_a_batch_that_fails_half_way_leaves_readable_stores(hip_lib):
    """the first items of a failing batch are indexed before the error is found; nothing was computed, so reading them
    back gives UndefValue() (this used to read past the end of a store that was never grown)"""
    L = cx._lib
    dev = cx.DeviceGraph(schedule=L.SCHED_FLOODING)
    m = cx.synth.ssm_chain(40)
    cx.synth.load_into_device(m, dev)
    dev.sweep(45)                                  # a chain without priors: defined everywhere after T sweeps
    v = int(m.x_ids[5])
    fac = int(np.sort(m.edge_fac[m.edge_var == v])[-1])
    with pytest.raises(cx.CortexHipError):
        dev.update_batch([L.ITEM_PRODUCT_OF_MESSAGES, L.ITEM_JOINT_MARGINAL, L.ITEM_MESSAGE_TO_FACTOR], [v, 0, v], [L.item_range(1, 2), fac, 987654321])
    assert np.all(np.isnan(dev.get_products([v], [1], [2])))
    assert np.all(np.isnan(dev.get_joint_marginals([fac])[0]))
    dev.update_batch([L.ITEM_PRODUCT_OF_MESSAGES, L.ITEM_JOINT_MARGINAL], [v, 0], [L.item_range(1, 2), fac])    # and the next good batch works
    assert np.all(np.isfinite(dev.get_products([v], [1], [2])))
    assert np.all(np.isfinite(dev.get_joint_marginals([fac])[0]))


def test_unknown_item_kind_is_an_error_not_a_crash(hip_lib):
    dev = cx.DeviceGraph()
    m = cx.synth.ssm_chain(4)
    cx.synth.load_into_device(m, dev)
    with pytest.raises(cx.CortexHipError) as e:
        dev.update_batch([32], [int(m.x_ids[0])], [0])
    assert e.value.code == cx._lib.ERR_UNSUPPORTED


# ------------------------------------------------------------------------------------------------ config C1 at its stated size
@pytest.mark.parametrize("mode", ["per_signal", "wavefront"])
def test_config_c1_T1000_through_the_plugin(hip_lib, mode):
    """BASELINE.json configs[0]: the scalar-Gaussian chain (Kalman smoother) at T = 1,000 (3,998 edges) through the
    processor plug-in: the host keeps the reference's scheduler and readiness bits, every process! is a device computation.
    Execution order == the same scheduler driving the reference-arithmetic CPU processor; marginals == Thomas solve."""
    n = 1000
    proc = cx.HipProcessor(mode=mode)
    engine, x, mean, var, dataset = _run(n, proc)
    em, ev = exact.ssm_chain_posterior(dataset, 1.0, 1.0)
    assert_close(mean, em, 1e-9, "C1 marginal mean vs exact smoother")
    assert_close(var, ev, 1e-9, "C1 marginal variance vs exact smoother")
    assert np.all(mean >= 0) and np.all(np.diff(mean) >= 0) and np.all(var >= 0)        # the reference's own assertions (:485-487)
    engine_cpu = _run(n, SSMBeliefPropagationProcessor())[0]
    order_cpu = [e.signal.variant for r in engine_cpu.get_trace().inference_requests[0].rounds for e in r.executions]
    assert len(order_cpu) == 5 * n - 4 + n                                                 # 4,996 messages + 1,000 marginals
    if mode == "per_signal":
        assert proc.execution_log == order_cpu and proc.launches == len(order_cpu)
    else:
        assert sorted(map(repr, proc.execution_log)) == sorted(map(repr, order_cpu))
        assert proc.launches <= 2 * n + 4


def test_config_c1_through_the_plugin_in_reference_mode(hip_lib):
    """mode "reference": update_marginals! = ONE cx_sweep_for under CX_SCHED_REFERENCE — the reference's own executions (what a trace = true
    engine records, signal by signal) replayed on the device in one launch; a second call with new data replays the standing plan"""
    from cortex.jl_amd import _lib as L

    n = 1000
    proc = cx.HipProcessor(mode="reference")
    engine, x, mean, var, dataset = _run(n, proc)
    em, ev = exact.ssm_chain_posterior(dataset, 1.0, 1.0)
    assert_close(mean, em, 1e-9, "C1 marginal mean vs exact smoother")
    assert_close(var, ev, 1e-9, "C1 marginal variance vs exact smoother")
    engine_cpu = _run(n, SSMBeliefPropagationProcessor())[0]
    order_cpu = [e.signal.variant for r in engine_cpu.get_trace().inference_requests[0].rounds for e in r.executions]
    kinds = {V.MessageToFactor: L.ITEM_MESSAGE_TO_FACTOR, V.MessageToVariable: L.ITEM_MESSAGE_TO_VARIABLE, V.IndividualMarginal: L.ITEM_INDIVIDUAL_MARGINAL}
    want = [(kinds[type(v)], int(v.variable_id), int(getattr(v, "factor_id", 0) or 0), 0, 0) for v in order_cpu]
    assert proc.dev.ref_trace() == want, "the library's execution trace == the host scheduler's with the CPU processor"
    st = proc.dev.ref_plan_stats()
    # (round 6) the forward and the backward pass are chains of pairs: two scan steps beside a handful of stages, not 1,001 stages
    assert proc.launches == 1 and st["executions"] == 5 * n - 4 + n and st["stages"] <= 4 and st["launches"] <= 6, st
    # new data, same request: the readiness state before the call is the one the second plan was made for from then on
    rng = np.random.default_rng(9)
    for it in range(3):
        ys = [float(2 * i + rng.standard_normal()) for i in range(1, n + 1)]
        for i in range(n):
            proc.set_value(engine.get_connection_message_to_factor(engine.get_variable_ids()[n + i], engine.get_factor_ids()[i]), ys[i])
        update_marginals(engine, x)
        got = np.array([[get_value(get_variable_marginal(engine.get_variable(v))).mean, get_value(get_variable_marginal(engine.get_variable(v))).variance] for v in x[::97]])
        em, ev = exact.ssm_chain_posterior(ys, 1.0, 1.0)
        assert_close(got[:, 0], em[::97], 1e-9, f"new data {it}: mean"); assert_close(got[:, 1], ev[::97], 1e-9, f"new data {it}: variance")
    st = proc.dev.ref_plan_stats()
    assert st["hits"] >= 1 and st["plans"] <= 3


# ------------------------------------------------------------------------------------------------ dim > 1 behind the plug-in
def _make_mv_ssm(n, d, processor, A, Q, R, trace=True):
    """the graph of make_ssm_model (test/inference_engine_tests.jl:436-462) with d-dimensional variables and linear-Gaussian
    factors: likelihood y_t = x_t + N(0, R), transition x_{t+1} = A x_t + N(0, Q)"""
    graph = cx.BipartiteFactorGraph()
    x = [graph.add_variable(cx.Variable(name="x", index=(i,))) for i in range(1, n + 1)]
    y = [graph.add_variable(cx.Variable(name="y", index=(i,))) for i in range(1, n + 1)]
    lik_form, tr_form = cx.MvGaussianLinear(np.eye(d), R), cx.MvGaussianLinear(A, Q)
    likelihood = [graph.add_factor(cx.Factor(functional_form=lik_form)) for _ in range(n)]
    transition = [graph.add_factor(cx.Factor(functional_form=tr_form)) for _ in range(n - 1)]
    for i in range(n):
        graph.add_edge(y[i], likelihood[i], cx.Connection(label="out"))
        graph.add_edge(x[i], likelihood[i], cx.Connection(label="in"))
    for i in range(n - 1):
        graph.add_edge(x[i], transition[i], cx.Connection(label="in"))
        graph.add_edge(x[i + 1], transition[i], cx.Connection(label="out"))
    engine = cx.InferenceEngine(model_engine=graph, dependency_resolver=cx.DefaultDependencyResolver(),
                                inference_request_processor=processor, trace=trace)
    return engine, x, y, likelihood, transition


@pytest.mark.parametrize("d,n,mode,tol", [(4, 50, "per_signal", 1e-9), (4, 50, "wavefront", 1e-9), (2, 7, "per_signal", 1e-9),
                                          (64, 6, "per_signal", 1e-8), (64, 6, "wavefront", 1e-8), (4, 50, "reference", 1e-9), (3, 9, "reference", 1e-9)])
def test_d_dimensional_chain_through_the_plugin(hip_lib, d, n, mode, tol):
    """dim > 1 behind the plug-in boundary.  The host mirror's scheduler (readiness bits and all) drives the device
    one signal (or one wavefront) at a time through cx_update_batch; execution order == the same scheduler on the same graph shape
    with the scalar reference-arithmetic processor (the order depends on the graph only); marginals == the exact smoother; the
    messages == oracle/mv.py at its fixed point."""
    from oracle.mv import MvFlood

    model = cx.synth.lgssm_chain(n, d=d, seed=9)
    A, Q, R = model.meta["A"], model.meta["Q"], model.meta["R"]
    proc = cx.HipProcessor(mode=mode, dim=d)
    engine, x, y, lik, tr = _make_mv_ssm(n, d, proc, A, Q, R)
    for i in range(n):
        proc.set_value(engine.get_connection_message_to_factor(y[i], lik[i]), model.data_y[i])
    update_marginals(engine, x)
    vals = [get_value(get_variable_marginal(engine.get_variable(v))) for v in x]
    mean = np.stack([v.mean for v in vals]); cov = np.stack([v.covariance for v in vals])
    em, ecov = exact.lgssm_posterior(model.data_y, A, Q, R)
    assert_close(mean, em, tol, f"d={d} {mode}: marginal means vs exact smoother", scale_by="max")
    assert_close(cov, ecov, tol, f"d={d} {mode}: marginal covariances vs exact smoother", scale_by="max")
    # the schedule: the same signals, and in per-signal mode the same ORDER, as the scalar SSM under the CPU processor
    engine_cpu, xs, *_ = _run(n, SSMBeliefPropagationProcessor())
    order_cpu = [e.signal.variant for r in engine_cpu.get_trace().inference_requests[0].rounds for e in r.executions]
    if mode == "per_signal":
        assert proc.execution_log == order_cpu and proc.launches == 5 * n - 4 + n
    elif mode == "reference":      # ONE cx_sweep_for: the library's own trace is the host scheduler's order
        from cortex.jl_amd import _lib as L
        kinds = {V.MessageToFactor: L.ITEM_MESSAGE_TO_FACTOR, V.MessageToVariable: L.ITEM_MESSAGE_TO_VARIABLE, V.IndividualMarginal: L.ITEM_INDIVIDUAL_MARGINAL}
        assert proc.dev.ref_trace() == [(kinds[type(v)], int(v.variable_id), int(getattr(v, "factor_id", 0) or 0), 0, 0) for v in order_cpu]
        assert proc.launches == 1
    else:
        assert sorted(map(repr, proc.execution_log)) == sorted(map(repr, order_cpu)) and proc.launches <= 2 * n + 4
    # messages into the latent variables vs the numpy restatement at its fixed point (ids coincide: same construction order)
    o = MvFlood(model)
    o.sweep(n + 2)
    g = o.g
    xi = set(np.searchsorted(g.var_ids, model.x_ids).tolist())
    for e in np.flatnonzero(g.partner >= 0):
        if int(np.searchsorted(g.var_ids, g.edge_var[e])) not in xi:
            continue
        got = proc.read(V.MessageToVariable(int(g.edge_var[e]), int(g.edge_fac[e])))
        m, S = o.f2v[e]
        assert_close(got.mean, m, tol, f"f2v mean edge {e}", scale_by="max"); assert_close(got.covariance, S, tol, f"f2v covariance edge {e}", scale_by="max")


@pytest.mark.parametrize("d,children,mode", [(4, 5, "per_signal"), (3, 7, "wavefront"), (64, 5, "per_signal"), (4, 100, "wavefront"), (2, 12, "per_signal"),
                                             (4, 100, "reference"), (2, 12, "reference"),
                                             (64, 40, "wavefront"), (64, 11, "per_signal"), (16, 9, "wavefront"), (64, 40, "reference")])      # (round 6) dim 64 — and 5 .. 63 inside it — at any degree
def test_a_d_dimensional_hub_through_the_plugin(hip_lib, d, children, mode):
    """a state with `children` child states, everybody observed: the hub has degree children + 1 > 5, so the reference's default
    resolver hangs its messages and its marginal off a segment tree of ProductOfMessages signals (src/dependencies.jl:90-173).  The host
    mirror's scheduler drives all of them — the product nodes included — through cx_update_batch; the marginals are the joint solve's"""
    rng = np.random.default_rng(17)
    A = 0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    Q, R = 0.2 * np.eye(d), np.eye(d)
    n = children + 1
    proc = cx.HipProcessor(mode=mode, dim=d)
    graph = cx.BipartiteFactorGraph()
    x = [graph.add_variable(cx.Variable(name="x", index=(i,))) for i in range(1, n + 1)]
    y = [graph.add_variable(cx.Variable(name="y", index=(i,))) for i in range(1, n + 1)]
    lik_form, tr_form = cx.MvGaussianLinear(np.eye(d), R), cx.MvGaussianLinear(A, Q)
    lik = [graph.add_factor(cx.Factor(functional_form=lik_form)) for _ in range(n)]
    tr = [graph.add_factor(cx.Factor(functional_form=tr_form)) for _ in range(children)]
    for i in range(n):
        graph.add_edge(y[i], lik[i], cx.Connection(label="out"))
        graph.add_edge(x[i], lik[i], cx.Connection(label="in"))
    for c in range(children):
        graph.add_edge(x[0], tr[c], cx.Connection(label="in"))
        graph.add_edge(x[c + 1], tr[c], cx.Connection(label="out"))
    engine = cx.InferenceEngine(model_engine=graph, dependency_resolver=cx.DefaultDependencyResolver(), inference_request_processor=proc, trace=True)
    data = rng.standard_normal((n, d))
    for i in range(n):
        proc.set_value(engine.get_connection_message_to_factor(y[i], lik[i]), data[i])
    update_marginals(engine, x)
    if mode == "reference":      # ONE cx_sweep_for: the call's executions are the device's (cx_ref_trace), the segment tree's nodes among them
        assert proc.launches == 1 and any(r[0] == L.ITEM_PRODUCT_OF_MESSAGES for r in proc.dev.ref_trace())
    else:
        assert any(isinstance(v, V.ProductOfMessages) for v in proc.execution_log), "the segment tree's nodes were processed on the device"
    vals = [get_value(get_variable_marginal(engine.get_variable(v))) for v in x]
    # joint solve
    Qi, Ri = np.linalg.inv(Q), np.linalg.inv(R)
    J = np.zeros((n * d, n * d)); hv = np.zeros(n * d)
    for i in range(n):
        J[i*d:(i+1)*d, i*d:(i+1)*d] += Ri; hv[i*d:(i+1)*d] += Ri @ data[i]
    for c in range(1, n):
        P, C = slice(0, d), slice(c * d, (c + 1) * d)
        J[P, P] += A.T @ Qi @ A; J[C, C] += Qi; J[P, C] -= A.T @ Qi; J[C, P] -= Qi @ A
    S = np.linalg.inv(J); mean = (S @ hv).reshape(n, d)
    assert_close(np.stack([v.mean for v in vals]), mean, 1e-8, f"d={d} {mode}: marginal means vs the joint solve", scale_by="max")
    assert_close(np.stack([v.covariance for v in vals]), np.stack([S[i*d:(i+1)*d, i*d:(i+1)*d] for i in range(n)]), 1e-8,
                 f"d={d} {mode}: marginal covariances vs the joint solve", scale_by="max")
    if mode == "reference":
        _k, var, _f, lo, hi = next(r for r in proc.dev.ref_trace() if r[0] == L.ITEM_PRODUCT_OF_MESSAGES)
        assert np.all(np.isfinite(proc.dev.get_products([var], [lo], [hi])))
        return
    # a product node reads back as the product of its range
    node = next(v for v in proc.execution_log if isinstance(v, V.ProductOfMessages))
    got = proc.read(node)
    assert got.mean.shape == (d,) and np.all(np.isfinite(got.covariance))


def test_d_dimensional_batch_errors(hip_lib):
    d = 4
    model = cx.synth.lgssm_chain(5, d=d, seed=2)
    dev = cx.DeviceGraph(dim=d)
    cx.synth.load_into_device(model, dev)
    L = cx._lib
    with pytest.raises(cx.CortexHipError) as e:        # a JointMarginal item has no dim > 1 form (ProductOfMessages has one: tests/test_gpu_mv.py)
        dev.update_batch([L.ITEM_JOINT_MARGINAL], [0], [int(model.factor_ids[-1])])
    assert e.value.code == L.ERR_UNSUPPORTED
    with pytest.raises(cx.CortexHipError) as e:        # a range beyond the variable's degree
        dev.update_batch([L.ITEM_PRODUCT_OF_MESSAGES], [int(model.x_ids[0])], [L.item_range(1, 3)])
    assert e.value.code == L.ERR_INVALID_ARGUMENT
    with pytest.raises(cx.CortexHipError) as e:
        dev.update_batch([L.ITEM_MESSAGE_TO_VARIABLE], [int(model.x_ids[0])], [int(model.factor_ids[-1]) + 99])
    assert e.value.code == L.ERR_NOT_FOUND
    # a message whose dependency is undefined is not stored (the signal is not pending): x_2 -> transition needs alpha_1 first
    tr0 = int(model.factor_ids[5])
    dev.update_batch([L.ITEM_MESSAGE_TO_FACTOR], [int(model.x_ids[1])], [tr0])
    assert np.all(np.isnan(dev.get_messages([model.x_ids[1]], [tr0], L.TO_FACTOR)))


@pytest.mark.parametrize("dim", [1, 4])
@pytest.mark.parametrize("n_items", [1, 48, 49, 130])
def test_batches_in_the_kernel_arguments_and_staged_batches_agree(hip_lib, dim, n_items):
    """Up to 48 items travel in the kernel arguments (cx_update_batch_async returns when the launch is queued), more go through the
    handle's staging buffer: the same wavefront of variable→factor messages either way, equal to the leave-one-out sums of the stored messages."""
    import struct

    from cortex.jl_amd import _lib as L

    T = 140
    model = cx.synth.lgssm_chain(T, dim, seed=5) if dim > 1 else cx.synth.ssm_chain(T, seed=5)
    sched = L.SCHED_FLOODING if dim == 1 else L.SCHED_FUSED
    dev = cx.DeviceGraph(dim=dim, schedule=sched)
    cx.synth.load_into_device(model, dev, seed_variance=100.0)     # every factor→variable message defined from the start
    # the variable→factor messages of the first n_items (state, transition) pairs, as ONE batch
    pairs = [(int(v), int(f)) for v, f in zip(model.edge_var, model.edge_fac) if int(v) in set(model.x_ids.tolist())][: n_items]
    assert len(pairs) == n_items
    recs = b"".join(struct.pack("<iiqq", L.ITEM_MESSAGE_TO_FACTOR, 0, v, f) for v, f in pairs)
    dev.update_batch_packed(recs, n_items)        # asynchronous for n_items <= 48
    got = dev.get_messages([p[0] for p in pairs], [p[1] for p in pairs], L.TO_FACTOR, L.FORM_NATURAL)      # waits for the stream
    dev2 = cx.DeviceGraph(dim=dim, schedule=sched)
    cx.synth.load_into_device(model, dev2, seed_variance=100.0)
    dev2.update_batch([L.ITEM_MESSAGE_TO_FACTOR] * n_items, [p[0] for p in pairs], [p[1] for p in pairs])  # complete at return
    want = dev2.get_messages([p[0] for p in pairs], [p[1] for p in pairs], L.TO_FACTOR, L.FORM_NATURAL)
    assert np.array_equal(got, want, equal_nan=True)
    # independent of both: the sum (natural form) of the OTHER factor→variable messages of the variable, from the stored messages
    inc = dev.get_messages(model.edge_var, model.edge_fac, L.TO_VARIABLE, L.FORM_NATURAL)
    by_var = {}
    for (v, f), m in zip(zip(model.edge_var.tolist(), model.edge_fac.tolist()), inc):
        by_var.setdefault(v, []).append((f, m))
    want2 = np.array([sum(m for f2, m in by_var[v] if f2 != f) for v, f in pairs])
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, want2, rtol=1e-12, atol=1e-300)
