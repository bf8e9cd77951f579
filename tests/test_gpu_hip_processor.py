"""-m gpu: the drop-in itself — the reference's SSM test (test/inference_engine_tests.jl:379-488) with the processor
swapped for HipProcessor, in its three modes.  The host keeps Signals/readiness/scheduler (cortex.jl_amd mirror of
src/), the arithmetic runs on the MI355X through the C ABI."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import InferenceSignalVariants as V
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
        proc._item(V.JointMarginal(1, (1, 2)))
