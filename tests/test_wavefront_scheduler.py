"""cortex.jl_amd.run_wavefronts — the batched mode of the plug-in boundary (one launch per wavefront of mutually independent
pending signals) — on the CPU, with the reference-arithmetic processors of the reference's tests doing the computing.

Pinned against the form it replaces: a full scan_inference_request (src/inference_engine.jl:540-546) before every wavefront.  On
trees the two produce the same sequence of frontiers (as sets); the incremental form does O(frontier) host work per wavefront."""
import time

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import get_value, get_variable_marginal, request_inference_for, scan_inference_request, set_value, update_marginals
from cortex.jl_amd.signal import is_pending
from tests.test_host_mirror import SSMBeliefPropagationProcessor, make_ssm


def _launcher(engine, processor, log):
    def launch(front):
        log.append(sorted(repr(s.variant) for s in front))     # a wavefront is a SET of independent signals: its order carries no meaning
        for s in front:
            processor.process(engine, None, s)        # compute! + set_value!, like a per-signal process!
    return launch


def _scan_every_wavefront(request, launch):
    """the round-2 form: a full scan of the request per wavefront"""
    while True:
        seen, front = set(), []
        for s in scan_inference_request(request):
            if id(s) not in seen:
                seen.add(id(s)); front.append(s)
        if not front:
            break
        launch(front)
    final = [m for m in request.marginals if is_pending(m)]
    if final:
        launch(final)


def _ssm(n, seed=1234):
    proc = SSMBeliefPropagationProcessor()
    engine, x, y, lik, tr = make_ssm(n, proc)
    rng = np.random.default_rng(seed)
    for i in range(n):
        set_value(engine.get_connection_message_to_factor(y[i], lik[i]), 2 * (i + 1) + rng.standard_normal())
    return engine, proc, x, y, lik


@pytest.mark.parametrize("n", [1, 2, 3, 40])
def test_incremental_frontiers_equal_repeated_full_scans_on_a_chain(n):
    e1, p1, x1, *_ = _ssm(n)
    e2, p2, x2, *_ = _ssm(n)
    log1, log2, stats = [], [], {}
    cx.run_wavefronts(request_inference_for(e1, x1), _launcher(e1, p1, log1), stats)
    _scan_every_wavefront(request_inference_for(e2, x2), _launcher(e2, p2, log2))
    assert log1 == log2
    assert stats["wavefronts"] == len(log1) and stats["full_scans"] == 2        # the first one and the confirming last one
    if n >= 3:
        assert len(log1) == 2 * (n - 1) + 1 + 1      # likelihoods + first v→f | 2(n-1)-1 alternating | ... | marginals: O(n) wavefronts
    # same marginals as the reference's sequential scheduler
    e3, p3, x3, *_ = _ssm(n)
    update_marginals(e3, x3)
    for a, b in zip(x1, x3):
        ma, mb = get_value(get_variable_marginal(e1.get_variable(a))), get_value(get_variable_marginal(e3.get_variable(b)))
        assert ma.mean == pytest.approx(mb.mean, rel=1e-12) and ma.variance == pytest.approx(mb.variance, rel=1e-12)


def test_second_request_after_new_data_and_partial_requests():
    """update_marginals! again after new data for some observations, and a request for a few variables only: every frontier of the
    incremental form equals the full-scan form's"""
    n = 25
    runs = []
    for form in ("incremental", "scan"):
        engine, proc, x, y, lik = _ssm(n)
        log = []
        launch = _launcher(engine, proc, log)
        run = (lambda req: cx.run_wavefronts(req, launch)) if form == "incremental" else (lambda req: _scan_every_wavefront(req, launch))
        run(request_inference_for(engine, x[5:9]))                      # a partial request first
        run(request_inference_for(engine, x))
        for i in (3, 17):
            set_value(engine.get_connection_message_to_factor(y[i], lik[i]), -4.0)
        run(request_inference_for(engine, x))
        runs.append((log, [get_value(get_variable_marginal(engine.get_variable(v))).mean for v in x]))
    assert runs[0][0] == runs[1][0]
    assert runs[0][1] == runs[1][1]


def test_beta_bernoulli_segment_tree_wavefronts():
    """degree 100 > 5: the marginal hangs off the segment tree of ProductOfMessages signals (dependencies.jl:90-173): O(log n) wavefronts"""
    from tests.test_gpu_hip_processor import BetaBernoulliProcessor, make_beta_bernoulli

    n = 100
    data = np.random.default_rng(7).random(n) < 0.5
    logs = []
    for form in ("incremental", "scan"):
        proc = BetaBernoulliProcessor()
        engine, p, o, f = make_beta_bernoulli(n, proc, trace=False)
        for i in range(n):
            set_value(engine.get_connection_message_to_factor(o[i], f[i]), bool(data[i]))
        log = []
        req = request_inference_for(engine, [p])
        (cx.run_wavefronts if form == "incremental" else _scan_every_wavefront)(req, _launcher(engine, proc, log))
        logs.append(log)
        ans = get_value(get_variable_marginal(engine.get_variable(p)))
        assert (ans.a, ans.b) == (1.0 + data.sum(), 1.0 + n - data.sum())
    assert logs[0] == logs[1] and len(logs[0]) <= 2 + int(np.ceil(np.log2(n))) + 2


def test_host_cost_is_linear_in_the_chain_length():
    """2T wavefronts of O(1) host work each, not O(T) each.  (Python on a shared box: a loose bound.)"""
    times = {}
    for n in (250, 1000):
        best = None
        for _ in range(3):                  # the best of three: a loaded box stretches single runs, not the scaling
            engine, proc, x, *_ = _ssm(n)
            t0 = time.perf_counter()
            cx.run_wavefronts(request_inference_for(engine, x), _launcher(engine, proc, []))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        times[n] = best
    assert times[1000] < 8 * times[250] + 0.05, times          # quadratic would be 16x
    assert times[1000] < 1.5, times
