"""-m gpu: the variational families (cx_set_marginals / cx_update_marginals) against the array form of the reference's
update_marginals! on its two variational SSM test models (oracle/vmp.py, itself pinned call by call against the restated
engine in tests/test_vmp_restatement.py) and against the engine restatement directly.

Tolerance: the device sums natural parameters (Normal) and rates (Gamma) in a fixed tree; the reference folds
mean/precision and shape/scale pairs left to right or through its segment tree — identical up to rounding, RTOL 1e-10."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import vmp
from tests import vmp_support as S

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def _device(model, family, schedule=L.SCHED_CHAIN_SCAN):
    dev = cx.DeviceGraph(family=family, schedule=schedule)
    cx.synth.load_vmp_into_device(model, dev)
    return dev


def _state(dev, model):
    xs = dev.get_marginals(model.x_ids)
    g = dev.get_marginals([model.ssnoise, model.obsnoise])
    return np.concatenate([xs[:, 0], xs[:, 1], g[0], g[1]])


def _array_state(arr):
    return np.concatenate([arr.xm, arr.xw, arr.ss, arr.obs])


def _ids_of(model, which):
    out = []
    for w in which:
        out += list(model.x_ids) if w == "x" else [model.ssnoise if w == "ssnoise" else model.obsnoise]
    return out


MF_SEQ = [["x"], ["ssnoise"], ["obsnoise"], ["obsnoise"], ["ssnoise"], ["ssnoise", "obsnoise"], ["x"], ["x", "ssnoise", "obsnoise"],
          ["obsnoise", "x"]]
# (the last three: states TOGETHER with precisions, as the reference's experiment ends an iteration, test/inference_engine_tests.jl:1113;
# accepted where the reference's emergent order is class by class — oracle/vmp.py: StructuredVMP.update — n - 1 > 5)
ST_SEQ = [["obsnoise"], ["ssnoise"], ["x"], ["ssnoise"], ["ssnoise"], ["ssnoise"], ["x"], ["x"], ["obsnoise"], ["obsnoise"],
          ["ssnoise", "obsnoise"]]
ST_TOGETHER = [["ssnoise", "obsnoise", "x"], ["obsnoise"], ["ssnoise", "x"], ["obsnoise"], ["obsnoise", "x"]]


@pytest.mark.parametrize("n", [2, 3, 8, 100, 5000])
def test_mean_field_calls_match_the_array_form(hip_lib, n):
    model = cx.synth.vmp_ssm(n, seed=11)
    dev = _device(model, L.FAMILY_VMP_MEAN_FIELD)
    arr = vmp.MeanFieldVMP(model.data_y)
    for it in range(4):
        for which in MF_SEQ:
            dev.update_marginals(_ids_of(model, which))
            arr.update(which)
            np.testing.assert_allclose(_state(dev, model), _array_state(arr), rtol=RTOL, atol=0, err_msg=f"n={n} it={it} {which}")


@pytest.mark.parametrize("schedule", [L.SCHED_CHAIN_SCAN, L.SCHED_TREE])      # (a chain is a tree: level by level for the short ones, over its heavy path for n = 5000)
@pytest.mark.parametrize("n", [7, 8, 100, 5000])
def test_structured_calls_match_the_array_form(hip_lib, n, schedule):
    model = cx.synth.vmp_ssm(n, seed=12)
    dev = _device(model, L.FAMILY_VMP_STRUCTURED, schedule)
    arr = vmp.StructuredVMP(model.data_y)
    for it in range(4):
        for which in ST_SEQ + (ST_TOGETHER if n - 1 > 5 else []):
            dev.update_marginals(_ids_of(model, which))
            arr.update(which)
            np.testing.assert_allclose(_state(dev, model), _array_state(arr), rtol=1e-9, atol=0, err_msg=f"n={n} it={it} {which}")


@pytest.mark.parametrize("kind,family,rule", [("mean_field", L.FAMILY_VMP_MEAN_FIELD, S.mean_field_rule),
                                              ("structured", L.FAMILY_VMP_STRUCTURED, S.structured_rule)])
def test_reference_experiment_against_the_restated_engine(hip_lib, kind, family, rule):
    """The reference's own experiment (n = 100, its sequence of update_marginals! calls, class by class) on the device and
    on the C restatement of the engine driven through the transcribed resolvers and rules."""
    n, iters = 100, 30
    model = cx.synth.vmp_ssm(n, seed=1234)
    dev = _device(model, family)
    be = S.OracleBackend(rule)
    calls_of = S.mean_field_calls if kind == "mean_field" else S.structured_calls      # every call of the reference's experiment, the last one (:1113) included
    checked = [0]

    def on_call(it, ids):
        dev.update_marginals(ids)
        if it in (1, 2, iters):
            got = _state(dev, model)
            xs = np.array([be.get_marginal(int(v))[1][:2] for v in model.x_ids])
            want = np.concatenate([xs[:, 0], xs[:, 1], be.get_marginal(1)[1][:2], be.get_marginal(2)[1][:2]])
            np.testing.assert_allclose(got, want, rtol=1e-8, atol=0, err_msg=f"{kind}: iteration {it}, request {ids[:3]}")
            checked[0] += 1

    ans = S.run_experiment(be, kind, list(model.data_y), iters, on_call=on_call, calls_of=calls_of)
    assert checked[0] > 20
    g = dev.get_marginals([model.ssnoise, model.obsnoise])
    assert g[0, 0] * g[0, 1] == pytest.approx(S.mean(ans["ssnoise"]), rel=1e-8)
    assert g[1, 0] * g[1, 1] == pytest.approx(S.mean(ans["obsnoise"]), rel=1e-8)


def test_class_selectors_equal_id_lists(hip_lib):
    model = cx.synth.vmp_ssm(300, seed=5)
    a, b = _device(model, L.FAMILY_VMP_STRUCTURED), _device(model, L.FAMILY_VMP_STRUCTURED)
    for _ in range(3):
        a.update_marginals(model.x_ids); b.update_marginals(L.VMP_ALL_NORMAL)
        a.update_marginals([model.ssnoise, model.obsnoise]); b.update_marginals(L.VMP_ALL_PRECISION)
    assert np.array_equal(_state(a, model), _state(b, model))


def test_vmp_recovers_the_noise_precisions_at_scale(hip_lib):
    """n = 200,000: structured VMP to convergence; the posterior means of both precisions land on the true value 100."""
    model = cx.synth.vmp_ssm(200_000, seed=3)
    dev = _device(model, L.FAMILY_VMP_STRUCTURED)
    for _ in range(60):
        dev.update_marginals(L.VMP_ALL_NORMAL)
        dev.update_marginals(L.VMP_ALL_PRECISION)
    g = dev.get_marginals([model.ssnoise, model.obsnoise])
    assert g[0, 0] * g[0, 1] == pytest.approx(100.0, rel=0.03)
    assert g[1, 0] * g[1, 1] == pytest.approx(100.0, rel=0.03)


def test_vmp_errors(hip_lib):
    model = cx.synth.vmp_ssm(10, seed=1)
    dev = _device(model, L.FAMILY_VMP_STRUCTURED)
    with pytest.raises(cx.CortexHipError, match="separate calls"):      # no state update yet
        dev.update_marginals([model.ssnoise] + list(model.x_ids))
    dev.update_marginals(model.x_ids)
    with pytest.raises(cx.CortexHipError, match="separate calls"):      # the transition precision after the first state
        dev.update_marginals(list(model.x_ids) + [model.ssnoise])
    with pytest.raises(cx.CortexHipError, match="separate calls"):      # q(obsnoise) not updated since the states were
        dev.update_marginals([model.obsnoise] + list(model.x_ids))
    before = _state(dev, model)
    dev.update_marginals([model.ssnoise] + list(model.x_ids))            # accepted: the precision first, nothing stale
    assert not np.array_equal(before, _state(dev, model))
    small = _device(cx.synth.vmp_ssm(5, seed=1), L.FAMILY_VMP_STRUCTURED)
    small.update_marginals(cx.synth.vmp_ssm(5, seed=1).x_ids)
    with pytest.raises(cx.CortexHipError, match="separate calls"):      # degree <= 5: the joints are read as the last state update left them
        small.update_marginals([1] + list(cx.synth.vmp_ssm(5, seed=1).x_ids))
    with pytest.raises(cx.CortexHipError, match="together"):
        dev.update_marginals(model.x_ids[:3])
    with pytest.raises(cx.CortexHipError, match="unknown variable id"):
        dev.update_marginals([10_000])
    with pytest.raises(cx.CortexHipError, match="not available for the variational families"):
        dev.sweep(1)
    with pytest.raises(cx.CortexHipError, match="precision variable"):
        dev.set_marginals([model.ssnoise], L.FORM_MEAN_PRECISION, [0.0, 1.0])
    with pytest.raises(cx.CortexHipError, match="shape > 0"):
        dev.set_marginals([model.ssnoise], L.FORM_GAMMA, [-1.0, 1.0])
    with pytest.raises(cx.CortexHipError, match="is observed"):
        dev.set_marginals([model.y_ids[0]], L.FORM_MEAN_PRECISION, [0.0, 1.0])
    plain = cx.DeviceGraph()
    cx.synth.load_into_device(cx.synth.ssm_chain(5), plain)
    with pytest.raises(cx.CortexHipError, match="variational families only"):
        plain.update_marginals([1])
    bad = cx.DeviceGraph(family=L.FAMILY_VMP_MEAN_FIELD)
    with pytest.raises(cx.CortexHipError, match="edge_role"):
        bad.graph_create(model.edge_var, model.edge_fac, model.factor_ids, np.full(len(model.factor_ids), L.FACTOR_NORMAL_PRECISION), np.zeros(len(model.factor_ids)))


@pytest.mark.parametrize("family,rule,kind", [("mean_field", S.mean_field_rule, "mean_field"), ("structured", S.structured_rule, "structured")])
def test_vmp_processor_behind_the_engine_api(hip_lib, family, rule, kind):
    """The reference's experiment written against the engine API (make_ssm_model + experiment,
    test/inference_engine_tests.jl:691-770) with the HIP processor plugged in, beside the restated engine."""
    n, iters = 40, 8
    data = S.dataset(n, seed=9)
    graph = cx.BipartiteFactorGraph()
    ssnoise = graph.add_variable(cx.Variable(name="ssnoise"))
    obsnoise = graph.add_variable(cx.Variable(name="obsnoise"))
    x = [graph.add_variable(cx.Variable(name="x", index=(i,))) for i in range(n)]
    y = [graph.add_variable(cx.Variable(name="y", index=(i,))) for i in range(n)]
    lik_form = cx.NormalPrecisionFactor(roles=(("y", "out"), ("x", "mean"), ("obsnoise", "precision")))
    tr_form = cx.NormalPrecisionFactor(roles=(("x", "both"), ("ssnoise", "precision")))
    likelihood = [graph.add_factor(cx.Factor(functional_form=lik_form)) for _ in range(n)]
    transition = [graph.add_factor(cx.Factor(functional_form=tr_form)) for _ in range(n - 1)]
    for i in range(n):
        graph.add_edge(y[i], likelihood[i], cx.Connection(label="out"))
        graph.add_edge(x[i], likelihood[i], cx.Connection(label="out"))
        graph.add_edge(obsnoise, likelihood[i], cx.Connection(label="out"))
    for i in range(n - 1):
        graph.add_edge(x[i], transition[i], cx.Connection(label="out"))
        graph.add_edge(x[i + 1], transition[i], cx.Connection(label="in"))
        graph.add_edge(ssnoise, transition[i], cx.Connection(label="out"))
    proc = cx.HipVmpProcessor(family=family)
    engine = cx.InferenceEngine(model_engine=graph, inference_request_processor=proc, resolve_dependencies=False)
    marginal = lambda v: cx.get_variable_marginal(engine.get_variable(v))  # noqa: E731
    proc.set_value(marginal(ssnoise), cx.Gamma(1.0, 1.0))
    proc.set_value(marginal(obsnoise), cx.Gamma(1.0, 1.0))
    for v in x:
        proc.set_value(marginal(v), cx.NormalMeanPrecision(0.0, 1.0))
    for v, d in zip(y, data):
        proc.set_value(marginal(v), d)
    be = S.OracleBackend(rule)
    calls_of = S.mean_field_calls if kind == "mean_field" else S.structured_calls      # the last call of an iteration names states and precisions together (:1113)

    def on_call(it, ids):
        cx.update_marginals(engine, ids)

    ans = S.run_experiment(be, kind, data, iters, on_call=on_call, calls_of=calls_of)
    got_ss, got_obs = cx.get_value(marginal(ssnoise)), cx.get_value(marginal(obsnoise))
    assert got_ss.shape * got_ss.scale == pytest.approx(S.mean(ans["ssnoise"]), rel=1e-8)
    assert got_obs.shape * got_obs.scale == pytest.approx(S.mean(ans["obsnoise"]), rel=1e-8)
    for v, want in zip(x, ans["x"]):
        got = cx.get_value(marginal(v))
        assert got.mean == pytest.approx(want[1][0], rel=1e-8, abs=1e-12) and got.precision == pytest.approx(want[1][1], rel=1e-8)


@pytest.mark.parametrize("family", [L.FAMILY_VMP_MEAN_FIELD, L.FAMILY_VMP_STRUCTURED])
def test_vmp_checkpoint_continues_bit_for_bit(hip_lib, family):
    """cx_state_export / cx_state_import on a variational handle (structured: the inner chain handle's state travels inside
    the blob): a restored handle continues exactly where the exporter stood; a blob of another graph is refused."""
    ma, mb = cx.synth.vmp_ssm(400, seed=1), cx.synth.vmp_ssm(400, seed=2)
    a, b = _device(ma, family), _device(mb, family)

    def iterate(dev, k):
        for _ in range(k):
            dev.update_marginals(L.VMP_ALL_NORMAL)
            dev.update_marginals(L.VMP_ALL_PRECISION)

    iterate(a, 3)
    blob = a.export_state()
    iterate(a, 4)
    iterate(b, 2)                       # other data, other state
    b.import_state(blob)                # ... replaced: observations, marginals, chain messages, counters
    iterate(b, 4)
    assert np.array_equal(_state(a, ma), _state(b, ma))
    assert a.stats()["sweeps_done"] == b.stats()["sweeps_done"]
    other = _device(cx.synth.vmp_ssm(401, seed=1), family)
    with pytest.raises(cx.CortexHipError, match="different graph"):
        other.import_state(blob)
    plain = cx.DeviceGraph()
    cx.synth.load_into_device(cx.synth.ssm_chain(5), plain)
    with pytest.raises(cx.CortexHipError, match="not a state blob"):
        plain.import_state(blob)


@pytest.mark.parametrize("schedule", [L.SCHED_FUSED, L.SCHED_TREE])
def test_structured_vmp_on_a_tree_with_the_fused_schedule(hip_lib, schedule):
    """Beyond chains: a random tree of latent states with one shared edge precision and one observation precision, the inner
    handle on the fused flooding schedule (one sweep per update call; `depth` calls settle the tree) or on the tree schedule
    (ONE call is the exact two passes).  Belief propagation is exact on a tree, so the variational fixed point must equal the
    one computed with dense linear algebra."""
    calls_per_state_update = 1 if schedule == L.SCHED_TREE else None
    rng = np.random.default_rng(4)
    n = 40
    parent = [-1] + [int(rng.integers(0, i)) for i in range(1, n)]
    edges = [(parent[i], i) for i in range(1, n)]
    truth = np.cumsum(rng.standard_normal(n)) * 0.3
    y = truth + rng.standard_normal(n) * 0.5
    ss, obs = 1, 2
    x = np.arange(3, 3 + n); yv = np.arange(3 + n, 3 + 2 * n)
    lik = np.arange(3 + 2 * n, 3 + 3 * n); tr = np.arange(3 + 3 * n, 3 + 3 * n + len(edges))
    ev, ef, role = [], [], []
    for i in range(n):
        ev += [yv[i], x[i], obs]; ef += [lik[i]] * 3; role += [L.ROLE_OUT, L.ROLE_IN, L.ROLE_PRECISION]
    for k, (a, b) in enumerate(edges):
        ev += [x[a], x[b], ss]; ef += [tr[k]] * 3; role += [L.ROLE_IN, L.ROLE_OUT, L.ROLE_PRECISION]
    fids = np.concatenate([lik, tr])
    dev = cx.DeviceGraph(family=L.FAMILY_VMP_STRUCTURED, schedule=schedule)
    dev.graph_create(ev, ef, fids, np.full(len(fids), L.FACTOR_NORMAL_PRECISION, np.int32), np.zeros(len(fids)), edge_role=role)
    dev.set_marginals([ss, obs], L.FORM_GAMMA, [1.0, 1.0, 1.0, 1.0])
    dev.set_marginals(x, L.FORM_MEAN_PRECISION, np.tile([0.0, 1.0], n))
    dev.set_marginals(yv, L.FORM_POINT, y)
    # dense reference of the same coordinate ascent
    ts, to = 1.0, 1.0
    for _ in range(60):
        J = np.eye(n) * to
        for a, b in edges:
            J[a, a] += ts; J[b, b] += ts; J[a, b] -= ts; J[b, a] -= ts
        S = np.linalg.inv(J)
        mu = S @ (to * y)
        spread_s = sum(S[a, a] + S[b, b] - 2 * S[a, b] + (mu[a] - mu[b]) ** 2 for a, b in edges)
        spread_o = float(np.sum(np.diag(S) + (y - mu) ** 2))
        ts = (1 + 0.5 * len(edges)) / (0.5 * spread_s)
        to = (1 + 0.5 * n) / (0.5 * spread_o)
    for _ in range(60):
        for _ in range(calls_per_state_update or n):                 # the tree's depth is < n: more than enough flooding sweeps to settle it
            dev.update_marginals(L.VMP_ALL_NORMAL)
        dev.update_marginals(L.VMP_ALL_PRECISION)
    g = dev.get_marginals([ss, obs])
    assert g[0, 0] * g[0, 1] == pytest.approx(ts, rel=1e-6)
    assert g[1, 0] * g[1, 1] == pytest.approx(to, rel=1e-6)
    for _ in range(calls_per_state_update or n):
        dev.update_marginals(L.VMP_ALL_NORMAL)
    q = dev.get_marginals(x)
    J = np.eye(n) * to
    for a, b in edges:
        J[a, a] += ts; J[b, b] += ts; J[a, b] -= ts; J[b, a] -= ts
    S = np.linalg.inv(J)
    np.testing.assert_allclose(q[:, 0], S @ (to * y), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(1 / q[:, 1], np.diag(S), rtol=1e-5)
