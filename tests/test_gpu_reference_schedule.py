"""-m gpu: CX_SCHED_REFERENCE — ONE cx_sweep on a LOOPY graph is ONE update_marginals! of the reference, call by call.

The device replays the order the reference's scheduler takes (found on a shadow of the readiness nibbles, csrc/cx_refsched.h; the
CPU twin tests/test_refsched.py pins that order against oracle/cortex_ref.c signal by signal and executes the levelled stages in
numpy) as stages of items in one graph launch.  Here: the values the device leaves after every call — every message in both
directions, every marginal — against the restated engine (<= 1e-9: natural-form against moment-form arithmetic), the execution trace
the library reports against the engine's, the plan cache, partial requests, a plug-in driving the same handle, and the checkpoint."""
import ctypes as C

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact, ref
from tests.helpers import assert_close, engine_oracle_from_model, random_loopy_model
from tests.loopy_support import pairwise_edges

pytestmark = pytest.mark.gpu
SEED_VARIANCE = 1e6


def _models():
    rnd, _ = random_loopy_model(11, 1, nv=60, extra=25)
    hubs, _ = random_loopy_model(5, 1, nv=40, extra=70)
    return {"grid8x9": cx.synth.gaussian_grid(8, 9, seed=5), "grid48x40": cx.synth.gaussian_grid(48, 40, seed=1), "random": rnd, "hubs": hubs}


def _oracle_trace(E):
    kinds = {ref.VAR_MSG_TO_FACTOR: L.ITEM_MESSAGE_TO_FACTOR, ref.VAR_MSG_TO_VARIABLE: L.ITEM_MESSAGE_TO_VARIABLE, ref.VAR_MARGINAL: L.ITEM_INDIVIDUAL_MARGINAL,
             ref.VAR_PRODUCT: L.ITEM_PRODUCT_OF_MESSAGES}
    out = []
    for _r, _v, s, _b, _a in E.trace():
        k, v, f, lo, hi = E.variant(s)
        out.append((kinds[k], v, f if k in (ref.VAR_MSG_TO_FACTOR, ref.VAR_MSG_TO_VARIABLE) else 0, lo if k == ref.VAR_PRODUCT else 0, hi if k == ref.VAR_PRODUCT else 0))
    return out


def _set_priors(dev, E, model):
    E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
    dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, np.stack([model.prior_mean, model.prior_variance], axis=1))


def _compare(dev, E, model, request, what, rtol=1e-9):
    for to_variable, direction in ((True, L.TO_VARIABLE), (False, L.TO_FACTOR)):
        tags, a, b = E.get_messages(model.edge_var, model.edge_fac, to_variable)
        got = dev.get_messages(model.edge_var, model.edge_fac, direction)
        und = tags == ref.UNDEF
        name = "f2v" if to_variable else "v2f"
        assert np.array_equal(np.isnan(got[:, 1]), und), f"{what}: the same {name} messages are defined"
        assert_close(got[~und, 0], a[~und], rtol, f"{what} {name} mean")
        ok = ~und & (tags != ref.REAL)
        assert_close(got[ok, 1], b[ok], rtol, f"{what} {name} variance")
    tags, em, ev = E.get_marginals(request)
    marg = dev.get_marginals(request)
    und = tags == ref.UNDEF
    assert np.array_equal(np.isnan(marg[:, 1]), und), f"{what}: the same marginals are defined"
    assert_close(marg[~und, 0], em[~und], rtol, f"{what} marginal mean")
    assert_close(marg[~und, 1], ev[~und], rtol, f"{what} marginal variance")


def _start(model):
    E = engine_oracle_from_model(model, trace=True)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev)
    pv, pf = pairwise_edges(model)
    E.set_messages_to_variable(pv, pf, np.zeros(len(pv)), np.full(len(pv), SEED_VARIANCE))
    dev.seed_messages(L.TO_VARIABLE, 0.0, SEED_VARIANCE)
    return E, dev


@pytest.mark.parametrize("name", ["grid8x9", "grid48x40", "random", "hubs"])
def test_one_sweep_on_a_loopy_graph_is_one_update_marginals_of_the_reference(hip_lib, name):
    model = _models()[name]
    E, dev = _start(model)
    for call in range(5):
        if call:
            _set_priors(dev, E, model)
        dev.sweep(1)
        E.update_marginals(model.x_ids)
        assert dev.ref_trace() == _oracle_trace(E), f"{name} call {call + 1}: the executions, in order"
        _compare(dev, E, model, model.x_ids, f"{name} call {call + 1}")
    st = dev.ref_plan_stats()
    assert st["executions"] == len(E.trace()) and st["stages"] >= 3 and st["launches"] <= st["stages"]
    assert st["hits"] + st["misses"] == 5
    if name.startswith("grid"):
        assert st["hits"] >= 3, "from the second call on the readiness state before a call repeats: a standing plan is replayed"
    dev.close()


def test_a_long_list_of_priors_re_set_every_call_is_translated_once(hip_lib):
    """cx_set_messages keeps the translation of a list of 4,096 ids or more (ids -> slots, variables, edges) and, under this schedule, the
    readiness state the list leads to from the state it started from: an iteration "set the priors, call" on 70 x 72 = 5,040 variables —
    with NEW prior values every time, the same list in another order once, and a shorter list once — against the restated engine, call by
    call: the executions in order, every message, every marginal"""
    model = cx.synth.gaussian_grid(70, 72, seed=4)
    E, dev = _start(model)
    rng = np.random.default_rng(0)
    n = len(model.prior_var)
    for call in range(7):
        if call:
            order = rng.permutation(n) if call == 4 else np.arange(n)
            keep = order[: n - 500] if call == 5 else order
            mean, var = model.prior_mean + 0.1 * call, model.prior_variance * (1.0 + 0.2 * call)
            E.set_messages_to_variable(model.prior_var[keep], model.prior_fac[keep], mean[keep], var[keep])
            dev.set_messages(model.prior_var[keep], model.prior_fac[keep], L.TO_VARIABLE, L.FORM_MOMENT, np.stack([mean[keep], var[keep]], axis=1))
        dev.sweep(1)
        E.update_marginals(model.x_ids)
        assert dev.ref_trace() == _oracle_trace(E), f"call {call + 1}: the executions, in order"
        _compare(dev, E, model, model.x_ids, f"call {call + 1}")
    dev.close()


def test_the_sweep_differs_from_a_jacobi_sweep_and_shares_its_fixed_point(hip_lib):
    """what the item asks for in one sentence: the fused schedule's sweep is NOT the reference's call (it reads old values), the
    reference-order schedule's is; both reach the same messages, and the exact posterior means"""
    model = _models()["grid8x9"]
    E, dev = _start(model)
    jac = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, jac, seed_variance=SEED_VARIANCE)
    dev.sweep(1); jac.sweep(1); E.update_marginals(model.x_ids)
    pv, pf = pairwise_edges(model)
    a, b = dev.get_messages(pv, pf, L.TO_VARIABLE), jac.get_messages(pv, pf, L.TO_VARIABLE)
    assert np.max(np.abs(a[:, 0] - b[:, 0])) > 1e-3, "after ONE sweep the two schedules hold different messages"
    for _ in range(250):
        _set_priors(dev, E, model)
        dev.sweep(1)
    jac.sweep(500)
    a, b = dev.get_messages(pv, pf, L.TO_VARIABLE), jac.get_messages(pv, pf, L.TO_VARIABLE)
    assert_close(a, b, 1e-9, "the common fixed point")
    me = exact.grid_posterior_mean(8, 9, model.meta["y"], model.meta["r"], model.meta["qh"], model.meta["qv"])
    assert_close(dev.get_marginals(model.x_ids)[:, 0], me, 1e-8, "converged means vs the sparse solve")
    assert dev.ref_plan_stats()["plans"] <= 4
    dev.close(); jac.close()


def test_a_grid_of_runs_and_launches_against_the_engine(hip_lib):
    """a 230 x 240 grid: its plan's stages run from a pair to ~1,100 records — runs of thin stages on flat records in one workgroup
    (cx_batch.hip: k_flat_run; between 960 and 1,024 records the fifteen item wavefronts take a second pass) next to stages that leave as
    launches; three calls, executions and every message and marginal against the restated engine"""
    model = cx.synth.gaussian_grid(230, 240, seed=7)
    E, dev = _start(model)
    for call in range(3):
        if call:
            _set_priors(dev, E, model)
        dev.sweep(1)
        E.update_marginals(model.x_ids)
        assert dev.ref_trace() == _oracle_trace(E), f"call {call + 1}: the executions"
        _compare(dev, E, model, model.x_ids, f"230 x 240 call {call + 1}")
    st = dev.ref_plan_stats()
    assert 1 < st["launches"] < st["stages"] and not dev.cluster_stats()["last_reference_call"], st
    dev.close()


def test_partial_requests_in_the_callers_order(hip_lib):
    model = _models()["grid8x9"]
    E, dev = _start(model)
    rng = np.random.default_rng(1)
    for call in range(4):
        if call:
            _set_priors(dev, E, model)
        request = rng.permutation(model.x_ids)[: 20 + 10 * call]
        dev.sweep_for(request)
        E.update_marginals(request)
        assert dev.ref_trace() == _oracle_trace(E)
        _compare(dev, E, model, request, f"partial request {call + 1}")
    dev.close()


def test_empty_and_repeated_requests(hip_lib):
    """update_marginals!(engine, ids) walks ids as given (src/inference_engine.jl:559-632): nothing for an empty list, and a variable named
    twice is processed where it stands first — the second mention finds its marginal computed"""
    model = _models()["grid8x9"]
    E, dev = _start(model)
    x = model.x_ids
    for request in ([], [x[5], x[5]], [x[40], x[7], x[40], x[7], x[8]], list(x) + list(x[:5])):
        _set_priors(dev, E, model)
        request = np.asarray(request, dtype=np.int64)
        dev.sweep_for(request)
        E.update_marginals(request)
        assert dev.ref_trace() == _oracle_trace(E), f"request {request[:6]}"
        if len(request):
            _compare(dev, E, model, np.unique(request), f"request of {len(request)} ids")
    dev.close()


@pytest.mark.parametrize("T", [1, 3, 200])
def test_the_reference_state_space_model(hip_lib, T):
    """test/inference_engine_tests.jl:379-488: one call is the Kalman smoother; a second call finds nothing pending"""
    model = cx.synth.ssm_chain(T, seed=4, random_variances=True)
    E = engine_oracle_from_model(model, trace=True)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    E.update_marginals(model.x_ids)
    assert dev.ref_trace() == _oracle_trace(E)
    _compare(dev, E, model, model.x_ids, f"chain T={T}")
    xm, xv = exact.ssm_chain_posterior(model.data_y, model.meta["r"], model.meta["q"])
    marg = dev.get_marginals(model.x_ids)
    assert_close(marg[:, 0], xm, 1e-9, "smoother mean"); assert_close(marg[:, 1], xv, 1e-9, "smoother variance")
    assert dev.ref_plan_stats()["executions"] == 5 * T - 4 + T
    dev.sweep(1)
    assert dev.ref_plan_stats()["executions"] == 0
    # new data for one observation: the reference recomputes what depends on it and is requested — here too
    y0 = model.data_y[:1] + 1.0
    dev.set_messages(model.data_var[:1], model.data_fac[:1], L.TO_FACTOR, L.FORM_POINT, y0)
    E.set_messages_to_factor(model.data_var[:1], model.data_fac[:1], y0, tag=ref.REAL)
    dev.sweep(1)
    E.update_marginals(model.x_ids)
    assert dev.ref_trace() == _oracle_trace(E)
    _compare(dev, E, model, model.x_ids, f"chain T={T}, new datum")
    dev.close()


def test_a_plugin_driving_the_same_handle_moves_the_shadow(hip_lib):
    """per-signal process! calls through cx_update_batch are set_value!s on the shadow too: after a call run signal by signal by the host
    scheduler, a cx_sweep finds nothing pending; with the priors re-set it computes exactly what the reference's next call computes"""
    from tests.loopy_support import HipBackend, run_calls

    model = _models()["grid8x9"]
    b = HipBackend("per_signal", schedule=L.SCHED_REFERENCE)
    executed = run_calls(model, b, n_calls=2)
    dev = b.proc.dev
    dev.sweep(1)
    assert dev.ref_plan_stats()["executions"] == 0
    dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, np.stack([model.prior_mean, model.prior_variance], axis=1))
    dev.sweep(1)
    assert dev.ref_plan_stats()["executions"] == executed[1]


def test_checkpoint_carries_the_readiness_state(hip_lib):
    model = _models()["hubs"]
    E, dev = _start(model)
    for call in range(2):
        if call:
            _set_priors(dev, E, model)
        dev.sweep(1); E.update_marginals(model.x_ids)
    blob = dev.export_state()
    other = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, other)
    other.import_state(blob)
    for d in (dev, other):
        d.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, np.stack([model.prior_mean, model.prior_variance], axis=1))
        d.sweep(1)
    E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
    E.update_marginals(model.x_ids)
    assert other.ref_trace() == dev.ref_trace() == _oracle_trace(E)
    assert np.array_equal(other.get_marginals(model.x_ids), dev.get_marginals(model.x_ids))
    _compare(other, E, model, model.x_ids, "after the import")
    dev.close(); other.close()


def test_refusals(hip_lib):
    model = _models()["grid8x9"]
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=SEED_VARIANCE)
    with pytest.raises(cx.CortexHipError) as ei:
        dev.sweep_for(model.x_ids[:3])
    assert ei.value.code == L.ERR_UNSUPPORTED and "CX_SCHED_REFERENCE" in ei.value.message
    dev.close()
    cx.DeviceGraph(schedule=L.SCHED_REFERENCE, dim=64).close()      # (round 6: accepted — dim 2, 3, 4 and 64: tests/test_gpu_reference_mv.py)
    with pytest.raises(cx.CortexHipError) as ei:
        cx.DeviceGraph(schedule=L.SCHED_REFERENCE, family=L.FAMILY_VMP_STRUCTURED)      # the variational rules run under it as a wiring
    assert ei.value.code == L.ERR_UNSUPPORTED
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev)
    with pytest.raises(cx.CortexHipError):
        dev.sweep_for([10 ** 9])
    with pytest.raises(cx.CortexHipError) as ei:
        dev.halo_configure_state([], [], [], [])
    assert ei.value.code == L.ERR_UNSUPPORTED
    dev.close()


def test_config_c4_full_size_two_calls(hip_lib):
    """BASELINE config C4 itself (1415 x 1415, 10,005,465 edges) under the reference's order: two consecutive calls against the restated
    engine at full size — 16,006,480 message executions + 2,002,225 marginals per call; every marginal, 400,000 sampled messages of
    each direction.  (The engine restatement takes ~30 s to wire 22 M signals and ~2 s per call on one core.)"""
    n = 1415
    model = cx.synth.gaussian_grid(n, n, seed=1234)
    E = engine_oracle_from_model(model)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev, seed_variance=SEED_VARIANCE)
    pv, pf = pairwise_edges(model)
    E.set_messages_to_variable(pv, pf, np.zeros(len(pv)), np.full(len(pv), SEED_VARIANCE))
    pick = np.random.default_rng(0).choice(len(pv), 400_000, replace=False)
    for call in range(2):
        if call:
            _set_priors(dev, E, model)
        c0 = E.counters()
        dev.sweep(1)
        E.update_marginals(model.x_ids)
        st = dev.ref_plan_stats()
        c1 = E.counters()
        assert st["messages"] == c1[0] - c0[0] == 16_006_480 and st["executions"] - st["messages"] == c1[1] - c0[1] == n * n
        for to_variable, direction in ((True, L.TO_VARIABLE), (False, L.TO_FACTOR)):
            _tags, a, b = E.get_messages(pv[pick], pf[pick], to_variable)
            got = dev.get_messages(pv[pick], pf[pick], direction)
            assert_close(got[:, 0], a, 1e-9, f"C4 call {call + 1} message mean"); assert_close(got[:, 1], b, 1e-9, f"C4 call {call + 1} message variance")
        _t, em, ev = E.get_marginals(model.x_ids)
        marg = dev.get_marginals(model.x_ids)
        assert_close(marg[:, 0], em, 1e-9, f"C4 call {call + 1} marginal mean"); assert_close(marg[:, 1], ev, 1e-9, f"C4 call {call + 1} marginal variance")
    assert 2_500 < st["stages"] < 3_300      # (≈ 2 N: a MessageToFactor and the MessageToVariable that reads it share a stage; ≈ 4 N with CX_REF_FUSE_PAIRS=0)
    dev.close()


@pytest.mark.parametrize("n", [1, 5, 6, 9, 33, 100, 3000])
def test_the_reference_beta_bernoulli_known_answer(hip_lib, n):
    """The reference's conjugate known answer (test/inference_engine_tests.jl:241-377): update_marginals!(engine, p) on the star of n
    Bernoulli factors leaves Beta(1 + Σ, 1 + n − Σ).  Under CX_SCHED_REFERENCE the device runs exactly the executions the reference
    runs — the n messages, the segment tree's ProductOfMessages nodes (n > 5) one by one from their stored children, the marginal —
    and the answer is exact (small integers in f64)."""
    rng = np.random.default_rng(n)
    data = rng.random(n) < 0.5
    p = 1
    o = 2 * np.arange(1, n + 1)          # ids as make_beta_bernoulli_model hands them out: o_i, f_i alternate (:311-327)
    f = o + 1
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE, family=L.FAMILY_NATURAL2)
    dev.graph_create(np.concatenate([np.full(n, p), o]), np.concatenate([f, f]), f, np.full(n, L.FACTOR_BERNOULLI, np.int32), np.ones(n))
    dev.set_messages(o, f, L.TO_FACTOR, L.FORM_POINT, data.astype(float))
    dev.sweep_for([p])
    nat = dev.get_marginals([p])[0]
    assert (1.0 + nat[0], 1.0 + nat[1]) == (1.0 + data.sum(), 1.0 + n - data.sum())
    st = dev.ref_plan_stats()
    assert st["executions"] == n + (n - 2 if n > 5 else 0) + 1 and st["messages"] == n
    if n <= 100:
        E = ref.Engine(ref.P_BETA_BERNOULLI, trace=True)
        assert E.add_variable() == p
        for i in range(n):
            assert (E.add_variable(), E.add_factor(ref.F_BERNOULLI)) == (o[i], f[i])
            E.add_edge(p, int(f[i])); E.add_edge(int(o[i]), int(f[i]))
        E.finalize()
        for i in range(n):
            E.set_value(E.message_to_factor(int(o[i]), int(f[i])), bool(data[i]))
        E.update_marginals([p])
        assert dev.ref_trace() == _oracle_trace(E)
        if n > 5:       # a node of the tree reads back as the product of its range
            lo, hi = 1, n // 2
            node = dev.get_products([p], [lo], [hi], L.FORM_NATURAL)[0]
            assert (node[0], node[1]) == (data[:hi].sum(), (~data[:hi]).sum())
    dev.sweep_for([p])
    assert dev.ref_plan_stats()["executions"] == 0       # nothing is pending: lazy
    dev.close()


# ---- cx_graph_wire: a user resolver's wiring in place of the default one ---------------------------------------------------------------
def _default_triples(model):
    """the default resolver's add_dependency! calls of a model of degree <= 5, as (signal, dependency) keys (kind, variable, factor)"""
    E0 = engine_oracle_from_model(model)
    kinds = {ref.VAR_MSG_TO_FACTOR: L.ITEM_MESSAGE_TO_FACTOR, ref.VAR_MSG_TO_VARIABLE: L.ITEM_MESSAGE_TO_VARIABLE, ref.VAR_MARGINAL: L.ITEM_INDIVIDUAL_MARGINAL}

    def key(sig):
        k, v, f, _lo, _hi = E0.variant(sig)
        return (kinds[k], int(v), int(f) if k != ref.VAR_MARGINAL else 0)
    out = []
    for v, f in zip(model.edge_var, model.edge_fac):
        for sig in (E0.message_to_factor(int(v), int(f)), E0.message_to_variable(int(v), int(f))):
            out += [(key(sig), key(d)) for d in E0.dependencies(sig)]
    for v in model.x_ids:
        out += [(key(E0.marginal(int(v))), key(d)) for d in E0.dependencies(E0.marginal(int(v)))]
    return out


def _engine_with_wiring(model, triples):
    n_nodes = int(max(model.edge_var.max(), model.factor_ids.max()))
    kind = np.zeros(n_nodes, dtype=np.int32); fkind = np.zeros(n_nodes, dtype=np.int32); p0 = np.ones(n_nodes)
    kind[np.unique(model.edge_var) - 1] = 1; kind[model.factor_ids - 1] = 2
    fkind[model.factor_ids - 1] = np.where(model.factor_kind == 1, ref.F_GAUSS_ADD, ref.F_OPAQUE)
    p0[model.factor_ids - 1] = np.asarray(model.factor_var).reshape(len(model.factor_ids), -1)[:, 0]
    E = ref.Engine(ref.P_SSM_BP, True)
    E.bulk_build(kind, fkind, p0, model.edge_var, model.edge_fac)
    E.finalize(resolve_dependencies=False)
    sig_of = lambda k: E.marginal(k[1]) if k[0] == L.ITEM_INDIVIDUAL_MARGINAL else (E.message_to_factor(k[1], k[2]) if k[0] == L.ITEM_MESSAGE_TO_FACTOR else E.message_to_variable(k[1], k[2]))      # noqa: E731
    for s, d, fl in triples:
        E.add_dependency(sig_of(s), sig_of(d), weak=bool(fl & 1), intermediate=bool(fl & 2), listen=not (fl & 4))
    return E


@pytest.mark.parametrize("seed", [0, 1])
def test_user_wiring_with_random_flags_on_a_loopy_grid(hip_lib, seed):
    """cx_graph_wire: the default wiring of a grid with random weak / intermediate / listen flags, a tenth of the product dependencies
    dropped, in a shuffled add_dependency! order — the device against the restated engine under the same calls: trace, messages, marginals"""
    rng = np.random.default_rng(seed)
    model = cx.synth.gaussian_grid(9, 8, seed=6)
    triples = []
    for s, d in _default_triples(model):
        if s[0] != L.ITEM_MESSAGE_TO_VARIABLE and rng.random() < 0.1:
            continue
        fl = (L.WIRE_WEAK if rng.random() < 0.3 else 0) | (L.WIRE_INTERMEDIATE if s[0] != L.ITEM_MESSAGE_TO_VARIABLE and rng.random() < 0.6 else 0) | (L.WIRE_NO_LISTEN if rng.random() < 0.1 else 0)
        triples.append((s, d, fl))
    triples = [triples[i] for i in rng.permutation(len(triples))]
    E = _engine_with_wiring(model, triples)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var)
    dev.graph_wire([s for s, _d, _f in triples], [d for _s, d, _f in triples], [f for _s, _d, f in triples])
    pv, pf = pairwise_edges(model)
    E.set_messages_to_variable(pv, pf, np.zeros(len(pv)), np.full(len(pv), SEED_VARIANCE))
    dev.seed_messages(L.TO_VARIABLE, 0.0, SEED_VARIANCE)
    for call in range(5):
        _set_priors(dev, E, model)
        request = model.x_ids if call % 2 == 0 else rng.permutation(model.x_ids)[:30]
        dev.sweep_for(request)
        E.update_marginals(request)
        assert dev.ref_trace() == _oracle_trace(E), f"call {call + 1}: the executions, in order"
        _compare(dev, E, model, request, f"user wiring, call {call + 1}")
    with pytest.raises(cx.CortexHipError) as ei:
        dev.graph_wire([], [], [])
    assert ei.value.code == L.ERR_STATE          # the wiring is fixed once values exist
    dev.close()


def test_a_filter_wiring_computes_the_kalman_filter(hip_lib):
    """a user resolver that wires the forward messages of the state-space model only: one call is the Kalman FILTER"""
    T = 400
    model = cx.synth.ssm_chain(T, seed=8, random_variances=True)
    x, y, lik, tr = model.x_ids, model.data_var, model.factor_ids[:T], model.factor_ids[T:]
    F, V, M, I = L.ITEM_MESSAGE_TO_FACTOR, L.ITEM_MESSAGE_TO_VARIABLE, L.ITEM_INDIVIDUAL_MARGINAL, L.WIRE_INTERMEDIATE
    triples = []
    for t in range(T):
        triples.append(((V, int(x[t]), int(lik[t])), (F, int(y[t]), int(lik[t])), 0))
        triples.append(((M, int(x[t]), 0), (V, int(x[t]), int(lik[t])), I))
        if t > 0:
            triples.append(((M, int(x[t]), 0), (V, int(x[t]), int(tr[t - 1])), I))
            triples.append(((V, int(x[t]), int(tr[t - 1])), (F, int(x[t - 1]), int(tr[t - 1])), 0))
        if t + 1 < T:
            triples.append(((F, int(x[t]), int(tr[t])), (V, int(x[t]), int(lik[t])), I))
            if t > 0:
                triples.append(((F, int(x[t]), int(tr[t])), (V, int(x[t]), int(tr[t - 1])), I))
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var)
    dev.graph_wire([s for s, _d, _f in triples], [d for _s, d, _f in triples], [f for _s, _d, f in triples])
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, model.data_y)
    dev.sweep_for(x)
    assert dev.ref_plan_stats()["executions"] == T + 2 * (T - 1) + T
    r, q = model.meta["r"], model.meta["q"]
    m, v = model.data_y[0], r[0]
    fm, fv = [m], [v]
    for t in range(1, T):
        pvar = v + q[t - 1]
        k = pvar / (pvar + r[t])
        m, v = m + k * (model.data_y[t] - m), (1 - k) * pvar
        fm.append(m); fv.append(v)
    marg = dev.get_marginals(x)
    assert_close(marg[:, 0], np.array(fm), 1e-9, "filtered means"); assert_close(marg[:, 1], np.array(fv), 1e-9, "filtered variances")
    em, ev = exact.ssm_chain_posterior(model.data_y, r, q)
    assert np.max(np.abs(marg[:-1, 1] - ev[:-1])) > 1e-3, "a filter is not the smoother (the last state is where they meet)"
    assert_close(marg[-1:], np.array([[em[-1], ev[-1]]]), 1e-9, "the last state: filter == smoother")
    with pytest.raises(cx.CortexHipError) as ei:      # a message that depends on a marginal: a rule the device does not have
        fresh = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        fresh.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var)
        fresh.graph_wire([(V, int(x[1]), int(tr[0]))], [(M, int(x[0]), 0)], [L.WIRE_WEAK])
    assert ei.value.code == L.ERR_UNSUPPORTED
    dev.close()
