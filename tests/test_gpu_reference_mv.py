"""-m gpu: CX_SCHED_REFERENCE for d-dimensional messages (dim 2, 3, 4): one cx_sweep / cx_sweep_for is the reference's one
update_marginals! on ANY graph of pairwise linear-Gaussian factors, loops included.

The reference has no d-dimensional rule (parity unpinned for d > 1); what CAN be pinned is the ORDER — the scheduler never
looks at a value, so the restated engine (oracle/cortex_ref.c) run on the scalar twin of a graph records exactly the executions the
reference would make on the d-dimensional one — and the VALUES those executions leave when each is computed, in that order, from the
newest stored values with the d-dimensional rules of oracle/mv.py (the numpy restatement the flooding sweeps are checked against).  On a
chain that is the exact smoother, checked against the chain-scan schedule and the block-tridiagonal solve as well."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import ref
from oracle.mv import MvFlood, product
from tests.helpers import assert_close, engine_oracle_from_model

pytestmark = pytest.mark.gpu
SEED_VARIANCE = 50.0


def loopy_lgssm(T, d, seed, skips=(3,)):
    """lgssm_chain plus transition-like factors x_t -> x_{t + s}: every state on cycles, degrees <= 5"""
    m = cx.synth.lgssm_chain(T, d=d, seed=seed)
    rng = np.random.default_rng(seed + 1)
    x = m.x_ids
    nxt = int(m.factor_ids.max()) + 1
    ev, ef, role, fids = [m.edge_var], [m.edge_fac], [m.edge_role], [m.factor_ids]
    for s in skips:
        f = np.arange(nxt, nxt + T - s, dtype=np.int64); nxt += T - s
        ev += [x[:-s], x[s:]]; ef += [f, f]
        role += [np.full(T - s, L.ROLE_IN, np.int32), np.full(T - s, L.ROLE_OUT, np.int32)]
        fids.append(f)
    A2 = 0.6 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    psets = dict(m.psets); psets[2] = (A2, 0.4 * np.eye(d))
    n_new = sum(len(f) for f in fids[1:])
    return cx.synth.Model(edge_var=np.concatenate(ev), edge_fac=np.concatenate(ef), factor_ids=np.concatenate(fids),
                          factor_kind=np.full(len(m.factor_ids) + n_new, L.FACTOR_GAUSS_LINEAR, dtype=np.int32),
                          factor_var=np.concatenate([m.factor_var, np.full(n_new, 2.0)]), x_ids=x, data_var=m.data_var, data_fac=m.data_fac,
                          data_y=m.data_y, dim=d, edge_role=np.concatenate(role), psets=psets, meta=dict(m.meta, kind="loopy_lgssm"))


def scalar_twin(model):
    """the same bipartite graph with scalar additive factors and scalar data: what the restated engine schedules"""
    return cx.synth.Model(edge_var=model.edge_var, edge_fac=model.edge_fac, factor_ids=model.factor_ids,
                          factor_kind=np.ones(len(model.factor_ids), np.int32), factor_var=np.ones(len(model.factor_ids)), x_ids=model.x_ids,
                          data_var=model.data_var, data_fac=model.data_fac, data_y=np.asarray(model.data_y)[:, 0])


class MvSequential(MvFlood):
    """oracle/mv.py's rules, one execution at a time in a given order (each from the newest stored values)"""

    def execute(self, kind, var, fac):
        g = self.g
        if kind == L.ITEM_INDIVIDUAL_MARGINAL:
            return
        e = int(g.edge_index([var], [fac])[0])
        if kind == L.ITEM_MESSAGE_TO_VARIABLE:
            r = self._rule(e)
            assert r is not None, "the reference computes a signal only when its dependencies are computed"
            self.f2v[e] = r
        else:
            v = int(np.searchsorted(g.var_ids, var))
            acc = None
            for o in range(int(g.var_off[v]), int(g.var_off[v + 1])):
                if o != e:
                    assert self.f2v[o] is not None
                    acc = self.f2v[o] if acc is None else product(acc, self.f2v[o])
            self.v2f[e] = acc


def _oracle_rows(E):
    kinds = {ref.VAR_MSG_TO_FACTOR: L.ITEM_MESSAGE_TO_FACTOR, ref.VAR_MSG_TO_VARIABLE: L.ITEM_MESSAGE_TO_VARIABLE, ref.VAR_MARGINAL: L.ITEM_INDIVIDUAL_MARGINAL}
    rows = []
    for _r, _v, s, _b, _a in E.trace():
        k, v, f, _lo, _hi = E.variant(s)
        rows.append((kinds[k], v, f if kinds[k] != L.ITEM_INDIVIDUAL_MARGINAL else 0, 0, 0))
    return rows


@pytest.mark.parametrize("d,T,skips", [(2, 12, (3,)), (3, 20, (2,)), (4, 30, (5,)), (64, 10, (3,)), (7, 9, (2,))])      # (round 6: dim 64, and 5 .. 63 inside it)
def test_one_call_on_a_loopy_d_dimensional_graph(hip_lib, d, T, skips):
    model = loopy_lgssm(T, d, seed=10 + d, skips=skips)
    twin = scalar_twin(model)
    E = engine_oracle_from_model(twin, trace=True)
    seq = MvSequential(model)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev, seed_variance=SEED_VARIANCE)
    # the seeding a user of the reference does by hand on a loopy graph: every message out of a pairwise factor that is still undefined
    E.set_messages_to_variable(twin.edge_var, twin.edge_fac, np.zeros(len(twin.edge_var)), np.full(len(twin.edge_var), SEED_VARIANCE))
    for e in range(seq.g.ne):
        seq.f2v[e] = (np.zeros(d), SEED_VARIANCE * np.eye(d))
    for call in range(3):
        if call:      # new data: the likelihood messages become pending again, and everything downstream of them
            y = np.asarray(model.data_y) + 0.1 * call
            dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y)
            E.set_messages_to_factor(twin.data_var, twin.data_fac, y[:, 0], tag=ref.REAL)
            for e, row in zip(seq.g.edge_index(model.data_var, model.data_fac), y):
                seq.point[int(e)] = row
        dev.sweep_for(model.x_ids)
        E.update_marginals(twin.x_ids)
        rows = _oracle_rows(E)
        assert dev.ref_trace() == rows, f"d={d} call {call + 1}: the executions, in the reference's order"
        for k, v, f, _lo, _hi in rows:
            seq.execute(k, v, f)
        # every message of both directions, every requested marginal
        g = seq.g
        for direction, store in ((L.TO_VARIABLE, seq.f2v), (L.TO_FACTOR, seq.v2f)):
            got = dev.get_messages(g.edge_var, g.edge_fac, direction)
            for e in range(g.ne):
                if store[e] is None or e in seq.point:
                    continue
                assert_close(got[e, :d], store[e][0], 1e-8, f"d={d} call {call + 1} direction {direction} edge {e}: mean", scale_by="max")
                assert_close(got[e, d:].reshape(d, d), store[e][1], 1e-8, f"d={d} call {call + 1} direction {direction} edge {e}: covariance", scale_by="max")
        marg = dev.get_marginals(model.x_ids)
        for i, xv in enumerate(model.x_ids):
            mm, SS = seq.marginal(int(np.searchsorted(g.var_ids, xv)))
            assert_close(marg[i, :d], mm, 1e-8, f"d={d} call {call + 1}: marginal mean of {xv}", scale_by="max")
            assert_close(marg[i, d:].reshape(d, d), SS, 1e-8, f"d={d} call {call + 1}: marginal covariance of {xv}", scale_by="max")
    st = dev.ref_plan_stats()
    assert st["hits"] + st["misses"] == 3 and st["executions"] == len(rows)


@pytest.mark.parametrize("d", [2, 4, 64])
def test_one_call_on_a_chain_is_the_exact_smoother(hip_lib, d):
    """no seeding, one call: 5 T - 4 messages in the reference's forward / backward order + T marginals == the chain-scan schedule's sweep
    and the block-tridiagonal posterior"""
    from oracle.exact import lgssm_posterior
    T = 60
    model = cx.synth.lgssm_chain(T, d=d, seed=3)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    st = dev.ref_plan_stats()
    assert st["messages"] == 5 * T - 4 and st["executions"] == 6 * T - 4
    scan = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, scan)
    scan.sweep(1)
    a, b = dev.get_marginals(model.x_ids), scan.get_marginals(model.x_ids)
    assert_close(a, b, 1e-9, "reference order vs chain scan", scale_by="max")
    means, covs = lgssm_posterior(np.asarray(model.data_y), model.meta["A"], model.meta["Q"], model.meta["R"])
    assert_close(a[:, :d], means, 1e-8, "means vs the block-tridiagonal solve", scale_by="max")
    assert_close(a[:, d:].reshape(T, d, d), covs, 1e-8, "covariances vs the block-tridiagonal solve", scale_by="max")
    # a lazy request in the caller's order, and a checkpoint that carries the shadow
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, np.asarray(model.data_y) + 1.0)
    blob = dev.export_state()
    other = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, other)
    other.import_state(blob)
    for h in (dev, other):
        h.sweep_for(model.x_ids[::-1][:7])
    assert dev.ref_trace() == other.ref_trace() and len(dev.ref_trace()) > 7
    assert np.array_equal(dev.get_marginals(model.x_ids[-7:]), other.get_marginals(model.x_ids[-7:]))


def test_refusals_for_dim_above_one(hip_lib):
    cx.DeviceGraph(dim=64, schedule=L.SCHED_REFERENCE).close()      # (round 6) accepted: the stages through the dim 64 kernels
    model = cx.synth.lgssm_chain(8, d=2, seed=1)
    dev = cx.DeviceGraph(dim=2, schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev)
    with pytest.raises(cx.CortexHipError, match="wiring is fixed"):  # (round 6: dim 2 .. 4 take user wirings — right after cx_graph_create, like dim 1)
        dev.graph_wire([], [], [])
    d64 = cx.DeviceGraph(dim=64, schedule=L.SCHED_REFERENCE)
    m64 = cx.synth.lgssm_chain(4, d=64, seed=1)
    cx.synth.load_into_device(m64, d64)
    with pytest.raises(cx.CortexHipError, match="dim 2, 3, 4"):
        d64.graph_wire([], [], [])


@pytest.mark.parametrize("d", [3])
def test_new_rule_matrices_under_a_standing_reference_plan(hip_lib, d):
    """cx_set_factor_matrices between calls (the parameter-learning flow): a replayed plan is a captured HIP graph with the rule tables'
    address baked in; the tables are rewritten in place while their size holds, and when one more parameter set makes them move, the
    plans' graphs are dropped with the old allocation — the next call equals a fresh handle's under the new (A, Q)."""
    import copy

    T = 40
    model = cx.synth.lgssm_chain(T, d=d, seed=5)
    y = np.asarray(model.data_y)
    rng = np.random.default_rng(3)
    A_new = 0.6 * np.linalg.qr(rng.standard_normal((d, d)))[0]

    def call(h):
        h.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y)
        h.sweep(1)
        return h.get_marginals(model.x_ids)

    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev)
    first = [call(dev) for _ in range(4)][-1]
    assert dev.ref_plan_stats()["hits"] >= 2, "the test needs a replayed plan"
    assert not np.any(np.isnan(first))
    nsets = len(model.psets)
    dev.set_factor_matrices(nsets, np.eye(d), np.eye(d))            # one more set: the tables move
    dev.set_factor_matrices(0, A_new, 0.5 * np.eye(d))
    hits = dev.ref_plan_stats()["hits"]
    got = call(dev)
    assert dev.ref_plan_stats()["hits"] == hits + 1, "the same plan, replayed over the new tables"
    changed = copy.copy(model)
    changed.psets = dict(model.psets); changed.psets[0] = (A_new, 0.5 * np.eye(d))
    fresh = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(changed, fresh)
    want = call(fresh)
    assert np.max(np.abs(first - want)) > 1e-3, "the two parameter sets must give different posteriors for this test to mean anything"
    assert_close(got, want, 1e-10, "marginals after new matrices vs a fresh handle", scale_by="max")
    dev.set_factor_matrices(0, *model.psets[0])                     # same size: in place
    assert_close(call(dev), first, 1e-10, "back to the first parameters", scale_by="max")
    dev.close(); fresh.close()


class MvBySignal(MvFlood):
    """oracle/mv.py's rules, one execution of the restated engine at a time, every product folded over the signal's OWN dependency list
    (E.dependencies): for a variable of degree above 5 those are segment-tree nodes, which may lag behind their leaves on a graph with loops"""

    def __init__(self, model, E):
        super().__init__(model)
        self.E, self.prod = E, {}

    def value_of(self, s):
        k, v, f, lo, hi = self.E.variant(s)
        if k == ref.VAR_MSG_TO_VARIABLE:
            return self.f2v[int(self.g.edge_index([v], [f])[0])]
        assert k == ref.VAR_PRODUCT
        return self.prod.get((v, lo, hi))

    def execute_signal(self, s):
        k, v, f, lo, hi = self.E.variant(s)
        if k == ref.VAR_MSG_TO_VARIABLE:
            e = int(self.g.edge_index([v], [f])[0])
            r = self._rule(e)
            assert r is not None
            self.f2v[e] = r
            return
        acc = None
        for d in self.E.dependencies(s):
            val = self.value_of(d)
            assert val is not None, "the reference computes a signal only when its dependencies are computed"
            acc = val if acc is None else product(acc, val)
        if k == ref.VAR_MSG_TO_FACTOR:
            self.v2f[int(self.g.edge_index([v], [f])[0])] = acc
        elif k == ref.VAR_PRODUCT:
            self.prod[(v, lo, hi)] = acc
        else:
            self.marg = getattr(self, "marg", {})
            self.marg[v] = acc


@pytest.mark.parametrize("d,T,skips", [(2, 16, (2, 3, 5)), (3, 14, (2, 3, 4)), (64, 12, (2, 3, 4))])
def test_one_call_with_variables_of_degree_above_five(hip_lib, d, T, skips):
    """three skip links per state: interior states have degree 9 — their messages and marginals hang off segment-tree nodes
    (dependencies.jl:90-173), list sums of k_batch_mv; executions, every message, every node and every marginal against the restated
    engine's order executed with d-dimensional arithmetic, three calls"""
    from tests.test_gpu_reference_schedule import _oracle_trace

    model = loopy_lgssm(T, d, seed=20 + d, skips=skips)
    twin = scalar_twin(model)
    E = engine_oracle_from_model(twin, trace=True)
    seq = MvBySignal(model, E)
    g = seq.g
    assert int(np.max(np.diff(g.var_off))) >= 7
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev, seed_variance=SEED_VARIANCE)
    E.set_messages_to_variable(twin.edge_var, twin.edge_fac, np.zeros(len(twin.edge_var)), np.full(len(twin.edge_var), SEED_VARIANCE))
    for e in range(g.ne):
        seq.f2v[e] = (np.zeros(d), SEED_VARIANCE * np.eye(d))
    for call in range(3):
        if call:
            y = np.asarray(model.data_y) + 0.1 * call
            dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y)
            E.set_messages_to_factor(twin.data_var, twin.data_fac, y[:, 0], tag=ref.REAL)
            for e, row in zip(g.edge_index(model.data_var, model.data_fac), y):
                seq.point[int(e)] = row
        dev.sweep_for(model.x_ids)
        E.update_marginals(twin.x_ids)
        assert dev.ref_trace() == _oracle_trace(E), f"d={d} call {call + 1}: the executions, in the reference's order"
        order = [s for _r, _v, s, _b, _a in E.trace()]
        assert any(E.variant(s)[0] == ref.VAR_PRODUCT for s in order)
        for s in order:
            seq.execute_signal(s)
        for direction, store in ((L.TO_VARIABLE, seq.f2v), (L.TO_FACTOR, seq.v2f)):
            got = dev.get_messages(g.edge_var, g.edge_fac, direction)
            for e in range(g.ne):
                if store[e] is None or e in seq.point:
                    continue
                assert_close(got[e, :d], store[e][0], 1e-8, f"d={d} call {call + 1} direction {direction} edge {e}: mean", scale_by="max")
                assert_close(got[e, d:].reshape(d, d), store[e][1], 1e-8, f"d={d} call {call + 1} direction {direction} edge {e}: covariance", scale_by="max")
        marg = dev.get_marginals(model.x_ids)
        for i, xv in enumerate(model.x_ids):
            mm, SS = seq.marg[int(xv)]
            assert_close(marg[i, :d], mm, 1e-8, f"d={d} call {call + 1}: marginal mean of {xv}", scale_by="max")
            assert_close(marg[i, d:].reshape(d, d), SS, 1e-8, f"d={d} call {call + 1}: marginal covariance of {xv}", scale_by="max")
        keys = sorted(seq.prod)
        got = dev.get_products([k[0] for k in keys], [k[1] for k in keys], [k[2] for k in keys])
        for row, k in zip(got, keys):
            mm, SS = seq.prod[k]
            assert_close(row[:d], mm, 1e-8, f"d={d} call {call + 1}: node {k} mean", scale_by="max")
            assert_close(row[d:].reshape(d, d), SS, 1e-8, f"d={d} call {call + 1}: node {k} covariance", scale_by="max")


@pytest.mark.parametrize("d,n_factors,seed", [(2, 6, 2), (4, 15, 3)])
def test_factors_of_three_to_seven_variables(hip_lib, d, n_factors, seed):
    """x_out = A_1 x_1 + ... + A_k x_k + N(0, Q) with up to six inputs on a tree: the message out of such a factor is one item (the entry of the
    factor's table, cx_kary_mv_core.h) that reads the stored messages of ALL the factor's other edges.  One call from the priors is the
    exact posterior (a dense joint solve), the executions are those of the scalar handle on the same graph (the order never depends on
    what a message is; the scalar plans are pinned against the restated engine), and a second call after new priors replays"""
    from tests.test_gpu_kary_mv import _kary_tree, _load

    model, prior, facs, fid, sets, mean, cov = _kary_tree(n_factors, d, seed, k_choices=(2, 3, 5, 6))
    n = len(model.x_ids)
    dev = _load(model, prior, facs, fid, sets, L.SCHED_REFERENCE)
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    assert not np.any(np.isnan(marg))
    assert_close(marg[:, :d], mean, 1e-8, "marginal mean vs the joint solve", scale_by="max")
    assert_close(marg[:, d:].reshape(n, d, d), cov, 1e-8, "marginal covariance vs the joint solve", scale_by="max")
    trace = dev.ref_trace()
    # the scalar handle on the same bipartite graph
    twin = cx.synth.Model(edge_var=model.edge_var, edge_fac=model.edge_fac, factor_ids=model.factor_ids, factor_kind=model.factor_kind,
                          factor_var=np.ones(len(model.factor_ids)), x_ids=model.x_ids, edge_role=model.edge_role)
    sc = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(twin, sc)
    sc.set_messages(model.x_ids, model.x_ids + n, L.TO_VARIABLE, L.FORM_NATURAL, np.ones((n, 2)))
    sc.sweep(1)
    assert trace == sc.ref_trace() and len(trace) > 3 * n
    # new priors: pending again, the same plan
    eta, W = prior
    for _ in range(2):
        dev.set_messages(model.x_ids, model.x_ids + n, L.TO_VARIABLE, L.FORM_NATURAL, np.concatenate([eta, W.reshape(n, d * d)], axis=1))
        dev.sweep(1)
    assert dev.ref_plan_stats()["hits"] >= 1
    assert_close(dev.get_marginals(model.x_ids), marg, 1e-10, "the same priors again", scale_by="max")
    dev.close(); sc.close()


def test_factors_of_several_variables_on_a_cycle(hip_lib):
    """two factors of four variables that share two of them (x2 - F1 - x3 - F2 - x2 is a cycle), every variable with a prior: three calls with
    new priors in between; the executions are the scalar handle's, and every message of both directions equals those executions carried
    out one at a time in numpy (the moment-form formulas of csrc/cx_kary_mv_core.h on the NEWEST stored messages of the other edges)"""
    d = 2
    rng = np.random.default_rng(4)
    sets = {s: (0.7 * np.linalg.qr(rng.standard_normal((d, d)))[0], (0.3 + 0.1 * s) * np.eye(d) + 0.05) for s in range(3)}
    n = 6
    x = np.arange(1, n + 1, dtype=np.int64); unary = x + n; F = np.array([2 * n + 1, 2 * n + 2], np.int64)
    members = {int(F[0]): (4, [1, 2, 3]), int(F[1]): (5, [2, 3, 6])}      # factor: (out, inputs)
    eset = {(1, int(F[0])): 0, (2, int(F[0])): 1, (3, int(F[0])): 2, (2, int(F[1])): 2, (3, int(F[1])): 0, (6, int(F[1])): 1}
    qset = {int(F[0]): 0, int(F[1]): 1}
    ev, ef, role = list(x), list(unary), [L.ROLE_OUT] * n
    for f, (out, ins) in members.items():
        ev.append(out); ef.append(f); role.append(L.ROLE_OUT)
        for i in ins:
            ev.append(i); ef.append(f); role.append(L.ROLE_IN)
    ev, ef = np.array(ev, np.int64), np.array(ef, np.int64)
    kinds = np.concatenate([np.zeros(n, np.int32), np.full(2, L.FACTOR_GAUSS_LINEAR_N, np.int32)])
    model = cx.synth.Model(edge_var=ev, edge_fac=ef, factor_ids=np.concatenate([unary, F]), factor_kind=kinds,
                           factor_var=np.concatenate([np.zeros(n), [qset[int(F[0])], qset[int(F[1])]]]).astype(float), x_ids=x, dim=d,
                           edge_role=np.array(role, np.int32), psets=sets)
    twin = cx.synth.Model(edge_var=ev, edge_fac=ef, factor_ids=model.factor_ids, factor_kind=kinds, factor_var=np.ones(n + 2), x_ids=x, edge_role=model.edge_role)
    dev, sc = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE), cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev); cx.synth.load_into_device(twin, sc)
    kv = [v for (v, f) in eset]; kf = [f for (v, f) in eset]
    dev.set_factor_edge_sets(kv, kf, [eset[k] for k in eset])
    # the seeds a user of the reference sets by hand on a loopy graph: every message out of the two factors
    big = ev >= 0
    big[:n] = False
    seed = np.concatenate([np.zeros(d), (1e-3 * np.eye(d)).reshape(-1)])
    dev.set_messages(ev[big], ef[big], L.TO_VARIABLE, L.FORM_NATURAL, np.tile(seed, (int(big.sum()), 1)))
    sc.set_messages(ev[big], ef[big], L.TO_VARIABLE, L.FORM_NATURAL, np.tile([0.0, 1e-3], (int(big.sum()), 1)))
    f2v = {(int(v), int(f)): (np.zeros(d), 1e-3 * np.eye(d)) for v, f in zip(ev[big], ef[big])}      # natural (eta, Lambda)
    v2f = {}

    def moment(nat):
        V = np.linalg.inv(nat[1])
        return V @ nat[0], V

    def execute(kind, v, f):
        if kind == L.ITEM_MESSAGE_TO_FACTOR:
            others = [f2v[(v, g)] for g in ef[ev == v] if int(g) != f]
            v2f[(v, f)] = (sum(o[0] for o in others), sum(o[1] for o in others))
        elif kind == L.ITEM_MESSAGE_TO_VARIABLE:
            out, ins = members[f]
            A = {i: sets[eset[(i, f)]][0] for i in ins}
            Q = sets[qset[f]][1]
            m = {i: moment(v2f[(i, f)]) for i in [out] + ins if i != v}
            if v == out:
                mu, S = sum(A[i] @ m[i][0] for i in ins), Q + sum(A[i] @ m[i][1] @ A[i].T for i in ins)
                Si = np.linalg.inv(S)
                f2v[(v, f)] = (Si @ mu, Si)
            else:
                mu = m[out][0] - sum(A[i] @ m[i][0] for i in ins if i != v)
                S = m[out][1] + Q + sum(A[i] @ m[i][1] @ A[i].T for i in ins if i != v)
                Si = np.linalg.inv(S)
                f2v[(v, f)] = (A[v].T @ Si @ mu, A[v].T @ Si @ A[v])

    for call in range(3):
        W = np.stack([np.eye(d) * rng.uniform(0.5, 2.0) + 0.1 for _ in range(n)]); eta = rng.standard_normal((n, d))
        dev.set_messages(x, unary, L.TO_VARIABLE, L.FORM_NATURAL, np.concatenate([eta, W.reshape(n, d * d)], axis=1))
        sc.set_messages(x, unary, L.TO_VARIABLE, L.FORM_NATURAL, np.ones((n, 2)))
        for i in range(n):
            f2v[(int(x[i]), int(unary[i]))] = (eta[i], W[i])
        dev.sweep(1); sc.sweep(1)
        trace = dev.ref_trace()
        assert trace == sc.ref_trace() and sum(1 for r in trace if r[0] == L.ITEM_MESSAGE_TO_VARIABLE) >= 8
        for k, v, f, _lo, _hi in trace:
            execute(k, int(v), int(f))
        for direction, store in ((L.TO_VARIABLE, f2v), (L.TO_FACTOR, v2f)):
            keys = sorted(store)
            got = dev.get_messages([k[0] for k in keys], [k[1] for k in keys], direction, L.FORM_NATURAL)
            for row, k in zip(got, keys):
                assert_close(row[:d], store[k][0], 1e-8, f"call {call + 1} direction {direction} {k}: eta", scale_by="max")
                assert_close(row[d:].reshape(d, d), store[k][1], 1e-8, f"call {call + 1} direction {direction} {k}: Lambda", scale_by="max")
        marg = dev.get_marginals(x)
        for i in range(n):
            tot = [f2v[(int(x[i]), int(g))] for g in ef[ev == x[i]]]
            mm, VV = moment((sum(t[0] for t in tot), sum(t[1] for t in tot)))
            assert_close(marg[i, :d], mm, 1e-8, f"call {call + 1}: marginal mean of x{i + 1}", scale_by="max")
            assert_close(marg[i, d:].reshape(d, d), VV, 1e-8, f"call {call + 1}: marginal covariance of x{i + 1}", scale_by="max")
    dev.close(); sc.close()


@pytest.mark.parametrize("d", [2, 4])
def test_a_filter_wiring_of_a_d_dimensional_state_space_model(hip_lib, d):
    """(round 6) cx_graph_wire for dim 2 .. 4: a user resolver that wires the forward messages only — one call is the Kalman FILTER
    of the linear-Gaussian state-space model (numpy, textbook form), not the smoother; the marginals are sums of the wiring's own
    dependency lists (the backward messages are never computed and never read)"""
    T = 150
    model = cx.synth.lgssm_chain(T, d=d, seed=6)
    A, Q, Rm = model.meta["A"], model.meta["Q"], model.meta["R"]
    x, y = model.x_ids, model.data_var
    lik, tr = model.data_fac, np.setdiff1d(model.factor_ids, model.data_fac)
    assert len(lik) == T and len(tr) == T - 1
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
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
    for k, (Ak, Qk) in model.psets.items():
        dev.set_factor_matrices(k, Ak, Qk)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, edge_role=model.edge_role)
    dev.graph_wire([s for s, _d, _f in triples], [dd for _s, dd, _f in triples], [f for _s, _d, f in triples])
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, model.data_y)
    dev.sweep_for(x)
    assert dev.ref_plan_stats()["executions"] == T + 2 * (T - 1) + T
    Y = np.asarray(model.data_y)
    m, P = Y[0].copy(), Rm.copy()
    fm, fP = [m], [P]
    for t in range(1, T):
        mp, Pp = A @ m, A @ P @ A.T + Q
        K = Pp @ np.linalg.inv(Pp + Rm)
        m, P = mp + K @ (Y[t] - mp), (np.eye(d) - K) @ Pp
        fm.append(m); fP.append(P)
    marg = dev.get_marginals(x)
    assert_close(marg[:, :d], np.array(fm), 1e-8, "filtered means", scale_by="max")
    assert_close(marg[:, d:].reshape(T, d, d), np.array(fP), 1e-8, "filtered covariances", scale_by="max")
    from oracle.exact import lgssm_posterior
    sm, sP = lgssm_posterior(Y, A, Q, Rm)
    assert np.max(np.abs(marg[:-1, d:].reshape(T - 1, d, d) - sP[:-1])) > 1e-4, "a filter is not the smoother"
    assert_close(marg[-1, :d], sm[-1], 1e-8, "the last state: filter == smoother", scale_by="max")
