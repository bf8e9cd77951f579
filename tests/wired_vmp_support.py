"""Test support: the variational models of tests/vmp_support.py run through cx_graph_wire (CX_SCHED_REFERENCE).

The resolvers of tests/vmp_support.py (transcriptions of test/inference_engine_tests.jl:599-621 and :816-897) are written against a small
back-end protocol.  `WiringRecorder` is a third back-end: it RECORDS the resolver's add_dependency! / link_signal_to_variable! calls as
cx_graph_wire triples.  Two executors sit behind it:

  * ShadowBackend — the GPU-free host logic (cx_refsched.h built for the CPU, tests/hostlogic.py): the scheduler's executions per call,
    levelled into stages, and the stages executed by a numpy restatement of the device items (cx_batch.hip: batch_item / vmp_item)
  * DeviceBackend — the product: DeviceGraph(schedule = CX_SCHED_REFERENCE) + graph_wire + set_marginals + sweep_for

Both are compared call by call with OracleBackend (oracle/cortex_ref.c driven by the transcribed rules)."""
import numpy as np

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import ref
from tests import vmp_support as vs

K2F, K2V, KMARG, KPROD, KJOINT = L.ITEM_MESSAGE_TO_FACTOR, L.ITEM_MESSAGE_TO_VARIABLE, L.ITEM_INDIVIDUAL_MARGINAL, L.ITEM_PRODUCT_OF_MESSAGES, L.ITEM_JOINT_MARGINAL
ORACLE_KIND = {ref.VAR_MSG_TO_FACTOR: K2F, ref.VAR_MSG_TO_VARIABLE: K2V, ref.VAR_MARGINAL: KMARG, ref.VAR_PRODUCT: KPROD, ref.VAR_JOINT: KJOINT}
FACTOR_NORMAL_PRECISION, FACTOR_OPAQUE = 3, 0
ROLE_OUT, ROLE_IN, ROLE_PRECISION = 0, 1, 2


class TracedOracleBackend(vs.OracleBackend):
    """OracleBackend with the engine's trace switched on: rows (kind, variable id, factor id, lo, hi) per execution of the last call"""

    def __init__(self, rule):
        super().__init__(rule)
        self.E = ref.Engine(ref.P_CALLBACK, trace=True)
        self.E.set_rule(lambda s: rule(self, s))

    def trace_rows(self):
        rows = []
        for _r, _v, s, _b, _a in self.E.trace():
            k, v, f, lo, hi = self.E.variant(s)
            kind = ORACLE_KIND[k]
            rows.append((kind, v if kind not in (KJOINT,) else 0, f if kind in (K2F, K2V, KJOINT) else 0, lo if kind == KPROD else 0, hi if kind == KPROD else 0))
        return rows


class WiringRecorder:
    """the resolver protocol of tests/vmp_support.py, recording cx_graph_wire triples.  Ids are handed out like the oracle's (variables and
    factors share one counter), so traces compare id by id.  gamma_names: the variable names that are precisions."""

    vectorised = None      # "mean_field" / "structured": resolve() takes the triples from cortex.jl_amd.wiring instead of running the resolvers

    def __init__(self, gamma_names=("ssnoise", "obsnoise")):
        self.names, self.forms, self.edges, self.roles = {}, {}, [], {}
        self.next_id = 1
        self.gamma_names = set(gamma_names)
        self.sig, self.dep, self.flags = [], [], []
        self._has_out = set()

    # graph
    def add_variable(self, name):
        v = self.next_id; self.next_id += 1
        self.names[v] = name
        return v

    def add_factor(self, form):
        f = self.next_id; self.next_id += 1
        self.forms[f] = form
        return f

    def add_edge(self, v, f, role=None):
        if role is None:      # (the reference labels every edge of its test models :out / :in without meaning: the rules go by variable name)
            role = ROLE_PRECISION if self.names[v] in self.gamma_names else (ROLE_IN if f in self._has_out else ROLE_OUT)
        if role == ROLE_OUT:
            self._has_out.add(f)
        self.edges.append((v, f)); self.roles[(v, f)] = role

    # resolver protocol
    def marginal(self, v): return (KMARG, v, 0)
    def message_to_variable(self, v, f): return (K2V, v, f)
    def message_to_factor(self, v, f): return (K2F, v, f)
    def connected_variables(self, f): return sorted(v for v, ff in self.edges if ff == f)
    def connected_factors(self, v): return sorted(f for vv, f in self.edges if vv == v)
    def variable_name(self, v): return self.names[v]
    def factor_form(self, f): return self.forms[f]
    def is_joint(self, s): return s[0] == KJOINT
    def same(self, a, b): return a == b
    def new_joint_marginal(self, factor_id, cluster): return (KJOINT, 0, factor_id)
    def add_local_marginal_to_factor(self, f, s): pass

    def add_dependency(self, s, d, weak=False, intermediate=False, listen=True):
        self.sig.append(s); self.dep.append(d)
        self.flags.append((L.WIRE_WEAK if weak else 0) | (L.WIRE_INTERMEDIATE if intermediate else 0) | (0 if listen else L.WIRE_NO_LISTEN))

    def link_signal_to_variable(self, v, s):
        self.sig.append(s); self.dep.append((KMARG, v, 0)); self.flags.append(L.WIRE_LINK)

    def resolve(self, factor_resolver, variable_resolver):
        """resolve_dependencies!, dependencies.jl:5-15: factors first, then variables, each in ascending id order"""
        self._index()
        if self.vectorised == "from_engine":
            # the resolvers run on the HOST mirror of the reference's API; the wiring is read back from its signals (what HipProcessor.attach does)
            mirror = vs.MirrorBackend(rule=None)
            ids = {}
            for i in range(1, self.next_id):
                ids[i] = mirror.add_variable(self.names[i]) if i in self.names else mirror.add_factor(self.forms[i])
                assert ids[i] == i, "the mirror hands out the same ids"
            for v, f in self.edges:
                mirror.add_edge(v, f)
            mirror.resolve(factor_resolver, variable_resolver)
            t = cx.wiring.from_engine(mirror.engine)
            self.sig, self.dep, self.flags = t.signals, t.dependencies, t.flags
            return self.wired()
        if self.vectorised:
            ev, ef, role, _fids, _kinds = self.graph_arrays()
            priors = [f for f, form in self.forms.items() if form == "prior"]
            if self.vectorised == "mean_field":
                t = cx.wiring.mean_field(ev, ef, role, prior_factors=priors)
            else:
                t = cx.wiring.structured(ev, ef, role, [f for f, form in self.forms.items() if form == "transition"], prior_factors=priors)
            self.sig, self.dep, self.flags = t.signals, t.dependencies, t.flags
            return self.wired()
        for f in sorted(self.forms):
            factor_resolver(self, f)
        for v in sorted(self.names):
            if variable_resolver is None:
                self.sig.append((KMARG, v, 0)); self.dep.append((KMARG, v, 0)); self.flags.append(L.WIRE_DEFAULT_VARIABLE)
            else:
                variable_resolver(self, v)
        self.wired()

    def _index(self):
        by_f, by_v = {}, {}
        for v, f in self.edges:
            by_f.setdefault(f, []).append(v); by_v.setdefault(v, []).append(f)
        self._by_f = {f: sorted(vs_) for f, vs_ in by_f.items()}
        self._by_v = {v: sorted(fs) for v, fs in by_v.items()}
        self.connected_variables = lambda f: self._by_f.get(f, [])
        self.connected_factors = lambda v: self._by_v.get(v, [])

    def graph_arrays(self, factor_kinds=None):
        ev = np.array([v for v, _f in self.edges], dtype=np.int64); ef = np.array([f for _v, f in self.edges], dtype=np.int64)
        role = np.array([self.roles[e] for e in self.edges], dtype=np.int32)
        fids = np.array(sorted(self.forms), dtype=np.int64)
        kinds = np.array([(factor_kinds or {}).get(self.forms[int(f)], FACTOR_NORMAL_PRECISION) for f in fids], dtype=np.int32)
        return ev, ef, role, fids, kinds

    def wired(self):
        raise NotImplementedError


def _is_gamma(be, v):
    return be.names[v] in be.gamma_names


class ShadowBackend(WiringRecorder):
    """the GPU-free scheduler + a numpy executor of the levelled items"""

    def __init__(self, gamma_names=("ssnoise", "obsnoise"), factor_kinds=None):
        super().__init__(gamma_names)
        self.factor_kinds = factor_kinds
        self.last_rows, self.last_stages = [], 0

    def wired(self):
        from tests.hostlogic import FlatGraph
        ev, ef, role, fids, kinds = self.graph_arrays(self.factor_kinds)
        g = FlatGraph(ev, ef, fids, kinds, np.zeros(len(fids)), edge_role=role, schedule=L.SCHED_REFERENCE)
        assert g.status == 0, g.error
        rc, err = g.ref_wire(self.sig, self.dep, self.flags)
        assert rc == 0, err
        self.g = g
        self.var_ids, self.vbase, self.var_off, self.vinfo = g.arr("var_ids"), g.arr("vbase"), g.arr("var_off"), g.arr("vinfo")
        edge_var, edge_fac = g.arr("edge_var"), g.arr("edge_fac_id")
        ns = g.scalar("nslots")
        self.f2v = np.full((ns, 2), np.nan); self.v2f = np.full((ns, 2), np.nan)
        self.marg = np.full((g.scalar("nv"), 2), np.nan)
        self.prod = np.full((max(1, g.ref_scalar("products")), 2), np.nan)
        self.joint = np.full((g.scalar("nf"), 6), np.nan)
        self.slot = {}
        for e in range(len(edge_var)):
            v = edge_var[e]
            big = (self.vinfo[v] & 0x0f) == 0x0f
            self.slot[(int(self.var_ids[v]), int(edge_fac[e]))] = int(self.vbase[v] + (e - self.var_off[v]) * (1 if big else 256))

    def local(self, v):
        return int(np.searchsorted(self.var_ids, v))

    def set_marginal(self, v, value):
        tag, p = value
        self.marg[self.local(v)] = (p[0], 0.0) if tag == ref.REAL else (p[0], 1.0 / p[1]) if tag == ref.NORMAL_MP else (p[0], p[1])
        self.g.ref_set_marginals([v])

    def set_message_to_variable(self, v, f, natural):
        self.f2v[self.slot[(v, f)]] = natural
        self.g.ref_set(L.TO_VARIABLE, [v], [f])

    def get_marginal(self, v):
        a, b = self.marg[self.local(v)]
        return (ref.GAMMA, [a, b]) if _is_gamma(self, v) else (ref.NORMAL_MP, [a, 1.0 / b])

    def update_marginals(self, ids):
        rows = self.g.ref_update(np.atleast_1d(ids))
        self.last_rows = [tuple(int(x) for x in r[:5]) for r in rows]
        rc, err = self.g.ref_level()
        assert rc == 0, err
        rec, off, lists = self.g.arr("ref_rec"), self.g.arr("ref_stage_off"), self.g.arr("ref_list")
        wrec, woff = self.g.arr("ref_wide_rec"), self.g.arr("ref_wide_off")      # list items of many sources travel apart (k_wide_sum): same stage, same meaning
        self.last_stages, self.last_wide = len(off) - 1, len(wrec) // 5
        self.run(rec, off, lists, wrec, woff)

    def trace_rows(self):
        return self.last_rows

    def run(self, rec, stage_off, lists, wide_rec=(), wide_off=()):
        """cx_batch.hip: batch_item for the kinds a wired plan holds; asserts that no item of a stage reads what another one writes.  A record
        that LEADS is followed by one the same thread computes behind it (cx_refsched.h: kRecLeads): the follower reads its leader's result"""
        LEADS, FOLLOWS, MASK = 0x40000000, 0x20000000, 0x0fffffff
        for s in range(len(stage_off) - 1):
            items = rec[5 * stage_off[s]:5 * stage_off[s + 1]].reshape(-1, 5)
            if len(wide_off):
                wide = np.asarray(wide_rec[5 * wide_off[s]:5 * wide_off[s + 1]]).reshape(-1, 5)
                assert all(int(k) in (64, 65, 66, 72) for k in wide[:, 0])
                items = np.concatenate([items, wide])
            reads, writes, new = set(), set(), []
            leader_out = None
            for k, idx, v, lo, hi in items:
                flags, k, idx, lo, hi = int(k) & ~MASK, int(k) & MASK, int(idx), int(lo), int(hi)
                assert bool(flags & FOLLOWS) == (leader_out is not None), f"stage {s}: a follower without its leader"
                ls = [int(t) for t in lists[lo:lo + hi]]

                def val(b, i):
                    if leader_out is not None and leader_out[0] == (b, i):
                        return np.asarray(leader_out[1], dtype=np.float64)
                    return getattr(self, b)[i]
                if k in (64, 65, 66, 72):
                    src = [("f2v", t) if t >= 0 else ("prod", ~t) for t in ls]
                    dst = ({64: "v2f", 65: "prod", 66: "marg", 72: "marg"}[k], idx)
                    acc = np.array([val(b, i) for b, i in src]).sum(axis=0)
                    out = (acc[0] / acc[1], 1.0 / acc[1]) if k == 66 else (acc[0] + 1.0, 1.0 / acc[1]) if k == 72 else tuple(acc)
                elif k == 67:
                    src, dst = [("marg", ls[0]), ("marg", ls[1])], ("f2v", idx)
                    eg = self.marg[ls[1]][0] * self.marg[ls[1]][1]
                    out = (self.marg[ls[0]][0] * eg, eg)
                elif k == 68:
                    src, dst = [("marg", ls[0]), ("marg", ls[1])], ("f2v", idx)
                    a, b = self.marg[ls[0]], self.marg[ls[1]]
                    out = (0.5, 0.5 * (a[1] + b[1] + (a[0] - b[0]) ** 2))
                elif k == 69:
                    src, dst = [("v2f", ls[0]), ("marg", ls[1])], ("f2v", idx)
                    m, eg = val("v2f", ls[0]), self.marg[ls[1]][0] * self.marg[ls[1]][1]
                    w = 1.0 / (1.0 / m[1] + 1.0 / eg)
                    out = (m[0] / m[1] * w, w)
                elif k == 70:
                    src, dst = [("v2f", ls[0]), ("v2f", ls[1]), ("marg", ls[2])], ("joint", idx)
                    m1, m2, eg = self.v2f[ls[0]], self.v2f[ls[1]], self.marg[ls[2]][0] * self.marg[ls[2]][1]
                    W = np.array([[m1[1] + eg, -eg], [-eg, m2[1] + eg]])
                    V = np.linalg.inv(W)
                    mu = V @ np.array([m1[0], m2[0]])
                    out = (mu[0], mu[1], V[0, 0], V[0, 1], V[1, 0], V[1, 1])
                elif k == 71:
                    src, dst = [("joint", ls[0])], ("f2v", idx)
                    o = self.joint[ls[0]]
                    out = (0.5, 0.5 * (o[2] - o[3] - o[4] + o[5] + (o[0] - o[1]) ** 2))
                else:
                    raise AssertionError(f"unexpected item kind {k} in a wired plan")
                for b, i in src:
                    assert not np.any(np.isnan(val(b, i))), f"stage {s}: an item of kind {k} reads an undefined {b}[{i}]"
                if flags & FOLLOWS:
                    assert leader_out[0] in src, f"stage {s}: a follower that does not read its leader"
                reads.update(t for t in src if leader_out is None or t != leader_out[0]); new.append((dst, out))
                assert dst not in writes, f"stage {s}: two items write {dst}"
                writes.add(dst)
                leader_out = (dst, out) if flags & LEADS else None
            assert leader_out is None
            assert not (reads & writes), f"stage {s}: an item reads what another item of the same stage writes: {sorted(reads & writes)[:3]}"
            for (b, i), out in new:
                getattr(self, b)[i] = out


class DeviceBackend(WiringRecorder):
    """the product under CX_SCHED_REFERENCE with the recorded wiring"""

    def __init__(self, gamma_names=("ssnoise", "obsnoise"), factor_kinds=None):
        super().__init__(gamma_names)
        self.factor_kinds = factor_kinds

    def wired(self):
        ev, ef, role, fids, kinds = self.graph_arrays(self.factor_kinds)
        self.dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        self.dev.graph_create(ev, ef, fids, kinds, np.zeros((len(fids), 4)), edge_role=role)
        self.dev.graph_wire(self.sig, self.dep, self.flags)

    def set_marginal(self, v, value):
        tag, p = value
        form = L.FORM_POINT if tag == ref.REAL else L.FORM_MEAN_PRECISION if tag == ref.NORMAL_MP else L.FORM_GAMMA
        self.dev.set_marginals([v], form, p[:1] if tag == ref.REAL else p[:2])

    def set_marginals(self, ids, tag, payload):
        form = L.FORM_POINT if tag == ref.REAL else L.FORM_MEAN_PRECISION if tag == ref.NORMAL_MP else L.FORM_GAMMA
        self.dev.set_marginals(ids, form, payload)

    def set_message_to_variable(self, v, f, natural):
        self.dev.set_messages([v], [f], L.TO_VARIABLE, L.FORM_NATURAL, np.asarray(natural, dtype=np.float64).reshape(1, 2))

    def get_marginal(self, v):
        a, b = self.dev.get_marginals([v])[0]
        return (ref.GAMMA, [a, b]) if _is_gamma(self, v) else (ref.NORMAL_MP, [a, 1.0 / b])

    def update_marginals(self, ids):
        self.dev.sweep_for(np.atleast_1d(ids))

    def trace_rows(self):
        return [(k, 0 if k == KJOINT else v, f if k in (K2F, K2V, KJOINT) else 0, lo, hi) for k, v, f, lo, hi in self.dev.ref_trace()]


def assert_same_value(got, want, rtol, what):
    assert got[0] == want[0], f"{what}: value type {got[0]} vs {want[0]}"
    g, w = np.asarray(got[1][:2], dtype=np.float64), np.asarray(want[1][:2], dtype=np.float64)
    assert np.allclose(g, w, rtol=rtol, atol=0.0), f"{what}: {g} vs {w}"


# ---- a third model: a TREE of latent means with grouped unknown precisions and priors --------------------------------------------------
# (neither of the reference's two test models: several precisions of each kind, a tree instead of a chain, states of any degree, proper
# priors as unary factors whose messages the caller sets — the same rules, another graph and another wiring instance)
def to_natural(value):
    tag, p = value
    return (p[0] * p[1], p[1]) if tag == ref.NORMAL_MP else (p[0] - 1.0, 1.0 / p[1])


def _set_prior(be, v, f, value):
    if hasattr(be, "set_prior"):
        be.set_prior(v, f, value)
    elif isinstance(be, vs.OracleBackend):
        be.E.set_value_ex(be.E.message_to_variable(v, f), value[0], value[1])
    else:
        be.set_message_to_variable(v, f, to_natural(value))


def tree_factor_resolver(api, f):
    """the structured resolver of the reference's test, with prior factors left alone (their messages are values the user sets)"""
    if api.factor_form(f) != "prior":
        vs.structured_factor(api, f)


class TreeModel:
    pass


def make_tree_model(be, K, seed, n_tp=2, n_op=2, max_obs=3):
    rng = np.random.default_rng(seed)
    m = TreeModel()
    m.tp = [be.add_variable("ssnoise") for _ in range(n_tp)]
    m.op = [be.add_variable("obsnoise") for _ in range(n_op)]
    m.x = [be.add_variable("x") for _ in range(K)]
    m.parent = [-1] + [int(rng.integers(max(0, j - 6), j)) for j in range(1, K)]      # a random recursive tree, bushy near every node
    m.tgroup = [0] + [(j - 1) if j <= n_tp else int(rng.integers(0, n_tp)) for j in range(1, K)]      # (every precision has a factor)
    tp_true, op_true = [25.0, 100.0, 60.0][:n_tp], [50.0, 200.0, 120.0][:n_op]
    xt = np.zeros(K)
    xt[0] = rng.standard_normal()
    for j in range(1, K):
        xt[j] = xt[m.parent[j]] + rng.standard_normal() / np.sqrt(tp_true[m.tgroup[j]])
    m.obs = []                                        # (y variable, state index, group, datum)
    for j in range(K):
        # (at least one observation each: a leaf state without one has a MessageToFactor with no dependencies — a variable of degree 1,
        # dependencies.jl:48-55 — which is never computed, and then nothing upstream of it ever becomes pending, in the reference as here)
        for _ in range(int(rng.integers(1, max_obs + 1))):
            g = len(m.obs) if len(m.obs) < n_op else int(rng.integers(0, n_op))
            m.obs.append((be.add_variable("y"), j, g, float(xt[j] + rng.standard_normal() / np.sqrt(op_true[g]))))
    m.prior_x = be.add_factor("prior"); be.add_edge(m.x[0], m.prior_x)
    m.prior_tp = [be.add_factor("prior") for _ in m.tp]; m.prior_op = [be.add_factor("prior") for _ in m.op]
    for v, f in zip(m.tp + m.op, m.prior_tp + m.prior_op):
        be.add_edge(v, f)
    m.trans = [None]
    for j in range(1, K):
        f = be.add_factor("transition")
        be.add_edge(m.x[j], f); be.add_edge(m.x[m.parent[j]], f); be.add_edge(m.tp[m.tgroup[j]], f)
        m.trans.append(f)
    m.lik = []
    for y, j, g, _d in m.obs:
        f = be.add_factor("likelihood")
        be.add_edge(y, f); be.add_edge(m.x[j], f); be.add_edge(m.op[g], f)
        m.lik.append(f)
    be.resolve(tree_factor_resolver, None)
    m.x0_prior = vs.normal_mp(0.3, 0.5)
    m.gamma_prior = vs.gamma(1.5, 2.0)                # shape 1.5, rate 0.5
    for v in m.tp + m.op:
        be.set_marginal(v, vs.gamma(1.0, 1.0))
    for v in m.x:
        be.set_marginal(v, vs.normal_mp(0.0, 1.0))
    for y, _j, _g, d in m.obs:
        be.set_marginal(y, vs.real(d))
    # the priors AFTER the initial marginals: set_value! on a marginal clears the fresh bits of its own dependencies (signal.jl:232-253:
    # unset_all_dependencies_fresh!), so a prior message set before it would not count as new and a marginal of degree <= 5 would never
    # become pending — in the reference as here (the call-by-call tests pass either way; the model would just never move)
    set_priors(be, m)
    return m


def set_priors(be, m):
    """the user's set_value! on the prior messages.  Before EVERY call, as an iteration on the reference does it (cf. the priors of the
    loopy-grid tests): the marginals and MessageToFactor signals that depend on a prior message do so strongly, so they only become pending
    again when the prior counts as new — a prior set once would freeze the root's messages after the first call, in the reference as here"""
    _set_prior(be, m.x[0], m.prior_x, m.x0_prior)
    for v, f in zip(m.tp + m.op, m.prior_tp + m.prior_op):
        _set_prior(be, v, f, m.gamma_prior)


def tree_calls(m, iteration):
    """the requests of one iteration: by class, one precision alone, and — every other iteration — states and precisions in ONE request"""
    calls = [list(m.x), m.tp + m.op, [m.tp[0]], list(reversed(m.x)), m.op + m.tp]
    if iteration % 2 == 0:
        calls.append(m.tp + list(m.x) + m.op)
    return calls


class DenseTreeVMP:
    """coordinate ascent on the same model with dense linear algebra: q(x) a K-variate Gaussian, q(precision) Gamma distributions"""

    def __init__(self, m):
        self.m, self.K = m, len(m.x)
        self.tp = [(1.0, 1.0)] * len(m.tp); self.op = [(1.0, 1.0)] * len(m.op)      # (shape, scale)
        self.mu, self.Sigma = np.zeros(self.K), np.eye(self.K)

    def update_x(self):
        m, K = self.m, self.K
        Lam, h = np.zeros((K, K)), np.zeros(K)
        mean0, w0 = m.x0_prior[1]
        Lam[0, 0] += w0; h[0] += w0 * mean0
        for j in range(1, K):
            t = self.tp[m.tgroup[j]][0] * self.tp[m.tgroup[j]][1]
            p = m.parent[j]
            Lam[j, j] += t; Lam[p, p] += t; Lam[j, p] -= t; Lam[p, j] -= t
        for _y, j, g, d in m.obs:
            t = self.op[g][0] * self.op[g][1]
            Lam[j, j] += t; h[j] += t * d
        self.Sigma = np.linalg.inv(Lam)
        self.mu = self.Sigma @ h

    def update_precisions(self):
        m = self.m
        a0, rate0 = m.gamma_prior[1][0], 1.0 / m.gamma_prior[1][1]
        cnt, rate = np.zeros(len(m.tp)), np.zeros(len(m.tp))
        for j in range(1, self.K):
            p, g = m.parent[j], m.tgroup[j]
            cnt[g] += 1
            rate[g] += 0.5 * (self.Sigma[j, j] + self.Sigma[p, p] - 2 * self.Sigma[j, p] + (self.mu[j] - self.mu[p]) ** 2)
        self.tp = [(a0 + 0.5 * c, 1.0 / (rate0 + r)) for c, r in zip(cnt, rate)]
        cnt, rate = np.zeros(len(m.op)), np.zeros(len(m.op))
        for _y, j, g, d in m.obs:
            cnt[g] += 1
            rate[g] += 0.5 * (self.Sigma[j, j] + (d - self.mu[j]) ** 2)
        self.op = [(a0 + 0.5 * c, 1.0 / (rate0 + r)) for c, r in zip(cnt, rate)]


# ---- the plug-in: the host mirror of the reference's API with HipProcessor(mode = "reference") -------------------------------------------
class PluginBackend(vs.MirrorBackend):
    """What a user of the reference does, on the product's host mirror: build the model engine, pass THEIR resolver and the HIP processor
    to InferenceEngine, set values, call update_marginals.  The resolver runs on the host engine's signals; HipProcessor.attach reads the
    wiring back (cortex.jl_amd.wiring.from_engine) and hands it to the device (cx_graph_wire); every update_marginals! is then one
    cx_sweep_for.  Values travel as the processor's value types."""

    ROLES = {"likelihood": (("y", "out"), ("x", "mean"), ("obsnoise", "precision")), "transition": (("x", "both"), ("ssnoise", "precision"))}

    def __init__(self):
        from cortex.jl_amd import hip_processor as hp
        self.hp = hp
        super().__init__(rule=None)
        self.forms = {}

    def add_factor(self, form):
        ff = self.hp.NormalPrecisionFactor(roles=self.ROLES[form]) if form in self.ROLES else form
        f = self.graph.add_factor(self.cx.Factor(functional_form=ff))
        self.forms[f] = form
        return f

    def factor_form(self, f): return self.forms[f]

    def resolve(self, factor_resolver, variable_resolver):
        cx, be = self.cx, self

        class Resolver(cx.AbstractDependencyResolver):
            def resolve_factor_dependencies(self, engine, factor_id):
                be.engine = engine
                factor_resolver(be, factor_id)

            def resolve_variable_dependencies(self, engine, variable_id):
                be.engine = engine
                if variable_resolver is None:
                    cx.DefaultDependencyResolver().resolve_variable_dependencies(engine, variable_id)
                else:
                    variable_resolver(be, variable_id)

        self.processor = self.hp.HipProcessor(mode="reference")
        self.engine = cx.InferenceEngine(model_engine=self.graph, dependency_resolver=Resolver(), inference_request_processor=self.processor)

    def _value(self, value):
        tag, p = value
        return float(p[0]) if tag == ref.REAL else self.hp.NormalMeanPrecision(p[0], p[1]) if tag == ref.NORMAL_MP else self.hp.Gamma(p[0], p[1])

    def set_marginal(self, v, value): self.processor.set_value(self.marginal(v), self._value(value))
    def set_message_to_variable(self, v, f, natural): raise NotImplementedError
    def set_prior(self, v, f, value): self.processor.set_value(self.message_to_variable(v, f), self._value(value))

    def get_marginal(self, v):
        m = self.processor.read(self.marginal(v).variant)
        if isinstance(m, self.hp.Gamma):
            return (ref.GAMMA, [m.shape, m.scale])
        return (ref.NORMAL_MP, [m.mean, 1.0 / m.variance])

    def trace_rows(self):
        return [(k, 0 if k == KJOINT else v, f if k in (K2F, K2V, KJOINT) else 0, lo, hi) for k, v, f, lo, hi in self.processor.dev.ref_trace()]
