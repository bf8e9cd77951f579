"""CPU: CX_SCHED_REFERENCE's host logic (cortex.jl_amd/csrc/cx_refsched.h, compiled by g++ into libcortex_hostlogic.so) against the
restated reference engine (oracle/cortex_ref.c).

(1) ORDER: the shadow scheduler's executions equal the restated engine's trace, signal by signal, over consecutive calls on loopy
    graphs, trees and chains, for full, partial and reversed requests, with data and priors set in between — bit-exact, as the tier
    asks of schedule / indexing work.
(2) LEVELLING: the recorded call, cut into stages, is executed here in numpy with every stage's items reading the values the stage
    STARTED with (items of a stage run concurrently on the device): no item may read or overwrite what another item of its stage
    writes, and the values after the last stage equal the restated engine's (moment-form arithmetic, the reference's operation order)
    — every message and every marginal.
The same file runs under -fsanitize=address,undefined (tests/test_hostlogic.py collects it with CXH_LIB set)."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import ref
from tests.helpers import assert_close, engine_oracle_from_model, random_loopy_model
from tests.hostlogic import FlatGraph

SEED_VARIANCE = 1e6
K2F, K2V, KMARG, KPROD, KKARY = L.ITEM_MESSAGE_TO_FACTOR, L.ITEM_MESSAGE_TO_VARIABLE, L.ITEM_INDIVIDUAL_MARGINAL, L.ITEM_PRODUCT_OF_MESSAGES, 32
KSUM2F, KSUM2P, KSUM2M = 64, 65, 66


def _flat(model):
    g = FlatGraph(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, schedule=L.SCHED_REFERENCE)
    assert g.status == 0, g.error
    rc, err = g.ref_build()
    assert rc == 0, err
    return g


def _oracle_rows(E):
    rows = []
    for _r, _v, s, _b, _a in E.trace():
        k, v, f, lo, hi = E.variant(s)
        kind = {ref.VAR_MSG_TO_FACTOR: K2F, ref.VAR_MSG_TO_VARIABLE: K2V, ref.VAR_MARGINAL: KMARG, ref.VAR_PRODUCT: KPROD}[k]
        rows.append((kind, v, f if kind in (K2F, K2V) else 0, lo if kind == KPROD else 0, hi if kind == KPROD else 0))
    return rows


class NumpyDevice:
    """the device's message store in natural form + the items of cx_batch.hip: batch_item, executed stage by stage"""

    def __init__(self, g, model):
        self.g = g
        self.partner, self.vbase, self.var_off, self.vinfo = g.arr("partner"), g.arr("vbase"), g.arr("var_off"), g.arr("vinfo")
        self.q = g.arr("q")
        ns = g.scalar("nslots")
        self.f2v = np.full((ns, 2), np.nan); self.v2f = np.full((ns, 2), np.nan)
        self.marg = np.full((g.scalar("nv"), 2), np.nan)
        self.prod = np.full((max(1, g.ref_scalar("products")), 2), np.nan)
        self.var_ids, self.edge_var, self.edge_fac = g.arr("var_ids"), g.arr("edge_var"), g.arr("edge_fac_id")
        self.slot = {}
        for e in range(len(self.edge_var)):
            v = self.edge_var[e]
            self.slot[(int(self.var_ids[v]), int(self.edge_fac[e]))] = int(self.vbase[v] + 256 * (e - self.var_off[v])) if (self.vinfo[v] & 0x0f) != 0x0f else int(self.vbase[v] + e - self.var_off[v])

    def set(self, buf, v, f, mean, variance):
        getattr(self, buf)[self.slot[(int(v), int(f))]] = (mean / variance, 1.0 / variance)

    def scan_links(self, s, reads, writes, new):
        """the scan steps of stage s (cx_refsched.h: ScanStep; cx_planscan.hip runs them as prefix scans): link by link, in chain order — the
        leader's sum over its settled sources and the message the pair before it stored, then the follower's rule"""
        g = self.g
        scans = g.arr("ref_scans").reshape(-1, 3)
        if not len(scans):
            return 0
        ld, lv, fd, pr, so, sr, hd = (g.arr("ref_sl_" + k) for k in ("lead_dst", "lead_var", "fol_dst", "prec", "src_off", "src", "head"))
        n = 0
        for st, lo, hi in scans:
            if st != s + 1:
                continue
            m = None
            for l in range(lo, hi):
                assert pr[l] < 0, "this runner holds the sum-product rules only"
                if hd[l]:
                    m = np.zeros(2)
                else:
                    assert m is not None
                src = [("f2v", int(t)) if t >= 0 else ("prod", int(~t)) for t in sr[so[l]:so[l + 1]]]
                vals = np.array([getattr(self, b)[i] for b, i in src]).reshape(-1, 2)
                assert not np.any(np.isnan(vals)), f"stage {s}: a link's settled source is undefined"
                v = m + vals.sum(axis=0)
                qq = self.q[fd[l]]
                m = np.array([(v[0] / v[1]) / (1.0 / v[1] + qq), 1.0 / (1.0 / v[1] + qq)])
                reads.update(src)
                for dst, out in ((("v2f", int(ld[l])), tuple(v)), (("f2v", int(fd[l])), tuple(m))):
                    assert dst not in writes, f"stage {s}: a link writes {dst} twice"
                    writes.add(dst); new.append((dst, out))
                n += 2
        return n

    def run(self, rec, stage_off, lists):
        LEADS, FOLLOWS, MASK = 0x40000000, 0x20000000, 0x0fffffff      # cx_refsched.h: a record that leads is followed by one the same thread computes behind it
        self.chain_executions = 0
        for s in range(len(stage_off) - 1):
            items = rec[5 * stage_off[s]:5 * stage_off[s + 1]].reshape(-1, 5)
            reads, writes, new = set(), set(), []
            self.chain_executions += self.scan_links(s, reads, writes, new)
            leader_out = None                 # (destination, value) of the leader the next record follows
            for k, idx, v, lo, hi in items:
                flags, k = int(k) & ~MASK, int(k) & MASK
                assert bool(flags & FOLLOWS) == (leader_out is not None), f"stage {s}: a follower without its leader (or a leader without its follower)"
                if k == K2F:
                    deg = self.var_off[v + 1] - self.var_off[v]
                    src = [("f2v", int(self.vbase[v] + 256 * j)) for j in range(deg)]
                    src = [t for t in src if t[1] != idx]
                    dst = ("v2f", int(idx))
                elif k == K2V:
                    src, dst = [("v2f", int(self.partner[idx]))], ("f2v", int(idx))
                elif k == KMARG:
                    deg = self.var_off[v + 1] - self.var_off[v]
                    src, dst = [("f2v", int(self.vbase[v] + 256 * j)) for j in range(deg)], ("marg", int(v))
                elif k in (KSUM2F, KSUM2P, KSUM2M):
                    src = [("f2v", int(t)) if t >= 0 else ("prod", int(~t)) for t in lists[lo:lo + hi]]
                    dst = ({KSUM2F: "v2f", KSUM2P: "prod", KSUM2M: "marg"}[int(k)], int(idx))
                else:
                    raise AssertionError(f"unexpected item kind {k}")

                def value(b, i):
                    if leader_out is not None and leader_out[0] == (b, i):
                        return leader_out[1]          # the follower reads what its leader just stored
                    return getattr(self, b)[i]
                vals = np.array([value(b, i) for b, i in src])
                assert not np.any(np.isnan(vals)), f"stage {s}: an item of kind {k} reads an undefined value"
                if k == K2V:
                    xi, w = vals[0]
                    qq = self.q[idx]
                    out = ((xi / w) / (1.0 / w + qq), 1.0 / (1.0 / w + qq)) if np.isfinite(w) else (xi / qq, 1.0 / qq)
                else:
                    out = tuple(vals.sum(axis=0))
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

    def moment(self, buf, idx):
        a = getattr(self, buf)[idx]
        point = np.isinf(a[:, 1])              # a datum is stored as (y, +inf)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(point, a[:, 0], a[:, 0] / a[:, 1]), np.where(point, 0.0, 1.0 / a[:, 1])


def _check_values(dev, E, model, request, what):
    slots = np.array([dev.slot[(int(v), int(f))] for v, f in zip(model.edge_var, model.edge_fac)])
    for to_variable, buf in ((True, "f2v"), (False, "v2f")):
        tags, a, b = E.get_messages(model.edge_var, model.edge_fac, to_variable)
        m, s = dev.moment(buf, slots)
        und = tags == ref.UNDEF
        # (data are point masses: variance 0 on the oracle's side, infinite precision here)
        assert np.array_equal(np.isnan(s), und), f"{what}: the same {buf} messages are defined"
        assert_close(m[~und], a[~und], 1e-10, f"{what} {buf} mean")
        ok = ~und & (tags != ref.REAL)
        assert_close(s[ok], b[ok], 1e-10, f"{what} {buf} variance")
    tags, em, ev = E.get_marginals(request)
    vi = np.searchsorted(dev.var_ids, request)
    m, s = dev.moment("marg", vi)
    und = tags == ref.UNDEF
    assert np.array_equal(np.isnan(s), und), f"{what}: the same marginals are defined"
    assert_close(m[~und], em[~und], 1e-10, f"{what} marginal mean")
    assert_close(s[~und], ev[~und], 1e-10, f"{what} marginal variance")


def _models():
    rnd, _ = random_loopy_model(11, 1, nv=60, extra=25)
    hubs, _ = random_loopy_model(5, 1, nv=40, extra=70)           # degrees up to ~12: segment trees over the CSR tail too
    return {"grid8x9": cx.synth.gaussian_grid(8, 9, seed=5), "grid17x23": cx.synth.gaussian_grid(17, 23, seed=2), "random": rnd, "hubs": hubs}


def _seed(g, E, dev, model):
    pw = set(int(f) for f, k in zip(model.factor_ids, model.factor_kind) if k == 1)
    keep = np.array([int(f) in pw for f in model.edge_fac])
    pv, pf = model.edge_var[keep], model.edge_fac[keep]
    E.set_messages_to_variable(pv, pf, np.zeros(len(pv)), np.full(len(pv), SEED_VARIANCE))
    g.ref_set(L.TO_VARIABLE, pv, pf)
    for v, f in zip(pv, pf):
        dev.set("f2v", v, f, 0.0, SEED_VARIANCE)


def _set_priors(g, E, dev, model):
    E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
    g.ref_set(L.TO_VARIABLE, model.prior_var, model.prior_fac)
    for v, f, m, s in zip(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance):
        dev.set("f2v", v, f, m, s)


@pytest.mark.parametrize("name", ["grid8x9", "grid17x23", "random", "hubs"])
def test_order_and_levelled_values_on_loopy_graphs(name):
    model = _models()[name]
    g = _flat(model)
    E = engine_oracle_from_model(model, trace=True)
    dev = NumpyDevice(g, model)
    _seed(g, E, dev, model)
    _set_priors(g, E, dev, model)          # engine_oracle_from_model set them on its side already: once more is the same state
    hashes = []
    rng = np.random.default_rng(3)
    for call in range(6):
        if call:
            _set_priors(g, E, dev, model)
        request = model.x_ids if call < 4 else rng.permutation(model.x_ids)[: len(model.x_ids) // 2]
        hashes.append(g.ref_scalar("hash"))
        rows = g.ref_update(request)
        E.update_marginals(request)
        assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E), f"{name} call {call + 1}: execution order"
        assert rows[-1, 5] < g.ref_scalar("rounds")
        rc, err = g.ref_level()
        assert rc == 0, err
        rec, off, lists = g.arr("ref_rec"), g.arr("ref_stage_off"), g.arr("ref_list")
        dev.run(rec, off, lists)
        assert off[-1] + dev.chain_executions == len(rows) and len(rec) == 5 * off[-1]
        _check_values(dev, E, model, request, f"{name} call {call + 1}")
    if name.startswith("grid"):
        assert hashes[2] == hashes[3], "on a grid the readiness state before a call repeats from the second call on: the plan is reused"
    assert hashes[0] != hashes[1]


@pytest.mark.parametrize("T", [1, 2, 3, 40])
def test_the_reference_state_space_model(T):
    """test/inference_engine_tests.jl:379-488: one call = the forward and the backward pass; 5T - 4 messages + T marginals"""
    model = cx.synth.ssm_chain(T, seed=4, random_variances=True)
    g = _flat(model)
    E = engine_oracle_from_model(model, trace=True)
    dev = NumpyDevice(g, model)
    g.ref_set(L.TO_FACTOR, model.data_var, model.data_fac)
    for v, f, y in zip(model.data_var, model.data_fac, model.data_y):
        dev.v2f[dev.slot[(int(v), int(f))]] = (y, np.inf)
    for call in range(2):
        rows = g.ref_update(model.x_ids)
        E.update_marginals(model.x_ids)
        assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E)
        if call == 0:
            assert len(rows) == 5 * T - 4 + T
            rc, err = g.ref_level()
            assert rc == 0, err
            off = g.arr("ref_stage_off")
            dev.run(g.arr("ref_rec"), off, g.arr("ref_list"))
            _check_values(dev, E, model, model.x_ids, f"chain T={T}")
            if T > 2:
                assert len(off) - 1 <= 4 * T, "stages grow with the depth of the dependency chains, not with the number of messages"
        else:
            assert len(rows) == 0, "a second call without new data computes nothing (lazy)"


@pytest.mark.parametrize("T,chain_min", [(40, 8), (300, 128), (300, 16), (1000, 128)])
def test_the_chains_of_a_state_space_model_run_as_scan_steps(monkeypatch, T, chain_min):
    """(round 6) the forward and the backward pass of the reference's call on a state-space model are two chains of pairs: level() hands
    them out as scan steps (every execution of a chain at the stage of its first pair), everything else is levelled around them — a
    handful of stages instead of T —, and the values are the engine's"""
    monkeypatch.setenv("CX_REF_CHAIN_MIN", str(chain_min))
    model = cx.synth.ssm_chain(T, seed=T, random_variances=True)
    g = _flat(model)
    E = engine_oracle_from_model(model, trace=True)
    dev = NumpyDevice(g, model)
    g.ref_set(L.TO_FACTOR, model.data_var, model.data_fac)
    for v, f, y in zip(model.data_var, model.data_fac, model.data_y):
        dev.v2f[dev.slot[(int(v), int(f))]] = (y, np.inf)
    rows = g.ref_update(model.x_ids)
    E.update_marginals(model.x_ids)
    assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E)
    rc, err = g.ref_level()
    assert rc == 0, err
    off, scans = g.arr("ref_stage_off"), g.arr("ref_scans").reshape(-1, 3)
    dev.run(g.arr("ref_rec"), off, g.arr("ref_list"))
    _check_values(dev, E, model, model.x_ids, f"chain T={T} as scans")
    assert len(scans) >= 1 and len(off) - 1 <= 8, (scans, len(off) - 1)
    assert dev.chain_executions >= 4 * (T - 1) - 8 and off[-1] + dev.chain_executions == len(rows)
    assert int(g.arr("ref_sl_head").sum()) == 2, "two chains: the forward and the backward pass"
    monkeypatch.setenv("CX_REF_CHAIN_MIN", "0")
    rc, err = g.ref_level()
    assert rc == 0 and len(g.arr("ref_scans")) == 0 and len(g.arr("ref_stage_off")) - 1 >= T - 1


@pytest.mark.parametrize("name", ["grid8x9", "grid17x23", "random", "hubs"])
def test_chains_found_inside_loopy_plans_keep_the_engines_values(monkeypatch, name):
    """the same with a threshold of three pairs on graphs with loops, where chains of pairs run along rows between executions that read
    and overwrite their neighbours: whatever level() accepts as a scan step must leave the engine's values, call after call"""
    monkeypatch.setenv("CX_REF_CHAIN_MIN", "3")
    model = _models()[name]
    g = _flat(model)
    E = engine_oracle_from_model(model, trace=True)
    dev = NumpyDevice(g, model)
    _seed(g, E, dev, model)
    _set_priors(g, E, dev, model)
    found = 0
    rng = np.random.default_rng(4)
    for call in range(5):
        if call:
            _set_priors(g, E, dev, model)
        request = model.x_ids if call < 3 else rng.permutation(model.x_ids)[: len(model.x_ids) // 2]
        rows = g.ref_update(request)
        E.update_marginals(request)
        assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E)
        rc, err = g.ref_level()
        assert rc == 0, err
        off = g.arr("ref_stage_off")
        dev.run(g.arr("ref_rec"), off, g.arr("ref_list"))
        assert off[-1] + dev.chain_executions == len(rows)
        found += dev.chain_executions
        _check_values(dev, E, model, request, f"{name} call {call + 1} with chains as scans")
    print(f"{name}: {found} executions ran inside scan steps")      # (on these graphs level() mostly refuses: a chain's neighbours are rewritten within the call)


def test_trees_partial_requests_compute_only_what_they_need():
    model, _ = random_loopy_model(9, 1, nv=60, extra=-1)          # the spanning tree alone
    g = _flat(model)
    E = engine_oracle_from_model(model, trace=True)
    dev = NumpyDevice(g, model)
    _set_priors(g, E, dev, model)
    first = model.x_ids[:3]
    rows = g.ref_update(first)
    E.update_marginals(first)
    assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E)
    n_first = len(rows)
    rc, err = g.ref_level()
    assert rc == 0, err
    dev.run(g.arr("ref_rec"), g.arr("ref_stage_off"), g.arr("ref_list"))
    _check_values(dev, E, model, first, "tree, three marginals")
    rows = g.ref_update(model.x_ids[::-1])
    E.update_marginals(model.x_ids[::-1])
    assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E)
    assert 0 < n_first < n_first + len(rows)
    # (a request for three variables of a tree computes what is pending for THEM: the messages next to the leaves; their marginals stay
    # undefined until a request walks the rest of the tree — the reference's laziness, reproduced, not repaired)
    rc, err = g.ref_level()
    assert rc == 0, err
    dev.run(g.arr("ref_rec"), g.arr("ref_stage_off"), g.arr("ref_list"))
    _check_values(dev, E, model, model.x_ids[::-1], "tree, the rest in reverse order")


def test_wiring_equals_the_restated_resolver():
    """dependency lists, in order, with the intermediate flags: dependencies.jl:17-173 on a graph with hubs"""
    model = _models()["hubs"]
    g = _flat(model)
    E = engine_oracle_from_model(model)
    dep_off, dep, inter = g.arr("ref_dep_off"), g.arr("ref_dep"), g.arr("ref_dep_inter")
    ne, nv = g.scalar("ne"), g.scalar("nv")
    var_ids, edge_var, edge_fac = g.arr("var_ids"), g.arr("edge_var"), g.arr("edge_fac_id")
    rows = g.ref_update([])          # nothing requested, nothing computed
    assert len(rows) == 0

    def key_of(s):
        if s < ne:
            return (ref.VAR_MSG_TO_FACTOR, int(var_ids[edge_var[s]]), int(edge_fac[s]))
        if s < 2 * ne:
            return (ref.VAR_MSG_TO_VARIABLE, int(var_ids[edge_var[s - ne]]), int(edge_fac[s - ne]))
        if s < 2 * ne + nv:
            return (ref.VAR_MARGINAL, int(var_ids[s - 2 * ne]), 0)
        return None

    def okey(sig):
        k, v, f, lo, hi = E.variant(sig)
        return (k, v, f if k in (ref.VAR_MSG_TO_FACTOR, ref.VAR_MSG_TO_VARIABLE) else 0) if k != ref.VAR_PRODUCT else None

    checked = 0
    for e in range(ne):
        v, f = int(var_ids[edge_var[e]]), int(edge_fac[e])
        for s, osig in ((e, E.message_to_factor(v, f)), (ne + e, E.message_to_variable(v, f))):
            mine = [key_of(int(d)) for d in dep[dep_off[s]:dep_off[s + 1]]]
            theirs = [okey(d) for d in E.dependencies(osig)]
            assert mine == theirs, (v, f)
            checked += len(mine)
    assert checked > 0 and g.ref_scalar("products") > 0
    assert g.ref_scalar("dependencies") == len(dep) and set(np.unique(inter)) <= {0, 1}


@pytest.mark.parametrize("n", [1, 2, 5, 6, 7, 8, 9, 33, 100])
def test_the_reference_beta_bernoulli_model(n):
    """test/inference_engine_tests.jl:241-377: a star of n Bernoulli factors around p; for n > 5 the marginal hangs off a segment tree of
    ProductOfMessages nodes (dependencies.jl:90-173).  The shadow scheduler's executions for update_marginals!(engine, p) — messages,
    tree nodes in the resolver's creation order, marginal — equal the restated engine's trace"""
    rng = np.random.default_rng(n)
    data = rng.random(n) < 0.5
    E = ref.Engine(ref.P_BETA_BERNOULLI, trace=True)
    p = E.add_variable()
    o, f = [], []
    for _ in range(n):
        oi, fi = E.add_variable(), E.add_factor(ref.F_BERNOULLI)
        o.append(oi); f.append(fi)
        E.add_edge(p, fi); E.add_edge(oi, fi)
    E.finalize()
    for i in range(n):
        E.set_value(E.message_to_factor(o[i], f[i]), bool(data[i]))
    g = FlatGraph(np.concatenate([np.full(n, p), o]), np.concatenate([f, f]), f, np.full(n, L.FACTOR_BERNOULLI, np.int32), np.ones(n),
                  schedule=L.SCHED_REFERENCE, family=L.FAMILY_NATURAL2)
    assert g.status == 0, g.error
    rc, err = g.ref_build()
    assert rc == 0, err
    assert g.ref_scalar("products") == (n - 2 if n > 5 else 0)
    g.ref_set(L.TO_FACTOR, o, f)
    rows = g.ref_update([p])
    E.update_marginals([p])
    assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E)
    assert len(rows) == n + (n - 2 if n > 5 else 0) + 1
    rc, err = g.ref_level()
    assert rc == 0, err
    off = g.arr("ref_stage_off")
    assert len(off) - 1 <= 2 + max(1, int(np.ceil(np.log2(max(n, 2))))) + 1, "messages, then the tree level by level, then the marginal"
    assert len(g.ref_update([p])) == 0


# ---- user wirings (cx_graph_wire): add_dependency!(signal, dependency; weak, listen, intermediate) triple by triple -----------------------
def _key(E, sig):
    k, v, f, _lo, _hi = E.variant(sig)
    return ({ref.VAR_MSG_TO_FACTOR: K2F, ref.VAR_MSG_TO_VARIABLE: K2V, ref.VAR_MARGINAL: KMARG}[k], int(v), int(f) if k != ref.VAR_MARGINAL else 0)


def _oracle_signal(E, key):
    k, v, f = key
    return E.marginal(v) if k == KMARG else (E.message_to_factor(v, f) if k == K2F else E.message_to_variable(v, f))


def _wire_both(model, triples):
    """the same add_dependency! calls into the restated engine (built without the default resolver) and into the shadow"""
    n_nodes = int(max(model.edge_var.max(), model.factor_ids.max()))
    kind = np.zeros(n_nodes, dtype=np.int32); fkind = np.zeros(n_nodes, dtype=np.int32); p0 = np.ones(n_nodes)
    kind[np.unique(model.edge_var) - 1] = 1; kind[model.factor_ids - 1] = 2
    fkind[model.factor_ids - 1] = np.where(model.factor_kind == 1, ref.F_GAUSS_ADD, ref.F_OPAQUE)
    p0[model.factor_ids - 1] = np.asarray(model.factor_var).reshape(len(model.factor_ids), -1)[:, 0]
    E = ref.Engine(ref.P_SSM_BP, True)
    E.bulk_build(kind, fkind, p0, model.edge_var, model.edge_fac)
    E.finalize(resolve_dependencies=False)
    for s, d, fl in triples:
        E.add_dependency(_oracle_signal(E, s), _oracle_signal(E, d), weak=bool(fl & 1), intermediate=bool(fl & 2), listen=not (fl & 4))
    g = FlatGraph(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, schedule=L.SCHED_REFERENCE)
    assert g.status == 0, g.error
    rc, err = g.ref_build()
    assert rc == 0, err
    rc, err = g.ref_wire([s for s, _d, _f in triples], [d for _s, d, _f in triples], [f for _s, _d, f in triples])
    assert rc == 0, err
    return E, g


def test_a_filter_wiring_on_the_state_space_model():
    """a user resolver that wires the FORWARD messages only: update_marginals! then computes the Kalman FILTER (every marginal conditions
    on the data up to its own time), not the smoother the default wiring gives; same executions as the restated engine under the same
    add_dependency! calls, values by the levelled plan == the restated engine's == a textbook Kalman filter"""
    T = 30
    model = cx.synth.ssm_chain(T, seed=8, random_variances=True)
    x, y, lik, tr = model.x_ids, model.data_var, model.factor_ids[:T], model.factor_ids[T:]
    INTER = L.WIRE_INTERMEDIATE
    triples = []
    for t in range(T):
        triples.append(((K2V, int(x[t]), int(lik[t])), (K2F, int(y[t]), int(lik[t])), 0))                  # lik_t→x_t  <-  y_t→lik_t
        triples.append(((KMARG, int(x[t]), 0), (K2V, int(x[t]), int(lik[t])), INTER))
        if t > 0:
            triples.append(((KMARG, int(x[t]), 0), (K2V, int(x[t]), int(tr[t - 1])), INTER))
            triples.append(((K2V, int(x[t]), int(tr[t - 1])), (K2F, int(x[t - 1]), int(tr[t - 1])), 0))    # tr_{t-1}→x_t  <-  x_{t-1}→tr_{t-1}
        if t + 1 < T:
            triples.append(((K2F, int(x[t]), int(tr[t])), (K2V, int(x[t]), int(lik[t])), INTER))           # x_t→tr_t  <-  lik_t→x_t, tr_{t-1}→x_t
            if t > 0:
                triples.append(((K2F, int(x[t]), int(tr[t])), (K2V, int(x[t]), int(tr[t - 1])), INTER))
    E, g = _wire_both(model, triples)
    dev = NumpyDevice(g, model)
    E.set_messages_to_factor(model.data_var, model.data_fac, model.data_y, tag=ref.REAL)
    g.ref_set(L.TO_FACTOR, model.data_var, model.data_fac)
    for v, f, yy in zip(model.data_var, model.data_fac, model.data_y):
        dev.v2f[dev.slot[(int(v), int(f))]] = (yy, np.inf)
    rows = g.ref_update(x)
    E.update_marginals(x)
    assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E)
    assert len(rows) == T + 2 * (T - 1) + T          # likelihood messages, forward chain, marginals: no backward message is ever computed
    rc, err = g.ref_level()
    assert rc == 0, err
    dev.run(g.arr("ref_rec"), g.arr("ref_stage_off"), g.arr("ref_list"))
    _check_values(dev, E, model, x, "filter wiring")
    # textbook Kalman filter: vague start, y_t = x_t + N(0, r_t), x_{t+1} = x_t + N(0, q_t)
    r, q = model.meta["r"], model.meta["q"]
    m, v = model.data_y[0], r[0]
    fm, fv = [m], [v]
    for t in range(1, T):
        pv = v + q[t - 1]
        k = pv / (pv + r[t])
        m, v = m + k * (model.data_y[t] - m), (1 - k) * pv
        fm.append(m); fv.append(v)
    gm, gv = dev.moment("marg", np.searchsorted(dev.var_ids, x))
    assert_close(gm, np.array(fm), 1e-10, "filtered means"); assert_close(gv, np.array(fv), 1e-10, "filtered variances")


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_flags_and_dropped_dependencies_on_a_loopy_grid(seed):
    """the default wiring of a grid with random weak / intermediate / listen flags and a tenth of the product dependencies dropped, in a
    shuffled add_dependency! order: the shadow scheduler and the restated engine execute the same signals in the same order over six calls
    with data re-set in between, and the levelled plan leaves the engine's values"""
    rng = np.random.default_rng(seed)
    model = cx.synth.gaussian_grid(6, 7, seed=4)
    E0 = engine_oracle_from_model(model)
    triples = []
    for v, f in zip(model.edge_var, model.edge_fac):
        for sig in (E0.message_to_factor(int(v), int(f)), E0.message_to_variable(int(v), int(f))):
            for d in E0.dependencies(sig):
                triples.append((_key(E0, sig), _key(E0, d)))
    for v in model.x_ids:
        for d in E0.dependencies(E0.marginal(int(v))):
            triples.append((_key(E0, E0.marginal(int(v))), _key(E0, d)))
    out = []
    for s, d in triples:
        if s[0] != K2V and rng.random() < 0.1:
            continue                                              # a product that leaves one of its natural inputs out
        # (intermediate flags on the dependencies of products only, as the default resolver places them: flagged on the factor side too they
        # close cycles on a loopy graph, which process_dependencies! never leaves — refused at wiring time, see the last test)
        fl = (L.WIRE_WEAK if rng.random() < 0.3 else 0) | (L.WIRE_INTERMEDIATE if s[0] != K2V and rng.random() < 0.6 else 0) | (L.WIRE_NO_LISTEN if rng.random() < 0.1 else 0)
        out.append((s, d, fl))
    order = rng.permutation(len(out))
    triples = [out[i] for i in order]
    E, g = _wire_both(model, triples)
    dev = NumpyDevice(g, model)
    _seed(g, E, dev, model)
    for call in range(6):
        _set_priors(g, E, dev, model)
        request = model.x_ids if call % 2 == 0 else rng.permutation(model.x_ids)[:20]
        rows = g.ref_update(request)
        E.update_marginals(request)
        assert [tuple(r[:5]) for r in rows.tolist()] == _oracle_rows(E), f"seed {seed} call {call + 1}: execution order"
        rc, err = g.ref_level()
        assert rc == 0, err
        dev.run(g.arr("ref_rec"), g.arr("ref_stage_off"), g.arr("ref_list"))
        _check_values(dev, E, model, request, f"seed {seed} call {call + 1}")


def test_wirings_the_device_has_no_rule_for_are_refused():
    model = cx.synth.gaussian_grid(3, 3, seed=1)
    g = _flat(model)
    v, f = int(model.edge_var[-1]), int(model.edge_fac[-1])
    other = int(model.x_ids[0]) if int(model.x_ids[0]) != v else int(model.x_ids[1])
    for s, d, what in (((K2V, v, f), (KMARG, other, 0), "a message that depends on a marginal (a variational wiring)"),
                       ((K2F, v, f), (K2V, v, f), "a MessageToFactor that depends on the message of its own edge"),
                       ((KMARG, v, 0), (K2F, v, f), "a marginal that depends on a MessageToFactor")):
        rc, err = g.ref_wire([s], [d], [0])
        assert rc == L.ERR_UNSUPPORTED and "cx_graph_wire" in err, what
    rc, err = g.ref_wire([(KMARG, v, 0), (KMARG, v, 0)], [(K2V, v, f), (K2V, v, f)], [0, 0])
    assert rc == L.ERR_INVALID_ARGUMENT and "twice" in err
    # intermediate flags all the way round a loop of the grid: process_dependencies! would recurse for ever
    E0 = engine_oracle_from_model(model)
    sig, dep = [], []
    for vv, ff in zip(model.edge_var, model.edge_fac):
        for s in (E0.message_to_factor(int(vv), int(ff)), E0.message_to_variable(int(vv), int(ff))):
            for d in E0.dependencies(s):
                sig.append(_key(E0, s)); dep.append(_key(E0, d))
    rc, err = g.ref_wire(sig, dep, [L.WIRE_INTERMEDIATE] * len(sig))
    assert rc == L.ERR_UNSUPPORTED and "cycle" in err


def test_plans_for_d_dimensional_messages():
    """dim 2 .. 4 (cx_mvbatch.hip: k_batch_mv takes the compact records only): the same executions as for scalar messages on the same graph,
    no pairs, the rule table of the sending slot in every MessageToVariable record; variables of degree above 5 as list sums"""
    T = 9
    x = np.arange(1, T + 1); y = x + T; lik = x + 2 * T; tr = np.arange(3 * T + 1, 4 * T)
    skip = np.arange(4 * T, 4 * T + T - 3)
    ev = np.concatenate([y, x, x[:-1], x[1:], x[:-3], x[3:]]); ef = np.concatenate([lik, lik, tr, tr, skip, skip])
    role = np.concatenate([np.zeros(T), np.ones(T), np.ones(T - 1), np.zeros(T - 1), np.ones(T - 3), np.zeros(T - 3)]).astype(np.int32)
    fids = np.concatenate([lik, tr, skip])
    traces = {}
    for dim in (1, 3):
        kinds = np.full(len(fids), 1 if dim == 1 else 2, np.int32)      # additive (scalar) / linear with a parameter set (dim > 1)
        g = FlatGraph(ev, ef, fids, kinds, np.ones(len(fids)) if dim == 1 else np.zeros(len(fids)), edge_role=role, dim=dim, schedule=L.SCHED_REFERENCE)
        assert g.status == 0, g.error
        rc, err = g.ref_build(); assert rc == 0, err
        g.ref_set(L.TO_FACTOR, y, lik)
        g.ref_set(L.TO_VARIABLE, ev, ef)
        rows = g.ref_update(x)
        traces[dim] = [tuple(int(t) for t in r[:3]) for r in rows]
        rc, err = g.ref_level(); assert rc == 0, err
        rec = g.arr("ref_rec").reshape(-1, 5)
        if dim > 1:
            assert not np.any(rec[:, 0] & 0x60000000), "no leader / follower flags: that kernel takes one record per thread"
            assert set(rec[:, 0].tolist()) <= {K2F, K2V, KMARG}
            spdir, partner = g.arr("spdir"), g.arr("partner")
            m2v = rec[rec[:, 0] == K2V]
            assert np.array_equal(m2v[:, 3], spdir[partner[m2v[:, 1]]])
        else:
            assert np.any(rec[:, 0] & 0x40000000)
    assert traces[1] == traces[3], "the order never depends on what a message is"
    # one more skip link per state: degree 7
    skip2 = np.arange(5 * T, 5 * T + T - 2)
    ev2 = np.concatenate([ev, x[:-2], x[2:]]); ef2 = np.concatenate([ef, skip2, skip2]); role2 = np.concatenate([role, np.ones(T - 2, np.int32), np.zeros(T - 2, np.int32)])
    fids2 = np.concatenate([fids, skip2])
    # ... whose signals hang off segment-tree nodes: list sums (64 MessageToFactor, 65 ProductOfMessages, 66 IndividualMarginal), the same
    # records as for scalar messages but for the pairs
    hub = {}
    for dim in (1, 2):
        g = FlatGraph(ev2, ef2, fids2, np.full(len(fids2), 1 if dim == 1 else 2, np.int32), np.ones(len(fids2)) if dim == 1 else np.zeros(len(fids2)), edge_role=role2, dim=dim,
                      schedule=L.SCHED_REFERENCE)
        assert g.status == 0, g.error
        rc, err = g.ref_build(); assert rc == 0, err
        g.ref_set(L.TO_FACTOR, y, lik); g.ref_set(L.TO_VARIABLE, ev2, ef2)
        rows = g.ref_update(x)
        rc, err = g.ref_level(); assert rc == 0, err
        rec, lst = g.arr("ref_rec").reshape(-1, 5).copy(), g.arr("ref_list").copy()
        hub[dim] = ([tuple(int(t) for t in r[:3]) for r in rows], rec, lst)
    assert hub[1][0] == hub[2][0]
    rec = hub[2][1]
    assert not np.any(rec[:, 0] & 0x60000000)
    assert {64, 65, 66} <= set(rec[:, 0].tolist()) <= {K2F, K2V, KMARG, 64, 65, 66}
    sums = rec[rec[:, 0] >= 64]
    assert np.all(sums[:, 4] >= 1) and np.all(sums[:, 3] + sums[:, 4] <= len(hub[2][2]))
    # the same multiset of list items (kind, destination, the sources in order) as the scalar plan's
    def items(rec, lst):
        return sorted((int(r[0] & 0x0fffffff), int(r[1]), tuple(int(t) for t in lst[r[3]:r[3] + r[4]])) for r in rec if (r[0] & 0x0fffffff) >= 64)
    assert items(rec, hub[2][2]) == items(hub[1][1], hub[1][2])
