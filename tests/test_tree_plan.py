"""CPU: the plan of CX_SCHED_TREE (csrc/cx_tree_plan.h, compiled without HIP by csrc/cx_hostlogic.cpp) EXECUTED in numpy.

Every stage's items are applied to message tables that start undefined except for what the caller set (priors, data), with the
reference's rules for a linear-Gaussian factor of any arity (test/inference_engine_tests.jl:415-432 generalised; one
variable→factor message = the product of the OTHER factor→variable messages, src/dependencies.jl:60-88).  An item that reads an
undefined input fails the test — so a plan that passes has every dependency produced by an earlier stage — each message is
produced once, and the marginals of the one pass must equal a dense solve of the joint Gaussian."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from tests.hostlogic import FlatGraph
from tests.kary_support import dense_posterior

ITEM_M2F, ITEM_M2V, ITEM_MARG = L.ITEM_MESSAGE_TO_FACTOR, L.ITEM_MESSAGE_TO_VARIABLE, L.ITEM_INDIVIDUAL_MARGINAL


def flat_of(model):
    g = FlatGraph(model.edge_var, model.edge_fac, model.factor_ids, model.factor_kind, model.factor_var, edge_role=model.edge_role, dim=1, schedule=L.SCHED_TREE)
    assert g.status == L.OK, g.error
    if len(model.data_var):
        g.clamp(model.data_var)
    return g


class PlanRun:
    """message tables by slot + the model's rules, driven by the plan's items"""

    def __init__(self, g, model):
        self.g, self.m = g, model
        self.var_ids, self.var_off = g.arr("var_ids"), g.arr("var_off")
        self.edge_var, self.edge_fac = g.arr("edge_var"), g.arr("edge_fac_id")
        vbase, deg = g.arr("vbase"), g.arr("var_deg")
        k = np.arange(len(self.edge_var)) - self.var_off[self.edge_var]
        self.slot_of_edge = np.where(deg[self.edge_var] <= 8, vbase[self.edge_var] + 256 * k, vbase[self.edge_var] + k)
        self.edge_of_slot = {int(s): e for e, s in enumerate(self.slot_of_edge)}
        ns = g.scalar("nslots")
        self.f2v = np.full((ns, 2), np.nan)     # natural form (xi, w)
        self.v2f = np.full((ns, 2), np.nan)
        self.clamped = (g.arr("vinfo") & 16) != 0
        meta = model.meta
        self.fac_edges = {}
        for e, f in enumerate(self.edge_fac):
            self.fac_edges.setdefault(int(f), []).append(e)
        self.coef = {}                          # (factor id, variable id) -> c: sum_e c_e x_e = b + N(0, q)
        for fi, fid in enumerate(meta["kary_ids"]):
            for v in meta["fac_vars"][fi]:
                self.coef[(int(fid), int(v))] = 1.0
        for v, f, a in zip(meta["all_coef_var"], meta["all_coef_fac"], meta["all_coef"]):
            self.coef[(int(f), int(v))] = -float(a)
        self.qb = {int(fid): (float(meta["q"][fi]), float(meta["b"][fi])) for fi, fid in enumerate(meta["kary_ids"])}
        self.written = set()
        eidx = {(int(self.var_ids[v]), int(f)): e for e, (v, f) in enumerate(zip(self.edge_var, self.edge_fac))}
        for v, f, mu, s2 in zip(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance):
            self.f2v[self.slot_of_edge[eidx[(int(v), int(f))]]] = (mu / s2, 1.0 / s2)
        for v, f, y in zip(model.data_var, model.data_fac, model.data_y):
            self.v2f[self.slot_of_edge[eidx[(int(v), int(f))]]] = (y, np.inf)

    def moments(self, nat):
        return (nat[0], 0.0) if np.isinf(nat[1]) else (nat[0] / nat[1], 1.0 / nat[1])

    def m2f(self, slot, v):
        e = self.edge_of_slot[slot]
        assert self.edge_var[e] == v and not self.clamped[v]
        others = [self.slot_of_edge[x] for x in range(self.var_off[v], self.var_off[v + 1]) if x != e]
        assert others, "a variable of degree 1 has no variable→factor item"
        tot = self.f2v[others].sum(axis=0)
        assert not np.any(np.isnan(tot)), f"variable {self.var_ids[v]} → factor {self.edge_fac[e]}: an input is undefined at this stage"
        self._store(("v2f", slot), self.v2f, slot, tot)

    def m2v(self, slot):
        e = self.edge_of_slot[slot]
        f, v = int(self.edge_fac[e]), int(self.var_ids[self.edge_var[e]])
        assert not self.clamped[self.edge_var[e]], "no message into an observed variable"
        q, b = self.qb[f]
        cj = self.coef[(f, v)]
        mean, var = b, q
        for x in self.fac_edges[f]:
            if x == e:
                continue
            nat = self.v2f[self.slot_of_edge[x]]
            assert not np.any(np.isnan(nat)), f"factor {f} → variable {v}: the message of variable {self.var_ids[self.edge_var[x]]} is undefined at this stage"
            mx, vx = self.moments(nat)
            c = self.coef[(f, int(self.var_ids[self.edge_var[x]]))]
            mean -= c * mx
            var += c * c * vx
        mean /= cj
        var /= cj * cj
        self._store(("f2v", slot), self.f2v, slot, (mean / var, 1.0 / var))

    def _store(self, key, table, slot, value):
        assert key not in self.written, f"{key} is produced twice in one sweep"
        self.written.add(key)
        table[slot] = value

    def run(self):
        g = self.g
        rec = g.arr("tree_rec").reshape(-1, 5)
        off, kary, koff = g.arr("tree_stage_off"), g.arr("tree_kary"), g.arr("tree_kary_off")
        kslot = g.arr("kary_slot_all")
        marg = {}
        for s in range(len(off) - 1):
            items, ents = rec[off[s]:off[s + 1]], kary[koff[s]:koff[s + 1]]
            kinds = set(int(k) for k in items[:, 0]) | ({ITEM_M2V} if len(ents) else set())
            # a stage is of ONE kind: variable→factor items read factor→variable messages and write the other table (and the other way
            # round), so the items of a stage cannot read what another item of the same stage writes
            assert len(kinds) <= 1, f"stage {s} mixes item kinds {kinds}"
            for kind, slot, var, _, _ in items:
                if kind == ITEM_M2F:
                    self.m2f(int(slot), int(var))
                elif kind == ITEM_M2V:
                    self.m2v(int(slot))
                else:
                    assert kind == ITEM_MARG and s == len(off) - 2, "marginals belong to the last stage"
                    v = int(var)
                    tot = self.f2v[[self.slot_of_edge[x] for x in range(self.var_off[v], self.var_off[v + 1])]].sum(axis=0)
                    assert not np.any(np.isnan(tot)), f"marginal of {self.var_ids[v]}: an input is undefined"
                    marg[int(self.var_ids[v])] = (tot[0] / tot[1], 1.0 / tot[1])
            for ent in ents:
                self.m2v(int(kslot[ent]))
        return marg


class HpRun(PlanRun):
    """the heavy-path plan (cx_tree_plan.h: build_hp): item stages as above, and scans — here a plain walk along every path of the
    scanned depth in both directions, reading the side sums (everything a position hears except its skipped slots) the way
    cx_chain.hip's side kernel forms them"""

    def link_params(self, L):
        """step kind 3: a factor with more than two edges on a path of depth L is, given the messages of its OTHER variables (its light
        children and its observed variables: all of them must be defined by now), a pairwise rule x_recv = a x_send + b + N(0, q) between
        the path's two variables — per direction, as cx_kary.hip: k_kary_link_params forms it"""
        g = self.g
        frm, to = g.arr("hp_from"), g.arr("hp_to")
        l0, l1 = g.arr("hp_link_off")[L:L + 2]
        n = 0
        for l in range(l0, l1):
            ef, et = self.edge_of_slot[int(frm[l])], self.edge_of_slot[int(to[l])]
            f = int(self.edge_fac[ef])
            assert f == int(self.edge_fac[et]), "a link's two slots belong to one factor"
            if len(self.fac_edges[f]) == 2:
                continue
            n += 1
            q, b = self.qb[f]
            sm = sv = 0.0
            for x in self.fac_edges[f]:
                if x in (ef, et):
                    continue
                nat = self.v2f[self.slot_of_edge[x]]
                assert not np.any(np.isnan(nat)), f"link through factor {f}: the message of variable {self.var_ids[self.edge_var[x]]} is undefined at this step"
                mx, vx = self.moments(nat)
                c = self.coef[(f, int(self.var_ids[self.edge_var[x]]))]
                sm += c * mx
                sv += c * c * vx
            cf, ct = self.coef[(f, int(self.var_ids[self.edge_var[ef]]))], self.coef[(f, int(self.var_ids[self.edge_var[et]]))]
            self.eff[int(to[l])] = (-cf / ct, (b - sm) / ct, (q + sv) / (ct * ct))
            self.eff[int(frm[l])] = (-ct / cf, (b - sm) / cf, (q + sv) / (cf * cf))
        assert n > 0, "a parameter step for a depth without such links"

    def pair_rule(self, recv_slot, m):
        """the message into `recv_slot` through its factor, given the sender's variable→factor message m (natural form): a two-edge
        factor's own rule, or the pairwise parameters link_params left for this slot"""
        e = self.edge_of_slot[recv_slot]
        f, v = int(self.edge_fac[e]), int(self.var_ids[self.edge_var[e]])
        mx, vx = self.moments(m)
        if len(self.fac_edges[f]) > 2:
            a, b, q = self.eff[recv_slot]
            mean, var = a * mx + b, a * a * vx + q
            return np.array([mean / var, 1.0 / var])
        (x,) = [x for x in self.fac_edges[f] if x != e]
        q, b = self.qb[f]
        cj, c = self.coef[(f, v)], self.coef[(f, int(self.var_ids[self.edge_var[x]]))]
        mean, var = (b - c * mx) / cj, (q + c * c * vx) / (cj * cj)
        return np.array([mean / var, 1.0 / var])

    def scan(self, L, final, marg):
        g = self.g
        pos_var, skip0 = g.arr("hp_pos_var"), g.arr("hp_skip0")
        skip1 = g.arr("hp_skip1_down" if final else "hp_skip1_up")
        link_pos, frm, to = g.arr("hp_link_pos"), g.arr("hp_from"), g.arr("hp_to")
        hf, hb = g.arr("hp_head_fwd"), g.arr("hp_head_bwd")
        p0, p1 = g.arr("hp_pos_off")[L:L + 2]
        l0, l1 = g.arr("hp_link_off")[L:L + 2]
        side = {}
        for p in range(p0, p1):
            v = int(pos_var[p])
            sl = [self.slot_of_edge[x] for x in range(self.var_off[v], self.var_off[v + 1])]
            keep = [x for x in sl if x != skip0[p] and x != skip1[p]]
            assert {int(skip0[p]), int(skip1[p])} - {-1} <= set(int(x) for x in sl), "skipped slots belong to the position's variable"
            tot = self.f2v[keep].sum(axis=0) if keep else np.zeros(2)
            assert not np.any(np.isnan(tot)), f"scan of depth {L}: a side input of variable {self.var_ids[v]} is undefined"
            side[p] = tot
        alpha = np.zeros(2)
        for l in range(l0, l1):                       # head -> tail
            if hf[l]:
                alpha = np.zeros(2)
            alpha = self.pair_rule(int(to[l]), alpha + side[int(link_pos[l])])
            if final:
                self._store(("f2v", int(to[l])), self.f2v, int(to[l]), alpha)
        beta = np.zeros(2)
        for l in range(l1 - 1, l0 - 1, -1):           # tail -> head
            if hb[l]:
                beta = np.zeros(2)
            beta = self.pair_rule(int(frm[l]), beta + side[int(link_pos[l]) + 1])
            if final:
                self._store(("f2v", int(frm[l])), self.f2v, int(frm[l]), beta)
            else:
                self.f2v[int(frm[l])] = beta         # the way up: overwritten by the final scan
        if final:
            for p in range(p0, p1):
                v = int(pos_var[p])
                tot = self.f2v[[self.slot_of_edge[x] for x in range(self.var_off[v], self.var_off[v + 1])]].sum(axis=0)
                assert int(self.var_ids[v]) not in marg
                marg[int(self.var_ids[v])] = (tot[0] / tot[1], 1.0 / tot[1])
            # the final scan also leaves the links' two variable→factor messages (what each end hears from everybody else): the
            # messages of a heavy factor with more than two edges to its light children read them
            for l in range(l0, l1):
                for slot_, p in ((int(frm[l]), int(link_pos[l])), (int(to[l]), int(link_pos[l]) + 1)):
                    v = int(pos_var[p])
                    others = [self.slot_of_edge[x] for x in range(self.var_off[v], self.var_off[v + 1]) if self.slot_of_edge[x] != slot_]
                    self._store(("v2f", slot_), self.v2f, slot_, self.f2v[others].sum(axis=0))

    def run(self):
        g = self.g
        rec = g.arr("hp_rec").reshape(-1, 5)
        off, kary, koff = g.arr("hp_stage_off"), g.arr("hp_kary"), g.arr("hp_kary_off")
        kslot = g.arr("kary_slot_all")
        steps = g.arr("hp_steps").reshape(-1, 2)
        marg = {}
        up_written = set()
        self.eff = {}
        scanned = set()
        for kind, idx in steps:
            if kind == 3:
                assert int(idx) not in scanned, "the parameters of a depth's links are formed before its first scan"
                self.link_params(int(idx))
                continue
            if kind != 0:
                scanned.add(int(idx))
                self.scan(int(idx), kind == 2, marg)
                continue
            s = int(idx)
            items, ents = rec[off[s]:off[s + 1]], kary[koff[s]:koff[s + 1]]
            kinds = set(int(k) for k in items[:, 0]) | ({ITEM_M2V} if len(ents) else set())
            assert len(kinds) == 1, f"stage {s} mixes item kinds {kinds} (or is empty)"
            for k, slot, var, _, _ in items:
                if k == ITEM_M2F:
                    self.m2f(int(slot), int(var))
                elif k == ITEM_M2V:
                    self.m2v(int(slot))
                else:
                    assert k == ITEM_MARG and s == g.scalar("hp_marginal_stage")
                    v = int(var)
                    tot = self.f2v[[self.slot_of_edge[x] for x in range(self.var_off[v], self.var_off[v + 1])]].sum(axis=0)
                    assert not np.any(np.isnan(tot)), f"marginal of {self.var_ids[v]}: an input is undefined"
                    assert int(self.var_ids[v]) not in marg
                    marg[int(self.var_ids[v])] = (tot[0] / tot[1], 1.0 / tot[1])
            for ent in ents:
                self.m2v(int(kslot[ent]))
        del up_written
        return marg


@pytest.mark.parametrize("shape", ["random", "deep", "star"])
@pytest.mark.parametrize("n_factors,components,seed", [(1, 1, 1), (7, 1, 2), (60, 1, 3), (60, 3, 4), (300, 2, 5), (1500, 1, 6)])
def test_heavy_path_plan_is_the_exact_posterior(shape, n_factors, components, seed):
    """cx_tree_plan.h: build_hp — heavy paths by scans, light edges by items: every marginal of the dense solve, each message produced
    once, every input defined when it is read; the number of light depths is logarithmic where the level schedule's depth is not"""
    m = cx.synth.tree_model(n_factors, seed=seed, shape=shape, components=components, observe=0.3)
    g = flat_of(m)
    rc, err = g.tree_hp()
    assert rc == L.OK, err
    marg = HpRun(g, m).run()
    ids, em, ev = dense_posterior(m)
    assert sorted(marg) == sorted(int(i) for i in ids), "a marginal for every non-observed variable and no other"
    got = np.array([marg[int(i)] for i in ids])
    assert np.allclose(got[:, 0], em, rtol=1e-9, atol=1e-12) and np.allclose(got[:, 1], ev, rtol=1e-9, atol=1e-12)
    n_free = len(ids)
    assert g.tree()[0] == L.OK
    assert g.scalar("hp_levels") <= g.scalar("tree_depth") // 2 + 1, "never more light depths than the tree has levels of variables"
    assert len(g.arr("hp_pos_var")) + g.scalar("hp_single") == n_free, "every free variable is on exactly one path or single"


@pytest.mark.parametrize("n_factors,components,observe", [(3, 1, 0.0), (30, 1, 0.5), (301, 2, 0.3), (3000, 1, 0.3)])
def test_heavy_paths_of_a_comb_need_three_light_depths(n_factors, components, observe):
    """a spine of states with a tooth of two variables below each (synth.tree_model(shape="comb")): the level schedule is ~ n / 3 levels
    deep, the heavy-path plan has the spine at light depths 0 and 1 and the teeth one below — a constant number of launches"""
    m = cx.synth.tree_model(n_factors, seed=n_factors, shape="comb", components=components, observe=observe)
    g = flat_of(m)
    rc, err = g.tree_hp()
    assert rc == L.OK, err
    assert g.tree()[0] == L.OK
    # (the heavy-path plan roots a component at an end of its longest path: the spine is the root's heavy path, the teeth one light edge down)
    assert g.scalar("hp_levels") <= 2 + (components > 1) and g.scalar("hp_launches") <= 30
    if n_factors >= 300:
        assert g.scalar("tree_depth") >= n_factors // (3 * components) and 2 * g.scalar("tree_depth") + 1 > 3 * g.scalar("hp_launches")
    marg = HpRun(g, m).run()
    ids, em, ev = dense_posterior(m)
    assert sorted(marg) == sorted(int(i) for i in ids)
    got = np.array([marg[int(i)] for i in ids])
    assert np.allclose(got[:, 0], em, rtol=1e-9, atol=1e-12) and np.allclose(got[:, 1], ev, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("k_choices,n_factors,seed", [((1, 1, 2, 3, 5), 1500, 1), ((2, 3, 5), 400, 2), ((2,), 300, 3), ((5,), 200, 4)])
def test_heavy_paths_run_through_factors_of_more_than_two_edges(k_choices, n_factors, seed):
    """long paths with side branches whose factors have 3..6 variables: the links through such factors get their pairwise parameters in a
    step of their own before the depth's first scan (every light child's message defined by then — the executor fails otherwise), the
    factor's messages to its light children are items after the final scan; a few dozen launches where the level schedule has two per
    level, the dense solve's marginals"""
    m = cx.synth.tree_model(n_factors, seed=seed, k_choices=k_choices, shape="deep", observe=0.3)
    g = flat_of(m)
    rc, err = g.tree_hp()
    assert rc == L.OK, err
    assert g.tree()[0] == L.OK
    assert g.scalar("hp_kary_links") > 0.3 * len(g.arr("hp_link_pos"))
    assert g.scalar("hp_launches") <= 80 < 2 * g.scalar("tree_depth") + 1 and g.scalar("hp_levels") <= 6
    steps = g.arr("hp_steps").reshape(-1, 2)
    assert 1 <= int((steps[:, 0] == 3).sum()) <= g.scalar("hp_levels")
    marg = HpRun(g, m).run()
    ids, em, ev = dense_posterior(m)
    assert sorted(marg) == sorted(int(i) for i in ids)
    got = np.array([marg[int(i)] for i in ids])
    assert np.allclose(got[:, 0], em, rtol=1e-9, atol=1e-12) and np.allclose(got[:, 1], ev, rtol=1e-9, atol=1e-12)


def test_sixty_small_random_forests_under_both_plans():
    """every shape, arity mix, number of components and share of observed leaves a seed produces: the level plan and the heavy-path plan,
    both executed in numpy, both the dense solve's marginals (an undefined input at any step fails inside the executor)"""
    rng = np.random.default_rng(2024)
    for trial in range(60):
        n_factors = int(rng.integers(1, 50))
        shape = ["random", "deep", "star", "comb"][trial % 4]
        m = cx.synth.tree_model(n_factors, seed=3000 + trial, shape=shape, components=int(rng.integers(1, 4)), observe=float(rng.uniform(0, 0.6)),
                                k_choices=tuple(int(k) for k in rng.integers(1, 7, size=3)))
        ids, em, ev = dense_posterior(m)
        for plan in ("level", "heavy paths"):
            g = flat_of(m)
            rc, err = g.tree() if plan == "level" else g.tree_hp()
            assert rc == L.OK, err
            marg = (PlanRun if plan == "level" else HpRun)(g, m).run()
            assert sorted(marg) == sorted(int(i) for i in ids), f"trial {trial} ({shape}), {plan}: a marginal for every non-observed variable"
            got = np.array([marg[int(i)] for i in ids])
            assert np.allclose(got[:, 0], em, rtol=1e-9, atol=1e-12) and np.allclose(got[:, 1], ev, rtol=1e-9, atol=1e-12), f"trial {trial} ({shape}), {plan}"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_heavy_paths_of_pairwise_trees_are_logarithmic(seed):
    m = cx.synth.tree_model(2000, seed=seed, k_choices=(1,), shape="deep", observe=0.3)
    g = flat_of(m)
    assert g.tree_hp()[0] == L.OK
    n_free = len(g.arr("hp_pos_var")) + g.scalar("hp_single")
    assert g.scalar("hp_levels") <= int(np.log2(n_free)) + 1
    marg = HpRun(g, m).run()
    ids, em, ev = dense_posterior(m)
    got = np.array([marg[int(i)] for i in ids])
    assert np.allclose(got[:, 0], em, rtol=1e-9, atol=1e-12) and np.allclose(got[:, 1], ev, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("shape", ["random", "deep", "star"])
@pytest.mark.parametrize("n_factors,components,seed", [(1, 1, 1), (7, 1, 2), (60, 1, 3), (60, 3, 4), (300, 2, 5)])
def test_one_pass_of_the_plan_is_the_exact_posterior(shape, n_factors, components, seed):
    m = cx.synth.tree_model(n_factors, seed=seed, shape=shape, components=components, observe=0.3)
    g = flat_of(m)
    rc, err = g.tree()
    assert rc == L.OK, err
    run = PlanRun(g, m)
    marg = run.run()
    ids, em, ev = dense_posterior(m)
    assert sorted(marg) == sorted(int(i) for i in ids), "a marginal for every non-observed variable and no other"
    got = np.array([marg[int(i)] for i in ids])
    assert np.allclose(got[:, 0], em, rtol=1e-9, atol=1e-12) and np.allclose(got[:, 1], ev, rtol=1e-9, atol=1e-12)
    assert g.scalar("tree_components") == components and g.scalar("tree_marginals") == len(ids)
    # stages: 2 x depth + 1; the depth is half the longest path of the deepest component, rounded up to a variable
    assert len(g.arr("tree_stage_off")) - 1 == 2 * g.scalar("tree_depth") + 1


def test_depth_is_half_the_diameter():
    """a path of T variables: rooted at its middle, the plan is T - 1 levels deep at most (variables and factors alternate), not 2 T"""
    for T in (2, 3, 10, 31):
        m = cx.synth.ssm_chain_linear(T, seed=T)
        m.meta.update(kary_ids=np.zeros(0, dtype=np.int64))
        g = FlatGraph(m.edge_var, m.edge_fac, m.factor_ids, m.factor_kind, m.factor_var, edge_role=m.edge_role, schedule=L.SCHED_TREE)
        g.clamp(m.data_var)
        rc, err = g.tree()
        assert rc == L.OK, err
        # longest path of the free part: likelihood factor - x_1 - ... - x_T - likelihood factor = 2 T nodes + ... edges: 2 (T - 1) + 2 hops
        assert g.scalar("tree_depth") in (T, T + 1), (T, g.scalar("tree_depth"))


def test_cycles_are_refused_and_observed_variables_cut_them():
    m = cx.synth.gaussian_grid(4, 4, seed=3)
    g = FlatGraph(m.edge_var, m.edge_fac, m.factor_ids, m.factor_kind, m.factor_var, edge_role=m.edge_role, schedule=L.SCHED_TREE)
    g.clamp(m.data_var)
    rc, err = g.tree()
    assert rc == L.ERR_UNSUPPORTED and "cycle" in err
    # a 2 x N ladder is loopy; observing one rail leaves a path
    lad = cx.synth.gaussian_grid(2, 6, seed=4)
    g2 = FlatGraph(lad.edge_var, lad.edge_fac, lad.factor_ids, lad.factor_kind, lad.factor_var, edge_role=lad.edge_role, schedule=L.SCHED_TREE)
    g2.clamp(lad.data_var)
    assert g2.tree()[0] == L.ERR_UNSUPPORTED
    g2.clamp(np.r_[lad.data_var, lad.x_ids[:6]])
    rc, err = g2.tree()
    assert rc == L.OK, err
    assert g2.scalar("tree_components") == 1 and g2.scalar("tree_marginals") == 6
