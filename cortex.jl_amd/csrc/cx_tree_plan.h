// cx_tree_plan.h — the GPU-free part of CX_SCHED_TREE: ONE cx_sweep on a graph whose non-observed part is a forest is what ONE
// update_marginals! of the reference leaves there — every message computed once from final inputs, leaves to root and back
// (/root/reference/src/inference_engine.jl:575-608: the forward and the reverse pass over the pending signals; on a tree the second
// call finds nothing pending).  The chain-scan schedule covers paths with prefix scans; this one covers any forest, level by level:
//   root     per component the centre of its longest path (two breadth-first searches), so that the depth — the number of stages — is
//            half the diameter
//   up       stages from the deepest level to level 1: the nodes of a level send to their parents — variables (even levels) their
//            variable→factor message, factors (odd levels) their factor→variable message
//   down     stages from level 0 to the last but one: the nodes of a level send to their children
//   last     the marginal of every non-observed variable
// A stage is a list of batch items (5 int32 each: kind, slot, variable, rule table (dim > 1), 0 — what cx_update_batch stages,
// cx_batch.hip / cx_mvbatch.hip) plus,
// for factors with more than two edges, a list of entries of the k-ary table (cx_kary.hip).  Items of a stage are independent.
// Lazy like the reference: no message into an observed variable or a stand-in, no variable→factor message towards a factor whose
// other variables are all observed, nothing out of a variable of degree 1 (its stored message has no dependencies,
// /root/reference/src/dependencies.jl:48-55).  A cycle among the non-observed variables is refused.
// Pure host C++ over any struct H with cx_handle's host fields (cx_flatten.h).
#pragma once

#include <algorithm>

#include "cx_flatten.h"

namespace cx {
namespace treeplan {

struct Out {
    std::vector<int32_t> rec;            // 5 per item
    std::vector<int64_t> stage_off;      // items of stage s: [stage_off[s], stage_off[s + 1])
    std::vector<int32_t> kary;           // k-ary entries (row * 8 + position), by stage
    std::vector<int64_t> kary_off;
    int32_t depth = 0;                   // levels below the roots (variables on even, factors on odd levels)
    int64_t n_components = 0, n_up = 0, n_down = 0, n_marginals = 0;
};

using flat::fail_;

// the forest of non-observed variables, rooted: shared by the level schedule below and the heavy-path schedule further down
struct Rooted {
    std::vector<int32_t> efac, foff, fedge;          // factor index of every edge; factor -> its edges
    std::vector<int32_t> level, parent_edge;         // per node (variable v -> v, factor f -> nv + f): -1 = not in the forest; the edge that leads to the parent (-1: root)
    std::vector<int32_t> members;                    // nodes of all components in breadth-first order (parents before children)
    std::vector<int32_t> ffree;                      // free variables a factor touches
    int32_t depth = 0;
    int64_t n_components = 0;
};

// at_end: root every component at an END of its longest path instead of its centre — the level schedule wants the centre (half the
// depth: half the stages), the heavy-path schedule the end: the longest path is then ONE heavy path from the root instead of two halves
// on two light depths (a chain: one scan per direction instead of two)
template <class H>
int32_t root_forest(const H *h, Rooted &R, std::string &err, bool at_end = false) {
    const int64_t nv = h->nv, nf = h->nf, ne = h->ne;
    R = Rooted();
    auto &efac = R.efac; auto &foff = R.foff; auto &fedge = R.fedge; auto &level = R.level; auto &parent_edge = R.parent_edge; auto &members = R.members;
    // factor index of every edge, factor CSR
    efac.resize(ne);
    for (int64_t e = 0; e < ne; e++) {
        auto it = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), h->edge_fac_id[e]);
        if (it == h->fac_ids.end() || *it != h->edge_fac_id[e]) return fail_(err, CX_ERR_STATE, "tree schedule: edge names an unknown factor");
        efac[e] = (int32_t)(it - h->fac_ids.begin());
    }
    foff.assign(nf + 1, 0);
    for (int64_t e = 0; e < ne; e++) foff[efac[e] + 1]++;
    for (int64_t f = 0; f < nf; f++) foff[f + 1] += foff[f];
    fedge.resize(ne);
    std::vector<int32_t> fill(foff.begin(), foff.end() - 1);
    for (int64_t e = 0; e < ne; e++) fedge[fill[efac[e]]++] = (int32_t)e;
    auto is_free = [&](int32_t v) { return !(h->vinfo[v] & (kClamped | kGhost)); };
    auto vdeg = [&](int32_t v) { return h->var_off[v + 1] - h->var_off[v]; };
    // nodes: variable v -> v, factor f -> nv + f.  Only free variables and the factors they touch take part.
    const int64_t nn = nv + nf;
    level.assign(nn, -1); parent_edge.assign(nn, -1);
    std::vector<int32_t> queue;
    queue.reserve(nn);
    // breadth-first search from `start` over the free part; fills level / parent_edge for the nodes it reaches when `keep`, returns
    // the last node reached (a farthest one) and reports a cycle
    std::vector<int32_t> seen_epoch(nn, 0);
    int32_t epoch = 0;
    std::vector<int32_t> tmp_level(nn, 0), tmp_parent(nn, -1);
    auto bfs = [&](int32_t start, bool &cycle) -> int32_t {
        epoch++;
        queue.clear();
        queue.push_back(start);
        seen_epoch[start] = epoch; tmp_level[start] = 0; tmp_parent[start] = -1;
        int32_t last = start;
        for (size_t qi = 0; qi < queue.size(); qi++) {
            const int32_t n = queue[qi];
            last = n;
            if (n < nv) {
                for (int32_t e = h->var_off[n]; e < h->var_off[n + 1]; e++) {
                    if (e == tmp_parent[n]) continue;
                    const int32_t m = (int32_t)nv + efac[e];
                    if (seen_epoch[m] == epoch) { cycle = true; continue; }
                    seen_epoch[m] = epoch; tmp_level[m] = tmp_level[n] + 1; tmp_parent[m] = e;
                    queue.push_back(m);
                }
            } else {
                const int32_t f = n - (int32_t)nv;
                for (int32_t k = foff[f]; k < foff[f + 1]; k++) {
                    const int32_t e = fedge[k];
                    if (e == tmp_parent[n]) continue;
                    const int32_t m = h->edge_var[e];
                    if (!is_free(m)) continue;                 // observed variables and stand-ins are constants hanging off the factor
                    if (seen_epoch[m] == epoch) { cycle = true; continue; }
                    seen_epoch[m] = epoch; tmp_level[m] = tmp_level[n] + 1; tmp_parent[m] = e;
                    queue.push_back(m);
                }
            }
        }
        return last;
    };
    members.reserve(nn);
    for (int64_t v0 = 0; v0 < nv; v0++) {
        if (!is_free((int32_t)v0) || level[v0] >= 0 || vdeg((int32_t)v0) == 0) continue;
        bool cycle = false;
        const int32_t a = bfs((int32_t)v0, cycle);
        if (cycle) return fail_(err, CX_ERR_UNSUPPORTED, "tree schedule: the non-observed variables of the graph form a cycle (through the component of variable " +
                                                             std::to_string(h->var_ids[v0]) + "): one pass up and down is not exact there; use the fused or flooding schedule");
        const int32_t b = bfs(a, cycle);
        // walk back from b half of the distance; the root is a variable (step one further when the middle is a factor)
        int32_t c = b;
        for (int32_t s = 0; s < (at_end ? 0 : tmp_level[b] / 2); s++) {
            const int32_t e = tmp_parent[c];
            c = c < nv ? (int32_t)nv + efac[e] : h->edge_var[e];
        }
        if (c >= nv) { const int32_t e = tmp_parent[c]; c = e >= 0 ? h->edge_var[e] : (int32_t)v0; }
        (void)bfs(c, cycle);
        for (int32_t n : queue) { level[n] = tmp_level[n]; parent_edge[n] = tmp_parent[n]; members.push_back(n); R.depth = std::max(R.depth, tmp_level[n]); }
        R.n_components++;
    }
    // how many free variables a factor touches (a variable→factor message towards a factor with a single free variable has no reader)
    R.ffree.assign(nf, 0);
    for (int64_t e = 0; e < ne; e++) if (is_free(h->edge_var[e])) R.ffree[efac[e]]++;
    return CX_OK;
}

template <class H>
int32_t build(const H *h, Out &out, std::string &err) {
    const int64_t nv = h->nv, nf = h->nf;
    out = Out();
    Rooted R;
    { const int32_t rc = root_forest(h, R, err); if (rc != CX_OK) return rc; }
    const auto &efac = R.efac; const auto &foff = R.foff; const auto &fedge = R.fedge; const auto &level = R.level; const auto &parent_edge = R.parent_edge;
    const auto &members = R.members; const auto &ffree = R.ffree;
    out.depth = R.depth; out.n_components = R.n_components;
    auto is_free = [&](int32_t v) { return !(h->vinfo[v] & (kClamped | kGhost)); };
    auto vdeg = [&](int32_t v) { return h->var_off[v + 1] - h->var_off[v]; };
    (void)nf;
    // stages: up (levels depth .. 1), down (levels 0 .. depth - 1), marginals
    const int32_t D = out.depth, nstages = 2 * D + 1;
    std::vector<std::vector<int32_t>> items(nstages), kents(nstages);
    auto push_item = [&](int32_t stage, int32_t kind, int32_t slot, int32_t var) {
        auto &r = items[stage];
        r.push_back(kind); r.push_back(slot); r.push_back(var); r.push_back(0); r.push_back(0);
    };
    auto m2v = [&](int32_t stage, int32_t e) {                 // the message of edge e's factor into edge e's variable
        const int32_t sl = flat::slot_of_edge_t(h, e);
        if (!h->slot_kary.empty() && h->slot_kary[sl] >= 0) kents[stage].push_back(h->slot_kary[sl]);
        else if (h->partner[sl] >= 0) {
            push_item(stage, CX_ITEM_MESSAGE_TO_VARIABLE, sl, h->edge_var[e]);
            if (h->cfg.dim > 1) items[stage][items[stage].size() - 2] = h->spdir[h->partner[sl]];      // dim > 1: the rule table of the SENDING slot travels in the item
        }
        // (a factor with one edge: its message is a stored constant)
    };
    auto m2f = [&](int32_t stage, int32_t e) {                 // the message of edge e's variable into edge e's factor
        const int32_t v = h->edge_var[e];
        if (vdeg(v) < 2 || ffree[efac[e]] < 2) return;
        push_item(stage, CX_ITEM_MESSAGE_TO_FACTOR, flat::slot_of_edge_t(h, e), v);
    };
    for (int32_t n : members) {
        const int32_t L = level[n];
        if (L > 0) {                       // up: to the parent, in stage D - L
            if (n < nv) { m2f(D - L, parent_edge[n]); out.n_up++; }
            else { m2v(D - L, parent_edge[n]); out.n_up++; }
        }
        // down: to every child, in stage D + L
        if (n < nv) {
            for (int32_t e = h->var_off[n]; e < h->var_off[n + 1]; e++)
                if (e != parent_edge[n]) { m2f(D + L, e); out.n_down++; }
            push_item(2 * D, CX_ITEM_INDIVIDUAL_MARGINAL, n, n);
            out.n_marginals++;
        } else {
            const int32_t f = n - (int32_t)nv;
            for (int32_t k = foff[f]; k < foff[f + 1]; k++) {
                const int32_t e = fedge[k];
                if (e != parent_edge[n] && is_free(h->edge_var[e])) { m2v(D + L, e); out.n_down++; }
            }
        }
    }
    // (level D nodes have no children; level 0 nodes no parent: stage indices stay inside [0, 2 D])
    out.stage_off.assign(1, 0); out.kary_off.assign(1, 0);
    for (int32_t s = 0; s < nstages; s++) {
        out.rec.insert(out.rec.end(), items[s].begin(), items[s].end());
        out.kary.insert(out.kary.end(), kents[s].begin(), kents[s].end());
        out.stage_off.push_back((int64_t)out.rec.size() / 5);
        out.kary_off.push_back((int64_t)out.kary.size());
    }
    return CX_OK;
}

// ---- heavy paths (scalar messages and dim 2 .. 4): the same exact sweep in O(log n) rounds of launches instead of one per level -------
// The level schedule pays two launches per level of the tree: a chain with side branches (a state-space model whose states carry a
// latent layer of their own: depth ~ T) costs ~ 2 T launches.  Here every variable picks its HEAVY child — the child with the largest
// subtree among those it reaches through a two-edge factor — and the heavy edges form disjoint paths; any root-to-leaf walk leaves a
// path (takes a LIGHT edge) at most log2(n) times.  The paths of one light depth are independent of each other and are what the
// chain-scan schedule handles: ONE segmented scan per light depth and direction (cx_chain.hip: side sums, tile totals, walks) computes
// every message along them, whatever their length; the light edges — from a path's head to the variable above it, factors with more
// than two edges, paths of a single variable — stay batch items as in the level schedule.  A sweep, with Lmax the largest light depth:
//   up    for l = Lmax + 1 .. 1:   [scan of the paths of depth l towards their heads: the message that leaves each head upwards is
//                                   complete because everything below the path is]
//                                  [items A: head (or single variable) of depth l -> its parent factor]
//                                  [items B: light factors below the variables of depth l - 1 -> those variables]
//   down  for l = 0 .. Lmax:       [items C: variables of depth l - 1 -> their light factors]   [items D: those factors -> the heads of depth l]
//                                  [scan of the paths of depth l, final: both directions exact, marginals written]
//   last  marginals of the variables that are on no path of two or more
// Exactness: every message is computed once from final inputs, as in the level schedule (the up scan's messages AWAY from the heads lack
// what comes from above and are overwritten by the down scan before anything reads them).
// A heavy edge may also run through a factor with MORE than two edges: given the messages of the factor's other variables — its light
// children, final once the depth below has been scanned, and its observed variables — the factor is a pairwise linear rule between the
// path's two variables (cx_kary.hip: k_kary_link_params forms its (a, b, q) per direction before the depth's first scan: step kind 3);
// on the way down the factor's messages to its light children are items that read the two variable→factor messages the final scan left.
struct HP {
    std::vector<int32_t> rec;            // item stages, as in Out
    std::vector<int64_t> stage_off;
    std::vector<int32_t> kary;
    std::vector<int64_t> kary_off;
    // every path of two or more variables, the light depths one after the other; positions run from the head (nearest the root) to the
    // tail, links likewise — the arrays of cx_chains.h, with the head's slot towards its parent as a second skipped slot on the way up
    std::vector<int32_t> pos_var, skip0, skip1_up, skip1_down, link_pos, from, to;
    std::vector<uint8_t> head_fwd, head_bwd;
    std::vector<int64_t> pos_off, link_off;      // per light depth, levels + 1 entries
    std::vector<int32_t> steps;                  // pairs (kind, index): 0 item stage, 1 scan of depth `index` on the way up (messages only), 2 the final scan of depth `index`,
                                                 // 3 the pairwise parameters of the links of depth `index` that run through factors with more than two edges
    int32_t levels = 0, depth = 0, marginal_stage = -1;
    int64_t n_components = 0, n_paths = 0, n_single = 0, launches = 0, n_kary_links = 0;
};

template <class H>
int32_t build_hp(const H *h, HP &out, std::string &err) {
    const int64_t nv = h->nv;
    out = HP();
    Rooted R;
    { const int32_t rc = root_forest(h, R, err, true); if (rc != CX_OK) return rc; }
    const auto &efac = R.efac; const auto &foff = R.foff; const auto &fedge = R.fedge; const auto &level = R.level; const auto &parent_edge = R.parent_edge;
    const auto &members = R.members; const auto &ffree = R.ffree;
    out.depth = R.depth; out.n_components = R.n_components;
    auto is_free = [&](int32_t v) { return !(h->vinfo[v] & (kClamped | kGhost)); };
    auto vdeg = [&](int32_t v) { return h->var_off[v + 1] - h->var_off[v]; };
    auto parent_of = [&](int32_t n) { const int32_t e = parent_edge[n]; return e < 0 ? -1 : (n < nv ? (int32_t)nv + efac[e] : h->edge_var[e]); };
    auto slot = [&](int32_t e) { return flat::slot_of_edge_t(h, e); };
    const int64_t nn = nv + h->nf;
    // subtree sizes (free variables), children before parents
    std::vector<int64_t> sz(nn, 0);
    for (int32_t n : members) if (n < nv) sz[n] = 1;
    for (size_t i = members.size(); i-- > 0;) { const int32_t n = members[i], p = parent_of(n); if (p >= 0) sz[p] += sz[n]; }
    // heavy child: through a two-edge factor with a rule of its own (partner slot) or a factor of the k-ary table, into a variable
    // that sends (degree >= 2)
    std::vector<int32_t> heavy(nv, -1), hfac(nv, -1);
    std::vector<uint8_t> fkary(h->nf, 0);                 // the factor is in the k-ary table (more than two edges)
    for (int32_t n : members) {
        if (n < nv) continue;
        const int32_t f = n - (int32_t)nv, ev = parent_edge[n];
        if (ev < 0) continue;
        const int32_t v = h->edge_var[ev];
        const bool kary = !h->slot_kary.empty() && h->slot_kary[slot(ev)] >= 0;
        if (!kary && (foff[f + 1] - foff[f] != 2 || h->partner[slot(ev)] < 0)) continue;
        fkary[f] = kary ? 1 : 0;
        for (int32_t k = foff[f]; k < foff[f + 1]; k++) {
            const int32_t ec = fedge[k];
            if (ec == ev) continue;
            const int32_t c = h->edge_var[ec];
            if (!is_free(c) || vdeg(c) < 2 || level[c] != level[n] + 1 || parent_edge[c] != ec) continue;
            if (heavy[v] < 0 || sz[c] > sz[heavy[v]]) { heavy[v] = c; hfac[v] = f; }
        }
    }
    // light depth of every variable; heads
    std::vector<int32_t> ld(nv, -1);
    std::vector<uint8_t> is_head(nv, 0);
    int32_t lmax = 0;
    for (int32_t n : members) {
        if (n >= nv) continue;
        const int32_t pf = parent_of(n);
        if (pf < 0) { ld[n] = 0; is_head[n] = 1; continue; }
        const int32_t v = parent_of(pf);
        const bool on = heavy[v] == n;
        ld[n] = ld[v] + (on ? 0 : 1);
        is_head[n] = on ? 0 : 1;
        lmax = std::max(lmax, ld[n]);
    }
    out.levels = lmax + 1;
    // paths by light depth
    std::vector<std::vector<int32_t>> heads(lmax + 1);
    std::vector<uint8_t> on_path(nv, 0);
    for (int32_t n : members) if (n < nv && is_head[n] && heavy[n] >= 0) heads[ld[n]].push_back(n);
    out.pos_off.assign(1, 0); out.link_off.assign(1, 0);
    std::vector<uint8_t> kary_depth(lmax + 1, 0);         // the depth has links through factors with more than two edges
    for (int32_t L = 0; L <= lmax; L++) {
        bool depth_has_kary = false;
        for (int32_t hd : heads[L]) {
            out.n_paths++;
            int32_t v = hd, prev_to = -1;
            while (true) {
                on_path[v] = 1;
                const int32_t c = heavy[v];
                const int32_t s_next = c >= 0 ? slot(parent_edge[nv + hfac[v]]) : -1;
                out.pos_var.push_back(v);
                if (v == hd) {
                    out.skip0.push_back(s_next);
                    out.skip1_down.push_back(-1);
                    out.skip1_up.push_back(parent_edge[v] >= 0 ? slot(parent_edge[v]) : -1);
                } else {
                    out.skip0.push_back(prev_to);
                    out.skip1_down.push_back(s_next);
                    out.skip1_up.push_back(s_next);
                }
                if (c < 0) break;
                out.link_pos.push_back((int32_t)out.pos_var.size() - 1);
                out.from.push_back(s_next);
                prev_to = slot(parent_edge[c]);
                out.to.push_back(prev_to);
                out.head_fwd.push_back(v == hd ? 1 : 0);
                out.head_bwd.push_back(heavy[c] < 0 ? 1 : 0);
                if (fkary[hfac[v]]) { out.n_kary_links++; depth_has_kary = true; }
                v = c;
            }
        }
        out.pos_off.push_back((int64_t)out.pos_var.size());
        out.link_off.push_back((int64_t)out.link_pos.size());
        kary_depth[L] = depth_has_kary ? 1 : 0;
    }
    // light factors by the depth of what hangs below them (= depth of the variable above + 1); heavy factors with more than two edges
    // likewise: their other children are light
    std::vector<std::vector<int32_t>> lfac(lmax + 2), hkfac(lmax + 2);
    for (int32_t n : members) {
        if (n < nv) continue;
        const int32_t f = n - (int32_t)nv, v = parent_of(n);
        if (v < 0) continue;
        if (hfac[v] == f) { if (fkary[f]) hkfac[ld[v] + 1].push_back(f); continue; }
        lfac[ld[v] + 1].push_back(f);
    }
    std::vector<std::vector<int32_t>> hv(lmax + 1);      // heads and single variables by depth (those with a parent)
    for (int32_t n : members) if (n < nv && is_head[n] && parent_edge[n] >= 0) hv[ld[n]].push_back(n);
    // ---- stages -------------------------------------------------------------------------------------------------------------------
    std::vector<int32_t> cur, curk;
    auto push_item = [&](int32_t kind, int32_t sl, int32_t var) { cur.push_back(kind); cur.push_back(sl); cur.push_back(var); cur.push_back(0); cur.push_back(0); };
    auto m2v = [&](int32_t e) {
        const int32_t sl = slot(e);
        if (!h->slot_kary.empty() && h->slot_kary[sl] >= 0) curk.push_back(h->slot_kary[sl]);
        else if (h->partner[sl] >= 0) {
            push_item(CX_ITEM_MESSAGE_TO_VARIABLE, sl, h->edge_var[e]);
            if (h->cfg.dim > 1) cur[cur.size() - 2] = h->spdir[h->partner[sl]];      // dim > 1: the rule table of the SENDING slot travels in the item
        }
    };
    auto m2f = [&](int32_t e) {
        const int32_t v = h->edge_var[e];
        if (vdeg(v) < 2 || ffree[efac[e]] < 2) return;
        push_item(CX_ITEM_MESSAGE_TO_FACTOR, slot(e), v);
    };
    out.stage_off.assign(1, 0); out.kary_off.assign(1, 0);
    auto close_stage = [&]() -> bool {
        if (cur.empty() && curk.empty()) return false;
        out.rec.insert(out.rec.end(), cur.begin(), cur.end());
        out.kary.insert(out.kary.end(), curk.begin(), curk.end());
        out.stage_off.push_back((int64_t)out.rec.size() / 5);
        out.kary_off.push_back((int64_t)out.kary.size());
        out.steps.push_back(0); out.steps.push_back((int32_t)out.stage_off.size() - 2);
        out.launches += 1;      // one launch per item stage: its k-ary entries ride in the same record list (cx_api_sweep.hip: build_tree, kind 32)
        cur.clear(); curk.clear();
        return true;
    };
    std::vector<uint8_t> params_done(lmax + 1, 0);
    auto scan = [&](int32_t kind, int32_t L) {
        if (out.link_off[L + 1] == out.link_off[L]) return;
        if (kary_depth[L] && !params_done[L]) {      // before the depth's first scan: everything below it is final
            out.steps.push_back(3); out.steps.push_back(L);
            out.launches += 1;
            params_done[L] = 1;
        }
        out.steps.push_back(kind); out.steps.push_back(L);
        out.launches += (h->cfg.dim > 1 && kind == 2) ? 4 : 3;      // side sums, thread totals, walks (+ the marginal pass of dim 2 .. 4)
    };
    for (int32_t l = lmax + 1; l >= 1; l--) {
        if (l <= lmax) {
            scan(1, l);
            for (int32_t c : hv[l]) m2f(parent_edge[c]);
            close_stage();
        }
        for (int32_t f : lfac[l]) m2v(parent_edge[nv + f]);
        close_stage();
    }
    for (int32_t l = 0; l <= lmax; l++) {
        if (l >= 1) {
            for (int32_t f : lfac[l]) m2f(parent_edge[nv + f]);
            close_stage();
            for (int32_t f : lfac[l])
                for (int32_t k = foff[f]; k < foff[f + 1]; k++) {
                    const int32_t e = fedge[k];
                    if (e != parent_edge[nv + f] && is_free(h->edge_var[e])) m2v(e);
                }
            for (int32_t f : hkfac[l]) {                 // the light children of a heavy factor: the final scan of depth l - 1 left both of its path variables' messages
                const int32_t v = h->edge_var[parent_edge[nv + f]], ec = parent_edge[heavy[v]];
                for (int32_t k = foff[f]; k < foff[f + 1]; k++) {
                    const int32_t e = fedge[k];
                    if (e != parent_edge[nv + f] && e != ec && is_free(h->edge_var[e])) m2v(e);
                }
            }
            close_stage();
        }
        scan(2, l);
    }
    for (int32_t n : members) if (n < nv && !on_path[n]) { push_item(CX_ITEM_INDIVIDUAL_MARGINAL, n, n); out.n_single++; }
    if (close_stage()) out.marginal_stage = (int32_t)out.stage_off.size() - 2;
    return CX_OK;
}

}  // namespace treeplan
}  // namespace cx
