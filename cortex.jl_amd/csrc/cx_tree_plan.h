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
// cx_kernels.hip / cx_mvbatch.hip) plus,
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

template <class H>
int32_t build(const H *h, Out &out, std::string &err) {
    const int64_t nv = h->nv, nf = h->nf, ne = h->ne;
    out = Out();
    // factor index of every edge, factor CSR
    std::vector<int32_t> efac(ne);
    for (int64_t e = 0; e < ne; e++) {
        auto it = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), h->edge_fac_id[e]);
        if (it == h->fac_ids.end() || *it != h->edge_fac_id[e]) return fail_(err, CX_ERR_STATE, "tree schedule: edge names an unknown factor");
        efac[e] = (int32_t)(it - h->fac_ids.begin());
    }
    std::vector<int32_t> foff(nf + 1, 0);
    for (int64_t e = 0; e < ne; e++) foff[efac[e] + 1]++;
    for (int64_t f = 0; f < nf; f++) foff[f + 1] += foff[f];
    std::vector<int32_t> fedge(ne), fill(foff.begin(), foff.end() - 1);
    for (int64_t e = 0; e < ne; e++) fedge[fill[efac[e]]++] = (int32_t)e;
    auto is_free = [&](int32_t v) { return !(h->vinfo[v] & (kClamped | kGhost)); };
    auto vdeg = [&](int32_t v) { return h->var_off[v + 1] - h->var_off[v]; };
    // nodes: variable v -> v, factor f -> nv + f.  Only free variables and the factors they touch take part.
    const int64_t nn = nv + nf;
    std::vector<int32_t> level(nn, -1), parent_edge(nn, -1), comp_of;      // parent_edge: the edge that leads to the parent (-1: root)
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
    std::vector<int32_t> members;       // nodes of all components, in final breadth-first order (parents before children)
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
        for (int32_t s = 0; s < tmp_level[b] / 2; s++) {
            const int32_t e = tmp_parent[c];
            c = c < nv ? (int32_t)nv + efac[e] : h->edge_var[e];
        }
        if (c >= nv) { const int32_t e = tmp_parent[c]; c = e >= 0 ? h->edge_var[e] : (int32_t)v0; }
        (void)bfs(c, cycle);
        for (int32_t n : queue) { level[n] = tmp_level[n]; parent_edge[n] = tmp_parent[n]; members.push_back(n); out.depth = std::max(out.depth, tmp_level[n]); }
        out.n_components++;
    }
    // how many free variables a factor touches (a variable→factor message towards a factor with a single free variable has no reader)
    std::vector<int32_t> ffree(nf, 0);
    for (int64_t e = 0; e < ne; e++) if (is_free(h->edge_var[e])) ffree[efac[e]]++;
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

}  // namespace treeplan
}  // namespace cx
