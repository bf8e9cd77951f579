// cx_chains.h — the GPU-free part of CX_SCHED_CHAIN_SCAN's set-up: the non-observed variables linked by 2-edge factors decomposed into
// disjoint simple paths (positions in path order, links, the two chain slots of every position, first / last link flags, and for
// dim > 1 the rule-table index of each link's two messages and the isolated positions).  A variable with more than two non-observed
// neighbours or a cycle is refused: the one-call result of the reference on such a graph is not a forward/backward pass
// (/root/reference/src/inference_engine.jl:575-608).  Pure host C++ over any struct H with cx_handle's host fields (cx_flatten.h).
#pragma once

#include "cx_flatten.h"

namespace cx {
namespace chains {

struct Out {
    std::vector<int32_t> pos_var, skip0, skip1, link_pos, from, to, tab_fwd, tab_bwd;
    std::vector<uint8_t> head_fwd, head_bwd;
    int64_t npos_linked = 0;
};

using flat::fail_;

template <class H>
int32_t decompose(const H *h, Out &out, std::string &err) {
    auto &pos_var = out.pos_var; auto &skip0 = out.skip0; auto &skip1 = out.skip1; auto &link_pos = out.link_pos; auto &from = out.from; auto &to = out.to;
    auto &head_fwd = out.head_fwd; auto &head_bwd = out.head_bwd; auto &tab_fwd = out.tab_fwd; auto &tab_bwd = out.tab_bwd;
    const int64_t nv = h->nv;
    std::vector<int32_t> slot_var(h->nslots, -1);
    for (int64_t e = 0; e < h->ne; e++) slot_var[flat::slot_of_edge_t(h, e)] = h->edge_var[e];
    auto is_free = [&](int32_t v) { return !(h->vinfo[v] & (kClamped | kGhost)) && (h->var_off[v + 1] - h->var_off[v]) >= 2; };
    std::vector<int32_t> dyn(2 * nv, -1);
    std::vector<uint8_t> ndyn(nv, 0);
    for (int64_t e = 0; e < h->ne; e++) {
        const int32_t v = h->edge_var[e];
        if (!is_free(v)) continue;
        const int32_t s = flat::slot_of_edge_t(h, e), p = h->partner[s];
        if (p < 0 || !is_free(slot_var[p])) continue;
        if (ndyn[v] == 2)
            return fail_(err, CX_ERR_UNSUPPORTED, "chain-scan schedule: variable " + std::to_string(h->var_ids[v]) + " has more than two non-observed neighbours (the graph is not a union of chains)");
        dyn[2 * v + ndyn[v]++] = s;
    }
    std::vector<uint8_t> visited(nv, 0);
    for (int64_t v0 = 0; v0 < nv; v0++) {
        if (!is_free((int32_t)v0) || visited[v0] || ndyn[v0] != 1) continue;
        int32_t cur = (int32_t)v0, incoming = -1;
        bool first = true;
        while (true) {
            visited[cur] = 1;
            int32_t out = -1;
            for (int k = 0; k < ndyn[cur]; k++) if (dyn[2 * cur + k] != incoming) out = dyn[2 * cur + k];
            pos_var.push_back(cur); skip0.push_back(incoming); skip1.push_back(out);
            if (out < 0) break;
            link_pos.push_back((int32_t)pos_var.size() - 1); from.push_back(out); to.push_back(h->partner[out]);
            head_fwd.push_back(first ? 1 : 0); head_bwd.push_back(0);
            first = false;
            incoming = h->partner[out];
            cur = slot_var[incoming];
            if (visited[cur]) return fail_(err, CX_ERR_UNSUPPORTED, "chain-scan schedule: the graph has a cycle");
        }
        if (!head_bwd.empty()) head_bwd.back() = 1;
    }
    for (int64_t v = 0; v < nv; v++)
        if (is_free((int32_t)v) && !visited[v] && ndyn[v] == 2)
            return fail_(err, CX_ERR_UNSUPPORTED, "chain-scan schedule: the graph has a cycle through variable " + std::to_string(h->var_ids[v]));
    if (h->cfg.dim > 1) {
        // a non-observed variable with no non-observed neighbour is a path of one position and no link: the side pass of
        // cx_mvchain.hip writes its marginal (the scalar path leaves such variables to its general variable phase)
        // (so is a non-observed variable of degree 1 whose one factor leads to no chain variable: a chain of one state)
        out.npos_linked = (int64_t)pos_var.size();
        for (int64_t v = 0; v < nv; v++) {
            if (visited[v] || (h->vinfo[v] & (kClamped | kGhost))) continue;
            const int32_t deg = h->var_off[v + 1] - h->var_off[v];
            bool alone = is_free((int32_t)v) && ndyn[v] == 0;
            if (deg == 1) {
                const int32_t pp = h->partner[flat::slot_of_edge_t(h, h->var_off[v])];
                alone = pp < 0 || !is_free(slot_var[pp]);
            }
            if (alone) { visited[v] = 1; pos_var.push_back((int32_t)v); skip0.push_back(-1); skip1.push_back(-1); }
        }
        // rule-table index of each link's two messages: spdir of the SENDING slot (2 * parameter set + direction)
        for (size_t l = 0; l < from.size(); l++) { tab_fwd.push_back(h->spdir[from[l]]); tab_bwd.push_back(h->spdir[to[l]]); }
    }
    return CX_OK;
}

}  // namespace chains
}  // namespace cx
