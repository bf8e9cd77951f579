// cx_api_sweep.hip — cx_sweep and what it needs: the chain decomposition of CX_SCHED_CHAIN_SCAN, the composed maps of a
// chain partition, the schedules' launch sequences, residuals.

#include "cx_host.h"
#include "cx_chains.h"
#include "cx_tree_plan.h"

using namespace cxh;

namespace cxh {

// In the fused schedule without materialisation the variable→factor messages of the last sweep exist only as
// "leave-one-out of the sweep's input buffer", which is retained in d_f2v_alt: recompute them on demand.
int32_t ensure_v2f(cx_handle *h) {
    if (!h->v2f_stale) return CX_OK;
    const double2 *src = h->d_f2v_alt ? h->d_f2v_alt : h->d_f2v;
    cx::launch_var_to_factor(h, src, false);
    cx::launch_big_var_to_factor(h, src, false);
    CX_HIP(h, hipGetLastError());
    h->v2f_stale = false;
    return CX_OK;
}

// ---- chain decomposition for CX_SCHED_CHAIN_SCAN -------------------------------------------------------------------
// Free variables (not observed, not ghosts, degree >= 2) linked by 2-edge factors must form disjoint simple paths.
int32_t build_chains(cx_handle *h) {
    if (!h->chains_dirty) return CX_OK;
    if (h->cfg.dim > 1) { int32_t rc0 = mv_ensure_chain_msgs(h); if (rc0 != CX_OK) return rc0; }   // before the old chains' buffers go
    try {
        cx::chains::Out co;
        {
            std::string cerr;
            const int32_t crc = cx::chains::decompose(h, co, cerr);
            if (crc != CX_OK) return fail(h, crc, cerr);
        }
        h->chain_npos_linked = co.npos_linked;
        auto &pos_var = co.pos_var; auto &skip0 = co.skip0; auto &skip1 = co.skip1; auto &link_pos = co.link_pos; auto &from = co.from; auto &to = co.to;
        auto &head_fwd = co.head_fwd; auto &head_bwd = co.head_bwd; auto &tab_fwd = co.tab_fwd; auto &tab_bwd = co.tab_bwd;
        const int64_t nv = h->nv;
        for (void *p : {(void *)h->d_chain_pos_var, (void *)h->d_chain_skip0, (void *)h->d_chain_skip1, (void *)h->d_chain_link_pos,
                        (void *)h->d_chain_from, (void *)h->d_chain_to, (void *)h->d_chain_head_fwd, (void *)h->d_chain_head_bwd,
                        (void *)h->d_chain_side, h->d_chain_totals, (void *)h->d_chain_tab_fwd, (void *)h->d_chain_tab_bwd, (void *)h->d_mvc_side,
                        (void *)h->d_mvc_totals, (void *)h->d_mvc_side_l, (void *)h->d_mvc_alpha, (void *)h->d_mvc_gamma, (void *)h->d_mvc_prefix, (void *)h->d_mvc_wave_carry, (void *)h->d_mvc_block, (void *)h->d_mvc_var_link}) if (p) (void)hipFree(p);
        h->d_chain_tab_fwd = h->d_chain_tab_bwd = nullptr; h->d_mvc_side = h->d_mvc_totals = nullptr;
        h->d_mvc_side_l = h->d_mvc_alpha = h->d_mvc_gamma = h->d_mvc_prefix = h->d_mvc_wave_carry = h->d_mvc_block = nullptr; h->d_mvc_var_link = nullptr;
        h->chain_npos = (int64_t)pos_var.size(); h->chain_nlinks = (int64_t)link_pos.size();
        h->chain_side_dirty = true; h->chain_linkpar_dirty = true;
        h->chain_pos0 = link_pos.empty() ? -1 : link_pos[0];      // one path: positions follow the links
        for (size_t l = 0; l < link_pos.size() && h->chain_pos0 >= 0; l++) if (link_pos[l] != h->chain_pos0 + (int64_t)l) h->chain_pos0 = -1;
        int64_t n_readers = 0;   // variables that read factor→variable messages: everything but observed variables and ghosts
        for (int64_t v = 0; v < nv; v++) n_readers += (h->vinfo[v] & (cx::kClamped | cx::kGhost)) ? 0 : 1;
        h->chain_covers_all = n_readers == h->chain_npos;
        int32_t rc;
        if ((rc = dev_upload(h, &h->d_chain_pos_var, pos_var)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_chain_skip0, skip0)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_chain_skip1, skip1)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_chain_link_pos, link_pos)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_chain_from, from)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_chain_to, to)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_chain_head_fwd, head_fwd)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_chain_head_bwd, head_bwd)) != CX_OK) return rc;
        if (cx::is_mfma_dim(h->cfg.dim)) {
            // wide messages: a plan of compositions and walks (cx_chain64_plan.h) instead of the per-thread scan of dim 2..4
            h->d_chain_side = nullptr; h->d_chain_totals = nullptr;
            if ((rc = cx::chain64_build(h, pos_var, skip0, skip1, link_pos, from, to, head_fwd, head_bwd, tab_fwd, tab_bwd)) != CX_OK) return rc;
        } else if (h->cfg.dim > 1) {
            h->d_chain_side = nullptr; h->d_chain_totals = nullptr;
            if ((rc = dev_upload(h, &h->d_chain_tab_fwd, tab_fwd)) != CX_OK) return rc;
            if ((rc = dev_upload(h, &h->d_chain_tab_bwd, tab_bwd)) != CX_OK) return rc;
            h->mvc_K = cx::mvc_links_per_thread(h->chain_nlinks);
            const int64_t il = cx::mvc_ntiles(h->chain_nlinks, h->mvc_K) * cx::kBlock * h->mvc_K;     // interleaved arrays, padded to whole tiles
            if ((rc = dev_alloc(h, &h->d_mvc_side, h->nc * h->chain_npos)) != CX_OK) return rc;
            if ((rc = dev_alloc(h, &h->d_mvc_side_l, h->ncs * il)) != CX_OK) return rc;
            if ((rc = dev_alloc(h, &h->d_mvc_alpha, h->ncs * il)) != CX_OK) return rc;
            if ((rc = dev_alloc(h, &h->d_mvc_gamma, h->ncs * il)) != CX_OK) return rc;
            if ((rc = dev_alloc(h, &h->d_mvc_prefix, (int64_t)cx::mvc_prefix_doubles(h->cfg.dim, h->chain_nlinks, h->mvc_K))) != CX_OK) return rc;
            if ((rc = dev_alloc(h, &h->d_mvc_wave_carry, (int64_t)cx::mvc_wave_carry_doubles(h->cfg.dim, h->chain_nlinks, h->mvc_K))) != CX_OK) return rc;
            if ((rc = dev_alloc(h, &h->d_mvc_block, (int64_t)cx::mvc_totals_doubles(h->cfg.dim, 1, 1))) != CX_OK) return rc;      // two maps
            if ((rc = dev_alloc(h, &h->d_mvc_totals, (int64_t)cx::mvc_totals_doubles(h->cfg.dim, h->chain_nlinks, h->mvc_K))) != CX_OK) return rc;
            {   // which link ends (on its right) in a variable: what a marginal on demand is formed from (cx_mvchain.hip: k_mvc_marg_gather)
                std::vector<int32_t> var_link(nv, -1);
                for (size_t l = 0; l < link_pos.size(); l++) var_link[pos_var[link_pos[l] + 1]] = (int32_t)l;
                if ((rc = dev_upload(h, &h->d_mvc_var_link, var_link)) != CX_OK) return rc;
            }
        } else {
            if ((rc = dev_alloc(h, &h->d_chain_side, h->chain_npos)) != CX_OK) return rc;
            char *tot = nullptr;
            if ((rc = dev_alloc(h, &tot, (int64_t)cx::chain_total_bytes(h->chain_nlinks))) != CX_OK) return rc;
            h->d_chain_totals = tot;
        }
        CX_HIP(h, hipStreamSynchronize(h->stream));
        h->chains_dirty = false;
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "chain decomposition: host allocation failed"); }
}


// ---- CX_SCHED_TREE: the stages of cx_tree_plan.h, built once per set of observed variables ----------------------------
void batch_graph_drop(cx_handle *h) {
    for (auto &g : h->batch_graph) { if (g.exec) (void)hipGraphExecDestroy(g.exec); g = cx_handle::BatchGraph(); }
    h->batch_epoch++;
}

void tree_graph_drop(cx_handle *h) {
    if (h->tree_exec) { (void)hipGraphExecDestroy(h->tree_exec); h->tree_exec = nullptr; }
}

int32_t build_tree(cx_handle *h) {
    if (!h->tree_dirty) return CX_OK;
    try {
        cx::treeplan::Out plan;
        std::string terr;
        const int32_t rc = cx::treeplan::build(h, plan, terr);
        if (rc != CX_OK) return fail(h, rc, terr);
        for (void *p : {(void *)h->d_tree_rec, (void *)h->d_tree_kary}) if (p) (void)hipFree(p);
        h->d_tree_rec = h->d_tree_kary = nullptr;
        tree_graph_drop(h);
        int32_t rc2;
        // Heavy paths (cx_tree_plan.h: build_hp; scalar messages): the same exact sweep as O(log n) rounds of scans and item stages.  Taken
        // when it needs fewer launches than the level schedule's 2 x depth + 1 — chains with side branches, not bushy trees.
        // CX_TREE_HP=0 / 1: never / whenever the graph allows (A/B, tests).
        h->tree_hp = false;
        const bool hp_mv = h->cfg.dim >= 2 && h->cfg.dim <= 4 && h->big_vars.empty() && h->n_kary == 0;      // (dim 2 .. 4: the scans of cx_mvchain.hip per light depth; their side sums take degree <= 8: a graph with a hub runs level by level)
        const bool hp64 = cx::is_mfma_dim(h->cfg.dim);                          // (dim 64: a plan of compositions and walks per light depth, cx_mv64chain.hip)
        cx::chain64_tree_free(h);
        h->tree_c64_up.clear(); h->tree_c64_final.clear();
        if ((h->cfg.dim == 1 && h->cfg.family != CX_FAMILY_NATURAL2) || hp_mv || hp64) {
            const char *hp_e = std::getenv("CX_TREE_HP");
            const int hp_env = hp_e ? std::atoi(hp_e) : -1;
            cx::treeplan::HP hp;
            if (hp_env != 0) {
                const int32_t rch = cx::treeplan::build_hp(h, hp, terr);
                if (rch != CX_OK) return fail(h, rch, terr);
            }
            bool take = hp_env != 0 && !hp.link_pos.empty() && (hp_env > 0 || hp.launches < 2 * (int64_t)plan.depth + 1);
            if (take && hp64) {
                // the scans' launches are known once the plans exist; a variable with more than three inputs besides its links
                // (cx_chain64_plan.h) leaves the tree to the level schedule
                h->tree_c64_up.assign(hp.levels, -1); h->tree_c64_final.assign(hp.levels, -1);
                int64_t launches = 0;
                for (size_t i = 0; i + 1 < hp.steps.size() && take; i += 2) {
                    const int32_t kind = hp.steps[i], L = hp.steps[i + 1];
                    if (kind == 0) { launches++; continue; }
                    const int64_t p0 = hp.pos_off[L], p1 = hp.pos_off[L + 1], l0 = hp.link_off[L], l1 = hp.link_off[L + 1];
                    auto cut = [](const auto &v, int64_t a, int64_t b) { return std::decay_t<decltype(v)>(v.begin() + a, v.begin() + b); };
                    std::vector<int32_t> lp = cut(hp.link_pos, l0, l1), tf(l1 - l0), tb(l1 - l0);
                    for (auto &x : lp) x -= (int32_t)p0;
                    for (int64_t l = l0; l < l1; l++) { tf[l - l0] = h->spdir[hp.from[l]]; tb[l - l0] = h->spdir[hp.to[l]]; }
                    int idx = -1;
                    const int32_t rcb = cx::chain64_tree_build(h, &idx, cut(hp.pos_var, p0, p1), cut(hp.skip0, p0, p1), cut(kind == 1 ? hp.skip1_up : hp.skip1_down, p0, p1),
                                                               lp, cut(hp.from, l0, l1), cut(hp.to, l0, l1), cut(hp.head_fwd, l0, l1), cut(hp.head_bwd, l0, l1), tf, tb);
                    if (rcb == CX_ERR_UNSUPPORTED) { take = false; break; }
                    if (rcb != CX_OK) return rcb;
                    (kind == 1 ? h->tree_c64_up : h->tree_c64_final)[L] = idx;
                    launches += cx::chain64_tree_launches(h, idx);
                }
                if (take && hp_env <= 0 && launches >= 2 * (int64_t)plan.depth + 1) take = false;
                if (!take) { cx::chain64_tree_free(h); h->tree_c64_up.clear(); h->tree_c64_final.clear(); }
                else hp.launches = launches;
            }
            if (take && hp64) {
                h->tree_hp = true;
                h->tree_hp_steps = hp.steps; h->tree_hp_pos_off = hp.pos_off; h->tree_hp_link_off = hp.link_off; h->tree_hp_marginal_stage = hp.marginal_stage;
                const int64_t hs[4] = {hp.levels, hp.n_paths, hp.n_single, hp.launches};
                std::memcpy(h->tree_hp_stats, hs, sizeof hs);
                plan.rec = std::move(hp.rec); plan.stage_off = std::move(hp.stage_off); plan.kary = std::move(hp.kary); plan.kary_off = std::move(hp.kary_off);
            } else if (take) {
                for (void *p : {(void *)h->d_chain_pos_var, (void *)h->d_chain_skip0, (void *)h->d_chain_skip1, (void *)h->d_tree_skip1_down, (void *)h->d_chain_link_pos,
                                (void *)h->d_chain_from, (void *)h->d_chain_to, (void *)h->d_chain_head_fwd, (void *)h->d_chain_head_bwd, (void *)h->d_chain_side,
                                h->d_chain_totals}) if (p) (void)hipFree(p);
                h->d_chain_pos_var = h->d_chain_skip0 = h->d_chain_skip1 = h->d_tree_skip1_down = h->d_chain_link_pos = h->d_chain_from = h->d_chain_to = nullptr;
                h->d_chain_head_fwd = h->d_chain_head_bwd = nullptr; h->d_chain_side = nullptr; h->d_chain_totals = nullptr;
                if ((rc2 = dev_upload(h, &h->d_chain_pos_var, hp.pos_var)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_chain_skip0, hp.skip0)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_chain_skip1, hp.skip1_up)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_tree_skip1_down, hp.skip1_down)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_chain_link_pos, hp.link_pos)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_chain_from, hp.from)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_chain_to, hp.to)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_chain_head_fwd, hp.head_fwd)) != CX_OK) return rc2;
                if ((rc2 = dev_upload(h, &h->d_chain_head_bwd, hp.head_bwd)) != CX_OK) return rc2;
                h->tree_hp_K.clear(); h->tree_hp_npos = (int64_t)hp.pos_var.size();
                if (!hp_mv) {
                    if ((rc2 = dev_alloc(h, &h->d_chain_side, (int64_t)hp.pos_var.size())) != CX_OK) return rc2;
                    char *tot = nullptr;
                    if ((rc2 = dev_alloc(h, &tot, (int64_t)cx::chain_total_bytes((int64_t)hp.link_pos.size()))) != CX_OK) return rc2;
                    h->d_chain_totals = tot;
                } else {
                    for (void *p : {(void *)h->d_chain_tab_fwd, (void *)h->d_chain_tab_bwd, (void *)h->d_mvc_side, (void *)h->d_mvc_totals, (void *)h->d_mvc_side_l,
                                    (void *)h->d_mvc_alpha, (void *)h->d_mvc_gamma, (void *)h->d_mvc_prefix, (void *)h->d_mvc_wave_carry}) if (p) (void)hipFree(p);
                    h->d_chain_tab_fwd = h->d_chain_tab_bwd = nullptr; h->d_mvc_side = h->d_mvc_totals = nullptr;
                    h->d_mvc_side_l = h->d_mvc_alpha = h->d_mvc_gamma = h->d_mvc_prefix = h->d_mvc_wave_carry = nullptr;
                    std::vector<int32_t> tab_fwd(hp.from.size()), tab_bwd(hp.from.size());
                    for (size_t l = 0; l < hp.from.size(); l++) { tab_fwd[l] = h->spdir[hp.from[l]]; tab_bwd[l] = h->spdir[hp.to[l]]; }      // the SENDING slot's table (cx_chains.h)
                    if ((rc2 = dev_upload(h, &h->d_chain_tab_fwd, tab_fwd)) != CX_OK) return rc2;
                    if ((rc2 = dev_upload(h, &h->d_chain_tab_bwd, tab_bwd)) != CX_OK) return rc2;
                    int64_t il = 0, pre = 0, wc = 0, tot = 0;
                    for (int32_t L = 0; L < hp.levels; L++) {
                        const int64_t nl = hp.link_off[L + 1] - hp.link_off[L];
                        const int K = cx::mvc_links_per_thread(nl);
                        h->tree_hp_K.push_back(K);
                        il = std::max<int64_t>(il, cx::mvc_ntiles(nl, K) * cx::kBlock * K);
                        pre = std::max<int64_t>(pre, (int64_t)cx::mvc_prefix_doubles(h->cfg.dim, nl, K));
                        wc = std::max<int64_t>(wc, (int64_t)cx::mvc_wave_carry_doubles(h->cfg.dim, nl, K));
                        tot = std::max<int64_t>(tot, (int64_t)cx::mvc_totals_doubles(h->cfg.dim, nl, K));
                    }
                    if ((rc2 = dev_alloc(h, &h->d_mvc_side, h->nc * (int64_t)hp.pos_var.size())) != CX_OK) return rc2;
                    if ((rc2 = dev_alloc(h, &h->d_mvc_side_l, h->ncs * il)) != CX_OK) return rc2;
                    if ((rc2 = dev_alloc(h, &h->d_mvc_alpha, h->ncs * il)) != CX_OK) return rc2;
                    if ((rc2 = dev_alloc(h, &h->d_mvc_gamma, h->ncs * il)) != CX_OK) return rc2;
                    if ((rc2 = dev_alloc(h, &h->d_mvc_prefix, pre)) != CX_OK) return rc2;
                    if ((rc2 = dev_alloc(h, &h->d_mvc_wave_carry, wc)) != CX_OK) return rc2;
                    if ((rc2 = dev_alloc(h, &h->d_mvc_totals, tot)) != CX_OK) return rc2;
                }
                for (void *p : {(void *)h->d_tree_a, (void *)h->d_tree_b}) if (p) (void)hipFree(p);
                h->d_tree_a = h->d_tree_b = nullptr;
                h->tree_hp_kary_links = hp.n_kary_links;
                if (hp.n_kary_links > 0 && !h->any_linear) {      // no (a, b) per slot on this graph: the links' own, 1 and 0 everywhere else
                    const std::vector<double> one((size_t)h->nslots, 1.0), zero((size_t)h->nslots, 0.0);
                    if ((rc2 = dev_upload(h, &h->d_tree_a, one)) != CX_OK) return rc2;
                    if ((rc2 = dev_upload(h, &h->d_tree_b, zero)) != CX_OK) return rc2;
                }
                h->tree_hp = true;
                h->tree_hp_steps = hp.steps; h->tree_hp_pos_off = hp.pos_off; h->tree_hp_link_off = hp.link_off; h->tree_hp_marginal_stage = hp.marginal_stage;
                const int64_t hs[4] = {hp.levels, hp.n_paths, hp.n_single, hp.launches};
                std::memcpy(h->tree_hp_stats, hs, sizeof hs);
                // the item stages of the heavy-path plan take the place of the level schedule's
                plan.rec = std::move(hp.rec); plan.stage_off = std::move(hp.stage_off); plan.kary = std::move(hp.kary); plan.kary_off = std::move(hp.kary_off);
            }
        }
        if (!h->tree_hp) std::memset(h->tree_hp_stats, 0, sizeof h->tree_hp_stats);
        // a stage's messages out of factors with more than two edges ride in the same list as its other items (cx_batch.hip:
        // kItemKaryEntry): one launch per stage
        std::vector<int32_t> rec;
        std::vector<int64_t> off(1, 0), koff(plan.kary_off.size(), 0);
        rec.reserve(plan.rec.size() + 5 * plan.kary.size());
        if (cx::is_mfma_dim(h->cfg.dim)) {
            // dim 64: a factor→variable message is ONE rule record of cx_mv64w.hip — the sending slot, the other slots of the sending
            // variable (the rule sums them itself: no stored variable→factor message), the rule table, the destination — so only the
            // plan's factor→variable items become work; messages out of observed variables are constants (k_point64 at data injection),
            // marginals of dim 64 are formed when read
            std::vector<int32_t> slot_var(h->nslots, -1), pre_s, pre_v;
            h->tree_pre_off.assign(1, 0);
            for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
            for (size_t st = 0; st + 1 < plan.stage_off.size(); st++) {
                for (int64_t i = plan.stage_off[st]; i < plan.stage_off[st + 1]; i++) {
                    if (plan.rec[5 * i] != CX_ITEM_MESSAGE_TO_VARIABLE) continue;
                    const int32_t dst = plan.rec[5 * i + 1], sl = h->partner[dst];
                    if (sl < 0) continue;
                    const int32_t u = slot_var[sl];
                    if (h->vinfo[u] & cx::kClamped) continue;
                    const int32_t deg = h->var_off[u + 1] - h->var_off[u];
                    int32_t others[3] = {-1, -1, -1};
                    int n_others = 0;
                    const int32_t stride = cx::slot_stride(h, u);
                    for (int32_t j = 0; j < deg; j++) {
                        const int32_t sj = h->vbase[u] + j * stride;
                        if (sj != sl && n_others < 3) others[n_others++] = sj;
                    }
                    if (deg > 4) {      // a sender of degree 5 or more: its other messages (final by now) are summed by k_v2f64 first, the rule reads the sum
                        pre_s.push_back(sl); pre_v.push_back(u);
                        rec.insert(rec.end(), {sl, -1, -1, -1, h->spdir[sl], dst, 1, 0});
                    } else
                    rec.insert(rec.end(), {sl, others[0], others[1], others[2], h->spdir[sl], dst, deg < 2 ? 1 : 0, 0});
                }
                off.push_back((int64_t)rec.size() / 8);
                h->tree_pre_off.push_back((int64_t)pre_s.size());
            }
            for (void *p : {(void *)h->d_tree_pre_slots, (void *)h->d_tree_pre_vars}) if (p) (void)hipFree(p);
            h->d_tree_pre_slots = h->d_tree_pre_vars = nullptr;
            if ((rc2 = dev_upload(h, &h->d_tree_pre_slots, pre_s)) != CX_OK) return rc2;
            if ((rc2 = dev_upload(h, &h->d_tree_pre_vars, pre_v)) != CX_OK) return rc2;
        } else
        for (size_t st = 0; st + 1 < plan.stage_off.size(); st++) {
            rec.insert(rec.end(), plan.rec.begin() + 5 * plan.stage_off[st], plan.rec.begin() + 5 * plan.stage_off[st + 1]);
            for (int64_t k = plan.kary_off[st]; k < plan.kary_off[st + 1]; k++) rec.insert(rec.end(), {32, plan.kary[k], 0, 0, 0});
            off.push_back((int64_t)rec.size() / 5);
        }
        if (!rec.empty() && (rc2 = dev_upload(h, &h->d_tree_rec, rec)) != CX_OK) return rc2;
        // (The XCD-resident cluster of the reference-order plans — cx_batch.hip: k_ref_cluster — was tried on these level plans too and
        // taken out again: a tree's levels are thin next to the roots, where one workgroup's runs cost 1 us a stage, and one huge level of
        // leaves, which wants the whole chip; the 1.09 M-edge forest took 1.24 ms on the cluster against 0.63 as launches, and no forest of
        // tools/bench_configs.py or the tests has the many stages of 1 - 16 k items the cluster wins on.)
        if (h->d_tree_stage_off) { (void)hipFree(h->d_tree_stage_off); h->d_tree_stage_off = nullptr; }
        if ((rc2 = dev_upload(h, &h->d_tree_stage_off, off)) != CX_OK) return rc2;
        CX_HIP(h, hipStreamSynchronize(h->stream));
        h->tree_stage_off = off; h->tree_kary_off = koff;
        const int64_t st[8] = {plan.depth, (int64_t)plan.stage_off.size() - 1, (int64_t)plan.rec.size() / 5, (int64_t)plan.kary.size(), plan.n_components,
                               plan.n_up, plan.n_down, plan.n_marginals};
        std::memcpy(h->tree_stats, st, sizeof st);
        h->tree_dirty = false;
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "tree schedule: host allocation failed"); }
}

// one exact sweep: every stage in order, on the handle's stream, in place (a stage's items are independent; a stage reads what
// earlier stages of this sweep and the stored constants left)
static void tree_issue(cx_handle *h) {
    if (h->tree_hp) {
        // heavy paths: item stages and scans in the plan's order.  (The final scans always write marginals — and the links'
        // variable→factor messages, which no item produces — whatever compute_marginals_in_sweep says; only the stage of the single
        // variables' marginals is left out then.)
        for (size_t i = 0; i + 1 < h->tree_hp_steps.size(); i += 2) {
            const int32_t kind = h->tree_hp_steps[i], idx = h->tree_hp_steps[i + 1];
            if (kind == 0 && cx::is_mfma_dim(h->cfg.dim)) {      // a stage of dim 64: one launch of the rule kernel over its records (marginals are formed when read)
                const int64_t n = h->tree_stage_off[idx + 1] - h->tree_stage_off[idx], np = h->tree_pre_off[idx + 1] - h->tree_pre_off[idx];
                if (np > 0) cx::mv64_launch_v2f(h, (int)np, h->d_tree_pre_slots + h->tree_pre_off[idx], h->d_tree_pre_vars + h->tree_pre_off[idx], h->d_mv_f2v);
                if (n > 0) cx::mv64_launch_rule(h, (int)n, h->d_tree_rec + 8 * h->tree_stage_off[idx], h->d_mv_f2v, h->d_mv_f2v, CX_KERNEL_BATCH);
            } else if (cx::is_mfma_dim(h->cfg.dim)) {
                const int32_t pi = (kind == 1 ? h->tree_c64_up : h->tree_c64_final)[idx];
                if (pi >= 0) (void)cx::chain64_tree_sweep(h, pi);
            } else if (kind == 0) {
                if (idx == h->tree_hp_marginal_stage && h->cfg.compute_marginals_in_sweep == 0) continue;
                const int64_t n = h->tree_stage_off[idx + 1] - h->tree_stage_off[idx];
                if (n > 0 && h->cfg.dim > 1) cx::mv_launch_batch(h, h->d_tree_rec + 5 * h->tree_stage_off[idx], n);
                else if (n > 0) cx::launch_batch(h, h->d_tree_rec + 5 * h->tree_stage_off[idx], n);
            } else if (h->cfg.dim > 1) {
                const int64_t l0 = h->tree_hp_link_off[idx];
                cx::mvc_launch_scan_range(h, h->tree_hp_npos, h->tree_hp_pos_off[idx + 1], l0, h->tree_hp_link_off[idx + 1] - l0, h->tree_hp_K[idx],
                                          kind == 1 ? h->d_chain_skip1 : h->d_tree_skip1_down, kind == 2);
            } else if (kind == 3) {
                const int64_t l0 = h->tree_hp_link_off[idx];
                cx::launch_kary_link_params(h, l0, h->tree_hp_link_off[idx + 1] - l0, h->d_tree_a ? h->d_tree_a : h->d_a, h->d_tree_b ? h->d_tree_b : h->d_b);
            } else {
                const int64_t p0 = h->tree_hp_pos_off[idx], l0 = h->tree_hp_link_off[idx];
                cx::launch_chain_scan_range(h, h->d_f2v, p0, h->tree_hp_pos_off[idx + 1] - p0, l0, h->tree_hp_link_off[idx + 1] - l0,
                                            kind == 1 ? h->d_chain_skip1 : h->d_tree_skip1_down, kind == 2);
            }
        }
        return;
    }
    size_t ns = h->tree_stage_off.empty() ? 0 : h->tree_stage_off.size() - 1;
    if (ns > 0 && h->cfg.compute_marginals_in_sweep == 0 && !cx::is_mfma_dim(h->cfg.dim)) ns--;      // the last stage is the marginals (the flag is fixed per handle)
    // CX_TREE_RUNS=0: every stage a launch of its own (A/B)
    static const bool runs = [] { const char *e = std::getenv("CX_TREE_RUNS"); return !(e && e[0] == '0'); }();
    for (size_t s = 0; s < ns;) {
        const int64_t n = h->tree_stage_off[s + 1] - h->tree_stage_off[s], nk = h->tree_kary_off[s + 1] - h->tree_kary_off[s];
        if (cx::is_mfma_dim(h->cfg.dim)) {
            const int64_t np = h->tree_pre_off[s + 1] - h->tree_pre_off[s];
            if (np > 0) cx::mv64_launch_v2f(h, (int)np, h->d_tree_pre_slots + h->tree_pre_off[s], h->d_tree_pre_vars + h->tree_pre_off[s], h->d_mv_f2v);
            if (n > 0) cx::mv64_launch_rule(h, (int)n, h->d_tree_rec + 8 * h->tree_stage_off[s], h->d_mv_f2v, h->d_mv_f2v, CX_KERNEL_BATCH);
            s++; continue;
        }
        if (h->cfg.dim > 1) { if (n > 0) cx::mv_launch_batch(h, h->d_tree_rec + 5 * h->tree_stage_off[s], n); s++; continue; }
        // dim 1: consecutive thin stages (the levels next to the roots) leave as ONE launch of one workgroup (cx_batch.hip: k_batch_run)
        size_t e = s;
        while (runs && e < ns && h->tree_stage_off[e + 1] - h->tree_stage_off[e] <= 1024) e++;
        if (e >= s + 2) { cx::launch_batch_run(h, h->d_tree_rec, h->d_tree_stage_off, (int)s, (int)e); s = e; continue; }
        if (n > 0) cx::launch_batch(h, h->d_tree_rec + 5 * h->tree_stage_off[s], n);
        if (nk > 0) cx::launch_kary_items(h, h->d_tree_kary + h->tree_kary_off[s], nk);
        s++;
    }
}

// The stages are hundreds of small launches whose arguments never change between sweeps (device-resident lists, the handle's
// buffers): they are captured ONCE into a HIP graph — on a stream of the handle's own, so that the caller's stream may be the null
// stream — and a sweep is one hipGraphLaunch on the caller's stream.  Measured on the 1.09 M-edge forest of tools/bench_configs.py
// (206 launches): the sweep was bound by the host's launch rate.  CX_TREE_GRAPH=0: plain launches (A/B); a refused capture or
// instantiation also falls back to them, for good.
int32_t tree_sweep(cx_handle *h) {
    static const bool graphs = [] { const char *e = std::getenv("CX_TREE_GRAPH"); return !(e && e[0] == '0'); }();
    if (graphs && !h->tree_graph_failed && !h->profiling && !h->tree_exec) {
        hipError_t e = hipSuccess;
        if (!h->tree_capture_stream) e = hipStreamCreateWithFlags(&h->tree_capture_stream, hipStreamNonBlocking);
        hipGraph_t g = nullptr;
        if (e == hipSuccess) e = hipStreamBeginCapture(h->tree_capture_stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            hipStream_t user = h->stream;
            h->stream = h->tree_capture_stream;
            tree_issue(h);
            h->stream = user;
            e = hipStreamEndCapture(h->tree_capture_stream, &g);
        }
        if (e == hipSuccess && g) e = hipGraphInstantiate(&h->tree_exec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (e != hipSuccess || !h->tree_exec) { (void)hipGetLastError(); h->tree_exec = nullptr; h->tree_graph_failed = true; }
    }
    if (h->tree_exec && !h->profiling) {
        if (hipGraphLaunch(h->tree_exec, h->stream) == hipSuccess) return CX_OK;
        (void)hipGetLastError();
        tree_graph_drop(h);
        h->tree_graph_failed = true;
    }
    tree_issue(h);
    return CX_OK;
}

// ---- the sweep ----------------------------------------------------------------------------------------------------
void sweep_main(cx_handle *h, bool skip_ghosts) {
    const bool marg = h->cfg.compute_marginals_in_sweep != 0;
    if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) {
        // messages out of observed leaves (data) into the chains: by the scan's side pass when every free variable is on a
        // chain, by a factor phase over all slots otherwise (free variables off the chains need theirs too)
        if (!h->chain_covers_all) cx::launch_factor_to_var(h, h->d_v2f, h->d_f2v);
        // When every reader of factor→variable messages sits on a chain and nobody asked for stored variable→factor messages,
        // the scan's second kernel writes the marginals itself and the variable phase is not launched (variable→factor messages
        // are recomputed from the stored messages on demand: ensure_v2f).  Variables off the chains — observed ones, stand-ins —
        // have marginals that depend on stored messages only: a full variable phase after those were set, none otherwise.
        const bool fast = h->chain_covers_all && h->big_vars.empty() && h->cfg.materialize_messages_to_factor == 0 && !h->offchain_marg_dirty;
        const int form = marg ? (h->d_split_mean ? 3 : (h->cfg.family == CX_FAMILY_NATURAL2 ? 2 : 1)) : 0;
        h->split_marg_written = fast && form == 3;
        cx::launch_chain_scan(h, h->d_f2v, h->chain_covers_all, fast ? form : 0, h->chain_v2f_from_scan);   // all forward and backward chain messages
        if (fast && marg) {
            h->v2f_stale = true;
        } else {
            cx::launch_var_to_factor(h, h->d_f2v, marg);       // every variable→factor message + marginals
            cx::launch_big_var_to_factor(h, h->d_f2v, marg);
            h->v2f_stale = false;
            if (marg) h->offchain_marg_dirty = false;
        }
    } else if (h->cfg.schedule == CX_SCHED_FLOODING) {
        cx::launch_var_to_factor(h, h->d_f2v, marg);
        cx::launch_big_var_to_factor(h, h->d_f2v, marg);
    } else {
        // a factor with more than two edges reads variable→factor messages that other threads compute: such a graph stores them
        // every sweep and runs the factors' own kernel behind the variable phase, into the same output buffer (cx_kary.hip)
        const bool store = h->cfg.materialize_messages_to_factor != 0 || h->n_kary > 0;
        cx::launch_fused(h, h->d_f2v, h->d_f2v_alt, marg, store, skip_ghosts);
        if (!h->big_vars.empty()) {
            cx::launch_big_var_to_factor(h, h->d_f2v, marg);
            cx::launch_push_slots(h, h->d_big_slots, (int64_t)h->big_slots.size(), h->d_f2v_alt, CX_KERNEL_BIG_VAR);
        }
        cx::launch_kary(h, h->d_v2f, h->d_f2v_alt);
    }
}

void sweep_finish(cx_handle *h) {
    if (h->cfg.schedule == CX_SCHED_FLOODING) {
        cx::launch_factor_to_var(h, h->d_v2f, h->d_f2v);
        cx::launch_kary(h, h->d_v2f, h->d_f2v);
    } else if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) {
        // nothing: the scans already produced every factor→variable message a free variable reads, from the same
        // variable→factor messages the variable phase just wrote (a factor phase here would only re-derive them)
    } else {
        std::swap(h->d_f2v, h->d_f2v_alt);
        h->v2f_stale = h->cfg.materialize_messages_to_factor == 0 && h->n_kary == 0;
    }
    h->sweeps_done++;
}

}  // namespace cxh

// ---- chain-scan partitions: the composed maps of a time block -------------------------------------------
// Host copy of cx_chain.hip's map algebra (projective-linear maps on (xi, w, 1), D normalised to 1)
namespace {
struct HLin { double e, f, g, A, B, C; int seg, pad; };
HLin hlin_compose(const HLin &first, const HLin &second) {
    if (second.seg) return second;
    HLin r;
    const double inv = 1.0 / (second.C * first.B + 1.0);
    r.A = (second.A * first.A + second.B * first.C) * inv;
    r.B = (second.A * first.B + second.B) * inv;
    r.C = (second.C * first.A + first.C) * inv;
    r.e = (second.e * first.e) * inv;
    r.f = (second.e * first.f + second.f * first.A + second.g * first.C) * inv;
    r.g = (second.e * first.g + second.f * first.B + second.g) * inv;
    r.seg = first.seg; r.pad = 0;
    return r;
}
}  // namespace


extern "C" {

int32_t cx_chain_block_maps(cx_handle *h, double *fwd6, double *bwd6, double *side_first2, double *side_last2,
                            int64_t *first_variable_id, int64_t *last_variable_id, int64_t *n_links) {
    CX_NOT_VMP(h, "cx_chain_block_maps");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_chain_block_maps: no graph");
    CX_REQUIRE(h, h->cfg.schedule == CX_SCHED_CHAIN_SCAN, CX_ERR_STATE, "cx_chain_block_maps: chain-scan handles only");
    CX_REQUIRE(h, fwd6 && bwd6 && side_first2 && side_last2, CX_ERR_INVALID_ARGUMENT, "cx_chain_block_maps: null argument");
    if (h->cfg.dim > 1) {
        try { h->chain_partition = true; return mv_chain_block_maps(h, fwd6, bwd6, side_first2, side_last2, first_variable_id, last_variable_id, n_links); }
        catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_chain_block_maps: host allocation failed"); }
    }
    int32_t rc = build_chains(h);
    if (rc != CX_OK) return rc;
    CX_REQUIRE(h, h->chain_npos >= 1 && h->chain_nlinks == h->chain_npos - 1, CX_ERR_UNSUPPORTED,
               "cx_chain_block_maps: the non-observed variables of this handle must form ONE path (a time block of a chain)");
    try {
        // side sums + tile totals only (no apply): the same kernels a sweep starts with
        h->chain_partition = true;
        int64_t ntiles = 0;
        // as in sweep_main: when variables off the chain read messages too (the stand-ins do), the leaf messages come from a
        // factor phase over all slots, otherwise from the side pass itself
        if (!h->chain_covers_all) cx::launch_factor_to_var(h, h->d_v2f, h->d_f2v);
        cx::launch_chain_totals(h, h->d_f2v, h->chain_covers_all, &ntiles);
        CX_HIP(h, hipGetLastError());
        std::vector<HLin> tot((size_t)2 * (ntiles + 1));
        if (ntiles) CX_HIP(h, hipMemcpyAsync(tot.data(), h->d_chain_totals, tot.size() * sizeof(HLin), hipMemcpyDeviceToHost, h->stream));
        double2 sf, sl;
        CX_HIP(h, hipMemcpyAsync(&sf, h->d_chain_side, 16, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipMemcpyAsync(&sl, h->d_chain_side + (h->chain_npos - 1), 16, hipMemcpyDeviceToHost, h->stream));
        int32_t pv[2] = {0, 0};
        CX_HIP(h, hipMemcpyAsync(&pv[0], h->d_chain_pos_var, 4, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipMemcpyAsync(&pv[1], h->d_chain_pos_var + (h->chain_npos - 1), 4, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        for (int dir = 0; dir < 2; dir++) {
            HLin t{1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0, 0};      // identity: a block of one variable has no link
            for (int64_t i = 0; i < ntiles; i++) t = (i == 0) ? tot[(size_t)dir * (ntiles + 1)] : hlin_compose(t, tot[(size_t)dir * (ntiles + 1) + i]);
            double *o = dir == 0 ? fwd6 : bwd6;
            o[0] = t.e; o[1] = t.f; o[2] = t.g; o[3] = t.A; o[4] = t.B; o[5] = t.C;
        }
        side_first2[0] = sf.x; side_first2[1] = sf.y; side_last2[0] = sl.x; side_last2[1] = sl.y;
        if (first_variable_id) *first_variable_id = h->var_ids[pv[0]];
        if (last_variable_id) *last_variable_id = h->var_ids[pv[1]];
        if (n_links) *n_links = h->chain_nlinks;
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_chain_block_maps: host allocation failed"); }
}

int32_t cx_tree_plan_stats(const cx_handle *h, int64_t *out8) {
    CX_REQUIRE(const_cast<cx_handle *>(h), h && out8, CX_ERR_INVALID_ARGUMENT, "cx_tree_plan_stats: null argument");
    for (int i = 0; i < 8; i++) out8[i] = h->tree_dirty ? 0 : h->tree_stats[i];
    return CX_OK;
}

int32_t cx_tree_heavy_path_stats(const cx_handle *h, int64_t *out4) {
    CX_REQUIRE(const_cast<cx_handle *>(h), h && out4, CX_ERR_INVALID_ARGUMENT, "cx_tree_heavy_path_stats: null argument");
    for (int i = 0; i < 4; i++) out4[i] = (h->tree_dirty || !h->tree_hp) ? 0 : h->tree_hp_stats[i];
    return CX_OK;
}

int32_t cx_chain_plan_stats(const cx_handle *h, int64_t *out8) {
    if (!h || !out8) return CX_ERR_INVALID_ARGUMENT;
    cx::chain64_stats(h, out8);
    return CX_OK;
}

int32_t cx_chain_scan_stats(const cx_handle *h, int64_t *out4) {
    if (!h || !out4) return CX_ERR_INVALID_ARGUMENT;
    out4[0] = h->chain_onepass_state; out4[1] = h->chain_onepass_launches; out4[2] = h->batch_graph_launches; out4[3] = 0;
    return CX_OK;
}

// Damped message passing: loopy Gaussian BP outside the walk-summable regime can oscillate; mixing every new factor→variable message
// with the one it replaces is the usual remedy.  The reference has no such knob (its rules are the user's: a user damps inside the
// rule); here the rules are the library's, so the knob is too.
int32_t cx_set_damping(cx_handle *h, double lambda) {
    CX_NOT_VMP(h, "cx_set_damping");
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "null handle");
    CX_REQUIRE(h, lambda >= 0.0 && lambda < 1.0, CX_ERR_INVALID_ARGUMENT, "cx_set_damping: 0 <= lambda < 1");
    CX_REQUIRE(h, lambda == 0.0 || h->cfg.schedule == CX_SCHED_FUSED || h->cfg.schedule == CX_SCHED_FLOODING, CX_ERR_UNSUPPORTED,
               "cx_set_damping: the fused and flooding schedules iterate to a fixed point and can be damped; the chain-scan, tree and reference-order schedules are exact or sequential passes");
    // (round 6: dim 64, and 5 .. 63 embedded in it, too: the rule's output mixed with the message it replaces, cx_mv64.hip: k_damp64)
    CX_REQUIRE(h, lambda == 0.0 || h->halo_state || (h->send_slots.empty() && h->recv_slots.empty()), CX_ERR_UNSUPPORTED,
               "cx_set_damping: not with per-sweep message halos (cx_halo_configure); state halos (cx_halo_configure_state) run plain sweeps and are damped like them");
    CX_HIP(h, hipStreamSynchronize(h->stream));
    h->damping = lambda;
    batch_graph_drop(h);
    return CX_OK;
}

int32_t cx_sweep(cx_handle *h, int32_t n_sweeps) {
    CX_NOT_VMP(h, "cx_sweep");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_sweep: no graph");
    CX_REQUIRE(h, n_sweeps >= 0, CX_ERR_INVALID_ARGUMENT, "cx_sweep: n_sweeps < 0");
    if (h->cfg.dim > 1) return mv_sweep(h, n_sweeps);
    CX_REQUIRE(h, h->halo_state || h->chain_partition || h->cfg.schedule == CX_SCHED_TREE || (h->recv_slots.empty() && h->send_slots.empty()), CX_ERR_STATE,
               "cx_sweep: this handle holds a partition (halo configured): use cx_sweep_begin / _main / _end");
    if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) { int32_t rc = build_chains(h); if (rc != CX_OK) return rc; }
    { int32_t rc = cx::kary_upload(h); if (rc != CX_OK) return rc; }      // coefficients set since the last sweep
    if (h->cfg.schedule == CX_SCHED_REFERENCE) {
        CX_REQUIRE(h, h->recv_slots.empty() && h->send_slots.empty(), CX_ERR_UNSUPPORTED, "cx_sweep: the reference-order schedule is not partitioned");
        return ref_sweep_all(h, n_sweeps);
    }
    if (h->cfg.schedule == CX_SCHED_TREE) {
        // (round 5) a forest cut at its variables: cx_halo_configure marks the far variables of the cut factors as stand-ins — constants hanging
        // off their factors, like observed variables — and the CALLER moves the boundary between exact local sweeps: the stand-ins' messages
        // with cx_set_messages, the boundary variables' with cx_get_messages (cortex.jl_amd/partition.py: TreeRegionExchange)
        int32_t rc = build_tree(h);
        if (rc != CX_OK) return rc;
        for (int32_t s = 0; s < n_sweeps; s++) { const int32_t rt = tree_sweep(h); if (rt != CX_OK) return rt; h->sweeps_done++; }
        h->v2f_stale = false;            // every variable→factor message somebody reads was stored by its stage
        CX_HIP(h, hipGetLastError());
        return CX_OK;
    }
    int32_t s = 0;
    // The sweeps between two exchanges of a deep-halo partition (cx_halo_configure_state + layers): 12 .. 32 launches of 10 us whose slice
    // ranges repeat batch after batch.  CX_HALO_GRAPH=1: the second time the same batch is asked for it is captured, from then on ONE graph
    // launch.  OFF by default: on the 1/8 strip of C4 (depth 16) the replayed graph measured 10.10 - 10.25 us per sweep against 9.90 - 9.92
    // for the plain launches, 11.74 against 11.18 with the exchange (profiles/r06_strip.md) — back-to-back launches from one thread already
    // overlap their launch latency, and a graph's kernel nodes are dispatched no closer together.  What a batch bakes in is in its key
    // (first sweep after the exchange, sweeps, the two buffers, batch_epoch); the host state the sweeps leave is applied by the same code.
    const char *bg_env = std::getenv("CX_HALO_GRAPH");      // (read per call: a test runs both forms in one process)
    const bool batch_graphs = bg_env && bg_env[0] == '1';
    cx_handle::BatchGraph *bg = nullptr;
    bool capturing = false;
    hipStream_t user_stream = h->stream;
    if (batch_graphs && s == 0 && n_sweeps >= 4 && h->halo_state && h->halo_depth > 0 && h->cfg.schedule == CX_SCHED_FUSED && h->big_vars.empty() && h->n_kary == 0 &&
        !h->profiling && h->cfg.materialize_messages_to_factor == 0) {
        uint64_t key = 0x9e3779b97f4a7c15ull;
        for (uint64_t x : {(uint64_t)h->sweeps_since_exchange, (uint64_t)n_sweeps, (uint64_t)(uintptr_t)h->d_f2v, (uint64_t)(uintptr_t)h->d_f2v_alt, h->batch_epoch, (uint64_t)(uintptr_t)h->stream})
            key = (key ^ x) * 0xbf58476d1ce4e5b9ull + (key >> 29);
        for (auto &g : h->batch_graph) if (g.key == key) bg = &g;
        if (!bg) {      // a new batch: takes the slot that is not the most recent one's
            bg = h->batch_graph[0].seen <= h->batch_graph[1].seen ? &h->batch_graph[0] : &h->batch_graph[1];
            if (bg->exec) (void)hipGraphExecDestroy(bg->exec);
            *bg = cx_handle::BatchGraph();
            bg->key = key;
        }
        bg->seen++;
        if (!bg->exec && !bg->failed && bg->seen >= 2) {
            hipError_t er = hipSuccess;
            if (!h->tree_capture_stream) er = hipStreamCreateWithFlags(&h->tree_capture_stream, hipStreamNonBlocking);
            if (er == hipSuccess) er = hipStreamBeginCapture(h->tree_capture_stream, hipStreamCaptureModeThreadLocal);
            if (er == hipSuccess) { capturing = true; h->stream = h->tree_capture_stream; }
            else { (void)hipGetLastError(); bg->failed = true; }
        }
    }
    const bool replay = bg && bg->exec && !capturing;
    for (; s < n_sweeps; s++) {
        h->run_slice0 = 0; h->run_nslices = 0;
        if (h->halo_state && h->halo_depth > 0 && h->cfg.schedule == CX_SCHED_FUSED && h->big_vars.empty()) {
            const int j = std::min(h->sweeps_since_exchange + 1, h->halo_depth);     // this is sweep j after the exchange
            const int L = h->halo_depth - j + 1;                                       // layers that have to run
            if (h->trim_hi[L] >= h->trim_lo[L]) { h->run_slice0 = h->trim_lo[L]; h->run_nslices = h->trim_hi[L] - h->trim_lo[L] + 1; }
        }
        if (!replay) sweep_main(h, false);      // (a replayed batch: the launches are in the graph, the host state below is not)
        sweep_finish(h);
        h->run_slice0 = 0; h->run_nslices = 0;
        h->sweeps_since_exchange++;
    }
    if (capturing) {
        h->stream = user_stream;
        hipGraph_t g = nullptr;
        hipError_t er = hipStreamEndCapture(h->tree_capture_stream, &g);
        if (er == hipSuccess && g) er = hipGraphInstantiate(&bg->exec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (er != hipSuccess || !bg->exec) {
            // nothing was launched while capturing and the host state has moved on: the batch cannot be repeated from here
            (void)hipGetLastError(); bg->exec = nullptr; bg->failed = true;
            return fail(h, CX_ERR_DEVICE, "cx_sweep: capturing the sweeps of a halo batch as a graph failed; set CX_HALO_GRAPH=0");
        }
    }
    if (bg && bg->exec) { CX_HIP(h, hipGraphLaunch(bg->exec, h->stream)); h->batch_graph_launches++; }
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

// Loopy graphs: sweep until the largest change of a factor→variable message over `check_every` sweeps falls below tol
// (the stopping rule a user of the reference writes around update_marginals!; the reference itself has none).
int32_t cx_sweep_until(cx_handle *h, double tol, int32_t max_sweeps, int32_t check_every, int32_t *sweeps_run, double *residual) {
    CX_NOT_VMP(h, "cx_sweep_until");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_sweep_until: no graph");
    CX_REQUIRE(h, tol >= 0 && max_sweeps >= 0 && check_every >= 1, CX_ERR_INVALID_ARGUMENT, "cx_sweep_until: tol >= 0, max_sweeps >= 0, check_every >= 1");
    double r = std::numeric_limits<double>::infinity();
    int32_t rc = cx_residual(h, &r);            // snapshot of the starting point
    if (rc != CX_OK) return rc;
    int32_t done = 0;
    r = std::numeric_limits<double>::infinity();
    while (done < max_sweeps) {
        const int32_t k = std::min(check_every, max_sweeps - done);
        if ((rc = cx_sweep(h, k)) != CX_OK) return rc;
        done += k;
        if ((rc = cx_residual(h, &r)) != CX_OK) return rc;
        if (r <= tol) break;
    }
    if (sweeps_run) *sweeps_run = done;
    if (residual) *residual = r;
    return CX_OK;
}

int32_t cx_sweep_begin(cx_handle *h) {
    CX_NOT_VMP(h, "cx_sweep_begin");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_sweep_begin: no graph");
    CX_REQUIRE(h, h->cfg.dim == 1, CX_ERR_UNSUPPORTED, "cx_sweep_begin: partitioned sweeps are implemented for dim == 1 only in this build");
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, "cx_sweep_begin: previous sweep not ended");
    CX_REQUIRE(h, h->damping == 0.0, CX_ERR_UNSUPPORTED, "cx_sweep_begin: damped sweeps (cx_set_damping) are not split into begin / main / end");
    CX_REQUIRE(h, !h->halo_state, CX_ERR_STATE, "cx_sweep_begin: the handle is configured for state halos (cx_halo_configure_state): use cx_sweep + cx_halo_state_exchange");
    CX_REQUIRE(h, h->cfg.schedule != CX_SCHED_CHAIN_SCAN && h->cfg.schedule != CX_SCHED_TREE && h->cfg.schedule != CX_SCHED_REFERENCE, CX_ERR_UNSUPPORTED, "cx_sweep_begin: the chain-scan, tree and reference-order schedules are not partitioned in this build");
    cx::launch_halo_export(h, h->d_f2v, h->stream);
    CX_HIP(h, hipGetLastError());
    h->in_sweep = true;
    return CX_OK;
}

int32_t cx_sweep_main(cx_handle *h) {
    CX_REQUIRE(h, h && h->has_graph && h->in_sweep, CX_ERR_STATE, "cx_sweep_main: call cx_sweep_begin first");
    sweep_main(h, true);
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

int32_t cx_sweep_end(cx_handle *h) {
    CX_REQUIRE(h, h && h->has_graph && h->in_sweep, CX_ERR_STATE, "cx_sweep_end: call cx_sweep_begin first");
    cx::launch_halo_import(h, h->d_f2v_alt, h->cfg.schedule == CX_SCHED_FUSED);
    sweep_finish(h);
    CX_HIP(h, hipGetLastError());
    h->in_sweep = false;
    return CX_OK;
}

int32_t cx_residual(cx_handle *h, double *out) {
    CX_NOT_VMP(h, "cx_residual");
    CX_REQUIRE(h, h && h->has_graph && out, CX_ERR_STATE, "cx_residual: no graph / null out");
    if (h->cfg.dim > 1) return mv_residual(h, out);
    if (!h->d_prev) {
        int32_t rc = dev_alloc(h, &h->d_prev, h->nslots);
        if (rc != CX_OK) return rc;
        CX_HIP(h, hipMemcpyAsync(h->d_prev, h->d_f2v, (size_t)h->nslots * 16, hipMemcpyDeviceToDevice, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        *out = std::numeric_limits<double>::infinity();
        return CX_OK;
    }
    cx::launch_residual(h, h->d_f2v, h->d_prev, h->nslots, h->d_scratch);
    std::vector<double> part(1024);
    CX_HIP(h, hipMemcpyAsync(part.data(), h->d_scratch, 1024 * 8, hipMemcpyDeviceToHost, h->stream));
    CX_HIP(h, hipMemcpyAsync(h->d_prev, h->d_f2v, (size_t)h->nslots * 16, hipMemcpyDeviceToDevice, h->stream));
    CX_HIP(h, hipStreamSynchronize(h->stream));
    double m = 0.0;
    for (double p : part) m = (p != p) ? kInf : std::max(m, p);
    *out = m;
    return CX_OK;
}

}  // extern "C"
