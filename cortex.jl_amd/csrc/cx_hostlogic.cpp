// cx_hostlogic.cpp — the GPU-free host logic of libcortex_hip.so as a plain C++ translation unit: it compiles with g++
// (no HIP), also under -fsanitize=address,undefined, and exports a small C interface for the CPU tests
// (tests/hostlogic.py).  The product does not link this file: the .hip translation units include the same headers.
//
//   cxh_plan64_*   the work plan of the chain-scan schedule for dim 64 (cx_chain64_plan.h)
//   cxh_ref_*      CX_SCHED_REFERENCE: the default resolver's wiring, the shadow of the readiness state, the recorded and levelled calls (cx_refsched.h)
//   cxh_flat_*     cx_graph_create's flattening (cx_flatten.h) and CX_SCHED_CHAIN_SCAN's chain decomposition (cx_chains.h) over a
//                  plain struct with cx_handle's host fields
#include <cstdio>
#include <cstring>
#include <new>

#include "cx_chain64_plan.h"
#include "cx_chains.h"
#include "cx_halo_plan.h"
#include "cx_tree_plan.h"
#include "cx_refsched.h"

using cx::plan64::Plan;

extern "C" {

void *cxh_plan64_create(int32_t d, int64_t npos, int64_t nlinks, const int32_t *link_pos, const int32_t *from, const int32_t *to,
                        const int32_t *tab_fwd, const int32_t *tab_bwd, const uint8_t *head_fwd, const uint8_t *head_bwd,
                        const int32_t *side, int32_t K0, int32_t fan, int64_t lanes, int32_t root, char *err, int32_t errlen) {
    try {
        cx::plan64::Input in;
        in.d = d; in.npos = npos; in.nlinks = nlinks; in.link_pos = link_pos; in.from = from; in.to = to;
        in.tab_fwd = tab_fwd; in.tab_bwd = tab_bwd; in.head_fwd = head_fwd; in.head_bwd = head_bwd; in.side = side;
        in.K0 = K0; in.fan = fan; in.lanes = lanes; in.root = root != 0;
        return new Plan(cx::plan64::build(in));
    } catch (const std::exception &e) {
        if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.what());
        return nullptr;
    }
}

void cxh_plan64_destroy(void *p) { delete (Plan *)p; }

// what: 0 n_pot, 1 n_ent, 2 children, 3 steps, 4 compose launches, 5 walk launches, 6 msg, 7 pot, 8 K0, 9 levels,
//       10 compositions per sweep, 11 rule applications per sweep, 12 paths with a root potential, 100 + i: root potential of path i
int64_t cxh_plan64_info(const void *pv, int32_t what) {
    const Plan *p = (const Plan *)pv;
    switch (what) {
    case 0: return p->n_pot;
    case 1: return p->n_ent;
    case 2: return (int64_t)p->children.size();
    case 3: return (int64_t)p->steps.size();
    case 4: return (int64_t)p->compose_launches.size();
    case 5: return (int64_t)p->walk_launches.size();
    case 6: return p->msg;
    case 7: return p->pot;
    case 8: return p->K0;
    case 9: return p->levels;
    case 10: return p->n_compositions;
    case 11: return p->n_rules;
    case 12: return (int64_t)p->root_pot.size();
    }
    if (what >= 100 && what - 100 < (int32_t)p->root_pot.size()) return p->root_pot[what - 100];
    return -1;
}

// jobs of launch `idx` (kind 0: compose, 1: walk) as rows {out, first, n}; returns the count (out may be NULL)
int64_t cxh_plan64_jobs(const void *pv, int32_t kind, int32_t idx, int64_t *out) {
    const Plan *p = (const Plan *)pv;
    const auto &L = kind == 0 ? p->compose_launches : p->walk_launches;
    if (idx < 0 || idx >= (int32_t)L.size()) return -1;
    if (out)
        for (size_t i = 0; i < L[idx].size(); i++) { out[3 * i] = L[idx][i].out; out[3 * i + 1] = L[idx][i].first; out[3 * i + 2] = L[idx][i].n; }
    return (int64_t)L[idx].size();
}

// children (what = 0) or steps (what = 1) as rows of ten int64
void cxh_plan64_records(const void *pv, int32_t what, int64_t *out) {
    const Plan *p = (const Plan *)pv;
    static_assert(sizeof(cx::plan64::Child) == 80 && sizeof(cx::plan64::Step) == 80, "ten words per record");
    if (what == 0) std::memcpy(out, p->children.data(), p->children.size() * sizeof(cx::plan64::Child));
    else std::memcpy(out, p->steps.data(), p->steps.size() * sizeof(cx::plan64::Step));
}

// ---- the flattened graph ------------------------------------------------------------------------------------------------------
struct HostGraph {                       // the host fields of cx_handle that cx_flatten.h / cx_chains.h touch, by the same names
    cx_config cfg{};
    int64_t nv = 0, nf = 0, ne = 0, nslots = 0, nslices = 0;
    std::vector<int64_t> var_ids, fac_ids, edge_fac_id;
    std::vector<int32_t> fac_kind, var_off, edge_var, vbase, slice_off, partner, big_vars, big_slots, fac_edges, spdir;
    std::vector<double> fac_params;
    std::vector<uint8_t> vinfo, lin_out_is_second, var_gamma;
    std::vector<int8_t> np_role;
    int64_t n_messages_per_sweep = 0, max_pset = -1;
    bool any_linear = false;
    int32_t big_start = 0;
    int64_t n_kary = 0;
    std::vector<int32_t> kary_slot, slot_kary, kary_pset;
    std::vector<double> kary_coef, kary_qb;
    bool kary_dirty = true;
    // deep halo (cx_halo_plan.h)
    std::vector<int32_t> trim_lo, trim_hi, send_slots;
    int own_slice_lo = 1, own_slice_hi = 0, halo_depth = 0, ipc_quiet_lo = 1, ipc_quiet_hi = 0;
    cx::flat::Out fo;
    cx::chains::Out co;
    cx::treeplan::Out to;
    cx::treeplan::HP hp;
    cx::refsched::Wiring rw;
    cx::refsched::State rs;
    cx::refsched::Call rcall;
    cx::refsched::Plan rplan;
    std::string err;
};

void *cxh_flat_create(int32_t dim, int32_t schedule, int32_t family, int64_t n_edges, const int64_t *edge_var, const int64_t *edge_fac, const int32_t *edge_role,
                      int64_t n_factors, const int64_t *factor_ids, const int32_t *factor_kind, const double *factor_params, int32_t *status, char *err, int32_t errlen) {
    HostGraph *g = new HostGraph();
    g->cfg.dim = dim; g->cfg.schedule = schedule; g->cfg.family = family;
    int32_t rc;
    try { rc = cx::flat::flatten(g, n_edges, edge_var, edge_fac, edge_role, n_factors, factor_ids, factor_kind, factor_params, g->fo, g->err); }
    catch (const std::exception &e) { rc = CX_ERR_INVALID_ARGUMENT; g->err = e.what(); }
    if (rc == CX_OK && dim > 1) g->spdir = g->fo.spdir;
    if (status) *status = rc;
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", g->err.c_str());
    return g;
}

void cxh_flat_destroy(void *p) { delete (HostGraph *)p; }

// observed variables (cx_set_messages with point-mass data marks them): what the chain decomposition excludes
void cxh_flat_clamp(void *p, int64_t n, const int64_t *variable_ids) {
    HostGraph *g = (HostGraph *)p;
    for (int64_t i = 0; i < n; i++) {
        auto it = std::lower_bound(g->var_ids.begin(), g->var_ids.end(), variable_ids[i]);
        if (it != g->var_ids.end() && *it == variable_ids[i]) g->vinfo[it - g->var_ids.begin()] |= cx::kClamped;
    }
}

int32_t cxh_flat_chains(void *p, char *err, int32_t errlen) {
    HostGraph *g = (HostGraph *)p;
    g->co = cx::chains::Out();
    std::string e;
    int32_t rc;
    try { rc = cx::chains::decompose(g, g->co, e); } catch (const std::exception &x) { rc = CX_ERR_INVALID_ARGUMENT; e = x.what(); }
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.c_str());
    return rc;
}

// CX_SCHED_TREE: the stages of one exact sweep over the forest of non-observed variables (cx_tree_plan.h)
int32_t cxh_flat_tree(void *p, char *err, int32_t errlen) {
    HostGraph *g = (HostGraph *)p;
    std::string e;
    int32_t rc;
    try { rc = cx::treeplan::build(g, g->to, e); } catch (const std::exception &x) { rc = CX_ERR_INVALID_ARGUMENT; e = x.what(); }
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.c_str());
    return rc;
}

// the same sweep over heavy paths (cx_tree_plan.h: build_hp)
int32_t cxh_flat_tree_hp(void *p, char *err, int32_t errlen) {
    HostGraph *g = (HostGraph *)p;
    std::string e;
    int32_t rc;
    try { rc = cx::treeplan::build_hp(g, g->hp, e); } catch (const std::exception &x) { rc = CX_ERR_INVALID_ARGUMENT; e = x.what(); }
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.c_str());
    return rc;
}

// ---- CX_SCHED_REFERENCE (cx_refsched.h) ----------------------------------------------------------------------------------------
int32_t cxh_ref_build(void *p, char *err, int32_t errlen) {
    HostGraph *g = (HostGraph *)p;
    std::string e;
    int32_t rc;
    try { rc = cx::refsched::build_wiring(g, g->rw, e); if (rc == CX_OK) cx::refsched::init_state(g->rw, g->rs); }
    catch (const std::exception &x) { rc = CX_ERR_INVALID_ARGUMENT; e = x.what(); }
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.c_str());
    return rc;
}

static int64_t href_edge(const HostGraph *g, int64_t var_id, int64_t fac_id) {
    auto it = std::lower_bound(g->var_ids.begin(), g->var_ids.end(), var_id);
    if (it == g->var_ids.end() || *it != var_id) return -1;
    const int64_t v = it - g->var_ids.begin();
    auto b = g->edge_fac_id.begin() + g->var_off[v], e = g->edge_fac_id.begin() + g->var_off[v + 1];
    auto jt = std::lower_bound(b, e, fac_id);
    return (jt == e || *jt != fac_id) ? -1 : (int64_t)(jt - g->edge_fac_id.begin());
}

// a user wiring: triples (signal, dependency, flags), signals as (kind CX_ITEM_*, variable id, factor id) rows of three int64
int32_t cxh_ref_wire(void *p, int64_t n, const int64_t *sig3, const int64_t *dep3, const int32_t *flags, char *err, int32_t errlen) {
    HostGraph *g = (HostGraph *)p;
    std::string e;
    int32_t rc = CX_OK;
    std::vector<int64_t> s((size_t)n), d((size_t)n);
    auto number = [&](const int64_t *t, int64_t *out) {
        if (t[0] == CX_ITEM_JOINT_MARGINAL) {
            auto it = std::lower_bound(g->fac_ids.begin(), g->fac_ids.end(), t[2]);
            if (it == g->fac_ids.end() || *it != t[2]) return false;
            *out = 2 * g->ne + g->nv + (it - g->fac_ids.begin());
            return true;
        }
        if (t[0] == CX_ITEM_INDIVIDUAL_MARGINAL) {
            auto it = std::lower_bound(g->var_ids.begin(), g->var_ids.end(), t[1]);
            if (it == g->var_ids.end() || *it != t[1]) return false;
            *out = 2 * g->ne + (it - g->var_ids.begin());
            return true;
        }
        const int64_t ed = href_edge(g, t[1], t[2]);
        if (ed < 0) return false;
        *out = t[0] == CX_ITEM_MESSAGE_TO_FACTOR ? ed : g->ne + ed;
        return true;
    };
    for (int64_t i = 0; i < n && rc == CX_OK; i++) {
        if (!number(sig3 + 3 * i, &s[i])) rc = CX_ERR_NOT_FOUND;
        d[i] = s[i];
        if (rc == CX_OK && !(flags[i] & cx::refsched::kWireDefaultVariable) && !number(dep3 + 3 * i, &d[i])) rc = CX_ERR_NOT_FOUND;
    }
    if (rc == CX_OK) {
        try { rc = cx::refsched::build_user_wiring(g, n, s.data(), d.data(), flags, g->rw, e); if (rc == CX_OK) cx::refsched::init_state(g->rw, g->rs); }
        catch (const std::exception &x) { rc = CX_ERR_INVALID_ARGUMENT; e = x.what(); }
    }
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.c_str());
    return rc;
}

// set_value! on message signals: direction CX_TO_FACTOR / CX_TO_VARIABLE
int32_t cxh_ref_set(void *p, int32_t direction, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids) {
    HostGraph *g = (HostGraph *)p;
    for (int64_t i = 0; i < n; i++) {
        const int64_t e = href_edge(g, variable_ids[i], factor_ids[i]);
        if (e < 0) return CX_ERR_NOT_FOUND;
        cx::refsched::set_value(g->rw, g->rs, direction == CX_TO_FACTOR ? g->rw.sig_v2f(e) : g->rw.sig_f2v(e));
    }
    return CX_OK;
}

// set_value! on marginal signals
int32_t cxh_ref_set_marginals(void *p, int64_t n, const int64_t *variable_ids) {
    HostGraph *g = (HostGraph *)p;
    for (int64_t i = 0; i < n; i++) {
        auto it = std::lower_bound(g->var_ids.begin(), g->var_ids.end(), variable_ids[i]);
        if (it == g->var_ids.end() || *it != variable_ids[i]) return CX_ERR_NOT_FOUND;
        cx::refsched::set_value(g->rw, g->rs, g->rw.sig_marg(it - g->var_ids.begin()));
    }
    return CX_OK;
}

// one update_marginals!(ids) on the shadow; returns the number of executions (or a negative status), leaves the call for cxh_ref_trace / cxh_ref_level
int64_t cxh_ref_update(void *p, int64_t n, const int64_t *variable_ids) {
    HostGraph *g = (HostGraph *)p;
    std::vector<int32_t> req((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        auto it = std::lower_bound(g->var_ids.begin(), g->var_ids.end(), variable_ids[i]);
        if (it == g->var_ids.end() || *it != variable_ids[i]) return CX_ERR_NOT_FOUND;
        req[i] = (int32_t)(it - g->var_ids.begin());
    }
    (void)cx::refsched::update_marginals(g->rw, g->rs, req.data(), n, g->rcall);
    return (int64_t)g->rcall.order.size();
}

// the executions of the last call as rows {kind (CX_ITEM_*), variable id, factor id, range lo, range hi, round}
void cxh_ref_trace(const void *p, int64_t *out6) {
    const HostGraph *g = (const HostGraph *)p;
    const auto &W = g->rw;
    for (size_t i = 0; i < g->rcall.order.size(); i++) {
        const int64_t s = g->rcall.order[i];
        int64_t *o = out6 + 6 * i;
        o[0] = o[1] = o[2] = o[3] = o[4] = 0; o[5] = g->rcall.round_of[i];
        if (s < 2 * W.ne) {
            const int64_t e = s < W.ne ? s : s - W.ne;
            o[0] = s < W.ne ? CX_ITEM_MESSAGE_TO_FACTOR : CX_ITEM_MESSAGE_TO_VARIABLE; o[1] = g->var_ids[g->edge_var[e]]; o[2] = g->edge_fac_id[e];
        } else if (s < 2 * W.ne + W.nv) { o[0] = CX_ITEM_INDIVIDUAL_MARGINAL; o[1] = g->var_ids[s - 2 * W.ne]; }
        else if (W.is_joint(s)) { o[0] = CX_ITEM_JOINT_MARGINAL; o[2] = g->fac_ids[s - W.sig_joint(0)]; }
        else { const auto &pr = W.prods[s - W.sig_prod(0)]; o[0] = CX_ITEM_PRODUCT_OF_MESSAGES; o[1] = g->var_ids[pr.var]; o[3] = pr.lo; o[4] = pr.hi; }
    }
}

// the last call levelled into stages of device items (segment-tree node i lives at index i of the product store)
int32_t cxh_ref_level(void *p, char *err, int32_t errlen) {
    HostGraph *g = (HostGraph *)p;
    std::string e;
    int32_t rc;
    const char *wl = std::getenv("CXH_REF_WIDE_LIST");      // (tests: a low threshold sends small lists down the wide path)
    try { rc = cx::refsched::level(g, g->rw, g->rcall, [](int64_t i) { return i; }, [](int64_t f) { return f; }, g->rplan, e, wl ? std::atoll(wl) : cx::refsched::kWideList); }
    catch (const std::exception &x) { rc = CX_ERR_INVALID_ARGUMENT; e = x.what(); }
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.c_str());
    return rc;
}

// what: 0 signals 1 dependencies 2 segment-tree nodes 3 state fingerprint (low 63 bits) 4 passes of the last call
int64_t cxh_ref_scalar(const void *p, int32_t what) {
    const HostGraph *g = (const HostGraph *)p;
    switch (what) {
    case 0: return g->rw.nsig;
    case 1: return (int64_t)g->rw.dep.size();
    case 2: return (int64_t)g->rw.prods.size();
    case 3: return (int64_t)(g->rs.hash & 0x7fffffffffffffffull);
    case 4: return g->rcall.rounds;
    }
    return -1;
}

// deep halo: layers by variable id (others 0), then the send list as slots; fills trim_lo / trim_hi / own and quiet runs
int32_t cxh_flat_halo(void *p, int64_t n, const int64_t *variable_ids, const int32_t *layer, int32_t depth, int64_t n_send, const int32_t *send_slots) {
    HostGraph *g = (HostGraph *)p;
    std::vector<int32_t> lay(g->nv, 0);
    for (int64_t i = 0; i < n; i++) {
        auto it = std::lower_bound(g->var_ids.begin(), g->var_ids.end(), variable_ids[i]);
        if (it == g->var_ids.end() || *it != variable_ids[i]) return CX_ERR_NOT_FOUND;
        lay[it - g->var_ids.begin()] = layer[i];
    }
    cx::haloplan::layers(g, lay, depth);
    g->send_slots.assign(send_slots, send_slots + n_send);
    cx::haloplan::quiet_run(g);
    return CX_OK;
}

// array `which` of the graph: returns its length, copies it as int64 (or as doubles for the floating-point ones) when out != NULL
//   0 var_ids 1 var_off 2 edge_var 3 edge_fac_id 4 vbase 5 vinfo 6 slice_off 7 partner 8 big_vars 9 spdir 10 var_deg
//   20 q 21 a 22 b 23 sq 24 sa 25 sb 26 kary_coef 27 kary_qb          30 kary_slot 31 slot_kary
//   60 tree items (5 per item) 61 tree stage offsets 62 tree k-ary entries 63 their stage offsets 64 partner 65 slot_kary 66 kary_slot
//   70.. heavy-path plan: 70 items 71 stage offsets 72 k-ary entries 73 their offsets 74 pos_var 75 skip0 76 skip1_up 77 skip1_down 78 link_pos 79 from 80 to
//        81 head_fwd 82 head_bwd 83 pos_off 84 link_off 85 steps
//   100 the plan's scan steps (stage, first link, end) 101 .. 107 the links: leader slot, leader variable, follower slot, precision variable, source offsets, sources, heads
//   90 reference plan items (5 per item) 91 its stage offsets 92 its source lists 93 dep_off 94 dep 95 intermediate flags 96 signal flags 97 the last call's executions (signal numbers)
//   40 pos_var 41 skip0 42 skip1 43 link_pos 44 from 45 to 46 head_fwd 47 head_bwd 48 tab_fwd 49 tab_bwd 50 trim_lo 51 trim_hi
int64_t cxh_flat_array(const void *p, int32_t which, void *out) {
    const HostGraph *g = (const HostGraph *)p;
    auto ints = [&](const auto &v) { if (out) for (size_t i = 0; i < v.size(); i++) ((int64_t *)out)[i] = (int64_t)v[i]; return (int64_t)v.size(); };
    auto dbls = [&](const std::vector<double> &v) { if (out) std::memcpy(out, v.data(), v.size() * 8); return (int64_t)v.size(); };
    switch (which) {
    case 0: return ints(g->var_ids); case 1: return ints(g->var_off); case 2: return ints(g->edge_var); case 3: return ints(g->edge_fac_id);
    case 4: return ints(g->vbase); case 5: return ints(g->vinfo); case 6: return ints(g->slice_off); case 7: return ints(g->partner);
    case 8: return ints(g->big_vars); case 9: return ints(g->fo.spdir); case 10: return ints(g->fo.var_deg);
    case 20: return dbls(g->fo.q); case 21: return dbls(g->fo.a); case 22: return dbls(g->fo.b); case 23: return dbls(g->fo.sq);
    case 24: return dbls(g->fo.sa); case 25: return dbls(g->fo.sb); case 26: return dbls(g->kary_coef); case 27: return dbls(g->kary_qb);
    case 30: return ints(g->kary_slot); case 31: return ints(g->slot_kary);
    case 40: return ints(g->co.pos_var); case 41: return ints(g->co.skip0); case 42: return ints(g->co.skip1); case 43: return ints(g->co.link_pos);
    case 44: return ints(g->co.from); case 45: return ints(g->co.to); case 46: return ints(g->co.head_fwd); case 47: return ints(g->co.head_bwd);
    case 48: return ints(g->co.tab_fwd); case 49: return ints(g->co.tab_bwd);
    case 50: return ints(g->trim_lo); case 51: return ints(g->trim_hi);
    case 60: return ints(g->to.rec); case 61: return ints(g->to.stage_off); case 62: return ints(g->to.kary); case 63: return ints(g->to.kary_off);
    case 64: return ints(g->partner); case 65: return ints(g->slot_kary); case 66: return ints(g->kary_slot);
    case 70: return ints(g->hp.rec); case 71: return ints(g->hp.stage_off); case 72: return ints(g->hp.kary); case 73: return ints(g->hp.kary_off);
    case 74: return ints(g->hp.pos_var); case 75: return ints(g->hp.skip0); case 76: return ints(g->hp.skip1_up); case 77: return ints(g->hp.skip1_down);
    case 78: return ints(g->hp.link_pos); case 79: return ints(g->hp.from); case 80: return ints(g->hp.to); case 81: return ints(g->hp.head_fwd);
    case 82: return ints(g->hp.head_bwd); case 83: return ints(g->hp.pos_off); case 84: return ints(g->hp.link_off); case 85: return ints(g->hp.steps);
    case 90: return ints(g->rplan.rec); case 91: return ints(g->rplan.stage_off); case 92: return ints(g->rplan.list); case 93: return ints(g->rw.dep_off); case 94: return ints(g->rw.dep);
    case 95: return ints(g->rw.dep_inter); case 96: return ints(g->rs.flags); case 97: return ints(g->rcall.order);
    case 98: return ints(g->rplan.wide_rec); case 99: return ints(g->rplan.wide_off);
    case 100: { if (out) for (size_t i = 0; i < g->rplan.scans.size(); i++) { ((int64_t *)out)[3 * i] = g->rplan.scans[i].stage; ((int64_t *)out)[3 * i + 1] = g->rplan.scans[i].lo; ((int64_t *)out)[3 * i + 2] = g->rplan.scans[i].hi; }
                return (int64_t)g->rplan.scans.size() * 3; }
    case 101: return ints(g->rplan.sl_lead_dst); case 102: return ints(g->rplan.sl_lead_var); case 103: return ints(g->rplan.sl_fol_dst); case 104: return ints(g->rplan.sl_prec);
    case 105: return ints(g->rplan.sl_src_off); case 106: return ints(g->rplan.sl_src); case 107: return ints(g->rplan.sl_head);
    }
    return -1;
}

// scalars: 0 nv 1 nf 2 ne 3 nslots 4 nslices 5 n_messages_per_sweep 6 any_linear 7 n_kary 8 big_start 9 npos_linked 10 own_slice_lo 11 own_slice_hi
//          12 ipc_quiet_lo 13 ipc_quiet_hi 14 tree depth 15 tree components 16 messages up 17 messages down 18 marginals
//          19 heavy-path plan: light depths 20 paths 21 variables on no path 22 launches per sweep 23 the marginal stage 24 links through factors of more than two edges
int64_t cxh_flat_scalar(const void *p, int32_t which) {
    const HostGraph *g = (const HostGraph *)p;
    const int64_t v[] = {g->nv, g->nf, g->ne, g->nslots, g->nslices, g->n_messages_per_sweep, g->any_linear ? 1 : 0, g->n_kary, g->big_start, g->co.npos_linked,
                         g->own_slice_lo, g->own_slice_hi, g->ipc_quiet_lo, g->ipc_quiet_hi, g->to.depth, g->to.n_components, g->to.n_up, g->to.n_down, g->to.n_marginals,
                         g->hp.levels, g->hp.n_paths, g->hp.n_single, g->hp.launches, g->hp.marginal_stage, g->hp.n_kary_links};
    return which >= 0 && which < 25 ? v[which] : -1;
}

}  // extern "C"
