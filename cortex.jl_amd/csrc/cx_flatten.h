// cx_flatten.h — the GPU-free part of cx_graph_create: the bipartite factor graph flattened ONCE into the tables the kernels read
// (reference: the accessor loops of src/inference_engine.jl:228-247 and src/dependencies.jl:5-126 over
// ext/BipartiteFactorGraphsExt/BipartiteFactorGraphsExt.jl:22-48): edges sorted by (variable id, factor id) — ascending-id neighbour
// order —, the SELL-256 slot layout, per-slot partners and rule parameters (the gather lists of dependencies.jl:17-31), the tables of
// the factors with more than two edges, the count of messages with a dependency and a listener.
//
// Pure host C++ over any struct H with cx_handle's host fields: libcortex_hip.so instantiates it for cx_handle (cx_api.hip), the CPU
// tests for a plain struct (cx_hostlogic.cpp), also under -fsanitize=address,undefined — the out-of-bounds read that lived here for
// three rounds (q has ONE element for dim > 1, and q[partner[s]] was read for every slot) is reproduced by that build with
// -DCX_REINTRODUCE_Q_PARTNER_READ and caught (tests/test_hostlogic.py).
#pragma once

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

#include "cortex_hip.h"
#include "cx_const.h"

namespace cx {
namespace flat {

struct Out {                       // what the upload needs beyond H's own fields
    std::vector<int32_t> var_deg, spdir;
    std::vector<double> q, a, b, sq, sa, sb;
    int64_t big_total = 0;
    bool mv = false;
};

inline int32_t fail_(std::string &err, int32_t code, const std::string &msg) { err = msg; return code; }
#define CX_FLAT_REQUIRE(cond, code, msg) do { if (!(cond)) return fail_(err, code, msg); } while (0)

template <class H>
inline int32_t slot_of_edge_t(const H *h, int64_t e) {
    const int32_t v = h->edge_var[e];
    const int32_t k = (int32_t)(e - h->var_off[v]);
    return ((h->vinfo[v] & kDegMask) == kBigDeg) ? h->vbase[v] + k : h->vbase[v] + k * kBlock;
}

template <class H>
int32_t flatten(H *h, int64_t n_edges, const int64_t *edge_var, const int64_t *edge_fac, const int32_t *edge_role, int64_t n_factors,
                const int64_t *factor_ids, const int32_t *factor_kind, const double *factor_params, Out &out, std::string &err) {
    const int64_t ne = n_edges;
    // ---- sort edges by (variable id, factor id): ascending-id neighbour order -------------------------------
    std::vector<int64_t> ord(ne);
    std::iota(ord.begin(), ord.end(), 0);
    bool sorted = true;
    for (int64_t e = 1; e < ne && sorted; e++)
        sorted = (edge_var[e - 1] < edge_var[e]) || (edge_var[e - 1] == edge_var[e] && edge_fac[e - 1] < edge_fac[e]);
    if (!sorted)
        std::sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) {
            return edge_var[a] != edge_var[b] ? edge_var[a] < edge_var[b] : edge_fac[a] < edge_fac[b];
        });
    for (int64_t e = 1; e < ne; e++)
        if (edge_var[ord[e]] == edge_var[ord[e - 1]] && edge_fac[ord[e]] == edge_fac[ord[e - 1]])
            return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: duplicate edge");
    // ---- variables (CSR) ----------------------------------------------------------------------------------------
    h->var_ids.clear(); h->var_off.clear(); h->edge_var.assign(ne, 0); h->edge_fac_id.assign(ne, 0);
    for (int64_t e = 0; e < ne; e++) {
        int64_t v = edge_var[ord[e]];
        CX_FLAT_REQUIRE(v >= 1, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: ids are 1-based");
        if (h->var_ids.empty() || h->var_ids.back() != v) { h->var_ids.push_back(v); h->var_off.push_back((int32_t)e); }
        h->edge_var[e] = (int32_t)h->var_ids.size() - 1;
        h->edge_fac_id[e] = edge_fac[ord[e]];
    }
    h->var_off.push_back((int32_t)ne);
    h->nv = (int64_t)h->var_ids.size(); h->ne = ne;
    const int64_t nv = h->nv;
    // ---- SELL-256 slot layout -----------------------------------------------------------------------------------
    h->nslices = (nv + kBlock - 1) / kBlock;
    h->vinfo.assign(nv, 0); h->vbase.assign(nv, 0); h->slice_off.assign(h->nslices + 1, 0);
    h->big_vars.clear(); h->big_slots.clear();
    std::vector<int32_t> var_deg(nv);
    int64_t slots = 0;
    for (int64_t s = 0; s < h->nslices; s++) {
        int32_t W = 0;
        const int64_t v0 = s * kBlock, v1 = std::min<int64_t>(nv, v0 + kBlock);
        for (int64_t v = v0; v < v1; v++) {
            const int32_t deg = h->var_off[v + 1] - h->var_off[v];
            var_deg[v] = deg;
            if (deg <= kSmallDeg) { W = std::max(W, deg); h->vinfo[v] = (uint8_t)deg; }
            else { h->vinfo[v] = kBigDeg; h->big_vars.push_back((int32_t)v); }
        }
        h->slice_off[s] = (int32_t)slots;
        for (int64_t v = v0; v < v1; v++) h->vbase[v] = (int32_t)(slots + (v - v0));
        slots += (int64_t)W * kBlock;
        CX_FLAT_REQUIRE(slots < (int64_t)0x7fffff00, CX_ERR_UNSUPPORTED, "cx_graph_create: slot space exceeds 2^31");
    }
    h->slice_off[h->nslices] = (int32_t)slots;
    h->big_start = (int32_t)slots;
    for (int32_t v : h->big_vars) {
        h->vbase[v] = (int32_t)slots;
        for (int32_t k = 0; k < var_deg[v]; k++) h->big_slots.push_back((int32_t)slots + k);
        slots += var_deg[v];
        CX_FLAT_REQUIRE(slots < (int64_t)0x7fffff00, CX_ERR_UNSUPPORTED, "cx_graph_create: slot space exceeds 2^31");
    }
    const int64_t big_total = slots - h->big_start;
    // dim 2..4 store a message as 16-byte pairs, block-major over blocks of 256 slots (cx_mv_core.h): the CSR tail of the variables of
    // degree > 8 ends on a whole block
    if (h->cfg.dim > 1 && !h->big_vars.empty()) slots = (slots + kBlock - 1) / kBlock * kBlock;
    h->nslots = slots;
    // ---- factors ------------------------------------------------------------------------------------------------
    std::vector<int64_t> ford(n_factors);
    std::iota(ford.begin(), ford.end(), 0);
    bool fsorted = true;
    for (int64_t f = 1; f < n_factors && fsorted; f++) fsorted = factor_ids[f - 1] < factor_ids[f];
    if (!fsorted) std::sort(ford.begin(), ford.end(), [&](int64_t a, int64_t b) { return factor_ids[a] < factor_ids[b]; });
    h->fac_ids.resize(n_factors); h->fac_kind.resize(n_factors); h->fac_params.resize(n_factors * CX_NPARAM);
    for (int64_t f = 0; f < n_factors; f++) {
        h->fac_ids[f] = factor_ids[ford[f]];
        if (f > 0 && h->fac_ids[f] == h->fac_ids[f - 1]) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: duplicate factor id");
        h->fac_kind[f] = factor_kind[ford[f]];
        for (int k = 0; k < CX_NPARAM; k++) h->fac_params[f * CX_NPARAM + k] = factor_params[ford[f] * CX_NPARAM + k];
    }
    h->nf = n_factors;
    h->lin_out_is_second.assign(n_factors, 0); h->fac_edges.clear();
    h->np_role.clear(); h->var_gamma.clear();
    h->n_kary = 0; h->kary_slot.clear(); h->kary_coef.clear(); h->kary_qb.clear(); h->slot_kary.clear(); h->kary_pset.clear(); h->kary_dirty = true;
    std::vector<int32_t> edge_fix(ne);      // local factor number per CSR edge
    for (int64_t e = 0; e < ne; e++) {
        auto it = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), h->edge_fac_id[e]);
        if (it == h->fac_ids.end() || *it != h->edge_fac_id[e]) return fail_(err, CX_ERR_NOT_FOUND, "cx_graph_create: edge names a factor id missing from factor_ids");
        edge_fix[e] = (int32_t)(it - h->fac_ids.begin());
    }
    // factor CSR by counting sort (edges of one factor come out in ascending variable order)
    std::vector<int32_t> foff(n_factors + 1, 0);
    for (int64_t e = 0; e < ne; e++) foff[edge_fix[e] + 1]++;
    for (int64_t f = 0; f < n_factors; f++) foff[f + 1] += foff[f];
    std::vector<int32_t> fedge(ne), fill(foff.begin(), foff.end() - 1);
    for (int64_t e = 0; e < ne; e++) fedge[fill[edge_fix[e]]++] = (int32_t)e;
    // ---- per-slot rule parameters and partners (the gather lists of dependencies.jl:17-31) ----------------------
    h->partner.assign(slots, -1);
    const bool mv = h->cfg.dim > 1;
    std::vector<int32_t> spdir;
    if (mv) {
        spdir.assign(slots, 0);
        // dim 2..4: a variable's incoming messages live in registers (k_sweep_mv<D, DEG>: DEG = 3, 4 or 8 by the graph's widest
        // variable); variables of degree > 8 live in the CSR tail like the scalar ones (round 5: k_big_mv, cx_mv.hip; fused and tree
        // schedules, batch items — the reference's resolver takes any degree, src/dependencies.jl:90-173).  dim 64: a rule sums at most
        // three sources itself (k_rule64w) — a sender of degree 5 or more has its other messages summed into its stored variable→factor
        // message first (k_v2f64; flooding and tree schedules, batch items); the chain-scan plans sum the side information of such a
        // path variable into one message of their own first (k_side64, cx_mv64chain.hip).
        // (round 6) dim 64 takes any degree too: a variable of degree > 8 lives in the CSR tail (consecutive slots), k_v2f64 and the marginal
        // kernel walk its slots by the variable's own degree and stride
        if (!h->big_vars.empty() && h->cfg.schedule == CX_SCHED_CHAIN_SCAN)
            return fail_(err, CX_ERR_UNSUPPORTED, std::string("cx_graph_create: the chain-scan schedule for dim > 1 handles variables of degree <= 8; variable ") +
                        std::to_string(h->var_ids[h->big_vars[0]]) + " has degree " + std::to_string(var_deg[h->big_vars[0]]) + " (the fused and tree schedules take any degree)");
    }
    std::vector<double> q(mv ? 1 : slots, 0.0), a, b, sq, sa, sb;
    h->any_linear = false;
    // (dim > 1: every factor is "linear" but its parameters are matrices in the rule tables — the per-slot scalar arrays below do
    // not exist; q has ONE element there, and reading q[partner] for every slot ran off its end: a latent out-of-bounds read
    // since round 1 that depended on what the heap held next to it)
#ifdef CX_REINTRODUCE_Q_PARTNER_READ      // the round-1 form: tests/test_hostlogic.py builds it under ASan to show that the sanitizer build catches it
    for (int64_t f = 0; f < n_factors; f++) if (h->fac_kind[f] == CX_FACTOR_GAUSS_LINEAR) h->any_linear = true;
#else
    for (int64_t f = 0; f < n_factors && !mv; f++) if (h->fac_kind[f] == CX_FACTOR_GAUSS_LINEAR) h->any_linear = true;
#endif
    if (h->any_linear) { a.assign(slots, 1.0); b.assign(slots, 0.0); sq.assign(slots, 0.0); sa.assign(slots, 1.0); sb.assign(slots, 0.0); }
    for (int64_t f = 0; f < n_factors; f++) {
        const int32_t deg = foff[f + 1] - foff[f], kind = h->fac_kind[f];
        const double *p = &h->fac_params[f * CX_NPARAM];
        if (kind == CX_FACTOR_OPAQUE) continue;
        if (h->cfg.family != CX_FAMILY_GAUSSIAN) {
            // the one device rule of the generic 2-parameter family: Bernoulli likelihood with an observed outcome
            if (kind != CX_FACTOR_BERNOULLI)
                return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_create: CX_FAMILY_NATURAL2 factors are CX_FACTOR_OPAQUE or CX_FACTOR_BERNOULLI");
            if (deg != 2) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_create: CX_FACTOR_BERNOULLI needs exactly 2 edges (factor id " + std::to_string(h->fac_ids[f]) + ")");
            const int32_t b1 = slot_of_edge_t(h, fedge[foff[f]]), b2 = slot_of_edge_t(h, fedge[foff[f] + 1]);
            h->partner[b1] = b2; h->partner[b2] = b1;
            continue;
        }
        if (kind == CX_FACTOR_NORMAL_PRECISION) {
            // out ~ N(in, 1 / precision) with a Gamma-distributed precision: no sum-product rule — its messages are computed by the variational
            // rules a user wiring selects (cx_graph_wire; cx_refsched.h: kRule*), so only the reference-order schedule can run it.  (The fused
            // calls of the two test models of the reference are the CX_FAMILY_VMP_* handles, cx_vmp.hip.)
            const std::string who = "cx_graph_create: CX_FACTOR_NORMAL_PRECISION (factor id " + std::to_string(h->fac_ids[f]) + ")";
            if (mv || h->cfg.schedule != CX_SCHED_REFERENCE)
                return fail_(err, CX_ERR_UNSUPPORTED, who + " has variational rules only: CX_SCHED_REFERENCE with a user wiring (cx_graph_wire, dim 1), or the CX_FAMILY_VMP_* families");
            if (deg != 3 || !edge_role) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + " needs three edges with roles CX_ROLE_OUT, CX_ROLE_IN, CX_ROLE_PRECISION");
            if (h->np_role.empty()) { h->np_role.assign(ne, -1); h->var_gamma.assign(h->nv, 0); }
            int seen = 0;
            for (int32_t k = 0; k < deg; k++) {
                const int32_t e = fedge[foff[f] + k], role = edge_role[ord[e]];
                if (role < CX_ROLE_OUT || role > CX_ROLE_PRECISION || (seen & (1 << role)))
                    return fail_(err, CX_ERR_INVALID_ARGUMENT, who + " needs three edges with roles CX_ROLE_OUT, CX_ROLE_IN, CX_ROLE_PRECISION, one each");
                seen |= 1 << role;
                h->np_role[e] = (int8_t)role;
                if (role == CX_ROLE_PRECISION) h->var_gamma[h->edge_var[e]] = 1;
            }
            continue;
        }
        if (kind == CX_FACTOR_GAUSS_LINEAR_N) {
            // more than two edges (cx_kary.hip): every edge's message reads all the others' (dependencies.jl:17-31).  Entry order: the OUT
            // edge, then the IN edges by ascending variable id; coefficients start at a_i = 1 (cx_set_factor_coefficients)
            const std::string who = "cx_graph_create: CX_FACTOR_GAUSS_LINEAR_N (factor id " + std::to_string(h->fac_ids[f]) + ")";
            // dim 2..4 (round 5): x_out = A_1 x_1 + ... + A_k x_k + N(0, Q); params[0] names the parameter set whose Q is the noise and whose A is
            // every input's matrix until cx_set_factor_edge_sets says otherwise (cx_kary_mv_core.h)
            if (mv && cx::is_mfma_dim(h->cfg.dim)) return fail_(err, CX_ERR_UNSUPPORTED, who + ": dim 64 (and 5 .. 63 with it) takes factors of two variables");
            if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) return fail_(err, CX_ERR_UNSUPPORTED, who + ": a factor of three or more variables is not a link of a chain (use the tree, the fused or the flooding schedule)");
            if (deg < 3 || deg > 7) return fail_(err, CX_ERR_UNSUPPORTED, who + " takes 2 to 6 inputs and one output (3 to 7 edges), not " + std::to_string(deg) + " edges");
            if (!mv && !(p[0] >= 0.0)) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + ": the variance q must be >= 0");
            if (mv && (p[0] < 0 || (double)(int64_t)p[0] != p[0])) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + ": params[0] must be a parameter-set index");
            if (mv) h->max_pset = std::max<int64_t>(h->max_pset, (int64_t)p[0]);
            if (!edge_role) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + " needs edge roles (one CX_ROLE_OUT, the rest CX_ROLE_IN)");
            const size_t row = (size_t)h->n_kary++;
            h->kary_slot.resize(8 * (row + 1), -1); h->kary_coef.resize(8 * (row + 1), 0.0);
            h->kary_pset.resize(8 * (row + 1), mv ? (int32_t)p[0] : -1);
            h->kary_qb.push_back(p[0]); h->kary_qb.push_back(p[1]);
            if (h->slot_kary.empty()) h->slot_kary.assign(slots, -1);
            int n_in = 0, n_out = 0;
            for (int32_t k = 0; k < deg; k++) {
                const int32_t e = fedge[foff[f] + k], sl = slot_of_edge_t(h, e), role = edge_role[ord[e]];
                if (role != CX_ROLE_OUT && role != CX_ROLE_IN) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + ": edge roles are CX_ROLE_OUT or CX_ROLE_IN");
                const int pos = role == CX_ROLE_OUT ? 0 : 1 + n_in++;
                if (role == CX_ROLE_OUT && n_out++) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + " has more than one CX_ROLE_OUT edge");
                if (pos > 7) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + " has no CX_ROLE_OUT edge");
                h->kary_slot[8 * row + pos] = sl;
                h->kary_coef[8 * row + pos] = role == CX_ROLE_OUT ? 1.0 : -1.0;
                h->slot_kary[sl] = (int32_t)(8 * row + pos);
            }
            if (n_out != 1) return fail_(err, CX_ERR_INVALID_ARGUMENT, who + " needs exactly one CX_ROLE_OUT edge");
            continue;
        }
        if (kind != CX_FACTOR_GAUSS_ADDITIVE && kind != CX_FACTOR_GAUSS_LINEAR)
            return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_create: unknown factor kind");
        if (deg != 2) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_create: Gaussian factor kinds need exactly 2 edges (factor id " + std::to_string(h->fac_ids[f]) + ")");
        const int32_t e1 = fedge[foff[f]], e2 = fedge[foff[f] + 1];
        const int32_t s1 = slot_of_edge_t(h, e1), s2 = slot_of_edge_t(h, e2);
        h->partner[s1] = s2; h->partner[s2] = s1;
        if (mv) {
            // dim > 1: x_out = A x_in + N(0, Q), (A, Q) = parameter set params[0] (cx_set_factor_matrices).
            // spdir[sending slot] = 2 * pset + direction of the RECEIVING edge (0: receiver = out, 1: receiver = in)
            if (kind != CX_FACTOR_GAUSS_LINEAR) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_create: dim > 1 supports CX_FACTOR_GAUSS_LINEAR (params[0] = parameter set) and CX_FACTOR_OPAQUE");
            const int64_t pset = (int64_t)p[0];
            if (pset < 0 || (double)pset != p[0]) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: params[0] must be a parameter-set index");
            const int32_t r1 = edge_role ? edge_role[ord[e1]] : CX_ROLE_OUT, r2 = edge_role ? edge_role[ord[e2]] : CX_ROLE_OUT;
            if (r1 == r2) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: GAUSS_LINEAR needs one ROLE_IN and one ROLE_OUT edge");
            const int32_t sin = r1 == CX_ROLE_IN ? s1 : s2, sout = r1 == CX_ROLE_IN ? s2 : s1;
            spdir[sin] = (int32_t)(2 * pset);       // sent by x_in, received on the out edge: forward
            spdir[sout] = (int32_t)(2 * pset + 1);  // sent by x_out, received on the in edge: backward
            h->max_pset = std::max<int64_t>(h->max_pset, pset);
            continue;
        }
        if (!(p[0] >= 0.0)) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: factor variance must be >= 0");
        q[s1] = q[s2] = p[0];
        if (kind == CX_FACTOR_GAUSS_LINEAR) {
            // x_out = a x_in + b + N(0,q): the edge with ROLE_IN carries x_in.  Effective parameters of the
            // RECEIVING edge: forward (receiver = out) {a, b, q}; backward (receiver = in) {1/a, -b/a, q/a²}.
            const double A = p[1], B = p[2];
            if (A == 0.0) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: GAUSS_LINEAR with a == 0");
            const int32_t r1 = edge_role ? edge_role[ord[e1]] : CX_ROLE_OUT, r2 = edge_role ? edge_role[ord[e2]] : CX_ROLE_OUT;
            if (r1 == r2) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: GAUSS_LINEAR needs one ROLE_IN and one ROLE_OUT edge");
            const int32_t sin = r1 == CX_ROLE_IN ? s1 : s2, sout = r1 == CX_ROLE_IN ? s2 : s1;
            h->lin_out_is_second[f] = sout == s2 ? 1 : 0;    // e1 < e2 in (variable, factor) order: "second" = the higher variable id
            a[sout] = A; b[sout] = B; q[sout] = p[0];
            a[sin] = 1.0 / A; b[sin] = -B / A; q[sin] = p[0] / (A * A);
        }
    }
    if (h->any_linear)
        for (int64_t s = 0; s < slots; s++)
            if (h->partner[s] >= 0) { sq[s] = q[h->partner[s]]; sa[s] = a[h->partner[s]]; sb[s] = b[h->partner[s]]; }
    // a precision variable is Gamma-distributed wherever it appears: the precision of its factors, or on opaque factors (a prior the caller sets)
    for (int64_t e = 0; e < ne && !h->var_gamma.empty(); e++)
        if (h->var_gamma[h->edge_var[e]] && h->np_role[e] != CX_ROLE_PRECISION && h->fac_kind[edge_fix[e]] != CX_FACTOR_OPAQUE)
            return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: variable " + std::to_string(h->var_ids[h->edge_var[e]]) + " is the precision of a CX_FACTOR_NORMAL_PRECISION factor "
                         "(Gamma-distributed) and a Normal variable of factor " + std::to_string(h->edge_fac_id[e]));
    // messages with >=1 dependency and >=1 listener (the metric's unit): both directions of every 2-edge Gaussian
    // factor, minus variable→factor messages of degree-1 variables (no dependencies, dependencies.jl:48-55)
    int64_t m = 0;
    for (int64_t e = 0; e < ne; e++) {
        const int32_t sl_ = slot_of_edge_t(h, e);
        if (h->partner[sl_] < 0 && (h->slot_kary.empty() || h->slot_kary[sl_] < 0)) continue;
        m += 1;
        if (var_deg[h->edge_var[e]] >= 2) m += 1;
    }
    h->n_messages_per_sweep = m;
    out.var_deg = std::move(var_deg); out.spdir = std::move(spdir); out.q = std::move(q); out.a = std::move(a); out.b = std::move(b);
    out.sq = std::move(sq); out.sa = std::move(sa); out.sb = std::move(sb); out.big_total = big_total; out.mv = mv;
    return CX_OK;
}

#undef CX_FLAT_REQUIRE

}  // namespace flat
}  // namespace cx
