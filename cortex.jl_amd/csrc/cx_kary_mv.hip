// cx_kary_mv.hip — linear-Gaussian factors of three to seven d-dimensional variables (d = 2, 3, 4) in the fused sweep: the host side of
// their table and the sweep's launches.  The rule: cx_kary_mv_core.h.  A sweep of a graph that has such factors (1) stores the
// variable→factor messages of their edges (the fused kernel keeps a variable's outgoing messages in registers; these factors read
// messages other threads compute: k_v2f_mv over the list of their slots), (2) computes every message out of them, one thread per
// (factor, edge), from the stored messages of the factor's other edges into the sweep's output buffer.
#include "cx_host.h"
#include "cx_kary_mv_core.h"

namespace cx {

namespace {
template <int D>
__global__ __launch_bounds__(kBlock) void k_kary_mv(int64_t n_entries, const KaryMvTab kt, const double *__restrict__ v2f, double *__restrict__ f2v_out,
                                                    const double *__restrict__ prev, double lam) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_entries) return;
    kary_item_mv<D>((int)i, kt, v2f, f2v_out, prev, lam);
}
}  // namespace

// the raw (A | Q) of every parameter set and the entries' set indices on the device; the list of slots whose variable→factor messages
// a sweep has to store.  Runs when the table is dirty (first use, cx_set_factor_edge_sets) or the matrices changed.
int32_t kary_mv_upload(cx_handle *h) {
    using namespace cxh;
    if (h->n_kary == 0 || h->cfg.dim == 1) return CX_OK;
    if (!h->kary_dirty && h->d_kary_aq) return CX_OK;
    const int d = h->cfg.dim;
    int32_t rc;
    CX_HIP(h, hipStreamSynchronize(h->stream));
    if (!h->d_kary_slot) {
        if ((rc = dev_upload(h, &h->d_kary_slot, h->kary_slot)) != CX_OK) return rc;
        if ((rc = dev_alloc(h, &h->d_kary_pset, (int64_t)h->kary_pset.size())) != CX_OK) return rc;
        std::vector<int32_t> sl, vr;
        std::vector<int32_t> slot_var(h->nslots, -1);
        for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
        for (int32_t s : h->kary_slot) if (s >= 0) { sl.push_back(s); vr.push_back(slot_var[s]); }
        h->n_kary_v2f = (int64_t)sl.size();
        if ((rc = dev_upload(h, &h->d_kary_v2f_slots, sl)) != CX_OK) return rc;
        if ((rc = dev_upload(h, &h->d_kary_v2f_vars, vr)) != CX_OK) return rc;
    }
    for (int32_t ps : h->kary_pset)
        if (ps >= 0 && (ps >= (int32_t)h->psets.size() || h->psets[ps].empty()))
            return fail(h, CX_ERR_STATE, "a CX_FACTOR_GAUSS_LINEAR_N factor names parameter set " + std::to_string(ps) + ", which was never set (cx_set_factor_matrices)");
    const size_t per = (size_t)2 * d * d, nsets = h->psets.size();
    std::vector<double> aq(per * std::max<size_t>(nsets, 1), 0.0);
    for (size_t i = 0; i < nsets; i++) if (!h->psets[i].empty()) std::memcpy(&aq[per * i], h->psets[i].data(), per * 8);
    if (h->d_kary_aq && h->kary_aq_sets < (int64_t)nsets) { tree_graph_drop(h); ref_graphs_drop(h); (void)hipFree(h->d_kary_aq); h->d_kary_aq = nullptr; }      // (captured launches hold the address)
    if (!h->d_kary_aq) { if ((rc = dev_alloc(h, &h->d_kary_aq, (int64_t)aq.size())) != CX_OK) return rc; h->kary_aq_sets = (int64_t)nsets; }
    CX_HIP(h, hipMemcpy(h->d_kary_aq, aq.data(), aq.size() * 8, hipMemcpyHostToDevice));
    CX_HIP(h, hipMemcpy(h->d_kary_pset, h->kary_pset.data(), h->kary_pset.size() * 4, hipMemcpyHostToDevice));
    h->kary_dirty = false;
    return CX_OK;
}

void kary_mv_free(cx_handle *h) {
    for (void *p : {(void *)h->d_kary_pset, (void *)h->d_kary_aq, (void *)h->d_kary_v2f_slots, (void *)h->d_kary_v2f_vars}) if (p) (void)hipFree(p);
    h->d_kary_pset = nullptr; h->d_kary_aq = nullptr; h->d_kary_v2f_slots = h->d_kary_v2f_vars = nullptr; h->n_kary_v2f = 0; h->kary_aq_sets = 0;
}

// the k-ary part of one fused sweep (input buffer d_mv_f2v, output f2v_out)
void mv_launch_kary(cx_handle *h, double *f2v_out) {
    if (h->n_kary == 0) return;
    if (!f2v_out) f2v_out = h->d_mv_f2v_alt;
    mv_launch_v2f(h, h->d_kary_v2f_slots, h->d_kary_v2f_vars, h->n_kary_v2f, h->d_mv_f2v);
    const int64_t n = 8 * h->n_kary;
    const KaryMvTab kt{h->d_kary_slot, h->d_kary_pset, h->d_kary_aq};
    const dim3 g((unsigned)((n + kBlock - 1) / kBlock)), b(kBlock);
#define CX_KM(DD) hipLaunchKernelGGL((k_kary_mv<DD>), g, b, 0, h->stream, n, kt, (const double *)h->d_mv_v2f, f2v_out, (const double *)h->d_mv_f2v, h->damping)
    if (h->cfg.dim == 2) CX_KM(2);
    else if (h->cfg.dim == 3) CX_KM(3);
    else CX_KM(4);
#undef CX_KM
}

}  // namespace cx

extern "C" {

// dim > 1: the A_i of the CX_ROLE_IN edge (variable, factor) of a CX_FACTOR_GAUSS_LINEAR_N factor is the A of parameter set sets[i]
int32_t cx_set_factor_edge_sets(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, const int64_t *parameter_sets) {
    using namespace cxh;
    CX_NOT_VMP(h, "cx_set_factor_edge_sets");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_set_factor_edge_sets: no graph");
    CX_REQUIRE(h, h->cfg.dim >= 2 && h->cfg.dim <= 4, CX_ERR_UNSUPPORTED, "cx_set_factor_edge_sets: dim 2, 3, 4 (scalar factors take cx_set_factor_coefficients)");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && factor_ids && parameter_sets, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_edge_sets: null argument");
    for (int64_t i = 0; i < n; i++) {
        const int64_t e = find_edge(h, variable_ids[i], factor_ids[i]);
        if (e < 0) return fail(h, CX_ERR_NOT_FOUND, "no connection between variable " + std::to_string(variable_ids[i]) + " and factor " + std::to_string(factor_ids[i]));
        const int32_t en = h->slot_kary.empty() ? -1 : h->slot_kary[cx::slot_of_edge(h, e)];
        if (en < 0 || (en & 7) == 0)
            return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_edge_sets: (variable " + std::to_string(variable_ids[i]) + ", factor " + std::to_string(factor_ids[i]) +
                        ") is not a ROLE_IN edge of a CX_FACTOR_GAUSS_LINEAR_N factor");
        if (parameter_sets[i] < 0 || parameter_sets[i] >= (1 << 20)) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_edge_sets: bad parameter set");
        h->kary_pset[en] = (int32_t)parameter_sets[i];
        h->max_pset = std::max<int64_t>(h->max_pset, parameter_sets[i]);
    }
    h->kary_dirty = true; h->tree_dirty = true;
    return CX_OK;
}

}  // extern "C"
