// cx_api.hip — host side of libcortex_hip.so: the C ABI of include/cortex_hip.h.
//
// Flattens the bipartite factor graph once (reference: the accessor loops of
// src/inference_engine.jl:228-247 and src/dependencies.jl:5-126 over
// ext/BipartiteFactorGraphsExt/BipartiteFactorGraphsExt.jl:22-48) into a CSR edge table sorted by
// (variable id, factor id), keeps Gaussian messages resident in HBM, and launches the kernels of
// cx_kernels.hip.  No exception leaves this file; every entry point returns a status.

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <numeric>

#include "cx_internal.h"

namespace {

thread_local std::string g_create_error;

int32_t fail(cx_handle *h, int32_t code, const std::string &msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define CX_HIP(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, e_ == hipErrorOutOfMemory ? CX_ERR_OUT_OF_MEMORY : CX_ERR_DEVICE,     \
                        std::string(#call) + ": " + hipGetErrorString(e_));                      \
    } while (0)

#define CX_REQUIRE(h, cond, code, msg) \
    do { if (!(cond)) return fail(h, code, msg); } while (0)

template <class T>
int32_t dev_alloc(cx_handle *h, T **p, int64_t count) {
    *p = nullptr;
    if (count <= 0) count = 1;
    CX_HIP(h, hipMalloc((void **)p, (size_t)count * sizeof(T)));
    h->device_bytes += count * (int64_t)sizeof(T);
    return CX_OK;
}

template <class T>
int32_t dev_upload(cx_handle *h, T **p, const std::vector<T> &v) {
    int32_t rc = dev_alloc(h, p, (int64_t)v.size());
    if (rc != CX_OK) return rc;
    if (!v.empty()) CX_HIP(h, hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return CX_OK;
}

void dev_free_all(cx_handle *h) {
    void *ptrs[] = {h->d_slice_off, h->d_partner, h->d_vbase, h->d_var_deg, h->d_big, h->d_big_slots, h->d_big_tmp,
                    h->d_vinfo, h->d_q, h->d_a, h->d_b, h->d_sq, h->d_sa, h->d_sb, h->d_f2v, h->d_v2f, h->d_marg,
                    h->d_f2v_alt, h->d_prev, h->d_scratch, h->d_send_slots, h->d_recv_slots, h->d_send_vars,
                    h->ext_halo_buffers ? nullptr : (void *)h->d_send_buf, h->ext_halo_buffers ? nullptr : (void *)h->d_recv_buf,
                    h->d_stage, h->d_spdir, h->d_ptab, h->d_ptab_bt, h->d_zero_msg, h->d_mv_f2v, h->d_mv_f2v_alt, h->d_mv_v2f, h->d_mv_marg, h->d_mv_prev, h->d_rule64_slots, h->d_rule64_vars, h->d_rule64_flags, h->d_point64_slots, h->d_rule64_rec, h->d_chain_pos_var, h->d_chain_skip0, h->d_chain_skip1, h->d_chain_link_pos, h->d_chain_from,
                    h->d_chain_to, h->d_chain_head_fwd, h->d_chain_head_bwd, h->d_chain_side, h->d_chain_totals};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    cx::tiles_free(h);
    if (h->d_f2v_tmp) (void)hipFree(h->d_f2v_tmp);
    h->d_f2v_tmp = nullptr; h->alt_two_back = false;
    if (h->d_prod) (void)hipFree(h->d_prod);
    if (h->d_joint) (void)hipFree(h->d_joint);
    h->d_prod = nullptr; h->d_joint = nullptr; h->prod_cap = h->joint_cap = 0; h->prod_index.clear(); h->joint_index.clear();
    h->d_rule64_slots = h->d_rule64_vars = h->d_rule64_flags = h->d_point64_slots = h->d_rule64_rec = nullptr; h->work64_dirty = h->point64_dirty = true;
    h->d_spdir = nullptr; h->d_ptab = nullptr; h->d_ptab_bt = nullptr; h->d_zero_msg = nullptr; h->d_mv_f2v = h->d_mv_f2v_alt = h->d_mv_v2f = h->d_mv_marg = h->d_mv_prev = nullptr; h->ptab_sets = 0;
    h->d_chain_pos_var = h->d_chain_skip0 = h->d_chain_skip1 = h->d_chain_link_pos = h->d_chain_from = h->d_chain_to = nullptr;
    h->d_chain_head_fwd = h->d_chain_head_bwd = nullptr; h->d_chain_side = nullptr; h->d_chain_totals = nullptr; h->chains_dirty = true;
    h->d_slice_off = h->d_partner = h->d_vbase = h->d_var_deg = h->d_big = h->d_big_slots = nullptr;
    h->d_big_tmp = nullptr; h->d_vinfo = nullptr;
    h->d_q = h->d_a = h->d_b = h->d_sq = h->d_sa = h->d_sb = nullptr;
    h->d_f2v = h->d_v2f = h->d_marg = h->d_f2v_alt = h->d_prev = nullptr;
    h->mv_max_deg = 0;
    h->d_scratch = nullptr; h->d_send_slots = h->d_recv_slots = h->d_send_vars = nullptr;
    h->d_send_buf = h->d_recv_buf = nullptr;
    h->d_stage = nullptr; h->stage_bytes = 0; h->device_bytes = 0;
}

int32_t ensure_stage(cx_handle *h, int64_t bytes) {
    if (bytes <= h->stage_bytes) return CX_OK;
    if (h->d_stage) { CX_HIP(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_stage); h->d_stage = nullptr; }
    int64_t want = std::max<int64_t>(bytes, 1 << 20);
    CX_HIP(h, hipMalloc(&h->d_stage, (size_t)want));
    h->stage_bytes = want;
    return CX_OK;
}

// (variable_id, factor_id) -> edge index; edges are sorted by (variable, factor)
int64_t find_edge(const cx_handle *h, int64_t var_id, int64_t fac_id) {
    auto it = std::lower_bound(h->var_ids.begin(), h->var_ids.end(), var_id);
    if (it == h->var_ids.end() || *it != var_id) return -1;
    int64_t v = it - h->var_ids.begin();
    auto b = h->edge_fac_id.begin() + h->var_off[v], e = h->edge_fac_id.begin() + h->var_off[v + 1];
    auto jt = std::lower_bound(b, e, fac_id);
    if (jt == e || *jt != fac_id) return -1;
    return jt - h->edge_fac_id.begin();
}

int64_t find_var(const cx_handle *h, int64_t var_id) {
    auto it = std::lower_bound(h->var_ids.begin(), h->var_ids.end(), var_id);
    if (it == h->var_ids.end() || *it != var_id) return -1;
    return it - h->var_ids.begin();
}

const double kNaN = std::numeric_limits<double>::quiet_NaN();
const double kInf = std::numeric_limits<double>::infinity();

// boundary form -> storage (natural) form
bool to_natural(int32_t form, const double *p, double2 *out) {
    switch (form) {
    case CX_FORM_MOMENT:
        if (std::isnan(p[1])) { *out = make_double2(kNaN, kNaN); return true; }
        if (p[1] == 0.0) { *out = make_double2(p[0], kInf); return true; }  // zero variance == point mass
        out->y = 1.0 / p[1]; out->x = p[0] * out->y; return true;
    case CX_FORM_POINT: *out = make_double2(p[0], kInf); return true;
    case CX_FORM_NATURAL: *out = make_double2(p[0], p[1]); return true;
    }
    return false;
}

void from_natural(int32_t form, double2 m, double *out) {
    if (form == CX_FORM_NATURAL) { out[0] = m.x; out[1] = m.y; return; }
    if (std::isnan(m.y)) { out[0] = kNaN; if (form == CX_FORM_MOMENT) out[1] = kNaN; return; }
    if (m.y == kInf) { out[0] = m.x; if (form == CX_FORM_MOMENT) out[1] = 0.0; return; }
    double var = 1.0 / m.y;
    out[0] = m.x * var;
    if (form == CX_FORM_MOMENT) out[1] = var;
}

}  // namespace

extern "C" {

int32_t cx_version(void) { return CX_ABI_VERSION; }

const char *cx_last_error(const cx_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int64_t cx_payload_doubles(int32_t dim, int32_t form) {
    if (dim < 1) return -1;
    if (form == CX_FORM_POINT) return dim;
    if (form == CX_FORM_MOMENT || form == CX_FORM_NATURAL) return (int64_t)dim + (int64_t)dim * dim;
    if ((form == CX_FORM_MEAN_PRECISION || form == CX_FORM_GAMMA) && dim == 1) return 2;
    return -1;
}

const char *cx_kernel_name(int32_t k) {
    switch (k) {
    case CX_KERNEL_VAR_TO_FACTOR: return "k_sweep<var_to_factor>";
    case CX_KERNEL_FACTOR_TO_VAR: return "k_factor_to_var";
    case CX_KERNEL_FUSED: return "k_sweep<fused>";
    case CX_KERNEL_BATCH: return "k_batch";
    case CX_KERNEL_BIG_VAR: return "k_big_var_to_factor";
    case CX_KERNEL_HALO_BEGIN: return "k_halo_export";
    case CX_KERNEL_HALO_END: return "k_halo_import";
    case CX_KERNEL_TILED: return "k_sweep2<two sweeps per launch>";
    }
    return "";
}

int32_t cx_create(const cx_config *config, cx_handle **out) {
    if (!out) return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: out is NULL");
    *out = nullptr;
    if (!config || config->struct_size != (int32_t)sizeof(cx_config))
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: config is NULL or struct_size mismatch");
    if (config->dim != 1 && config->dim != 2 && config->dim != 3 && config->dim != 4 && config->dim != 64)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: this build implements dim in {1, 2, 3, 4, 64}");
    const bool is_vmp = config->family == CX_FAMILY_VMP_MEAN_FIELD || config->family == CX_FAMILY_VMP_STRUCTURED;
    if (config->family != CX_FAMILY_GAUSSIAN && config->family != CX_FAMILY_NATURAL2 && !is_vmp)
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: unknown family");
    if (is_vmp && config->dim != 1)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: the variational families need dim == 1");
    if (config->family == CX_FAMILY_NATURAL2 && (config->dim != 1 || config->schedule == CX_SCHED_CHAIN_SCAN))
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: CX_FAMILY_NATURAL2 needs dim == 1 and the flooding or fused schedule");
    if (config->dim > 1 && config->schedule != CX_SCHED_FUSED)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: dim > 1 runs the fused schedule only");
    if (config->schedule != CX_SCHED_FLOODING && config->schedule != CX_SCHED_FUSED && config->schedule != CX_SCHED_CHAIN_SCAN)
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: unknown schedule");
    if (config->sweeps_per_launch < 0 || config->sweeps_per_launch > 2)
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: sweeps_per_launch must be 0 (automatic), 1 or 2");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CX_ERR_NO_DEVICE, "cx_create: no HIP device visible (this library has no CPU fallback)");
    if (config->device < 0 || config->device >= ndev) return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: bad device ordinal");
    e = hipSetDevice(config->device);
    if (e != hipSuccess) return fail(nullptr, CX_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    cx_handle *h = new (std::nothrow) cx_handle();
    if (!h) return fail(nullptr, CX_ERR_OUT_OF_MEMORY, "cx_create: host allocation failed");
    h->cfg = *config;
    h->nc = config->dim == 1 ? 2 : (config->dim == 64 ? 64 + 64 * 64 : config->dim + config->dim * (config->dim + 1) / 2);
    h->stream = nullptr;  // default stream until cx_set_stream
    *out = h;
    return CX_OK;
}

int32_t cx_destroy(cx_handle *h) {
    if (!h) return CX_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    for (auto &r : h->recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    cx::comm_destroy(h);
    cx::vmp_free(h);
    dev_free_all(h);
    delete h;
    return CX_OK;
}

int32_t cx_sync(cx_handle *h) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "null handle");
    CX_HIP(h, hipStreamSynchronize(h->stream));
    return CX_OK;
}

int32_t cx_set_stream(cx_handle *h, void *hip_stream) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "null handle");
    CX_HIP(h, hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)hip_stream;
    return cx::vmp_set_stream(h);
}

// (re)upload the (P, B, C) rule tables of every registered parameter set (dim > 1)
static int32_t upload_ptab(cx_handle *h) {
    const int d = h->cfg.dim;
    const int64_t nsets = std::max<int64_t>((int64_t)h->psets.size(), h->max_pset + 1);
    if (nsets == 0) return CX_OK;
    const size_t per = (size_t)6 * d * d;
    std::vector<double> tab(per * nsets, std::numeric_limits<double>::quiet_NaN());
    for (int64_t i = 0; i < (int64_t)h->psets.size(); i++) {
        if (h->psets[i].empty()) continue;
        if (!cx::mv_rule_tables(d, h->psets[i].data(), h->psets[i].data() + d * d, &tab[per * i]))
            return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_matrices: Q of parameter set " + std::to_string(i) + " is not positive definite");
    }
    CX_HIP(h, hipStreamSynchronize(h->stream));
    if (h->d_ptab && h->ptab_sets < nsets) { (void)hipFree(h->d_ptab); h->d_ptab = nullptr; }
    if (!h->d_ptab) { int32_t rc = dev_alloc(h, &h->d_ptab, (int64_t)(per * nsets)); if (rc != CX_OK) return rc; h->ptab_sets = nsets; }
    CX_HIP(h, hipMemcpy(h->d_ptab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    if (d == 64) {   // the wave-per-message rule kernel reads B transposed (tile rows are its contraction index)
        const size_t dd = (size_t)d * d;
        std::vector<double> bt(2 * (size_t)nsets * dd);
        for (int64_t t = 0; t < 2 * nsets; t++) {
            const double *B = &tab[(size_t)t * 3 * dd + dd];
            for (int r = 0; r < d; r++) for (int c = 0; c < d; c++) bt[(size_t)t * dd + (size_t)r * d + c] = B[(size_t)c * d + r];
        }
        if (!h->d_zero_msg) {
            int32_t rc0 = dev_alloc(h, &h->d_zero_msg, (int64_t)(d + d * d));
            if (rc0 != CX_OK) return rc0;
            CX_HIP(h, hipMemset(h->d_zero_msg, 0, (size_t)(d + d * d) * 8));
        }
        if (h->d_ptab_bt) { (void)hipFree(h->d_ptab_bt); h->d_ptab_bt = nullptr; }
        int32_t rc = dev_alloc(h, &h->d_ptab_bt, (int64_t)bt.size());
        if (rc != CX_OK) return rc;
        CX_HIP(h, hipMemcpy(h->d_ptab_bt, bt.data(), bt.size() * 8, hipMemcpyHostToDevice));
    }
    return CX_OK;
}

int32_t cx_set_factor_matrices(cx_handle *h, int64_t parameter_set, const double *A, const double *Q) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "null handle");
    CX_REQUIRE(h, h->cfg.dim > 1, CX_ERR_STATE, "cx_set_factor_matrices: dim == 1 factors take scalar parameters");
    CX_REQUIRE(h, parameter_set >= 0 && parameter_set < (1 << 20) && A && Q, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_matrices: bad argument");
    try {
        const int d = h->cfg.dim;
        if ((int64_t)h->psets.size() <= parameter_set) h->psets.resize(parameter_set + 1);
        auto &ps = h->psets[parameter_set];
        ps.assign(A, A + d * d);
        ps.insert(ps.end(), Q, Q + d * d);
        std::vector<double> chk((size_t)6 * d * d);
        if (!cx::mv_rule_tables(d, A, Q, chk.data())) {
            ps.clear();
            return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_matrices: Q is not symmetric positive definite");
        }
        if (!h->has_graph) return CX_OK;
        // the messages out of observed variables, N(A y, Q), are cached in both Jacobi buffers: new (A, Q) invalidates them
        h->observed_passes_due = 2;
        h->point64_dirty = true;
        return upload_ptab(h);
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_factor_matrices: host allocation failed"); }
}

static inline bool is_vmp(const cx_handle *h) {
    return h->cfg.family == CX_FAMILY_VMP_MEAN_FIELD || h->cfg.family == CX_FAMILY_VMP_STRUCTURED;
}
#define CX_NOT_VMP(h, name) CX_REQUIRE(h, !(h) || !is_vmp(h), CX_ERR_UNSUPPORTED, name ": not available for the variational families (their state is the set of marginals: cx_set_marginals / cx_update_marginals)")

int32_t cx_graph_create(cx_handle *h, int64_t n_edges, const int64_t *edge_var, const int64_t *edge_fac,
                        const int32_t *edge_role, int64_t n_factors, const int64_t *factor_ids,
                        const int32_t *factor_kind, const double *factor_params) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "null handle");
    CX_REQUIRE(h, !h->has_graph, CX_ERR_STATE, "cx_graph_create: handle already has a graph");
    CX_REQUIRE(h, n_edges > 0 && edge_var && edge_fac, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: empty edge list");
    CX_REQUIRE(h, n_edges < (int64_t)0x0fffffff, CX_ERR_UNSUPPORTED, "cx_graph_create: more than 2^28-1 edges per handle");
    if (is_vmp(h)) {
        CX_REQUIRE(h, n_factors > 0 && factor_ids && factor_kind, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: factor table missing");
        int32_t rc = cx::vmp_graph_create(h, n_edges, edge_var, edge_fac, edge_role, n_factors, factor_ids, factor_kind);
        if (rc != CX_OK) cx::vmp_free(h);
        return rc;
    }
    CX_REQUIRE(h, n_factors > 0 && factor_ids && factor_kind && factor_params, CX_ERR_INVALID_ARGUMENT,
               "cx_graph_create: factor table missing");
    try {
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
                return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: duplicate edge");
        // ---- variables (CSR) ----------------------------------------------------------------------------------------
        h->var_ids.clear(); h->var_off.clear(); h->edge_var.assign(ne, 0); h->edge_fac_id.assign(ne, 0);
        for (int64_t e = 0; e < ne; e++) {
            int64_t v = edge_var[ord[e]];
            CX_REQUIRE(h, v >= 1, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: ids are 1-based");
            if (h->var_ids.empty() || h->var_ids.back() != v) { h->var_ids.push_back(v); h->var_off.push_back((int32_t)e); }
            h->edge_var[e] = (int32_t)h->var_ids.size() - 1;
            h->edge_fac_id[e] = edge_fac[ord[e]];
        }
        h->var_off.push_back((int32_t)ne);
        h->nv = (int64_t)h->var_ids.size(); h->ne = ne;
        const int64_t nv = h->nv;
        // ---- SELL-256 slot layout -----------------------------------------------------------------------------------
        h->nslices = (nv + cx::kBlock - 1) / cx::kBlock;
        h->vinfo.assign(nv, 0); h->vbase.assign(nv, 0); h->slice_off.assign(h->nslices + 1, 0);
        h->big_vars.clear(); h->big_slots.clear();
        std::vector<int32_t> var_deg(nv);
        int64_t slots = 0;
        for (int64_t s = 0; s < h->nslices; s++) {
            int32_t W = 0;
            const int64_t v0 = s * cx::kBlock, v1 = std::min<int64_t>(nv, v0 + cx::kBlock);
            for (int64_t v = v0; v < v1; v++) {
                const int32_t deg = h->var_off[v + 1] - h->var_off[v];
                var_deg[v] = deg;
                if (deg <= cx::kSmallDeg) { W = std::max(W, deg); h->vinfo[v] = (uint8_t)deg; }
                else { h->vinfo[v] = cx::kBigDeg; h->big_vars.push_back((int32_t)v); }
            }
            h->slice_off[s] = (int32_t)slots;
            for (int64_t v = v0; v < v1; v++) h->vbase[v] = (int32_t)(slots + (v - v0));
            slots += (int64_t)W * cx::kBlock;
            CX_REQUIRE(h, slots < (int64_t)0x7fffff00, CX_ERR_UNSUPPORTED, "cx_graph_create: slot space exceeds 2^31");
        }
        h->slice_off[h->nslices] = (int32_t)slots;
        h->big_start = (int32_t)slots;
        for (int32_t v : h->big_vars) {
            h->vbase[v] = (int32_t)slots;
            for (int32_t k = 0; k < var_deg[v]; k++) h->big_slots.push_back((int32_t)slots + k);
            slots += var_deg[v];
            CX_REQUIRE(h, slots < (int64_t)0x7fffff00, CX_ERR_UNSUPPORTED, "cx_graph_create: slot space exceeds 2^31");
        }
        h->nslots = slots;
        const int64_t big_total = slots - h->big_start;
        // ---- factors ------------------------------------------------------------------------------------------------
        std::vector<int64_t> ford(n_factors);
        std::iota(ford.begin(), ford.end(), 0);
        bool fsorted = true;
        for (int64_t f = 1; f < n_factors && fsorted; f++) fsorted = factor_ids[f - 1] < factor_ids[f];
        if (!fsorted) std::sort(ford.begin(), ford.end(), [&](int64_t a, int64_t b) { return factor_ids[a] < factor_ids[b]; });
        h->fac_ids.resize(n_factors); h->fac_kind.resize(n_factors); h->fac_params.resize(n_factors * CX_NPARAM);
        for (int64_t f = 0; f < n_factors; f++) {
            h->fac_ids[f] = factor_ids[ford[f]];
            if (f > 0 && h->fac_ids[f] == h->fac_ids[f - 1]) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: duplicate factor id");
            h->fac_kind[f] = factor_kind[ford[f]];
            for (int k = 0; k < CX_NPARAM; k++) h->fac_params[f * CX_NPARAM + k] = factor_params[ford[f] * CX_NPARAM + k];
        }
        h->nf = n_factors;
        h->lin_out_is_second.assign(n_factors, 0); h->fac_edges.clear();
        std::vector<int32_t> edge_fac(ne);
        for (int64_t e = 0; e < ne; e++) {
            auto it = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), h->edge_fac_id[e]);
            if (it == h->fac_ids.end() || *it != h->edge_fac_id[e]) return fail(h, CX_ERR_NOT_FOUND, "cx_graph_create: edge names a factor id missing from factor_ids");
            edge_fac[e] = (int32_t)(it - h->fac_ids.begin());
        }
        // factor CSR by counting sort (edges of one factor come out in ascending variable order)
        std::vector<int32_t> foff(n_factors + 1, 0);
        for (int64_t e = 0; e < ne; e++) foff[edge_fac[e] + 1]++;
        for (int64_t f = 0; f < n_factors; f++) foff[f + 1] += foff[f];
        std::vector<int32_t> fedge(ne), fill(foff.begin(), foff.end() - 1);
        for (int64_t e = 0; e < ne; e++) fedge[fill[edge_fac[e]]++] = (int32_t)e;
        // ---- per-slot rule parameters and partners (the gather lists of dependencies.jl:17-31) ----------------------
        h->partner.assign(slots, -1);
        const bool mv = h->cfg.dim > 1;
        std::vector<int32_t> spdir;
        if (mv) {
            spdir.assign(slots, 0);
            if (!h->big_vars.empty())
                return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_create: dim > 1 handles variables of degree <= 8 only; variable " +
                            std::to_string(h->var_ids[h->big_vars[0]]) + " has more");
            for (int64_t v = 0; v < nv; v++)
                if (var_deg[v] > 4)
                    return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_create: dim > 1 handles variables of degree <= 4 in this build; variable " +
                                std::to_string(h->var_ids[v]) + " has degree " + std::to_string(var_deg[v]));
        }
        std::vector<double> q(mv ? 1 : slots, 0.0), a, b, sq, sa, sb;
        h->any_linear = false;
        for (int64_t f = 0; f < n_factors; f++) if (h->fac_kind[f] == CX_FACTOR_GAUSS_LINEAR) h->any_linear = true;
        if (h->any_linear) { a.assign(slots, 1.0); b.assign(slots, 0.0); sq.assign(slots, 0.0); sa.assign(slots, 1.0); sb.assign(slots, 0.0); }
        for (int64_t f = 0; f < n_factors; f++) {
            const int32_t deg = foff[f + 1] - foff[f], kind = h->fac_kind[f];
            const double *p = &h->fac_params[f * CX_NPARAM];
            if (kind == CX_FACTOR_OPAQUE) continue;
            if (h->cfg.family != CX_FAMILY_GAUSSIAN) {
                // the one device rule of the generic 2-parameter family: Bernoulli likelihood with an observed outcome
                if (kind != CX_FACTOR_BERNOULLI)
                    return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_create: CX_FAMILY_NATURAL2 factors are CX_FACTOR_OPAQUE or CX_FACTOR_BERNOULLI");
                if (deg != 2) return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_create: CX_FACTOR_BERNOULLI needs exactly 2 edges (factor id " + std::to_string(h->fac_ids[f]) + ")");
                const int32_t b1 = cx::slot_of_edge(h, fedge[foff[f]]), b2 = cx::slot_of_edge(h, fedge[foff[f] + 1]);
                h->partner[b1] = b2; h->partner[b2] = b1;
                continue;
            }
            if (kind != CX_FACTOR_GAUSS_ADDITIVE && kind != CX_FACTOR_GAUSS_LINEAR)
                return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_create: unknown factor kind");
            if (deg != 2) return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_create: Gaussian factor kinds need exactly 2 edges (factor id " + std::to_string(h->fac_ids[f]) + ")");
            const int32_t e1 = fedge[foff[f]], e2 = fedge[foff[f] + 1];
            const int32_t s1 = cx::slot_of_edge(h, e1), s2 = cx::slot_of_edge(h, e2);
            h->partner[s1] = s2; h->partner[s2] = s1;
            if (mv) {
                // dim > 1: x_out = A x_in + N(0, Q), (A, Q) = parameter set params[0] (cx_set_factor_matrices).
                // spdir[sending slot] = 2 * pset + direction of the RECEIVING edge (0: receiver = out, 1: receiver = in)
                if (kind != CX_FACTOR_GAUSS_LINEAR) return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_create: dim > 1 supports CX_FACTOR_GAUSS_LINEAR (params[0] = parameter set) and CX_FACTOR_OPAQUE");
                const int64_t pset = (int64_t)p[0];
                if (pset < 0 || (double)pset != p[0]) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: params[0] must be a parameter-set index");
                const int32_t r1 = edge_role ? edge_role[ord[e1]] : CX_ROLE_OUT, r2 = edge_role ? edge_role[ord[e2]] : CX_ROLE_OUT;
                if (r1 == r2) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: GAUSS_LINEAR needs one ROLE_IN and one ROLE_OUT edge");
                const int32_t sin = r1 == CX_ROLE_IN ? s1 : s2, sout = r1 == CX_ROLE_IN ? s2 : s1;
                spdir[sin] = (int32_t)(2 * pset);       // sent by x_in, received on the out edge: forward
                spdir[sout] = (int32_t)(2 * pset + 1);  // sent by x_out, received on the in edge: backward
                h->max_pset = std::max<int64_t>(h->max_pset, pset);
                continue;
            }
            if (!(p[0] >= 0.0)) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: factor variance must be >= 0");
            q[s1] = q[s2] = p[0];
            if (kind == CX_FACTOR_GAUSS_LINEAR) {
                // x_out = a x_in + b + N(0,q): the edge with ROLE_IN carries x_in.  Effective parameters of the
                // RECEIVING edge: forward (receiver = out) {a, b, q}; backward (receiver = in) {1/a, -b/a, q/a²}.
                const double A = p[1], B = p[2];
                if (A == 0.0) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: GAUSS_LINEAR with a == 0");
                const int32_t r1 = edge_role ? edge_role[ord[e1]] : CX_ROLE_OUT, r2 = edge_role ? edge_role[ord[e2]] : CX_ROLE_OUT;
                if (r1 == r2) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: GAUSS_LINEAR needs one ROLE_IN and one ROLE_OUT edge");
                const int32_t sin = r1 == CX_ROLE_IN ? s1 : s2, sout = r1 == CX_ROLE_IN ? s2 : s1;
                h->lin_out_is_second[f] = sout == s2 ? 1 : 0;    // e1 < e2 in (variable, factor) order: "second" = the higher variable id
                a[sout] = A; b[sout] = B; q[sout] = p[0];
                a[sin] = 1.0 / A; b[sin] = -B / A; q[sin] = p[0] / (A * A);
            }
        }
        if (h->any_linear)
            for (int64_t s = 0; s < slots; s++)
                if (h->partner[s] >= 0) { sq[s] = q[h->partner[s]]; sa[s] = a[h->partner[s]]; sb[s] = b[h->partner[s]]; }
        // messages with >=1 dependency and >=1 listener (the metric's unit): both directions of every 2-edge Gaussian
        // factor, minus variable→factor messages of degree-1 variables (no dependencies, dependencies.jl:48-55)
        int64_t m = 0;
        for (int64_t e = 0; e < ne; e++) {
            if (h->partner[cx::slot_of_edge(h, e)] < 0) continue;
            m += 1;
            if (var_deg[h->edge_var[e]] >= 2) m += 1;
        }
        h->n_messages_per_sweep = m;
        // ---- upload ------------------------------------------------------------------------------------------------
        CX_HIP(h, hipSetDevice(h->cfg.device));
        int32_t rc;
#define CX_TRY(x) do { rc = (x); if (rc != CX_OK) { dev_free_all(h); return rc; } } while (0)
        CX_TRY(dev_upload(h, &h->d_slice_off, h->slice_off));
        CX_TRY(dev_upload(h, &h->d_partner, h->partner));
        CX_TRY(dev_upload(h, &h->d_vbase, h->vbase));
        CX_TRY(dev_upload(h, &h->d_var_deg, var_deg));
        CX_TRY(dev_upload(h, &h->d_vinfo, h->vinfo));
        CX_TRY(dev_upload(h, &h->d_big, h->big_vars));
        CX_TRY(dev_upload(h, &h->d_big_slots, h->big_slots));
        CX_TRY(dev_alloc(h, &h->d_big_tmp, big_total));
        CX_TRY(dev_alloc(h, &h->d_scratch, 4096));
        if (mv) {
            const int64_t nc = h->nc;
            h->spdir = spdir; h->spdir_dirty = true;
            CX_TRY(dev_upload(h, &h->d_spdir, spdir));
            CX_TRY(dev_alloc(h, &h->d_mv_f2v, nc * slots)); CX_TRY(dev_alloc(h, &h->d_mv_f2v_alt, nc * slots));
            CX_TRY(dev_alloc(h, &h->d_mv_v2f, nc * slots)); CX_TRY(dev_alloc(h, &h->d_mv_marg, h->cfg.dim == 64 ? 1 : nc * nv));
            CX_HIP(h, hipMemsetAsync(h->d_mv_f2v, 0xff, (size_t)(nc * slots) * 8, h->stream));
            CX_HIP(h, hipMemsetAsync(h->d_mv_f2v_alt, 0xff, (size_t)(nc * slots) * 8, h->stream));
            CX_HIP(h, hipMemsetAsync(h->d_mv_v2f, 0xff, (size_t)(nc * slots) * 8, h->stream));
            if (h->cfg.dim != 64) CX_HIP(h, hipMemsetAsync(h->d_mv_marg, 0xff, (size_t)(nc * nv) * 8, h->stream));
            CX_HIP(h, hipStreamSynchronize(h->stream));
            h->has_graph = true;
            return upload_ptab(h);
        }
        CX_TRY(dev_upload(h, &h->d_q, q));
        if (h->any_linear) {
            CX_TRY(dev_upload(h, &h->d_a, a)); CX_TRY(dev_upload(h, &h->d_b, b));
            CX_TRY(dev_upload(h, &h->d_sq, sq)); CX_TRY(dev_upload(h, &h->d_sa, sa)); CX_TRY(dev_upload(h, &h->d_sb, sb));
        }
        CX_TRY(dev_alloc(h, &h->d_f2v, slots)); CX_TRY(dev_alloc(h, &h->d_v2f, slots)); CX_TRY(dev_alloc(h, &h->d_marg, nv));
        // every message starts as UndefValue(): all-ones bytes are a NaN in both halves of each double2
        CX_HIP(h, hipMemsetAsync(h->d_f2v, 0xff, (size_t)slots * sizeof(double2), h->stream));
        CX_HIP(h, hipMemsetAsync(h->d_v2f, 0xff, (size_t)slots * sizeof(double2), h->stream));
        CX_HIP(h, hipMemsetAsync(h->d_marg, 0xff, (size_t)nv * sizeof(double2), h->stream));
        if (h->cfg.schedule == CX_SCHED_FUSED) {
            CX_TRY(dev_alloc(h, &h->d_f2v_alt, slots));
            CX_HIP(h, hipMemsetAsync(h->d_f2v_alt, 0xff, (size_t)slots * sizeof(double2), h->stream));
        }
#undef CX_TRY
        CX_HIP(h, hipStreamSynchronize(h->stream));
        h->has_graph = true; h->offchain_marg_dirty = true;
        return CX_OK;
    } catch (const std::bad_alloc &) {
        return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_graph_create: host allocation failed");
    } catch (const std::exception &ex) {
        return fail(h, CX_ERR_INVALID_ARGUMENT, std::string("cx_graph_create: ") + ex.what());
    }
}

int32_t cx_graph_stats(const cx_handle *h, cx_stats *out) {
    if (!h || !out) return CX_ERR_INVALID_ARGUMENT;
    out->n_variables = h->nv; out->n_factors = h->nf; out->n_edges = h->ne;
    out->n_messages_per_sweep = h->n_messages_per_sweep;
    out->n_slices = h->nslices; out->n_big_variables = (int64_t)h->big_vars.size(); out->n_slots = h->nslots;
    out->device_bytes = h->device_bytes; out->sweeps_done = h->sweeps_done;
    return CX_OK;
}

// tiles of the two-sweep launches (built on the first cx_sweep(n >= 2) of a fused scalar handle)
int32_t cx_tile_stats(const cx_handle *h, int64_t *n_tiles, double *variables_loaded_per_owned, int64_t *lds_bytes_per_workgroup) {
    if (!h) return CX_ERR_INVALID_ARGUMENT;
    if (n_tiles) *n_tiles = h->tiles_state > 0 ? h->n_tiles : 0;
    if (variables_loaded_per_owned) *variables_loaded_per_owned = h->tiles_state > 0 ? h->tile_redundancy : 0.0;
    if (lds_bytes_per_workgroup) *lds_bytes_per_workgroup = h->tiles_state > 0 ? h->tile_lds : 0;
    return CX_OK;
}

int32_t cx_edge_index(const cx_handle *hc, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, int64_t *out_edge) {
    cx_handle *h = const_cast<cx_handle *>(hc);
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_edge_index: no graph");
    CX_REQUIRE(h, n >= 0 && (n == 0 || (variable_ids && factor_ids && out_edge)), CX_ERR_INVALID_ARGUMENT, "cx_edge_index: null argument");
    for (int64_t i = 0; i < n; i++) {
        out_edge[i] = find_edge(h, variable_ids[i], factor_ids[i]);
        if (out_edge[i] < 0)
            return fail(h, CX_ERR_NOT_FOUND, "no connection between variable " + std::to_string(variable_ids[i]) + " and factor " + std::to_string(factor_ids[i]));
    }
    return CX_OK;
}

// (variable_id, factor_id) lists -> slots (+ optionally the local variable numbers)
namespace {
int32_t stage_slots(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, std::vector<int32_t> &slots,
                std::vector<int32_t> *vars) {
    slots.resize(n);
    if (vars) vars->resize(n);
    for (int64_t i = 0; i < n; i++) {
        int64_t e = find_edge(h, variable_ids[i], factor_ids[i]);
        if (e < 0) return fail(h, CX_ERR_NOT_FOUND, "no connection between variable " + std::to_string(variable_ids[i]) + " and factor " + std::to_string(factor_ids[i]));
        slots[i] = cx::slot_of_edge(h, e);
        if (vars) (*vars)[i] = h->edge_var[e];
    }
    return CX_OK;
}
}  // namespace


// ==================================================================================================================
// dim > 1 (cx_mv.hip): host side of the data path.  Payload rows at the ABI: MOMENT mean[d] + covariance[d*d];
// NATURAL eta[d] + Lambda[d*d]; POINT y[d].  Device: eta[d] + packed upper triangle of Lambda, component-major.
// ==================================================================================================================
namespace {

void mv_pack(int d, const double *eta, const double *lam_full, double *out) {
    for (int i = 0; i < d; i++) out[i] = eta[i];
    int c = d;
    for (int i = 0; i < d; i++) for (int j = i; j < d; j++) out[c++] = 0.5 * (lam_full[i * d + j] + lam_full[j * d + i]);
}

void mv_unpack(int d, const double *in, double *eta, double *lam_full) {
    for (int i = 0; i < d; i++) eta[i] = in[i];
    int c = d;
    for (int i = 0; i < d; i++) for (int j = i; j < d; j++) { lam_full[i * d + j] = in[c]; lam_full[j * d + i] = in[c]; c++; }
}

bool mv_to_natural(int d, int32_t form, const double *p, double *out /* nc */) {
    const int nc = d + d * (d + 1) / 2;
    if (form == CX_FORM_POINT) {
        for (int i = 0; i < nc; i++) out[i] = 0.0;
        for (int i = 0; i < d; i++) out[i] = p[i];
        out[d] = kInf;
        return true;
    }
    if (form == CX_FORM_NATURAL) { mv_pack(d, p, p + d, out); return true; }
    bool undef = false;
    for (int i = 0; i < d * d; i++) undef = undef || std::isnan(p[d + i]);
    if (undef) { for (int i = 0; i < nc; i++) out[i] = kNaN; return true; }
    std::vector<double> lam((size_t)d * d), eta(d);
    if (!cx::spd_inverse(d, p + d, lam.data())) return false;
    for (int i = 0; i < d; i++) { double s = 0; for (int j = 0; j < d; j++) s += lam[i * d + j] * p[j]; eta[i] = s; }
    mv_pack(d, eta.data(), lam.data(), out);
    return true;
}

void mv_from_natural(int d, int32_t form, const double *in /* nc */, double *out /* d + d*d */) {
    std::vector<double> lam((size_t)d * d), eta(d);
    mv_unpack(d, in, eta.data(), lam.data());
    if (form == CX_FORM_NATURAL) { for (int i = 0; i < d; i++) out[i] = eta[i]; for (int i = 0; i < d * d; i++) out[d + i] = lam[i]; return; }
    if (std::isnan(in[d])) { for (int i = 0; i < d + d * d; i++) out[i] = kNaN; return; }
    if (in[d] == kInf) { for (int i = 0; i < d; i++) out[i] = eta[i]; for (int i = 0; i < d * d; i++) out[d + i] = 0.0; return; }
    std::vector<double> cov((size_t)d * d);
    if (!cx::spd_inverse(d, lam.data(), cov.data())) { for (int i = 0; i < d + d * d; i++) out[i] = kNaN; return; }
    for (int i = 0; i < d; i++) { double s = 0; for (int j = 0; j < d; j++) s += cov[i * d + j] * eta[j]; out[i] = s; }
    for (int i = 0; i < d * d; i++) out[d + i] = cov[i];
}

// variable→factor messages are never stored by the dim > 1 sweep: recompute the requested ones from the input buffer
// the last sweep read (retained in d_mv_f2v_alt after the swap)
int32_t mv_refresh_v2f(cx_handle *h, const std::vector<int32_t> &slots, const std::vector<int32_t> &vars) {
    const int64_t n = (int64_t)slots.size();
    int32_t rc = ensure_stage(h, n * 8);
    if (rc != CX_OK) return rc;
    int32_t *d_s = (int32_t *)h->d_stage, *d_v = d_s + n;
    CX_HIP(h, hipMemcpyAsync(d_s, slots.data(), n * 4, hipMemcpyHostToDevice, h->stream));
    CX_HIP(h, hipMemcpyAsync(d_v, vars.data(), n * 4, hipMemcpyHostToDevice, h->stream));
    if (h->cfg.dim == 64) cx::mv64_launch_v2f(h, (int)n, d_s, d_v, h->sweeps_done > 0 ? h->d_mv_f2v_alt : h->d_mv_f2v);
    else cx::mv_launch_v2f(h, d_s, d_v, n, h->sweeps_done > 0 ? h->d_mv_f2v_alt : h->d_mv_f2v);
    CX_HIP(h, hipGetLastError());
    CX_HIP(h, hipStreamSynchronize(h->stream));
    return CX_OK;
}

int32_t mv64_set_messages(cx_handle *h, int64_t n, const std::vector<int32_t> &idx, const std::vector<int32_t> &vars, int32_t direction,
                          int32_t form, const double *payload) {
    const int d = 64, nc = h->nc;
    const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16;
    int32_t rc;
    if (form == CX_FORM_POINT) {
        rc = ensure_stage(h, bytes_idx + n * d * 8);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double *d_val = (double *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        CX_HIP(h, hipMemcpyAsync(d_val, payload, (size_t)n * d * 8, hipMemcpyHostToDevice, h->stream));
        cx::mv64_set_point(h, h->d_mv_v2f, d_idx, d_val, n);
        for (int64_t i = 0; i < n; i++) h->vinfo[vars[i]] |= cx::kClamped;
        CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream)); h->tile_info_dirty = true;
        h->work64_dirty = h->point64_dirty = true;
    } else {
        std::vector<double> val((size_t)n * nc);
        for (int64_t i = 0; i < n; i++) {
            const double *p = payload + i * (d + d * d);
            double *o = &val[(size_t)i * nc];
            if (form == CX_FORM_NATURAL) { std::memcpy(o, p, (size_t)nc * 8); continue; }
            bool undef = false;
            for (int k = 0; k < d * d; k++) undef = undef || std::isnan(p[d + k]);
            if (undef) { for (int k = 0; k < nc; k++) o[k] = kNaN; continue; }
            if (!cx::spd_inverse(d, p + d, o + d))
                return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_messages: covariance of row " + std::to_string(i) + " is not positive definite");
            for (int r = 0; r < d; r++) { double s = 0; for (int c = 0; c < d; c++) s += o[d + r * d + c] * p[c]; o[r] = s; }
        }
        rc = ensure_stage(h, bytes_idx + n * nc * 8);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double *d_val = (double *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        CX_HIP(h, hipMemcpyAsync(d_val, val.data(), (size_t)n * nc * 8, hipMemcpyHostToDevice, h->stream));
        if (direction == CX_TO_FACTOR) { cx::mv64_rows_scatter(h, h->d_mv_v2f, d_idx, d_val, n); h->point64_dirty = true; }
        else { cx::mv64_rows_scatter(h, h->d_mv_f2v, d_idx, d_val, n); cx::mv64_rows_scatter(h, h->d_mv_f2v_alt, d_idx, d_val, n); }
    }
    CX_HIP(h, hipGetLastError());
    CX_HIP(h, hipStreamSynchronize(h->stream));
    return CX_OK;
}

int32_t mv_set_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, int32_t direction,
                        int32_t form, const double *payload) {
    const int d = h->cfg.dim, nc = h->nc;
    std::vector<int32_t> idx, vars;
    int32_t rc = stage_slots(h, n, variable_ids, factor_ids, idx, &vars);
    if (rc != CX_OK) return rc;
    if (d == 64) return mv64_set_messages(h, n, idx, vars, direction, form, payload);
    const int64_t stride = form == CX_FORM_POINT ? d : d + d * d;
    std::vector<double> val((size_t)n * nc);
    for (int64_t i = 0; i < n; i++)
        if (!mv_to_natural(d, form, payload + i * stride, &val[(size_t)i * nc]))
            return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_messages: covariance of row " + std::to_string(i) + " is not positive definite");
    const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16;
    rc = ensure_stage(h, bytes_idx + n * nc * 8);
    if (rc != CX_OK) return rc;
    int32_t *d_idx = (int32_t *)h->d_stage;
    double *d_val = (double *)((char *)h->d_stage + bytes_idx);
    CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
    CX_HIP(h, hipMemcpyAsync(d_val, val.data(), (size_t)n * nc * 8, hipMemcpyHostToDevice, h->stream));
    if (direction == CX_TO_FACTOR) {
        cx::mv_launch_scatter(h, h->d_mv_v2f, h->nslots, nc, d_idx, d_val, n);
        h->observed_passes_due = 2;   // a stored variable→factor message changed: observed senders are refreshed
        if (form == CX_FORM_POINT) {
            for (int64_t i = 0; i < n; i++) h->vinfo[vars[i]] |= cx::kClamped;
            CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream)); h->tile_info_dirty = true;
            h->spdir_dirty = true;
        }
    } else {
        cx::mv_launch_scatter(h, h->d_mv_f2v, h->nslots, nc, d_idx, d_val, n);
        cx::mv_launch_scatter(h, h->d_mv_f2v_alt, h->nslots, nc, d_idx, d_val, n);
    }
    CX_HIP(h, hipGetLastError());
    CX_HIP(h, hipStreamSynchronize(h->stream));
    return CX_OK;
}

int32_t mv_get(cx_handle *h, const double *src, int64_t stride, const std::vector<int32_t> &idx, int32_t form, bool already_moment,
               double *out) {
    const int d = h->cfg.dim, nc = h->nc;
    const int64_t n = (int64_t)idx.size();
    const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16;
    int32_t rc = ensure_stage(h, bytes_idx + n * nc * 8);
    if (rc != CX_OK) return rc;
    int32_t *d_idx = (int32_t *)h->d_stage;
    double *d_val = (double *)((char *)h->d_stage + bytes_idx);
    CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
    if (d == 64) {
        if (already_moment) cx::mv64_launch_marginals(h, (int)n, d_idx, h->d_mv_f2v, d_val);   // idx = variables: computed on demand
        else cx::mv64_rows_gather(h, src, d_idx, d_val, n);
        CX_HIP(h, hipGetLastError());
        std::vector<double> val((size_t)n * nc);
        CX_HIP(h, hipMemcpyAsync(val.data(), d_val, (size_t)n * nc * 8, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < n; i++) {
            double *o = out + i * nc;
            const double *in = &val[(size_t)i * nc];
            if (already_moment || form == CX_FORM_NATURAL) { std::memcpy(o, in, (size_t)nc * 8); continue; }
            if (std::isnan(in[d])) { for (int k = 0; k < nc; k++) o[k] = kNaN; continue; }
            if (in[d] == kInf) { for (int k = 0; k < nc; k++) o[k] = 0.0; for (int k = 0; k < d; k++) o[k] = in[k]; continue; }
            if (!cx::spd_inverse(d, in + d, o + d)) { for (int k = 0; k < nc; k++) o[k] = kNaN; continue; }
            for (int r = 0; r < d; r++) { double s = 0; for (int c = 0; c < d; c++) s += o[d + r * d + c] * in[c]; o[r] = s; }
        }
        return CX_OK;
    }
    cx::mv_launch_gather(h, src, stride, nc, d_idx, d_val, n);
    std::vector<double> val((size_t)n * nc);
    CX_HIP(h, hipMemcpyAsync(val.data(), d_val, (size_t)n * nc * 8, hipMemcpyDeviceToHost, h->stream));
    CX_HIP(h, hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < n; i++) {
        double *o = out + i * (d + d * d);
        if (already_moment) mv_unpack(d, &val[(size_t)i * nc], o, o + d);
        else mv_from_natural(d, form, &val[(size_t)i * nc], o);
    }
    return CX_OK;
}

int32_t mv_get_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, int32_t direction,
                        int32_t form, double *out) {
    std::vector<int32_t> idx, vars;
    int32_t rc = stage_slots(h, n, variable_ids, factor_ids, idx, &vars);
    if (rc != CX_OK) return rc;
    if (direction == CX_TO_FACTOR) { rc = mv_refresh_v2f(h, idx, vars); if (rc != CX_OK) return rc; }
    return mv_get(h, direction == CX_TO_FACTOR ? h->d_mv_v2f : h->d_mv_f2v, h->nslots, idx, form, false, out);
}

int32_t mv_get_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, double *out) {
    std::vector<int32_t> idx(n);
    for (int64_t i = 0; i < n; i++) {
        int64_t v = find_var(h, variable_ids[i]);
        if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
        idx[i] = (int32_t)v;
    }
    return mv_get(h, h->d_mv_marg, h->nv, idx, CX_FORM_MOMENT, true, out);
}

// d = 64: which messages need the full MFMA rule, which come from observed variables (constant), which nobody reads
int32_t build_work64(cx_handle *h) {
    if (!h->work64_dirty) return CX_OK;
    std::vector<int32_t> rs, rv, rf, ps, rec, slot_var(h->nslots, -1);
    for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
    for (int64_t e = 0; e < h->ne; e++) {
        const int32_t s = cx::slot_of_edge(h, e), p = h->partner[s], v = h->edge_var[e];
        if (p < 0) continue;
        const int32_t rvz = slot_var[p];
        if (h->vinfo[rvz] & cx::kClamped) continue;                 // a message into an observed variable has no reader
        const int32_t deg = h->var_off[v + 1] - h->var_off[v];
        if (h->vinfo[v] & cx::kClamped) { ps.push_back(s); continue; }
        rs.push_back(s); rv.push_back(v); rf.push_back(deg < 2 ? 1 : 0);   // degree-1 leaf: its stored message is the input
        // the record the rule kernel reads: sender slot, the other incoming slots in ascending neighbour order (the fold
        // order), rule-table index, destination slot, flags
        int32_t others[3] = {-1, -1, -1};
        int n_others = 0;
        for (int32_t j = 0; j < deg; j++) {
            const int32_t sj = h->vbase[v] + j * cx::kBlock;
            if (sj != s && n_others < 3) others[n_others++] = sj;
        }
        rec.insert(rec.end(), {s, others[0], others[1], others[2], h->spdir[s], p, deg < 2 ? 1 : 0, 0});
    }
    for (void *p : {(void *)h->d_rule64_slots, (void *)h->d_rule64_vars, (void *)h->d_rule64_flags, (void *)h->d_point64_slots, (void *)h->d_rule64_rec}) if (p) (void)hipFree(p);
    h->d_rule64_slots = h->d_rule64_vars = h->d_rule64_flags = h->d_point64_slots = h->d_rule64_rec = nullptr;
    h->n_rule64 = (int64_t)rs.size(); h->n_point64 = (int64_t)ps.size();
    int32_t rc;
    if ((rc = dev_upload(h, &h->d_rule64_slots, rs)) != CX_OK) return rc;
    if ((rc = dev_upload(h, &h->d_rule64_vars, rv)) != CX_OK) return rc;
    if ((rc = dev_upload(h, &h->d_rule64_flags, rf)) != CX_OK) return rc;
    if ((rc = dev_upload(h, &h->d_point64_slots, ps)) != CX_OK) return rc;
    if ((rc = dev_upload(h, &h->d_rule64_rec, rec)) != CX_OK) return rc;
    CX_HIP(h, hipStreamSynchronize(h->stream));
    h->work64_dirty = false;
    h->point64_dirty = true;
    return CX_OK;
}

int32_t mv_sweep(cx_handle *h, int32_t n_sweeps) {
    CX_REQUIRE(h, (int64_t)h->psets.size() > h->max_pset, CX_ERR_STATE, "cx_sweep: a factor names a parameter set that was never set (cx_set_factor_matrices)");
    for (int64_t i = 0; i <= h->max_pset; i++)
        CX_REQUIRE(h, !h->psets[i].empty(), CX_ERR_STATE, "cx_sweep: parameter set " + std::to_string(i) + " was never set (cx_set_factor_matrices)");
    if (h->cfg.dim == 64) {
        int32_t rc = build_work64(h);
        if (rc != CX_OK) return rc;
        if (h->point64_dirty) {   // messages out of observed variables are constant: computed once, into both buffers
            cx::mv64_launch_point(h, (int)h->n_point64, h->d_point64_slots, h->d_mv_f2v, h->d_mv_f2v_alt);
            h->point64_dirty = false;
        }
    }
    if (h->cfg.dim != 64 && h->spdir_dirty) {   // mask the rules whose receiver is an observed variable
        std::vector<int32_t> eff(h->spdir), slot_var(h->nslots, -1);
        for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
        for (int64_t sl = 0; sl < h->nslots; sl++) {
            const int32_t p = h->partner[sl];
            if (p >= 0 && (h->vinfo[slot_var[p]] & cx::kClamped)) eff[sl] = -1;
        }
        CX_HIP(h, hipMemcpy(h->d_spdir, eff.data(), eff.size() * 4, hipMemcpyHostToDevice));
        h->spdir_dirty = false;
        h->observed_passes_due = 2;   // the data (or the set of observed variables) changed: refresh both buffers
    }
    for (int32_t s = 0; s < n_sweeps; s++) {
        if (h->cfg.dim == 64)
            cx::mv64_launch_rule(h, (int)h->n_rule64, h->d_rule64_rec, h->d_mv_f2v, h->d_mv_f2v_alt, CX_KERNEL_FUSED);
        else {
            if (h->observed_passes_due > 0) { cx::mv_launch_sweep(h, false, true); h->observed_passes_due--; }
            cx::mv_launch_sweep(h, h->cfg.compute_marginals_in_sweep != 0, false);
        }
        std::swap(h->d_mv_f2v, h->d_mv_f2v_alt);
        h->sweeps_done++;
    }
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

int32_t mv_residual(cx_handle *h, double *out) {
    const int64_t n = h->nc * h->nslots;
    if (!h->d_mv_prev) {
        int32_t rc = dev_alloc(h, &h->d_mv_prev, n);
        if (rc != CX_OK) return rc;
        CX_HIP(h, hipMemcpyAsync(h->d_mv_prev, h->d_mv_f2v, (size_t)n * 8, hipMemcpyDeviceToDevice, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        *out = kInf;
        return CX_OK;
    }
    cx::mv_launch_residual(h, h->d_mv_f2v, h->d_mv_prev, n, h->d_scratch);
    std::vector<double> part(1024);
    CX_HIP(h, hipMemcpyAsync(part.data(), h->d_scratch, 1024 * 8, hipMemcpyDeviceToHost, h->stream));
    CX_HIP(h, hipMemcpyAsync(h->d_mv_prev, h->d_mv_f2v, (size_t)n * 8, hipMemcpyDeviceToDevice, h->stream));
    CX_HIP(h, hipStreamSynchronize(h->stream));
    double m = 0.0;
    for (double p : part) m = (p != p) ? kInf : std::max(m, p);
    *out = m;
    return CX_OK;
}

}  // namespace

// In the fused schedule without materialisation the variable→factor messages of the last sweep exist only as
// "leave-one-out of the sweep's input buffer", which is retained in d_f2v_alt: recompute them on demand.
// After a two-sweep launch (cx_tiles.hip) the retained buffer d_f2v_alt holds time t while d_f2v holds t+2: the buffer of
// time t+1 — the input of the last sweep, which is what variable→factor messages and checkpoints are defined from — never
// existed.  Regenerate it with one plain sweep from time t into a third buffer and make that the retained buffer.
static int32_t normalize_alt(cx_handle *h) {
    if (!h->alt_two_back) return CX_OK;
    if (!h->d_f2v_tmp) { int32_t rc = dev_alloc(h, &h->d_f2v_tmp, h->nslots); if (rc != CX_OK) return rc; }
    // slots no sweep writes (priors of unary factors, padding) are equal in every buffer: start from a copy
    CX_HIP(h, hipMemcpyAsync(h->d_f2v_tmp, h->d_f2v_alt, (size_t)h->nslots * sizeof(double2), hipMemcpyDeviceToDevice, h->stream));
    const bool prof = h->profiling;
    h->profiling = false;
    cx::launch_fused(h, h->d_f2v_alt, h->d_f2v_tmp, false, false, false);
    h->profiling = prof;
    CX_HIP(h, hipGetLastError());
    std::swap(h->d_f2v_alt, h->d_f2v_tmp);
    h->alt_two_back = false;
    return CX_OK;
}

static int32_t ensure_v2f(cx_handle *h) {
    if (!h->v2f_stale) return CX_OK;
    { int32_t rc = normalize_alt(h); if (rc != CX_OK) return rc; }
    const double2 *src = h->d_f2v_alt ? h->d_f2v_alt : h->d_f2v;
    cx::launch_var_to_factor(h, src, false);
    cx::launch_big_var_to_factor(h, src, false);
    CX_HIP(h, hipGetLastError());
    h->v2f_stale = false;
    return CX_OK;
}

int32_t cx_set_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, int32_t direction,
                        int32_t form, const double *payload) {
    // data injection (variable→factor messages of observed variables) changes the chains' leaf messages but no marginal of a
    // variable off the chains: those depend on stored factor→variable messages only
    if (h) { h->chain_side_dirty = true; if (direction != CX_TO_FACTOR) h->offchain_marg_dirty = true; }
    CX_NOT_VMP(h, "cx_set_messages");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_set_messages: no graph");
    CX_REQUIRE(h, direction == CX_TO_FACTOR || direction == CX_TO_VARIABLE, CX_ERR_INVALID_ARGUMENT, "cx_set_messages: bad direction");
    CX_REQUIRE(h, form == CX_FORM_MOMENT || form == CX_FORM_POINT || form == CX_FORM_NATURAL, CX_ERR_INVALID_ARGUMENT, "cx_set_messages: bad form");
    CX_REQUIRE(h, !(form == CX_FORM_POINT && direction == CX_TO_VARIABLE), CX_ERR_UNSUPPORTED, "cx_set_messages: point-mass data is a variable→factor message");
    CX_REQUIRE(h, h->cfg.family == CX_FAMILY_GAUSSIAN || form == CX_FORM_NATURAL || form == CX_FORM_POINT, CX_ERR_UNSUPPORTED,
               "cx_set_messages: CX_FAMILY_NATURAL2 takes CX_FORM_NATURAL payloads (and CX_FORM_POINT data)");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && factor_ids && payload, CX_ERR_INVALID_ARGUMENT, "cx_set_messages: null argument");
    if (h->cfg.dim > 1) { try { return mv_set_messages(h, n, variable_ids, factor_ids, direction, form, payload); } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_messages: host allocation failed"); } }
    try {
        std::vector<int32_t> idx, vars;
        int32_t rc = stage_slots(h, n, variable_ids, factor_ids, idx, &vars);
        if (rc != CX_OK) return rc;
        const int64_t stride = form == CX_FORM_POINT ? 1 : 2;
        std::vector<double2> val(n);
        for (int64_t i = 0; i < n; i++) to_natural(form, payload + i * stride, &val[i]);
        const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16, bytes = bytes_idx + n * 16;
        rc = ensure_stage(h, bytes);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double2 *d_val = (double2 *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        CX_HIP(h, hipMemcpyAsync(d_val, val.data(), n * 16, hipMemcpyHostToDevice, h->stream));
        if (direction == CX_TO_FACTOR) {
            rc = ensure_v2f(h);
            if (rc != CX_OK) return rc;
            cx::launch_scatter(h, h->d_v2f, d_idx, d_val, n);
            if (form == CX_FORM_POINT) {
                // a variable that carries a point-mass datum is observed: its messages are never recomputed.  New data for variables
                // that were observed already leaves the structure (chains, tiles) as it is.
                bool newly = false;
                for (int64_t i = 0; i < n; i++)
                    if (!(h->vinfo[vars[i]] & cx::kClamped)) { h->vinfo[vars[i]] |= cx::kClamped; newly = true; }
                if (newly) {
                    h->chains_dirty = true; h->offchain_marg_dirty = true;
                    CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream)); h->tile_info_dirty = true;
                }
            }
        } else {
            cx::launch_scatter(h, h->d_f2v, d_idx, d_val, n);
            if (h->d_f2v_alt) cx::launch_scatter(h, h->d_f2v_alt, d_idx, d_val, n);
        }
        CX_HIP(h, hipGetLastError());
        CX_HIP(h, hipStreamSynchronize(h->stream));  // host staging vectors die here
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_messages: host allocation failed"); }
}

int32_t cx_get_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, int32_t direction,
                        int32_t form, double *out) {
    CX_NOT_VMP(h, "cx_get_messages");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_get_messages: no graph");
    CX_REQUIRE(h, direction == CX_TO_FACTOR || direction == CX_TO_VARIABLE, CX_ERR_INVALID_ARGUMENT, "cx_get_messages: bad direction");
    CX_REQUIRE(h, form == CX_FORM_MOMENT || form == CX_FORM_NATURAL, CX_ERR_INVALID_ARGUMENT, "cx_get_messages: bad form");
    CX_REQUIRE(h, h->cfg.family == CX_FAMILY_GAUSSIAN || form == CX_FORM_NATURAL, CX_ERR_UNSUPPORTED, "cx_get_messages: CX_FAMILY_NATURAL2 returns CX_FORM_NATURAL payloads only");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && factor_ids && out, CX_ERR_INVALID_ARGUMENT, "cx_get_messages: null argument");
    if (h->cfg.dim > 1) { try { return mv_get_messages(h, n, variable_ids, factor_ids, direction, form, out); } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_messages: host allocation failed"); } }
    try {
        std::vector<int32_t> idx;
        int32_t rc = stage_slots(h, n, variable_ids, factor_ids, idx, nullptr);
        if (rc != CX_OK) return rc;
        if (direction == CX_TO_FACTOR) { rc = ensure_v2f(h); if (rc != CX_OK) return rc; }
        const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16, bytes = bytes_idx + n * 16;
        rc = ensure_stage(h, bytes);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double2 *d_val = (double2 *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        cx::launch_gather(h, direction == CX_TO_FACTOR ? h->d_v2f : h->d_f2v, d_idx, d_val, n);
        std::vector<double2> val(n);
        CX_HIP(h, hipMemcpyAsync(val.data(), d_val, n * 16, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < n; i++) from_natural(form, val[i], out + 2 * i);
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_messages: host allocation failed"); }
}

int32_t cx_seed_messages(cx_handle *h, int32_t direction, double mean, double variance) {
    if (h) { h->chain_side_dirty = true; h->offchain_marg_dirty = true; }
    CX_NOT_VMP(h, "cx_seed_messages");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_seed_messages: no graph");
    CX_REQUIRE(h, direction == CX_TO_FACTOR || direction == CX_TO_VARIABLE, CX_ERR_INVALID_ARGUMENT, "cx_seed_messages: bad direction");
    CX_REQUIRE(h, variance > 0.0, CX_ERR_INVALID_ARGUMENT, "cx_seed_messages: variance must be > 0");
    if (h->cfg.dim > 1) {
        CX_REQUIRE(h, direction == CX_TO_VARIABLE, CX_ERR_UNSUPPORTED, "cx_seed_messages: dim > 1 seeds factor→variable messages only");
        if (h->cfg.dim == 64) {
            cx::mv64_launch_seed(h, h->d_mv_f2v, mean / variance, 1.0 / variance);
            cx::mv64_launch_seed(h, h->d_mv_f2v_alt, mean / variance, 1.0 / variance);
        } else {
            cx::mv_launch_seed(h, h->d_mv_f2v, mean / variance, 1.0 / variance);
            cx::mv_launch_seed(h, h->d_mv_f2v_alt, mean / variance, 1.0 / variance);
        }
        CX_HIP(h, hipGetLastError());
        return CX_OK;
    }
    double2 v = make_double2(mean / variance, 1.0 / variance);
    if (direction == CX_TO_VARIABLE) {
        cx::launch_seed(h, h->d_f2v, h->nslots, v, h->d_partner);
        if (h->d_f2v_alt) cx::launch_seed(h, h->d_f2v_alt, h->nslots, v, h->d_partner);
    } else {
        int32_t rc = ensure_v2f(h);
        if (rc != CX_OK) return rc;
        cx::launch_seed(h, h->d_v2f, h->nslots, v, h->d_partner);
    }
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

int32_t cx_get_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, double *out) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_get_marginals: no graph");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && out, CX_ERR_INVALID_ARGUMENT, "cx_get_marginals: null argument");
    if (is_vmp(h)) return cx::vmp_get_marginals(h, n, variable_ids, out);
    if (h->cfg.dim > 1) { try { return mv_get_marginals(h, n, variable_ids, out); } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_marginals: host allocation failed"); } }
    try {
        std::vector<int32_t> idx(n);
        for (int64_t i = 0; i < n; i++) {
            int64_t v = find_var(h, variable_ids[i]);
            if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
            idx[i] = (int32_t)v;
        }
        const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16, bytes = bytes_idx + n * 16;
        int32_t rc = ensure_stage(h, bytes);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double2 *d_val = (double2 *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        cx::launch_gather(h, h->d_marg, d_idx, d_val, n);
        CX_HIP(h, hipMemcpyAsync(out, d_val, n * 16, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_marginals: host allocation failed"); }
}

// grow a device store to hold `need` records of `per_record` elements, keeping its contents; new records read as UndefValue()
extern "C++" {
template <class T>
static int32_t grow_store(cx_handle *h, T **buf, int64_t *cap, int64_t need, int64_t per_record) {
    if (need <= *cap) return CX_OK;
    int64_t ncap = std::max<int64_t>(need, std::max<int64_t>(256, *cap * 2));
    T *nb = nullptr;
    CX_HIP(h, hipMalloc((void **)&nb, (size_t)(ncap * per_record) * sizeof(T)));
    hipError_t e = hipMemsetAsync(nb, 0xff, (size_t)(ncap * per_record) * sizeof(T), h->stream);
    if (e == hipSuccess && *buf) {
        e = hipMemcpyAsync(nb, *buf, (size_t)(*cap * per_record) * sizeof(T), hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (e != hipSuccess) { (void)hipFree(nb); CX_HIP(h, e); }
    if (*buf) (void)hipFree(*buf);
    *buf = nb; *cap = ncap;
    return CX_OK;
}
}  // extern "C++"

// JointMarginal(factor): the two slots of a pairwise Gaussian factor, the OUT edge first (its per-slot parameters are the
// forward rule's), and whether the OUT edge's variable has the lower id
static int32_t joint_slots(cx_handle *h, int64_t factor_id, int32_t *s_out, int32_t *s_in, int32_t *out_first, int32_t *fidx) {
    auto it = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), factor_id);
    if (it == h->fac_ids.end() || *it != factor_id) return fail(h, CX_ERR_NOT_FOUND, "unknown factor id " + std::to_string(factor_id));
    const int32_t f = (int32_t)(it - h->fac_ids.begin());
    if (h->fac_kind[f] != CX_FACTOR_GAUSS_ADDITIVE && h->fac_kind[f] != CX_FACTOR_GAUSS_LINEAR)
        return fail(h, CX_ERR_UNSUPPORTED, "cx_update_batch: JointMarginal is implemented for pairwise Gaussian factors (factor " + std::to_string(factor_id) + " is not one)");
    // the factor's edges: scan the variables' CSR rows (construction-time cost, cached by the caller's store index)
    int32_t found[2] = {-1, -1}, vars[2] = {-1, -1}; int nfound = 0;
    if (h->fac_edges.empty()) {   // factor -> its (up to two) edges, built once
        h->fac_edges.assign((size_t)2 * h->nf, -1);
        for (int64_t e = 0; e < h->ne; e++) {
            auto jt = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), h->edge_fac_id[e]);
            const int64_t ff = jt - h->fac_ids.begin();
            if (h->fac_edges[2 * ff] < 0) h->fac_edges[2 * ff] = (int32_t)e; else if (h->fac_edges[2 * ff + 1] < 0) h->fac_edges[2 * ff + 1] = (int32_t)e;
        }
    }
    for (int k = 0; k < 2; k++) { const int32_t e = h->fac_edges[2 * (size_t)f + k]; if (e >= 0) { found[nfound] = cx::slot_of_edge(h, e); vars[nfound] = h->edge_var[e]; nfound++; } }
    if (nfound != 2) return fail(h, CX_ERR_UNSUPPORTED, "cx_update_batch: JointMarginal needs a 2-edge factor");
    // OUT edge: for a linear factor the slot whose receiving-edge parameters are the forward (a, b, q); host copy of the roles
    // is not kept, but the forward slot is the one with b-parameters (a, b, q) == params: compare with fac_params
    int out = 1;   // additive: either; take the higher-id variable as "out" (x_b = x_a + noise is symmetric)
    if (h->fac_kind[f] == CX_FACTOR_GAUSS_LINEAR) out = h->lin_out_is_second[f] ? 1 : 0;
    *s_out = found[out]; *s_in = found[1 - out];
    *out_first = vars[out] < vars[1 - out] ? 1 : 0;
    *fidx = f;
    return CX_OK;
}

int32_t cx_update_batch(cx_handle *h, const cx_item *items, int64_t n) {
    if (h) { h->chain_side_dirty = true; h->offchain_marg_dirty = true; }
    CX_NOT_VMP(h, "cx_update_batch");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_update_batch: no graph");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && items, CX_ERR_INVALID_ARGUMENT, "cx_update_batch: null argument");
    CX_REQUIRE(h, h->cfg.dim == 1, CX_ERR_UNSUPPORTED, "cx_update_batch: batched mode is implemented for dim == 1 only in this build");
    try {
        std::vector<int32_t> buf(5 * n, 0);
        for (int64_t i = 0; i < n; i++) {
            const cx_item &it = items[i];
            int64_t idx, var, lo = 0, hi = 0;
            if (it.kind == CX_ITEM_INDIVIDUAL_MARGINAL) {
                var = idx = find_var(h, it.variable_id);
                if (idx < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(it.variable_id));
            } else if (it.kind == CX_ITEM_MESSAGE_TO_FACTOR || it.kind == CX_ITEM_MESSAGE_TO_VARIABLE) {
                int64_t e = find_edge(h, it.variable_id, it.factor_id);
                if (e < 0) return fail(h, CX_ERR_NOT_FOUND, "no connection between variable " + std::to_string(it.variable_id) + " and factor " + std::to_string(it.factor_id));
                idx = cx::slot_of_edge(h, e); var = h->edge_var[e];
            } else if (it.kind == CX_ITEM_PRODUCT_OF_MESSAGES) {
                // ProductOfMessages(variable_id, range, ...), inference_signal.jl:62-66: the range travels in factor_id
                var = find_var(h, it.variable_id);
                if (var < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(it.variable_id));
                lo = (int64_t)((uint64_t)it.factor_id >> 32); hi = (int64_t)((uint64_t)it.factor_id & 0xffffffffu);
                const int64_t deg = h->var_off[var + 1] - h->var_off[var];
                if (lo < 1 || hi < lo || hi > deg)
                    return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_update_batch: ProductOfMessages range " + std::to_string(lo) + ":" + std::to_string(hi) +
                                " outside 1:" + std::to_string(deg) + " (variable " + std::to_string(it.variable_id) + ")");
                auto key = std::make_tuple((int32_t)var, (int32_t)lo, (int32_t)hi);
                auto pit = h->prod_index.find(key);
                if (pit == h->prod_index.end()) pit = h->prod_index.emplace(key, (int32_t)h->prod_index.size()).first;
                idx = pit->second;
            } else if (it.kind == CX_ITEM_JOINT_MARGINAL) {
                CX_REQUIRE(h, h->cfg.family == CX_FAMILY_GAUSSIAN, CX_ERR_UNSUPPORTED, "cx_update_batch: JointMarginal needs the Gaussian family");
                int32_t s_out, s_in, out_first, f;
                int32_t rc = joint_slots(h, it.factor_id, &s_out, &s_in, &out_first, &f);
                if (rc != CX_OK) return rc;
                auto jit = h->joint_index.find(f);
                if (jit == h->joint_index.end()) jit = h->joint_index.emplace(f, (int32_t)h->joint_index.size()).first;
                idx = jit->second; var = s_out; lo = s_in; hi = out_first;
            } else {
                return fail(h, CX_ERR_UNSUPPORTED, "cx_update_batch: unknown item kind " + std::to_string(it.kind));
            }
            buf[5 * i] = it.kind; buf[5 * i + 1] = (int32_t)idx; buf[5 * i + 2] = (int32_t)var; buf[5 * i + 3] = (int32_t)lo; buf[5 * i + 4] = (int32_t)hi;
        }
        int32_t rc = grow_store(h, &h->d_prod, &h->prod_cap, (int64_t)h->prod_index.size(), 1);
        if (rc != CX_OK) return rc;
        rc = grow_store(h, &h->d_joint, &h->joint_cap, (int64_t)h->joint_index.size(), 6);
        if (rc != CX_OK) return rc;
        rc = ensure_v2f(h);
        if (rc != CX_OK) return rc;
        rc = ensure_stage(h, 5 * n * 4);
        if (rc != CX_OK) return rc;
        CX_HIP(h, hipMemcpyAsync(h->d_stage, buf.data(), 5 * n * 4, hipMemcpyHostToDevice, h->stream));
        cx::launch_batch(h, (const int32_t *)h->d_stage, n);
        CX_HIP(h, hipGetLastError());
        CX_HIP(h, hipStreamSynchronize(h->stream));  // synchronous: the host sets readiness bits next (signal.jl:232-253)
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_update_batch: host allocation failed"); }
}

// stored ProductOfMessages values (natural form for CX_FORM_NATURAL, (mean, variance) for CX_FORM_MOMENT); a node that was
// never computed reads as UndefValue() (NaN)
int32_t cx_get_products(cx_handle *h, int64_t n, const int64_t *variable_ids, const int32_t *range_lo, const int32_t *range_hi,
                        int32_t form, double *out) {
    CX_NOT_VMP(h, "cx_get_products");
    CX_REQUIRE(h, h && h->has_graph && h->cfg.dim == 1, CX_ERR_STATE, "cx_get_products: no scalar graph");
    CX_REQUIRE(h, form == CX_FORM_MOMENT || form == CX_FORM_NATURAL, CX_ERR_INVALID_ARGUMENT, "cx_get_products: bad form");
    CX_REQUIRE(h, h->cfg.family == CX_FAMILY_GAUSSIAN || form == CX_FORM_NATURAL, CX_ERR_UNSUPPORTED, "cx_get_products: CX_FAMILY_NATURAL2 returns CX_FORM_NATURAL payloads only");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && range_lo && range_hi && out, CX_ERR_INVALID_ARGUMENT, "cx_get_products: null argument");
    try {
        // a batch that failed half-way may have indexed nodes the store was never grown for: they read as UndefValue()
        std::vector<double2> store((size_t)std::min<int64_t>((int64_t)h->prod_index.size(), h->prod_cap));
        if (!store.empty()) {
            CX_HIP(h, hipMemcpyAsync(store.data(), h->d_prod, store.size() * 16, hipMemcpyDeviceToHost, h->stream));
            CX_HIP(h, hipStreamSynchronize(h->stream));
        }
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = find_var(h, variable_ids[i]);
            if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
            auto it = h->prod_index.find(std::make_tuple((int32_t)v, range_lo[i], range_hi[i]));
            const double2 m = (it == h->prod_index.end() || (size_t)it->second >= store.size()) ? make_double2(kNaN, kNaN) : store[it->second];
            from_natural(form, m, out + 2 * i);
        }
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_products: host allocation failed"); }
}

// stored JointMarginal values: 6 doubles per factor — mean[2] then covariance[4] row-major, variables in ascending id order
int32_t cx_get_joint_marginals(cx_handle *h, int64_t n, const int64_t *factor_ids, double *out) {
    CX_NOT_VMP(h, "cx_get_joint_marginals");
    CX_REQUIRE(h, h && h->has_graph && h->cfg.dim == 1, CX_ERR_STATE, "cx_get_joint_marginals: no scalar graph");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && factor_ids && out, CX_ERR_INVALID_ARGUMENT, "cx_get_joint_marginals: null argument");
    try {
        std::vector<double> store((size_t)6 * std::min<int64_t>((int64_t)h->joint_index.size(), h->joint_cap));
        if (!store.empty()) {
            CX_HIP(h, hipMemcpyAsync(store.data(), h->d_joint, store.size() * 8, hipMemcpyDeviceToHost, h->stream));
            CX_HIP(h, hipStreamSynchronize(h->stream));
        }
        for (int64_t i = 0; i < n; i++) {
            auto ft = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), factor_ids[i]);
            if (ft == h->fac_ids.end() || *ft != factor_ids[i]) return fail(h, CX_ERR_NOT_FOUND, "unknown factor id " + std::to_string(factor_ids[i]));
            auto it = h->joint_index.find((int32_t)(ft - h->fac_ids.begin()));
            const bool have = it != h->joint_index.end() && (size_t)6 * it->second + 5 < store.size();
            for (int k = 0; k < 6; k++) out[6 * i + k] = have ? store[(size_t)6 * it->second + k] : kNaN;
        }
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_joint_marginals: host allocation failed"); }
}

// ---- chain decomposition for CX_SCHED_CHAIN_SCAN -------------------------------------------------------------------
// Free variables (not observed, not ghosts, degree >= 2) linked by 2-edge factors must form disjoint simple paths.
static int32_t build_chains(cx_handle *h) {
    if (!h->chains_dirty) return CX_OK;
    try {
        const int64_t nv = h->nv;
        std::vector<int32_t> slot_var(h->nslots, -1);
        for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
        auto is_free = [&](int32_t v) { return !(h->vinfo[v] & (cx::kClamped | cx::kGhost)) && (h->var_off[v + 1] - h->var_off[v]) >= 2; };
        std::vector<int32_t> dyn(2 * nv, -1);
        std::vector<uint8_t> ndyn(nv, 0);
        for (int64_t e = 0; e < h->ne; e++) {
            const int32_t v = h->edge_var[e];
            if (!is_free(v)) continue;
            const int32_t s = cx::slot_of_edge(h, e), p = h->partner[s];
            if (p < 0 || !is_free(slot_var[p])) continue;
            if (ndyn[v] == 2)
                return fail(h, CX_ERR_UNSUPPORTED, "chain-scan schedule: variable " + std::to_string(h->var_ids[v]) + " has more than two non-observed neighbours (the graph is not a union of chains)");
            dyn[2 * v + ndyn[v]++] = s;
        }
        std::vector<int32_t> pos_var, skip0, skip1, link_pos, from, to;
        std::vector<uint8_t> head_fwd, head_bwd, visited(nv, 0);
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
                if (visited[cur]) return fail(h, CX_ERR_UNSUPPORTED, "chain-scan schedule: the graph has a cycle");
            }
            if (!head_bwd.empty()) head_bwd.back() = 1;
        }
        for (int64_t v = 0; v < nv; v++)
            if (is_free((int32_t)v) && !visited[v] && ndyn[v] == 2)
                return fail(h, CX_ERR_UNSUPPORTED, "chain-scan schedule: the graph has a cycle through variable " + std::to_string(h->var_ids[v]));
        for (void *p : {(void *)h->d_chain_pos_var, (void *)h->d_chain_skip0, (void *)h->d_chain_skip1, (void *)h->d_chain_link_pos,
                        (void *)h->d_chain_from, (void *)h->d_chain_to, (void *)h->d_chain_head_fwd, (void *)h->d_chain_head_bwd,
                        (void *)h->d_chain_side, h->d_chain_totals}) if (p) (void)hipFree(p);
        h->chain_npos = (int64_t)pos_var.size(); h->chain_nlinks = (int64_t)link_pos.size();
        h->chain_side_dirty = true;
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
        if ((rc = dev_alloc(h, &h->d_chain_side, h->chain_npos)) != CX_OK) return rc;
        char *tot = nullptr;
        if ((rc = dev_alloc(h, &tot, (int64_t)cx::chain_total_bytes(h->chain_nlinks))) != CX_OK) return rc;
        h->d_chain_totals = tot;
        CX_HIP(h, hipStreamSynchronize(h->stream));
        h->chains_dirty = false;
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "chain decomposition: host allocation failed"); }
}

// CX_TILED=0 in the environment turns the two-sweep launches off (A/B measurements)
static bool tiled_env_enabled() {
    static const int on = [] { const char *e = std::getenv("CX_TILED"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

// ---- chain-scan partitions: the composed maps of a time block -------------------------------------------
// Host copy of cx_chain.hip's map algebra (projective-linear maps on (xi, w, 1), D normalised to 1)
extern "C++" {
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
}  // extern "C++"

int32_t cx_chain_block_maps(cx_handle *h, double *fwd6, double *bwd6, double *side_first2, double *side_last2,
                            int64_t *first_variable_id, int64_t *last_variable_id, int64_t *n_links) {
    CX_NOT_VMP(h, "cx_chain_block_maps");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_chain_block_maps: no graph");
    CX_REQUIRE(h, h->cfg.dim == 1 && h->cfg.schedule == CX_SCHED_CHAIN_SCAN, CX_ERR_STATE, "cx_chain_block_maps: scalar chain-scan handles only");
    CX_REQUIRE(h, fwd6 && bwd6 && side_first2 && side_last2, CX_ERR_INVALID_ARGUMENT, "cx_chain_block_maps: null argument");
    CX_REQUIRE(h, !h->any_linear, CX_ERR_UNSUPPORTED, "cx_chain_block_maps: additive factors only in this build");
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

// ---- the sweep ----------------------------------------------------------------------------------------------------
static void sweep_main(cx_handle *h, bool skip_ghosts) {
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
        const int form = marg ? (h->cfg.family == CX_FAMILY_NATURAL2 ? 2 : 1) : 0;
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
        const bool store = h->cfg.materialize_messages_to_factor != 0;
        cx::launch_fused(h, h->d_f2v, h->d_f2v_alt, marg, store, skip_ghosts);
        if (!h->big_vars.empty()) {
            cx::launch_big_var_to_factor(h, h->d_f2v, marg);
            cx::launch_push_slots(h, h->d_big_slots, (int64_t)h->big_slots.size(), h->d_f2v_alt, CX_KERNEL_BIG_VAR);
        }
    }
}

static void sweep_finish(cx_handle *h) {
    if (h->cfg.schedule == CX_SCHED_FLOODING) {
        cx::launch_factor_to_var(h, h->d_v2f, h->d_f2v);
    } else if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) {
        // nothing: the scans already produced every factor→variable message a free variable reads, from the same
        // variable→factor messages the variable phase just wrote (a factor phase here would only re-derive them)
    } else {
        std::swap(h->d_f2v, h->d_f2v_alt);
        h->v2f_stale = h->cfg.materialize_messages_to_factor == 0;
    }
    h->sweeps_done++;
}

int32_t cx_sweep(cx_handle *h, int32_t n_sweeps) {
    CX_NOT_VMP(h, "cx_sweep");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_sweep: no graph");
    CX_REQUIRE(h, n_sweeps >= 0, CX_ERR_INVALID_ARGUMENT, "cx_sweep: n_sweeps < 0");
    if (h->cfg.dim > 1) return mv_sweep(h, n_sweeps);
    CX_REQUIRE(h, h->halo_state || h->chain_partition || (h->recv_slots.empty() && h->send_slots.empty()), CX_ERR_STATE,
               "cx_sweep: this handle holds a partition (halo configured): use cx_sweep_begin / _main / _end");
    if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) { int32_t rc = build_chains(h); if (rc != CX_OK) return rc; }
    int32_t s = 0;
    // pairs of sweeps as ONE launch each (cx_tiles.hip), when the schedule and the graph allow it
    // (opt-in: measured SLOWER than single sweeps on MI355X, see DESIGN.md §4c — kept as a tested experiment, not the default)
    const bool want_pairs = n_sweeps >= 2 && h->cfg.schedule == CX_SCHED_FUSED && h->cfg.sweeps_per_launch == 2 &&
                            h->cfg.family == CX_FAMILY_GAUSSIAN && h->cfg.materialize_messages_to_factor == 0 && tiled_env_enabled();
    if (want_pairs && h->tiles_state == 0) {
        std::string why;
        if (cx::tiles_build(h, why) && !cx::tiles_prepare_kernel(h)) { cx::tiles_free(h); h->tiles_state = -1; }
    }
    if (want_pairs && h->tiles_state > 0) {
        const bool marg = h->cfg.compute_marginals_in_sweep != 0;
        for (; s + 2 <= n_sweeps; s += 2) {
            cx::launch_tiled2(h, h->d_f2v, h->d_f2v_alt, marg);
            std::swap(h->d_f2v, h->d_f2v_alt);     // d_f2v: time t+2; d_f2v_alt: time t
            h->alt_two_back = true;
            h->v2f_stale = true;
            h->sweeps_done += 2;
        }
    }
    for (; s < n_sweeps; s++) {
        h->run_slice0 = 0; h->run_nslices = 0;
        if (h->halo_state && h->halo_depth > 0 && h->cfg.schedule == CX_SCHED_FUSED && h->big_vars.empty()) {
            const int j = std::min(h->sweeps_since_exchange + 1, h->halo_depth);     // this is sweep j after the exchange
            const int L = h->halo_depth - j + 1;                                       // layers that have to run
            if (h->trim_hi[L] >= h->trim_lo[L]) { h->run_slice0 = h->trim_lo[L]; h->run_nslices = h->trim_hi[L] - h->trim_lo[L] + 1; }
        }
        sweep_main(h, false); sweep_finish(h); h->alt_two_back = false;
        h->run_slice0 = 0; h->run_nslices = 0;
        h->sweeps_since_exchange++;
    }
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
    CX_REQUIRE(h, !h->halo_state, CX_ERR_STATE, "cx_sweep_begin: the handle is configured for state halos (cx_halo_configure_state): use cx_sweep + cx_halo_state_exchange");
    CX_REQUIRE(h, h->cfg.schedule != CX_SCHED_CHAIN_SCAN, CX_ERR_UNSUPPORTED, "cx_sweep_begin: the chain-scan schedule is not partitioned in this build");
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

// ---- halo -------------------------------------------------------------------------------------------------------
// doubles per message in the halo buffers: the storage form (natural): 2 | d + d(d+1)/2 (packed symmetric) | 64 + 64*64
static inline int64_t halo_doubles(const cx_handle *h) { return h->cfg.dim == 1 ? 2 : h->nc; }

int32_t cx_halo_configure(cx_handle *h, int64_t n_send, const int64_t *sv, const int64_t *sf, int64_t n_recv,
                          const int64_t *rv, const int64_t *rf) {
    CX_NOT_VMP(h, "cx_halo_configure");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_halo_configure: no graph");
    CX_REQUIRE(h, h->cfg.dim == 1, CX_ERR_UNSUPPORTED, "cx_halo_configure: partitioned sweeps are implemented for dim == 1 only in this build");
    CX_REQUIRE(h, n_send >= 0 && n_recv >= 0, CX_ERR_INVALID_ARGUMENT, "cx_halo_configure: negative count");
    CX_REQUIRE(h, (n_send == 0 || (sv && sf)) && (n_recv == 0 || (rv && rf)), CX_ERR_INVALID_ARGUMENT, "cx_halo_configure: null argument");
    try {
        std::vector<int32_t> send_vars, recv_vars;
        int32_t rc = stage_slots(h, n_send, sv, sf, h->send_slots, &send_vars);
        if (rc != CX_OK) return rc;
        rc = stage_slots(h, n_recv, rv, rf, h->recv_slots, &recv_vars);
        if (rc != CX_OK) return rc;
        for (int32_t v : recv_vars)
            if (h->var_off[v + 1] - h->var_off[v] != 1)
                return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_halo_configure: an imported edge must belong to a degree-1 ghost variable");
        for (uint8_t &b : h->vinfo) b &= (uint8_t)~cx::kGhost;
        for (int32_t v : recv_vars) h->vinfo[v] |= cx::kGhost;
        h->halo_state = false;
        h->chains_dirty = true;
        CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream)); h->tile_info_dirty = true;
        for (void *p : {(void *)h->d_send_slots, (void *)h->d_recv_slots, (void *)h->d_send_vars}) if (p) (void)hipFree(p);
        if (!h->ext_halo_buffers) { if (h->d_send_buf) (void)hipFree(h->d_send_buf); if (h->d_recv_buf) (void)hipFree(h->d_recv_buf); }
        h->d_send_slots = h->d_recv_slots = h->d_send_vars = nullptr; h->d_send_buf = h->d_recv_buf = nullptr;
        h->ext_halo_buffers = false;
        rc = dev_upload(h, &h->d_send_slots, h->send_slots); if (rc != CX_OK) return rc;
        rc = dev_upload(h, &h->d_send_vars, send_vars); if (rc != CX_OK) return rc;
        rc = dev_upload(h, &h->d_recv_slots, h->recv_slots); if (rc != CX_OK) return rc;
        rc = dev_alloc(h, &h->d_send_buf, n_send); if (rc != CX_OK) return rc;
        rc = dev_alloc(h, &h->d_recv_buf, n_recv); if (rc != CX_OK) return rc;
        CX_HIP(h, hipStreamSynchronize(h->stream));
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_halo_configure: host allocation failed"); }
}

// ---- state halos (deep halo): the partition keeps `depth` redundant rows of its neighbours' variables; between exchanges
// the handle runs plain sweeps, an exchange overwrites the factor→variable messages of the redundant variables with the
// owner's values.  After k <= depth sweeps every message of an owned variable equals the un-partitioned sweep's bit for bit
// (the error of the frozen outer edge advances one row per sweep).
int32_t cx_halo_configure_state(cx_handle *h, int64_t n_send, const int64_t *sv, const int64_t *sf, int64_t n_recv,
                                const int64_t *rv, const int64_t *rf) {
    CX_NOT_VMP(h, "cx_halo_configure_state");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_halo_configure_state: no graph");
    CX_REQUIRE(h, h->cfg.schedule != CX_SCHED_CHAIN_SCAN, CX_ERR_UNSUPPORTED,
               "cx_halo_configure_state: fused / flooding schedules (a chain-scan partition exchanges block maps: cx_chain_block_maps)");
    CX_REQUIRE(h, n_send >= 0 && n_recv >= 0, CX_ERR_INVALID_ARGUMENT, "cx_halo_configure_state: negative count");
    CX_REQUIRE(h, (n_send == 0 || (sv && sf)) && (n_recv == 0 || (rv && rf)), CX_ERR_INVALID_ARGUMENT, "cx_halo_configure_state: null argument");
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, "cx_halo_configure_state: a cx_sweep_begin is still open");
    try {
        int32_t rc = stage_slots(h, n_send, sv, sf, h->send_slots, nullptr);
        if (rc != CX_OK) return rc;
        rc = stage_slots(h, n_recv, rv, rf, h->recv_slots, nullptr);
        if (rc != CX_OK) return rc;
        if (std::any_of(h->vinfo.begin(), h->vinfo.end(), [](uint8_t b) { return (b & cx::kGhost) != 0; })) {
            for (uint8_t &b : h->vinfo) b &= (uint8_t)~cx::kGhost;
            CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream)); h->tile_info_dirty = true;
        }
        for (void *p : {(void *)h->d_send_slots, (void *)h->d_recv_slots, (void *)h->d_send_vars}) if (p) (void)hipFree(p);
        if (!h->ext_halo_buffers) { if (h->d_send_buf) (void)hipFree(h->d_send_buf); if (h->d_recv_buf) (void)hipFree(h->d_recv_buf); }
        h->d_send_slots = h->d_recv_slots = h->d_send_vars = nullptr; h->d_send_buf = h->d_recv_buf = nullptr;
        h->ext_halo_buffers = false;
        rc = dev_upload(h, &h->d_send_slots, h->send_slots); if (rc != CX_OK) return rc;
        rc = dev_upload(h, &h->d_recv_slots, h->recv_slots); if (rc != CX_OK) return rc;
        const int64_t per = halo_doubles(h);     // doubles per message: 2 (scalar), packed natural form for dim 2..4, 4160 for dim 64
        rc = dev_alloc(h, &h->d_send_buf, (n_send * per + 1) / 2); if (rc != CX_OK) return rc;
        rc = dev_alloc(h, &h->d_recv_buf, (n_recv * per + 1) / 2); if (rc != CX_OK) return rc;
        CX_HIP(h, hipStreamSynchronize(h->stream));
        h->halo_state = true;
        h->halo_depth = 0; h->trim_lo.clear(); h->trim_hi.clear(); h->sweeps_since_exchange = 0;
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_halo_configure_state: host allocation failed"); }
}

static void state_pack(cx_handle *h) {
    const int64_t n = (int64_t)h->send_slots.size();
    if (h->cfg.dim == 1) cx::launch_gather(h, h->d_f2v, h->d_send_slots, h->d_send_buf, n);
    else if (h->cfg.dim == 64) cx::mv64_rows_gather(h, h->d_mv_f2v, h->d_send_slots, (double *)h->d_send_buf, n);
    else cx::mv_launch_gather(h, h->d_mv_f2v, h->nslots, h->nc, h->d_send_slots, (double *)h->d_send_buf, n);
}
static void state_unpack(cx_handle *h) {
    const int64_t n = (int64_t)h->recv_slots.size();
    if (h->cfg.dim == 1) cx::launch_scatter(h, h->d_f2v, h->d_recv_slots, h->d_recv_buf, n);
    else if (h->cfg.dim == 64) cx::mv64_rows_scatter(h, h->d_mv_f2v, h->d_recv_slots, (const double *)h->d_recv_buf, n);
    else cx::mv_launch_scatter(h, h->d_mv_f2v, h->nslots, h->nc, h->d_recv_slots, (const double *)h->d_recv_buf, n);
}

// Deep halo, trimmed sweeps.  `layer` = distance of a redundant variable from the owned set (1 .. depth; the stand-ins beyond are
// depth + 1; owned variables 0 and need not be listed).  After an exchange every layer is valid; sweep j (1-based) leaves layers
// <= depth - j valid and, to do so, has to RUN the variables of layers <= depth - j + 1 (a variable's new messages are pushed by
// its neighbours' threads).  cx_sweep therefore launches only the slices that hold such variables: on row strips of a grid
// (depth + 1) / 2 redundant rows per side per sweep on average instead of depth.  Results of owned variables are unchanged.
int32_t cx_halo_set_layers(cx_handle *h, int64_t n, const int64_t *variable_ids, const int32_t *layer, int32_t depth) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state, CX_ERR_STATE, "cx_halo_set_layers: call cx_halo_configure_state first");
    CX_REQUIRE(h, depth >= 1 && n >= 0 && (n == 0 || (variable_ids && layer)), CX_ERR_INVALID_ARGUMENT, "cx_halo_set_layers: bad argument");
    try {
        std::vector<int32_t> lay(h->nv, 0);
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = find_var(h, variable_ids[i]);
            if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
            if (layer[i] < 0) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_halo_set_layers: negative layer");
            lay[v] = layer[i];
        }
        h->trim_lo.assign(depth + 1, (int32_t)h->nslices); h->trim_hi.assign(depth + 1, -1);
        for (int64_t v = 0; v < h->nv; v++) {
            const int32_t s = (int32_t)(v >> cx::kSliceShift);
            for (int32_t L = std::min<int32_t>(lay[v], depth + 1); L <= depth; L++) {   // a variable of layer l belongs to every set "layer <= L", L >= l
                h->trim_lo[L] = std::min(h->trim_lo[L], s); h->trim_hi[L] = std::max(h->trim_hi[L], s);
            }
        }
        h->halo_depth = depth;
        h->sweeps_since_exchange = 0;
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_halo_set_layers: host allocation failed"); }
}

int32_t cx_halo_state_pack(cx_handle *h) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state, CX_ERR_STATE, "cx_halo_state_pack: call cx_halo_configure_state first");
    state_pack(h);
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

int32_t cx_halo_state_unpack(cx_handle *h) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state, CX_ERR_STATE, "cx_halo_state_unpack: call cx_halo_configure_state first");
    state_unpack(h);
    h->sweeps_since_exchange = 0;
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

int32_t cx_halo_buffers(cx_handle *h, void **send_ptr, int64_t *send_bytes, void **recv_ptr, int64_t *recv_bytes) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_halo_buffers: no graph");
    if (send_ptr) *send_ptr = h->d_send_buf;
    if (send_bytes) *send_bytes = (int64_t)h->send_slots.size() * 8 * halo_doubles(h);
    if (recv_ptr) *recv_ptr = h->d_recv_buf;
    if (recv_bytes) *recv_bytes = (int64_t)h->recv_slots.size() * 8 * halo_doubles(h);
    return CX_OK;
}

int32_t cx_halo_set_buffers(cx_handle *h, void *send_ptr, void *recv_ptr) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_halo_set_buffers: no graph");
    CX_REQUIRE(h, (send_ptr || h->send_slots.empty()) && (recv_ptr || h->recv_slots.empty()), CX_ERR_INVALID_ARGUMENT,
               "cx_halo_set_buffers: null buffer for a non-empty halo list");
    CX_HIP(h, hipStreamSynchronize(h->stream));
    if (!h->ext_halo_buffers) { if (h->d_send_buf) (void)hipFree(h->d_send_buf); if (h->d_recv_buf) (void)hipFree(h->d_recv_buf); }
    h->d_send_buf = (double2 *)send_ptr; h->d_recv_buf = (double2 *)recv_ptr;
    h->ext_halo_buffers = true;
    return CX_OK;
}

// ---- RCCL exchange issued by the library ----------------------------------------------------------------------------
int32_t cx_comm_unique_id(void *out128) {
    if (!out128) return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_comm_unique_id: null buffer");
    std::string err;
    if (!cx::comm_unique_id(out128, err)) return fail(nullptr, CX_ERR_DEVICE, "cx_comm_unique_id: " + err);
    return CX_OK;
}

int32_t cx_comm_init(cx_handle *h, int32_t world, int32_t rank, const void *id128) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "null handle");
    CX_REQUIRE(h, id128 && world >= 1 && rank >= 0 && rank < world, CX_ERR_INVALID_ARGUMENT, "cx_comm_init: bad world / rank / id");
    CX_REQUIRE(h, !h->comm, CX_ERR_STATE, "cx_comm_init: communicator already initialised");
    CX_HIP(h, hipSetDevice(h->cfg.device));
    std::string err;
    if (!cx::comm_init(h, world, rank, id128, err)) return fail(h, CX_ERR_DEVICE, "cx_comm_init: " + err);
    return CX_OK;
}

int32_t cx_halo_peers(cx_handle *h, int32_t n_peers, const int32_t *peer_rank, const int64_t *send_offset, const int64_t *send_count,
                      const int64_t *recv_offset, const int64_t *recv_count) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_halo_peers: no graph");
    CX_REQUIRE(h, n_peers >= 0 && (n_peers == 0 || (peer_rank && send_offset && send_count && recv_offset && recv_count)),
               CX_ERR_INVALID_ARGUMENT, "cx_halo_peers: null argument");
    std::vector<cx_handle::Peer> peers;
    for (int32_t i = 0; i < n_peers; i++) {
        cx_handle::Peer p{peer_rank[i], send_offset[i], send_count[i], recv_offset[i], recv_count[i]};
        if (p.send_off < 0 || p.send_count < 0 || p.send_off + p.send_count > (int64_t)h->send_slots.size() || p.recv_off < 0 ||
            p.recv_count < 0 || p.recv_off + p.recv_count > (int64_t)h->recv_slots.size())
            return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_halo_peers: segment outside the halo lists of cx_halo_configure");
        peers.push_back(p);
    }
    h->peers.swap(peers);
    return CX_OK;
}

int32_t cx_sweep_exchange(cx_handle *h, int32_t n_sweeps) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_sweep_exchange: no graph");
    CX_REQUIRE(h, n_sweeps >= 0, CX_ERR_INVALID_ARGUMENT, "cx_sweep_exchange: n_sweeps < 0");
    CX_REQUIRE(h, h->comm || h->peers.empty(), CX_ERR_STATE, "cx_sweep_exchange: call cx_comm_init first");
    for (auto &p : h->peers)
        CX_REQUIRE(h, p.rank >= 0 && p.rank < h->comm_world, CX_ERR_INVALID_ARGUMENT, "cx_sweep_exchange: bad peer rank");   // a rank may be its own neighbour (periodic cut)
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, "cx_sweep_exchange: a cx_sweep_begin is still open");
    CX_REQUIRE(h, !h->halo_state, CX_ERR_STATE, "cx_sweep_exchange: the handle is configured for state halos: use cx_sweep + cx_halo_state_exchange");
    CX_REQUIRE(h, h->cfg.dim == 1 && h->cfg.schedule != CX_SCHED_CHAIN_SCAN, CX_ERR_UNSUPPORTED, "cx_sweep_exchange: scalar fused / flooding schedules only");
    const bool overlap = !h->peers.empty();
    for (int32_t s = 0; s < n_sweeps; s++) {
        // The export reads the same input buffer as the main kernel and writes only the send buffer, so it runs on the
        // communication stream, beside the main kernel:   comm: [wait swept] export, group{send, recv}, record recv
        //                                                  main: main kernel, [wait recv] import+push, record swept
        if (overlap) {
            CX_HIP(h, hipEventRecord(h->ev_swept, h->stream));
            CX_HIP(h, hipStreamWaitEvent(h->comm_stream, h->ev_swept, 0));
            cx::launch_halo_export(h, h->d_f2v, h->comm_stream);
            std::string err;
            if (!cx::comm_exchange(h, err, true)) return fail(h, CX_ERR_DEVICE, "cx_sweep_exchange: " + err);
        }
        sweep_main(h, true);
        if (overlap) CX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_recv, 0));
        cx::launch_halo_import(h, h->d_f2v_alt, h->cfg.schedule == CX_SCHED_FUSED);
        sweep_finish(h);
        CX_HIP(h, hipGetLastError());
    }
    return CX_OK;
}

int32_t cx_halo_state_exchange(cx_handle *h) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state, CX_ERR_STATE, "cx_halo_state_exchange: call cx_halo_configure_state first");
    CX_REQUIRE(h, h->comm || h->peers.empty(), CX_ERR_STATE, "cx_halo_state_exchange: call cx_comm_init first");
    for (auto &p : h->peers)
        CX_REQUIRE(h, p.rank >= 0 && p.rank < h->comm_world, CX_ERR_INVALID_ARGUMENT, "cx_halo_state_exchange: bad peer rank");
    if (h->peers.empty()) return CX_OK;
    // pack, send/recv and unpack in stream order on the handle's own stream: no cross-stream hand-off at all (each one
    // costs ≈6 µs on this stack); the exchange happens once per `depth` sweeps, so it need not hide behind a kernel
    state_pack(h);
    std::string err;
    if (!cx::comm_exchange_on(h, h->stream, err)) return fail(h, CX_ERR_DEVICE, "cx_halo_state_exchange: " + err);
    state_unpack(h);
    h->sweeps_since_exchange = 0;
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

// ---- variational families (cx_vmp.hip) --------------------------------------------------------------------------------
int32_t cx_set_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, int32_t form, const double *payload) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_set_marginals: no graph");
    CX_REQUIRE(h, is_vmp(h), CX_ERR_UNSUPPORTED, "cx_set_marginals: only the variational families keep settable marginals (sum-product marginals are products of messages: cx_set_messages)");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && payload, CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: null argument");
    return cx::vmp_set_marginals(h, n, variable_ids, form, payload);
}

int32_t cx_update_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_update_marginals: no graph");
    CX_REQUIRE(h, is_vmp(h), CX_ERR_UNSUPPORTED, "cx_update_marginals: variational families only (sum-product handles run cx_sweep / cx_update_batch)");
    if (n == 0) return CX_OK;
    return cx::vmp_update_marginals(h, n, variable_ids);
}

// ---- checkpoint: the mutable state of a handle as one relocatable blob ------------------------------
// The reference keeps no persistent state (nothing to mirror); with all messages resident in HBM a long loopy run needs
// a way to stop and resume.  The blob holds the message buffers, marginals and observed-variable flags bit for bit, plus
// a fingerprint of the flattened graph so that it can only be restored into a handle built from the same graph.
extern "C++" {
namespace {

struct StateHeader {
    char magic[8];
    int32_t abi, dim, family, schedule;
    int64_t nv, ne, nslots, nc, sweeps_done;
    int32_t v2f_stale, n_sections;
    uint64_t fingerprint;
};
struct StateSection { int32_t id, reserved; int64_t bytes; };
struct StatePart { int32_t id; void *dev; int64_t bytes; };
const char kStateMagic[8] = {'C', 'X', 'S', 'T', 'A', 'T', 'E', '1'};

uint64_t fnv1a(uint64_t hsh, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { hsh ^= b[i]; hsh *= 1099511628211ull; }
    return hsh;
}

uint64_t graph_fingerprint(const cx_handle *h) {
    uint64_t f = 1469598103934665603ull;
    f = fnv1a(f, h->var_ids.data(), h->var_ids.size() * 8);
    f = fnv1a(f, h->var_off.data(), h->var_off.size() * 4);
    f = fnv1a(f, h->edge_fac_id.data(), h->edge_fac_id.size() * 8);
    f = fnv1a(f, h->fac_kind.data(), h->fac_kind.size() * 4);
    // the rule parameters: a blob continues under the parameters it was exported with, or not at all
    f = fnv1a(f, h->fac_params.data(), h->fac_params.size() * 8);
    f = fnv1a(f, h->spdir.data(), h->spdir.size() * 4);
    for (const auto &ps : h->psets) { const uint64_t n = ps.size(); f = fnv1a(f, &n, 8); f = fnv1a(f, ps.data(), ps.size() * 8); }
    return f;
}

std::vector<StatePart> state_parts(cx_handle *h) {
    std::vector<StatePart> parts;
    const int64_t slots = h->nslots, nv = h->nv;
    parts.push_back({1, h->d_vinfo, nv});
    if (h->cfg.dim == 1) {
        parts.push_back({2, h->d_f2v, slots * 16});
        if (h->d_f2v_alt) parts.push_back({3, h->d_f2v_alt, slots * 16});
        parts.push_back({4, h->d_v2f, slots * 16});
        parts.push_back({5, h->d_marg, nv * 16});
    } else {
        const int64_t nc = h->nc;
        parts.push_back({2, h->d_mv_f2v, nc * slots * 8});
        parts.push_back({3, h->d_mv_f2v_alt, nc * slots * 8});
        parts.push_back({4, h->d_mv_v2f, nc * slots * 8});
        if (h->cfg.dim != 64) parts.push_back({5, h->d_mv_marg, nc * nv * 8});
    }
    return parts;
}

}  // namespace
}  // extern "C++"

int32_t cx_state_bytes(const cx_handle *hc, int64_t *bytes) {
    cx_handle *h = const_cast<cx_handle *>(hc);
    CX_REQUIRE(h, h && h->has_graph && bytes, CX_ERR_STATE, "cx_state_bytes: no graph or null argument");
    if (is_vmp(h)) return cx::vmp_state_bytes(h, bytes);
    int64_t n = (int64_t)sizeof(StateHeader);
    for (auto &p : state_parts(h)) n += (int64_t)sizeof(StateSection) + p.bytes;
    *bytes = n;
    return CX_OK;
}

int32_t cx_state_export(cx_handle *h, void *buf, int64_t bytes) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_state_export: no graph");
    if (is_vmp(h)) return cx::vmp_state_export(h, buf, bytes);
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, "cx_state_export: a cx_sweep_begin is still open");
    int64_t need = 0;
    (void)cx_state_bytes(h, &need);
    CX_REQUIRE(h, buf && bytes >= need, CX_ERR_INVALID_ARGUMENT, "cx_state_export: buffer smaller than cx_state_bytes");
    CX_HIP(h, hipSetDevice(h->cfg.device));
    { int32_t rc = normalize_alt(h); if (rc != CX_OK) return rc; }   // the blob holds the buffers of time n and n - 1
    CX_HIP(h, hipStreamSynchronize(h->stream));
    auto parts = state_parts(h);
    StateHeader hd{};
    std::memcpy(hd.magic, kStateMagic, 8);
    hd.abi = CX_ABI_VERSION; hd.dim = h->cfg.dim; hd.family = h->cfg.family; hd.schedule = h->cfg.schedule;
    hd.nv = h->nv; hd.ne = h->ne; hd.nslots = h->nslots; hd.nc = h->nc; hd.sweeps_done = h->sweeps_done;
    hd.v2f_stale = (h->v2f_stale ? 1 : 0) | (h->offchain_marg_dirty ? 2 : 0);      // bit 1: marginals off the chains are due
    hd.n_sections = (int32_t)parts.size();
    hd.fingerprint = graph_fingerprint(h);
    char *o = (char *)buf;
    std::memcpy(o, &hd, sizeof hd); o += sizeof hd;
    for (auto &p : parts) {
        StateSection sc{p.id, 0, p.bytes};
        std::memcpy(o, &sc, sizeof sc); o += sizeof sc;
        if (p.bytes) CX_HIP(h, hipMemcpy(o, p.dev, (size_t)p.bytes, hipMemcpyDeviceToHost));
        o += p.bytes;
    }
    return CX_OK;
}

int32_t cx_state_import(cx_handle *h, const void *buf, int64_t bytes) {
    if (h) { h->chain_side_dirty = true; h->offchain_marg_dirty = true; }
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_state_import: no graph");
    if (is_vmp(h)) return cx::vmp_state_import(h, buf, bytes);
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, "cx_state_import: a cx_sweep_begin is still open");
    CX_REQUIRE(h, buf && bytes >= (int64_t)sizeof(StateHeader), CX_ERR_INVALID_ARGUMENT, "cx_state_import: blob too short");
    StateHeader hd;
    std::memcpy(&hd, buf, sizeof hd);
    CX_REQUIRE(h, std::memcmp(hd.magic, kStateMagic, 8) == 0 && hd.abi == CX_ABI_VERSION, CX_ERR_INVALID_ARGUMENT,
               "cx_state_import: not a state blob of this ABI version");
    CX_REQUIRE(h, hd.dim == h->cfg.dim && hd.family == h->cfg.family && hd.schedule == h->cfg.schedule, CX_ERR_INVALID_ARGUMENT,
               "cx_state_import: the blob was exported with a different dim / family / schedule");
    CX_REQUIRE(h, hd.nv == h->nv && hd.ne == h->ne && hd.nslots == h->nslots && hd.nc == h->nc && hd.fingerprint == graph_fingerprint(h),
               CX_ERR_INVALID_ARGUMENT, "cx_state_import: the blob belongs to a different graph");
    auto parts = state_parts(h);
    CX_REQUIRE(h, hd.n_sections == (int32_t)parts.size(), CX_ERR_INVALID_ARGUMENT, "cx_state_import: section count mismatch");
    // validate the whole layout before touching the device
    const char *o = (const char *)buf + sizeof hd, *end = (const char *)buf + bytes;
    for (auto &p : parts) {
        CX_REQUIRE(h, end - o >= (int64_t)sizeof(StateSection), CX_ERR_INVALID_ARGUMENT, "cx_state_import: truncated blob");
        StateSection sc;
        std::memcpy(&sc, o, sizeof sc); o += sizeof sc;
        CX_REQUIRE(h, sc.id == p.id && sc.bytes == p.bytes && end - o >= sc.bytes, CX_ERR_INVALID_ARGUMENT, "cx_state_import: truncated or foreign blob");
        o += sc.bytes;
    }
    // the vinfo section: only the observed flag is state; degree class and ghost flag are structure the kernels index by
    {
        const unsigned char *vi = (const unsigned char *)buf + sizeof hd + sizeof(StateSection);
        for (int64_t v = 0; v < h->nv; v++)
            CX_REQUIRE(h, (vi[v] & (uint8_t)~cx::kClamped) == (h->vinfo[v] & (uint8_t)~cx::kClamped), CX_ERR_INVALID_ARGUMENT,
                       "cx_state_import: the blob's variable table does not match this handle's graph");
    }
    CX_HIP(h, hipSetDevice(h->cfg.device));
    CX_HIP(h, hipStreamSynchronize(h->stream));
    o = (const char *)buf + sizeof hd;
    for (auto &p : parts) {
        o += sizeof(StateSection);
        if (p.id == 1) std::memcpy(h->vinfo.data(), o, (size_t)p.bytes);
        if (p.bytes) CX_HIP(h, hipMemcpy(p.dev, o, (size_t)p.bytes, hipMemcpyHostToDevice));
        o += p.bytes;
    }
    h->sweeps_done = hd.sweeps_done;
    h->v2f_stale = (hd.v2f_stale & 1) != 0;
    h->offchain_marg_dirty = (hd.v2f_stale & 2) != 0;     // the marginals themselves travelled in section 5
    h->alt_two_back = false; h->tile_info_dirty = true;
    h->spdir_dirty = h->work64_dirty = h->point64_dirty = h->chains_dirty = true;   // derived from the observed flags
    if (h->d_prev) { (void)hipFree(h->d_prev); h->d_prev = nullptr; }               // residual snapshots restart
    if (h->d_mv_prev) { (void)hipFree(h->d_mv_prev); h->d_mv_prev = nullptr; }
    return CX_OK;
}

// ---- profiling ----------------------------------------------------------------------------------------------------
int32_t cx_profile_enable(cx_handle *h, int32_t on) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "null handle");
    h->profiling = on != 0;
    h->prof_stride = on > 1 ? on : 1;   // on = n > 1: bracket every n-th launch of each kernel only
    for (auto &c : h->prof_count) c = 0;
    return CX_OK;
}

int32_t cx_profile_read(cx_handle *h, int32_t kernel, double *total_ms, int64_t *launches) {
    CX_REQUIRE(h, h && total_ms && launches, CX_ERR_INVALID_ARGUMENT, "cx_profile_read: null argument");
    CX_HIP(h, hipStreamSynchronize(h->stream));
    double tot = 0.0; int64_t cnt = 0;
    std::vector<cx::ProfileRec> keep;
    for (auto &r : h->recs) {
        if (r.kernel == kernel) {
            float ms = 0.f;
            CX_HIP(h, hipEventElapsedTime(&ms, r.start, r.stop));
            tot += ms; cnt++;
            (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop);
        } else keep.push_back(r);
    }
    h->recs.swap(keep);
    *total_ms = tot; *launches = cnt;
    return CX_OK;
}

}  // extern "C"
