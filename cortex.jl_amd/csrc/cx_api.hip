// cx_api.hip — host side of libcortex_hip.so: the C ABI of include/cortex_hip.h.
//
// Flattens the bipartite factor graph once (reference: the accessor loops of
// src/inference_engine.jl:228-247 and src/dependencies.jl:5-126 over
// ext/BipartiteFactorGraphsExt/BipartiteFactorGraphsExt.jl:22-48) into a CSR edge table sorted by
// (variable id, factor id), keeps Gaussian messages resident in HBM, and launches the kernels of
// cx_kernels.hip.  No exception leaves this file; every entry point returns a status.


#include "cx_host.h"
#include "cx_flatten.h"

using namespace cxh;

namespace cxh {

void dev_free_all(cx_handle *h) {
    void *ptrs[] = {h->d_slice_off, h->d_partner, h->d_vbase, h->d_var_deg, h->d_big, h->d_big_slots, h->d_big_slot_var, h->d_big_tmp,
                    h->d_vinfo, h->d_q, h->d_a, h->d_b, h->d_sq, h->d_sa, h->d_sb, h->d_f2v, h->d_v2f, h->d_marg,
                    h->d_f2v_alt, h->d_prev, h->d_scratch, h->d_send_slots, h->d_recv_slots, h->d_send_vars,
                    h->ext_halo_buffers ? nullptr : (void *)h->d_send_buf, h->ext_halo_buffers ? nullptr : (void *)h->d_recv_buf,
                    h->d_stage, h->d_spdir, h->d_ptab, h->d_ptab_bt, h->d_zero_msg, h->d_mv_f2v, h->d_mv_f2v_alt, h->d_mv_v2f, h->d_mv_marg, h->d_mv_prev, h->d_mv_prod, h->d_point64_slots, h->d_rule64_rec, h->d_chain_pos_var, h->d_chain_skip0, h->d_chain_skip1, h->d_chain_link_pos, h->d_chain_from,
                    h->d_chain_to, h->d_chain_head_fwd, h->d_chain_head_bwd, h->d_chain_side, h->d_chain_totals, h->d_chain_tab_fwd, h->d_chain_tab_bwd,
                    h->d_mvc_side, h->d_mvc_totals, h->d_mvc_side_l, h->d_mvc_alpha, h->d_mvc_gamma, h->d_mvc_prefix, h->d_mvc_wave_carry, h->d_mvc_block,
                    h->d_tree_rec, h->d_tree_kary, h->d_partner16, h->d_mvc_var_link, h->d_tree_stage_off, h->d_tree_skip1_down, h->d_tree_a, h->d_tree_b, h->d_pre64_slots, h->d_pre64_vars, h->d_tree_pre_slots, h->d_tree_pre_vars};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    h->d_tree_rec = h->d_tree_kary = nullptr; h->d_tree_skip1_down = nullptr; h->d_tree_a = h->d_tree_b = nullptr; h->d_pre64_slots = h->d_pre64_vars = h->d_tree_pre_slots = h->d_tree_pre_vars = nullptr; h->n_pre64 = 0; h->tree_hp = false; h->tree_dirty = true; h->d_partner16 = nullptr; h->d_mvc_var_link = nullptr; h->d_tree_stage_off = nullptr;
    tree_graph_drop(h); h->tree_graph_failed = false;
    batch_graph_drop(h);
    if (h->d_cluster_ctl) { (void)hipFree(h->d_cluster_ctl); h->d_cluster_ctl = nullptr; }
    h->cluster_state = 0;
    h->set_memos.clear();
    ref_free(h);
    cx::chain64_free(h);
    cx::chain_onepass_free(h);
    for (void *p : {(void *)h->d_marg64_sums, (void *)h->d_marg64_tab, (void *)h->d_marg64_rec}) if (p) (void)hipFree(p);
    h->d_marg64_sums = h->d_marg64_tab = nullptr; h->d_marg64_rec = nullptr; h->marg64_cap = 0;
    cx::chain64_tree_free(h);
    cx::kary_free(h);
    if (h->d_prod) (void)hipFree(h->d_prod);
    if (h->d_joint) (void)hipFree(h->d_joint);
    h->d_prod = nullptr; h->d_joint = nullptr; h->prod_cap = h->joint_cap = 0; h->prod_index.clear(); h->joint_index.clear();
    h->d_point64_slots = h->d_rule64_rec = nullptr; h->work64_dirty = h->point64_dirty = true;
    h->d_spdir = nullptr; h->d_ptab = nullptr; h->d_ptab_bt = nullptr; h->d_zero_msg = nullptr; h->d_mv_f2v = h->d_mv_f2v_alt = h->d_mv_v2f = h->d_mv_marg = h->d_mv_prev = nullptr; h->d_mv_prod = nullptr; h->mv_prod_cap = 0; h->ptab_sets = 0; h->ptab_bt_sets = 0;
    h->d_chain_pos_var = h->d_chain_skip0 = h->d_chain_skip1 = h->d_chain_link_pos = h->d_chain_from = h->d_chain_to = nullptr;
    h->d_chain_head_fwd = h->d_chain_head_bwd = nullptr; h->d_chain_side = nullptr; h->d_chain_totals = nullptr; h->chains_dirty = true; h->tree_dirty = true;
    h->d_chain_tab_fwd = h->d_chain_tab_bwd = nullptr; h->d_mvc_side = h->d_mvc_totals = nullptr; h->d_mvc_side_l = h->d_mvc_alpha = h->d_mvc_gamma = h->d_mvc_prefix = h->d_mvc_wave_carry = h->d_mvc_block = nullptr;
    h->d_slice_off = h->d_partner = h->d_vbase = h->d_var_deg = h->d_big = h->d_big_slots = h->d_big_slot_var = nullptr;
    h->d_big_tmp = nullptr; h->d_vinfo = nullptr;
    h->d_q = h->d_a = h->d_b = h->d_sq = h->d_sa = h->d_sb = nullptr;
    h->d_f2v = h->d_v2f = h->d_marg = h->d_f2v_alt = h->d_prev = nullptr;
    h->mv_max_deg = 0; h->sweep_max_w = 0;
    h->d_scratch = nullptr; h->d_send_slots = h->d_recv_slots = h->d_send_vars = nullptr;
    h->d_send_buf = h->d_recv_buf = nullptr;
    h->d_stage = nullptr; h->stage_bytes = 0; h->device_bytes = 0;
}

}  // namespace cxh

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
    }
    return "";
}

int32_t cx_create(const cx_config *config, cx_handle **out) {
    if (!out) return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: out is NULL");
    *out = nullptr;
    if (!config || config->struct_size != (int32_t)sizeof(cx_config))
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: config is NULL or struct_size mismatch");
    if (config->dim < 1 || config->dim > 64)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: this build implements dim 1 .. 64");
    // dim 5 .. 64 run on the matrix-core path in 1 x 1, 2 x 2 or 4 x 4 tiles of 16 (cx_const.h: is_mfma_dim): internal dim 16, 32 or 64,
    // the smallest that holds the user's (round 6; until then always 64: a d = 8 message moved 64 x its bytes).  A dim in between is
    // embedded: x -> (x, u) with u a unit random walk observed nowhere — every rule matrix, message and datum block-diagonal (real block,
    // identity block), so the real block of every result is exact and the identity block keeps every joint positive definite.  The
    // chain-scan and tree schedules run the same plans on kernels of their own for 1 x 1 and 2 x 2 tiles (cx_mv64chain.hip: k_compose_nt,
    // k_walk_nt).  CX_MFMA_DIM=64 forces 64 everywhere (A/B; a chain cut into time blocks: cx_chain_block_maps exchanges 64 x 64 potentials).
    cx_config padded;
    int user_dim = 0;
    if (config->dim > 4) {
        int internal = 64;
        const char *force = std::getenv("CX_MFMA_DIM");
        if (!(force && std::atoi(force) == 64)) internal = config->dim <= 16 ? 16 : config->dim <= 32 ? 32 : 64;
        user_dim = config->dim != internal ? config->dim : 0;
        padded = *config; padded.dim = internal; config = &padded;
    }
    const bool is_vmp = config->family == CX_FAMILY_VMP_MEAN_FIELD || config->family == CX_FAMILY_VMP_STRUCTURED;
    if (config->family != CX_FAMILY_GAUSSIAN && config->family != CX_FAMILY_NATURAL2 && !is_vmp)
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: unknown family");
    if (is_vmp && config->dim != 1)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: the variational families need dim == 1");
    if (config->family == CX_FAMILY_NATURAL2 && (config->dim != 1 || config->schedule == CX_SCHED_CHAIN_SCAN))
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: CX_FAMILY_NATURAL2 needs dim == 1 and the flooding, fused, tree or reference schedule");
    // (round 6: the reference-order schedule for dim 64 too — and 5 .. 63 inside it: the stages through k_v2f64 / k_rule64w, cx_api_ref.hip)
    if (config->schedule == CX_SCHED_REFERENCE && is_vmp)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: CX_SCHED_REFERENCE replays the reference's execution order for the Gaussian and natural-pair families (the variational rules run under it as a wiring: cx_graph_wire)");
    if (config->dim > 1 && config->schedule != CX_SCHED_FUSED && config->schedule != CX_SCHED_CHAIN_SCAN && config->schedule != CX_SCHED_TREE && config->schedule != CX_SCHED_REFERENCE)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: dim > 1 runs the fused, the chain-scan, the tree and the reference-order schedule");
    if (config->schedule != CX_SCHED_FLOODING && config->schedule != CX_SCHED_FUSED && config->schedule != CX_SCHED_CHAIN_SCAN && config->schedule != CX_SCHED_TREE && config->schedule != CX_SCHED_REFERENCE)
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: unknown schedule");
    if (config->schedule == CX_SCHED_TREE && config->family == CX_FAMILY_VMP_MEAN_FIELD)
        return fail(nullptr, CX_ERR_UNSUPPORTED, "cx_create: the mean-field family has no inner sweep to schedule (flooding, fused or chain-scan are accepted and ignored; "
                                                 "the tree schedule is the structured family's, for state variables that form a forest)");
    if (config->reserved != 0 && config->reserved != 1)
        return fail(nullptr, CX_ERR_INVALID_ARGUMENT, "cx_create: cx_config.reserved must be 0 (ABI 2 - 3 called it sweeps_per_launch; the two-sweep launch, measured slower, was removed in ABI 4)");
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
    h->user_dim = user_dim;
    h->nc = config->dim == 1 ? 2 : (cx::is_mfma_dim(config->dim) ? config->dim + config->dim * config->dim : config->dim + config->dim * (config->dim + 1) / 2);
    h->ncs = (config->dim >= 2 && config->dim <= 4) ? 2 * ((h->nc + 1) / 2) : h->nc;
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
    cx::ipc_destroy(h);
    tree_graph_drop(h);
    if (h->tree_capture_stream) { (void)hipStreamDestroy(h->tree_capture_stream); h->tree_capture_stream = nullptr; }
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
    // The tree schedule's stages are captured into a HIP graph with the table pointers baked in by value (cx_api_sweep.hip:
    // tree_sweep): the tables are rewritten IN PLACE while their size holds, and a graph captured over a table that has to move is
    // dropped before the old allocation goes (the next sweep captures again).
    if (h->d_ptab && h->ptab_sets < nsets) { tree_graph_drop(h); ref_graphs_drop(h); (void)hipFree(h->d_ptab); h->d_ptab = nullptr; }      // (the reference-order plans' graphs too: cx_api_ref.hip)
    if (!h->d_ptab) { int32_t rc = dev_alloc(h, &h->d_ptab, (int64_t)(per * nsets)); if (rc != CX_OK) return rc; h->ptab_sets = nsets; }
    CX_HIP(h, hipMemcpy(h->d_ptab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    h->pot64_fresh = false;
    if (cx::is_mfma_dim(d)) {   // the wave-per-message rule kernel reads B transposed (tile rows are its contraction index)
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
        if (h->d_ptab_bt && h->ptab_bt_sets < nsets) { tree_graph_drop(h); (void)hipFree(h->d_ptab_bt); h->d_ptab_bt = nullptr; }
        if (!h->d_ptab_bt) {
            int32_t rc = dev_alloc(h, &h->d_ptab_bt, (int64_t)bt.size());
            if (rc != CX_OK) return rc;
            h->ptab_bt_sets = nsets;
        }
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
        std::vector<double> Ap, Qp;
        if (h->user_dim) {      // blockdiag(A, I), blockdiag(Q, I)
            const int u = h->user_dim;
            Ap.assign((size_t)d * d, 0.0); Qp.assign((size_t)d * d, 0.0);
            for (int r = 0; r < d; r++) { Ap[(size_t)r * d + r] = 1.0; Qp[(size_t)r * d + r] = 1.0; }
            for (int r = 0; r < u; r++) for (int c = 0; c < u; c++) { Ap[(size_t)r * d + c] = A[(size_t)r * u + c]; Qp[(size_t)r * d + c] = Q[(size_t)r * u + c]; }
            A = Ap.data(); Q = Qp.data();
        }
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
        if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) { int32_t rc0 = mv_ensure_chain_msgs(h); if (rc0 != CX_OK) return rc0; }   // under the old tables
        // the messages out of observed variables, N(A y, Q), are cached in both Jacobi buffers: new (A, Q) invalidates them
        h->observed_passes_due = 2;
        h->kary_dirty = true;             // dim 2..4 factors of more than two variables read the raw (A, Q)
        h->point64_dirty = true;
        h->chain_side_dirty = true;
        return upload_ptab(h);
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_factor_matrices: host allocation failed"); }
}


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
        cx::flat::Out fo;
        std::string ferr;
        {
            const int32_t frc = cx::flat::flatten(h, n_edges, edge_var, edge_fac, edge_role, n_factors, factor_ids, factor_kind, factor_params, fo, ferr);
            if (frc != CX_OK) return fail(h, frc, ferr);
        }
        const int64_t ne = h->ne, nv = h->nv, slots = h->nslots, big_total = fo.big_total;
        const bool mv = fo.mv;
        std::vector<int32_t> &var_deg = fo.var_deg, &spdir = fo.spdir;
        std::vector<double> &q = fo.q, &a = fo.a, &b = fo.b, &sq = fo.sq, &sa = fo.sa, &sb = fo.sb;
        (void)ne;
        // ---- upload ------------------------------------------------------------------------------------------------
        CX_HIP(h, hipSetDevice(h->cfg.device));
        int32_t rc;
#define CX_TRY(x) do { rc = (x); if (rc != CX_OK) { dev_free_all(h); return rc; } } while (0)
        CX_TRY(dev_upload(h, &h->d_slice_off, h->slice_off));
        CX_TRY(dev_upload(h, &h->d_partner, h->partner));
        {   // the partners as 16-bit differences, where they all fit (grids, chains, banded graphs: the fused sweep then reads 2 bytes per slot)
            std::vector<int16_t> p16(h->partner.size());
            bool fits = !h->partner.empty();
            for (size_t sl = 0; sl < h->partner.size() && fits; sl++) {
                const int64_t pp = h->partner[sl], dl = pp - (int64_t)sl;
                if (pp < 0) p16[sl] = -32768;
                else if (dl < -32767 || dl > 32767) fits = false;
                else p16[sl] = (int16_t)dl;
            }
            if (fits) CX_TRY(dev_upload(h, &h->d_partner16, p16));
        }
        CX_TRY(dev_upload(h, &h->d_vbase, h->vbase));
        CX_TRY(dev_upload(h, &h->d_var_deg, var_deg));
        CX_TRY(dev_upload(h, &h->d_vinfo, h->vinfo));
        CX_TRY(dev_upload(h, &h->d_big, h->big_vars));
        CX_TRY(dev_upload(h, &h->d_big_slots, h->big_slots));
        if (mv && !h->big_vars.empty()) {
            std::vector<int32_t> sv;
            sv.reserve(h->big_slots.size());
            for (int32_t v : h->big_vars) sv.insert(sv.end(), (size_t)var_deg[v], v);
            CX_TRY(dev_upload(h, &h->d_big_slot_var, sv));
        }
        CX_TRY(dev_alloc(h, &h->d_big_tmp, big_total));
        CX_TRY(dev_alloc(h, &h->d_scratch, 4096));
        if (mv) {
            const int64_t nc = h->nc, ncs = h->ncs;
            h->spdir = spdir; h->spdir_dirty = true;
            CX_TRY(dev_upload(h, &h->d_spdir, spdir));
            CX_TRY(dev_alloc(h, &h->d_mv_f2v, ncs * slots)); CX_TRY(dev_alloc(h, &h->d_mv_f2v_alt, ncs * slots));
            CX_TRY(dev_alloc(h, &h->d_mv_v2f, ncs * slots)); CX_TRY(dev_alloc(h, &h->d_mv_marg, cx::is_mfma_dim(h->cfg.dim) ? 1 : ncs * h->nslices * cx::kBlock));      // pair form by variable, whole 256-blocks
            CX_HIP(h, hipMemsetAsync(h->d_mv_f2v, 0xff, (size_t)(ncs * slots) * 8, h->stream));
            CX_HIP(h, hipMemsetAsync(h->d_mv_f2v_alt, 0xff, (size_t)(ncs * slots) * 8, h->stream));
            CX_HIP(h, hipMemsetAsync(h->d_mv_v2f, 0xff, (size_t)(ncs * slots) * 8, h->stream));
            if (!cx::is_mfma_dim(h->cfg.dim)) CX_HIP(h, hipMemsetAsync(h->d_mv_marg, 0xff, (size_t)(ncs * h->nslices * cx::kBlock) * 8, h->stream));
            CX_HIP(h, hipStreamSynchronize(h->stream));
            if (h->cfg.schedule == CX_SCHED_REFERENCE) CX_TRY(ref_build(h));      // dim 2 .. 4: the same wiring and shadow, the stages through k_batch_mv
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
#define CX_TRY2(x) do { int32_t rc2_ = (x); if (rc2_ != CX_OK) { dev_free_all(h); return rc2_; } } while (0)
        CX_HIP(h, hipStreamSynchronize(h->stream));
        if (h->n_kary) { if (h->slot_kary.empty()) h->slot_kary.assign(slots, -1); CX_TRY2(cx::kary_upload(h)); }
        if (h->cfg.schedule == CX_SCHED_REFERENCE) CX_TRY2(ref_build(h));      // the default resolver's wiring + the shadow of the readiness state
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


}  // extern "C"
