// cx_api_msg.hip — data injection / read-back and the batched per-signal entry point (cx_update_batch) of the C ABI.

#include <limits>
#include <cmath>

#include "cx_host.h"

using namespace cxh;

extern "C" {

// ---- a dim embedded in the next tile size (16, 32, 64; cx_create): payloads in, block-diagonal with an identity block; results out, their real block ----
namespace {
// vector[u] (+ matrix[u][u]) -> vector[64] (+ matrix[64][64] = blockdiag(matrix, I)); a NaN matrix (UndefValue) stays all NaN
void pad_payload(int u, int d, bool with_matrix, const double *in, double *out) {
    for (int k = 0; k < d; k++) out[k] = k < u ? in[k] : 0.0;
    if (!with_matrix) return;
    bool undef = false;
    for (int k = 0; k < u * u; k++) undef = undef || std::isnan(in[u + k]);
    for (int r = 0; r < d; r++)
        for (int c = 0; c < d; c++)
            out[d + r * d + c] = undef ? std::numeric_limits<double>::quiet_NaN() : (r < u && c < u) ? in[u + r * u + c] : (r == c ? 1.0 : 0.0);
    if (undef) for (int k = 0; k < d; k++) out[k] = std::numeric_limits<double>::quiet_NaN();
}
void unpad_payload(int u, int d, const double *in, double *out) {
    for (int k = 0; k < u; k++) out[k] = in[k];
    for (int r = 0; r < u; r++) for (int c = 0; c < u; c++) out[u + r * u + c] = in[d + r * d + c];
}
}  // namespace

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
    if (h->user_dim) {
        try {
            const int u = h->user_dim, d = h->cfg.dim;
            const bool mat = form != CX_FORM_POINT;
            const size_t si = mat ? (size_t)u + (size_t)u * u : (size_t)u, so = mat ? (size_t)d + (size_t)d * d : (size_t)d;
            std::vector<double> big((size_t)n * so);
            for (int64_t i = 0; i < n; i++) pad_payload(u, d, mat, payload + (size_t)i * si, &big[(size_t)i * so]);
            return mv_set_messages(h, n, variable_ids, factor_ids, direction, form, big.data());
        } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_messages: host allocation failed"); }
    }
    if (h->cfg.dim > 1) { try { return mv_set_messages(h, n, variable_ids, factor_ids, direction, form, payload); } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_messages: host allocation failed"); } }
    try {
        // a long list is usually the list of the call before (an iteration's priors, a stream's data): its translation is kept
        std::vector<int32_t> idx_local, vars_local;
        const std::vector<int32_t> *idx_p = &idx_local, *vars_p = &vars_local;
        const std::vector<int64_t> *edges_p = nullptr;
        uint64_t set_key = 0;
        constexpr int64_t kMemoMin = 4096;
        if (n >= kMemoMin) {
            uint64_t k = 0x9e3779b97f4a7c15ull ^ (uint64_t)n ^ ((uint64_t)direction << 56);
            for (int64_t i = 0; i < n; i++) { k ^= (uint64_t)variable_ids[i] * 0xff51afd7ed558ccdull; k = (k << 23 | k >> 41) + (uint64_t)factor_ids[i] * 0xc4ceb9fe1a85ec53ull; }
            k |= 1;
            cx_handle::SetMemo *m = nullptr;
            for (auto &c : h->set_memos)
                if (c.key == k && c.direction == direction && (int64_t)c.var_ids.size() == n && std::memcmp(c.var_ids.data(), variable_ids, (size_t)n * 8) == 0 &&
                    std::memcmp(c.fac_ids.data(), factor_ids, (size_t)n * 8) == 0) { m = &c; break; }
            if (!m) {
                cx_handle::SetMemo c;
                c.key = k; c.direction = direction;
                int32_t rc0 = stage_slots(h, n, variable_ids, factor_ids, c.idx, &c.vars);
                if (rc0 != CX_OK) return rc0;
                c.edges.resize((size_t)n);
                for (int64_t i = 0; i < n; i++) c.edges[i] = find_edge(h, variable_ids[i], factor_ids[i]);
                c.var_ids.assign(variable_ids, variable_ids + n); c.fac_ids.assign(factor_ids, factor_ids + n);
                if (h->set_memos.size() >= 2) {      // the least recently used one goes
                    size_t lru = 0;
                    for (size_t j = 1; j < h->set_memos.size(); j++) if (h->set_memos[j].used < h->set_memos[lru].used) lru = j;
                    h->set_memos.erase(h->set_memos.begin() + lru);
                }
                h->set_memos.push_back(std::move(c));
                m = &h->set_memos.back();
            }
            m->used = ++h->set_memo_tick;
            idx_p = &m->idx; vars_p = &m->vars; edges_p = &m->edges; set_key = k;
        } else {
            int32_t rc0 = stage_slots(h, n, variable_ids, factor_ids, idx_local, &vars_local);
            if (rc0 != CX_OK) return rc0;
        }
        const std::vector<int32_t> &idx = *idx_p, &vars = *vars_p;
        int32_t rc = CX_OK;
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
                    h->vinfo_epoch++;
                    h->chains_dirty = true; h->tree_dirty = true; h->offchain_marg_dirty = true;
                    CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream));
                }
            }
        } else {
            cx::launch_scatter(h, h->d_f2v, d_idx, d_val, n);
            if (h->d_f2v_alt) cx::launch_scatter(h, h->d_f2v_alt, d_idx, d_val, n);
        }
        CX_HIP(h, hipGetLastError());
        CX_HIP(h, hipStreamSynchronize(h->stream));  // host staging vectors die here
        if (h->ref) {      // CX_SCHED_REFERENCE: the user's set_value! on the shadow of the readiness state, in list order
            if (edges_p) ref_on_set(h, n, edges_p->data(), direction, set_key);
            else {
                std::vector<int64_t> edges((size_t)n);
                for (int64_t i = 0; i < n; i++) edges[i] = find_edge(h, variable_ids[i], factor_ids[i]);
                ref_on_set(h, n, edges.data(), direction, 0);
            }
        }
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
    if (h->user_dim) {
        try {
            const int u = h->user_dim, d = h->cfg.dim;
            const size_t so = (size_t)u + (size_t)u * u, sb = (size_t)d + (size_t)d * d;
            std::vector<double> big((size_t)n * sb);
            const int32_t rc = mv_get_messages(h, n, variable_ids, factor_ids, direction, form, big.data());
            if (rc != CX_OK) return rc;
            for (int64_t i = 0; i < n; i++) unpad_payload(u, d, &big[(size_t)i * sb], out + (size_t)i * so);
            return CX_OK;
        } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_messages: host allocation failed"); }
    }
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
        if (cx::is_mfma_dim(h->cfg.dim)) {
            h->pot64_fresh = false;
            cx::mv64_launch_seed(h, h->d_mv_f2v, mean / variance, 1.0 / variance);
            cx::mv64_launch_seed(h, h->d_mv_f2v_alt, mean / variance, 1.0 / variance);
        } else {
            cx::mv_launch_seed(h, h->d_mv_f2v, mean / variance, 1.0 / variance);
            cx::mv_launch_seed(h, h->d_mv_f2v_alt, mean / variance, 1.0 / variance);
        }
        CX_HIP(h, hipGetLastError());
        try { ref_on_seed(h, direction); } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_seed_messages: host allocation failed"); }
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
    try { ref_on_seed(h, direction); } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_seed_messages: host allocation failed"); }
    return CX_OK;
}

int32_t cx_get_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, double *out) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_get_marginals: no graph");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && out, CX_ERR_INVALID_ARGUMENT, "cx_get_marginals: null argument");
    if (is_vmp(h)) return cx::vmp_get_marginals(h, n, variable_ids, out);
    if (h->user_dim) {
        try {
            const int u = h->user_dim, d = h->cfg.dim;
            const size_t so = (size_t)u + (size_t)u * u, sb = (size_t)d + (size_t)d * d;
            std::vector<double> big((size_t)n * sb);
            const int32_t rc = mv_get_marginals(h, n, variable_ids, big.data());
            if (rc != CX_OK) return rc;
            for (int64_t i = 0; i < n; i++) unpad_payload(u, d, &big[(size_t)i * sb], out + (size_t)i * so);
            return CX_OK;
        } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_marginals: host allocation failed"); }
    }
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

extern "C++" {
namespace cxh {
int32_t ensure_prod_store(cx_handle *h) {
    const double2 *before = h->d_prod;
    const int32_t rc = grow_store(h, &h->d_prod, &h->prod_cap, (int64_t)h->prod_index.size(), 1);
    if (before && h->d_prod != before) { tree_graph_drop(h); ref_graphs_drop(h); }      // captured launches hold the store's address by value
    return rc;
}
int32_t ensure_joint_store(cx_handle *h) {
    const double *before = h->d_joint;
    const int32_t rc = grow_store(h, &h->d_joint, &h->joint_cap, (int64_t)h->joint_index.size(), 6);
    if (before && h->d_joint != before) { tree_graph_drop(h); ref_graphs_drop(h); }
    return rc;
}
}  // namespace cxh
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

static int32_t update_batch(cx_handle *h, const cx_item *items, int64_t n) {
    if (h) { h->chain_side_dirty = true; h->offchain_marg_dirty = true; }
    CX_NOT_VMP(h, "cx_update_batch");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_update_batch: no graph");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && items, CX_ERR_INVALID_ARGUMENT, "cx_update_batch: null argument");
    if (h->cfg.dim > 1) { try { return mv_update_batch(h, items, n); } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_update_batch: host allocation failed"); } }
    try {
        std::vector<int32_t> buf(5 * n, 0), kary_entries;
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
                // a message out of a factor with more than two edges: computed from ALL its other stored variable→factor messages
                // (cx_kary.hip) behind the batch's other items; the pairwise path sees a slot without a partner and leaves it alone
                if (it.kind == CX_ITEM_MESSAGE_TO_VARIABLE && !h->slot_kary.empty() && h->slot_kary[idx] >= 0) kary_entries.push_back(h->slot_kary[idx]);
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
        int32_t rc = ensure_prod_store(h);
        if (rc != CX_OK) return rc;
        rc = ensure_joint_store(h);
        if (rc != CX_OK) return rc;
        rc = ensure_v2f(h);
        if (rc != CX_OK) return rc;
        if (!kary_entries.empty()) {
            if ((rc = cx::kary_upload(h)) != CX_OK) return rc;
            const int64_t nk = (int64_t)kary_entries.size();
            int32_t *d_en = nullptr;
            CX_HIP(h, hipMalloc((void **)&d_en, (size_t)nk * 4));
            hipError_t ce = hipMemcpyAsync(d_en, kary_entries.data(), (size_t)nk * 4, hipMemcpyHostToDevice, h->stream);
            if (ce == hipSuccess) { cx::launch_kary_items(h, d_en, nk); ce = hipStreamSynchronize(h->stream); }
            (void)hipFree(d_en);
            CX_HIP(h, ce);
        }
        if (n <= cx::kSmallBatch) {
            // a per-signal process! or a wavefront of a few signals: the records ride in the kernel arguments and the call returns
            // as soon as the launch is queued.  What the host does next — setting readiness bits (signal.jl:232-253) — does not read
            // the device; whatever does (cx_get_*, cx_residual, ...) waits for the stream first.
            cx::SmallBatch sb{};
            std::memcpy(sb.r, buf.data(), (size_t)(5 * n) * 4);
            cx::launch_batch_small(h, sb, (int)n);
            CX_HIP(h, hipGetLastError());
            ref_on_batch(h, items, n);
            return CX_OK;
        }
        rc = ensure_stage(h, 5 * n * 4);
        if (rc != CX_OK) return rc;
        CX_HIP(h, hipMemcpyAsync(h->d_stage, buf.data(), 5 * n * 4, hipMemcpyHostToDevice, h->stream));
        cx::launch_batch(h, (const int32_t *)h->d_stage, n);
        CX_HIP(h, hipGetLastError());
        CX_HIP(h, hipStreamSynchronize(h->stream));  // the staging buffer is the handle's: the next call may overwrite it
        ref_on_batch(h, items, n);
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_update_batch: host allocation failed"); }
}

// cx_update_batch: complete at return.  cx_update_batch_async: a batch of up to kSmallBatch items (dim 1 - 4) returns when its launch
// is queued — the form a scheduler uses between two set_value!s; larger batches and dim 64 are complete at return either way.
int32_t cx_update_batch_async(cx_handle *h, const cx_item *items, int64_t n) { return update_batch(h, items, n); }
int32_t cx_update_batch(cx_handle *h, const cx_item *items, int64_t n) {
    const int32_t rc = update_batch(h, items, n);
    if (rc != CX_OK || n == 0) return rc;
    CX_HIP(h, hipStreamSynchronize(h->stream));
    return CX_OK;
}

// stored ProductOfMessages values (natural form for CX_FORM_NATURAL, (mean, variance) for CX_FORM_MOMENT); a node that was
// never computed reads as UndefValue() (NaN)
int32_t cx_get_products(cx_handle *h, int64_t n, const int64_t *variable_ids, const int32_t *range_lo, const int32_t *range_hi,
                        int32_t form, double *out) {
    CX_NOT_VMP(h, "cx_get_products");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_get_products: no graph");
    CX_REQUIRE(h, form == CX_FORM_MOMENT || form == CX_FORM_NATURAL, CX_ERR_INVALID_ARGUMENT, "cx_get_products: bad form");
    CX_REQUIRE(h, h->cfg.family == CX_FAMILY_GAUSSIAN || form == CX_FORM_NATURAL, CX_ERR_UNSUPPORTED, "cx_get_products: CX_FAMILY_NATURAL2 returns CX_FORM_NATURAL payloads only");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && range_lo && range_hi && out, CX_ERR_INVALID_ARGUMENT, "cx_get_products: null argument");
    if (h->cfg.dim > 1) {
        // dim > 1: rows of the device table (a node that was never computed reads as UndefValue(): the table's spare entry 0 .. cap - 1 are
        // NaN until written; an unknown node gets a NaN row)
        try {
            const int d = h->cfg.dim, u = h->user_dim ? h->user_dim : d;
            const size_t sb = (size_t)d + (size_t)d * d, so = (size_t)u + (size_t)u * u;
            std::vector<int32_t> idx;
            std::vector<int64_t> where;
            for (int64_t i = 0; i < n; i++) {
                const int64_t v = find_var(h, variable_ids[i]);
                if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
                auto it = h->prod_index.find(std::make_tuple((int32_t)v, range_lo[i], range_hi[i]));
                if (it != h->prod_index.end() && it->second < h->mv_prod_cap) { idx.push_back(it->second); where.push_back(i); }
                for (size_t k = 0; k < so; k++) out[(size_t)i * so + k] = kNaN;
            }
            if (idx.empty()) return CX_OK;
            std::vector<double> rows(idx.size() * sb);
            const int32_t rc = mv_get(h, h->d_mv_prod, h->mv_prod_cap, idx, form, false, rows.data());
            if (rc != CX_OK) return rc;
            for (size_t k = 0; k < idx.size(); k++) {
                if (h->user_dim) unpad_payload(u, d, &rows[k * sb], out + (size_t)where[k] * so);
                else std::memcpy(out + (size_t)where[k] * so, &rows[k * sb], sb * 8);
            }
            return CX_OK;
        } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_products: host allocation failed"); }
    }
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

}  // extern "C"
