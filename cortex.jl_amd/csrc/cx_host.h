// cx_host.h — what the host-side translation units of libcortex_hip.so share (cx_api*.hip): status helpers, device
// allocation, id lookup, payload conversion, and the few functions one section calls in another.
// No exception leaves an entry point; every one returns a status (include/cortex_hip.h).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <numeric>

#include "cx_internal.h"

namespace cxh {

inline thread_local std::string g_create_error;

// The one-launch chain scan (cx_chain.hip: k_chain_onepass) raises a word in mapped host memory when one of its bounded waits times out
// (a workgroup of the launch never became resident: CU mask, another tenant's persistent kernel).  Whoever checks a HIP call next
// finds it: the call fails, the handle uses the two-launch scan from then on.  A chain-scan sweep is exact whatever state it starts
// from, so the caller repeats the sweep.
constexpr const char *kChainAbortMessage = "the one-launch chain scan timed out waiting for a workgroup that never became resident; the last chain-scan sweep "
                                           "stored nothing — call cx_sweep again (this handle now uses the two-launch scan)";
template <class H>
inline bool chain_abort_take(H *h) {
    if (!h || !h->chain_abort_host || !*h->chain_abort_host) return false;
    *const_cast<volatile unsigned *>(h->chain_abort_host) = 0;
    const_cast<cx_handle *>(static_cast<const cx_handle *>(h))->chain_onepass_state = -1;
    return true;
}
inline bool chain_abort_take(std::nullptr_t) { return false; }
inline int32_t fail(cx_handle *h, int32_t code, const std::string &msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define CX_HIP(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return cxh::fail(h, e_ == hipErrorOutOfMemory ? CX_ERR_OUT_OF_MEMORY : CX_ERR_DEVICE, \
                             std::string(#call) + ": " + hipGetErrorString(e_));                 \
        if (cxh::chain_abort_take(h))                                                            \
            return cxh::fail(h, CX_ERR_DEVICE, cxh::kChainAbortMessage);                         \
    } while (0)

#define CX_REQUIRE(h, cond, code, msg) \
    do { if (!(cond)) return cxh::fail(h, code, msg); } while (0)

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

inline int32_t ensure_stage(cx_handle *h, int64_t bytes) {
    if (bytes <= h->stage_bytes) return CX_OK;
    if (h->d_stage) { CX_HIP(h, hipStreamSynchronize(h->stream)); (void)hipFree(h->d_stage); h->d_stage = nullptr; }
    int64_t want = std::max<int64_t>(bytes, 1 << 20);
    CX_HIP(h, hipMalloc(&h->d_stage, (size_t)want));
    h->stage_bytes = want;
    return CX_OK;
}

// (variable_id, factor_id) -> edge index; edges are sorted by (variable, factor)
inline int64_t find_var(const cx_handle *h, int64_t var_id);
inline int64_t find_edge(const cx_handle *h, int64_t var_id, int64_t fac_id) {
    const int64_t v = find_var(h, var_id);
    if (v < 0) return -1;
    auto b = h->edge_fac_id.begin() + h->var_off[v], e = h->edge_fac_id.begin() + h->var_off[v + 1];
    auto jt = std::lower_bound(b, e, fac_id);
    if (jt == e || *jt != fac_id) return -1;
    return jt - h->edge_fac_id.begin();
}

inline int64_t find_var(const cx_handle *h, int64_t var_id) {
    // ids handed out by one counter (BipartiteFactorGraphs add_variable!) are often 1, 2, 3, ...: look there first
    if (var_id >= 1 && var_id <= (int64_t)h->var_ids.size() && h->var_ids[var_id - 1] == var_id) return var_id - 1;
    auto it = std::lower_bound(h->var_ids.begin(), h->var_ids.end(), var_id);
    if (it == h->var_ids.end() || *it != var_id) return -1;
    return it - h->var_ids.begin();
}

inline int64_t find_factor(const cx_handle *h, int64_t fac_id) {
    auto it = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), fac_id);
    if (it == h->fac_ids.end() || *it != fac_id) return -1;
    return it - h->fac_ids.begin();
}

const double kNaN = std::numeric_limits<double>::quiet_NaN();
const double kInf = std::numeric_limits<double>::infinity();

// boundary form -> storage (natural) form
inline bool to_natural(int32_t form, const double *p, double2 *out) {
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

inline void from_natural(int32_t form, double2 m, double *out) {
    if (form == CX_FORM_NATURAL) { out[0] = m.x; out[1] = m.y; return; }
    if (std::isnan(m.y)) { out[0] = kNaN; if (form == CX_FORM_MOMENT) out[1] = kNaN; return; }
    if (m.y == kInf) { out[0] = m.x; if (form == CX_FORM_MOMENT) out[1] = 0.0; return; }
    double var = 1.0 / m.y;
    out[0] = m.x * var;
    if (form == CX_FORM_MOMENT) out[1] = var;
}

inline bool is_vmp(const cx_handle *h) {
    return h->cfg.family == CX_FAMILY_VMP_MEAN_FIELD || h->cfg.family == CX_FAMILY_VMP_STRUCTURED;
}
#define CX_NOT_VMP(h, name) CX_REQUIRE(h, !(h) || !cxh::is_vmp(h), CX_ERR_UNSUPPORTED, name ": not available for the variational families (their state is the set of marginals: cx_set_marginals / cx_update_marginals)")

// (variable_id, factor_id) lists -> slots (+ optionally the local variable numbers)
inline int32_t stage_slots(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, std::vector<int32_t> &slots,
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

// ---- cx_api.hip -----------------------------------------------------------------------------------------------------
void dev_free_all(cx_handle *h);
// ---- cx_api_mv.hip: host side of dim > 1 ----------------------------------------------------------------------------
int32_t mv_set_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, int32_t direction,
                        int32_t form, const double *payload);
int32_t mv_get(cx_handle *h, const double *src, int64_t stride, const std::vector<int32_t> &idx, int32_t form, bool already_moment, double *out);      // rows of a message-form buffer by index
int32_t mv_get_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, int32_t direction,
                        int32_t form, double *out);
int32_t mv_get_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, double *out);
int32_t mv_sweep(cx_handle *h, int32_t n_sweeps);
int32_t mv_check_psets(cx_handle *h);         // every parameter set a factor names has been set
int32_t mv_residual(cx_handle *h, double *out);
int32_t mv_update_batch(cx_handle *h, const cx_item *items, int64_t n);
int32_t mv_ensure_prod_store(cx_handle *h);     // room in the dim > 1 product table for every key of prod_index
int32_t mv_chain_block_maps(cx_handle *h, double *fwd, double *bwd, double *side_first, double *side_last, int64_t *first_variable_id,
                            int64_t *last_variable_id, int64_t *n_links);
int32_t mv_ensure_marginals(cx_handle *h);    // chain scan, dim 2..4, marginals on demand: form them from the last sweep's alpha and gamma
int32_t mv_ensure_chain_msgs(cx_handle *h);   // chain scan, dim 2..4: materialise the chain messages in d_mv_f2v (every reader of it calls this)
// ---- cx_api_sweep.hip -----------------------------------------------------------------------------------------------
int32_t ensure_v2f(cx_handle *h);
int32_t build_chains(cx_handle *h);
int32_t build_tree(cx_handle *h);
int32_t tree_sweep(cx_handle *h);        // every stage of the plan on the handle's stream (one graph launch, or an XCD-resident cluster for scalar plans of wide stages)
void batch_graph_drop(cx_handle *h);       // the captured sweeps of a deep-halo batch (cx_api_sweep.hip: cx_sweep)
void tree_graph_drop(cx_handle *h);        // CX_SCHED_TREE: the stages of cx_tree_plan.h on the device (rebuilt when the set of observed variables changed)
void sweep_main(cx_handle *h, bool skip_ghosts);
void sweep_finish(cx_handle *h);
// ---- cx_api_msg.hip -------------------------------------------------------------------------------------------------
int32_t ref_set_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, int32_t form, const double *payload);   // cx_api_ref.hip
void ref_on_set_marginals(cx_handle *h, int64_t n, const int32_t *vars);
int32_t ensure_joint_store(cx_handle *h);  // the same for the joint-marginal store (registered factors: cx_handle::joint_index)
int32_t ensure_prod_store(cx_handle *h);   // the product store holds every registered ProductOfMessages node (graphs that captured its address are dropped when it moves)
// ---- cx_api_ref.hip: CX_SCHED_REFERENCE -----------------------------------------------------------------------------
int32_t ref_build(cx_handle *h);
void ref_free(cx_handle *h);
void ref_graphs_drop(cx_handle *h);
void ref_on_set(cx_handle *h, int64_t n, const int64_t *edges, int32_t direction, uint64_t set_key = 0);      // set_key != 0: the list is one cx_set_messages keeps (the state it leads to is kept too)
void ref_on_seed(cx_handle *h, int32_t direction);
void ref_on_batch(cx_handle *h, const cx_item *items, int64_t n);
// the XCD-resident cluster (cx_api_ref.hip)
bool cluster_prepare(cx_handle *h);                                        // control block + compute-unit count + environment switches; false: use launches
bool cluster_fits(const cx_handle *h, const std::vector<int64_t> &stage_off, int64_t n_stages);      // wide and deep enough, arrays below 2 GiB
void flat_records(const cx_handle *h, const std::vector<int32_t> &rec, const std::vector<int32_t> &list, const std::vector<int64_t> &stage_off, std::vector<int32_t> &flat);
// stages [0, n_stages) of a plan (host copy of the offsets: stage_off) — runs of stages on the cluster, stages wider than the chip as launches;
// synchronous; *launches (may be NULL) counts them.  An error: a barrier timed out, the plan ran in part
int32_t cluster_run(cx_handle *h, const int32_t *d_flat, const int32_t *d_rec, const int64_t *d_stage_off, const std::vector<int64_t> &stage_off, int64_t n_stages, int64_t *launches);
int32_t ref_sweep(cx_handle *h, const int32_t *req, int64_t n, const uint64_t *key_known = nullptr);
int32_t ref_sweep_all(cx_handle *h, int32_t n_sweeps);
int64_t ref_state_bytes(cx_handle *h);
void ref_state_write(cx_handle *h, char *out);
bool ref_state_read(cx_handle *h, const char *in, int64_t bytes);

}  // namespace cxh
