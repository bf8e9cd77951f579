// cx_api_state.hip — variational-family entry points, checkpoint (cx_state_*), profiling.

#include "cx_host.h"

using namespace cxh;

extern "C" {

// ---- variational families (cx_vmp.hip) --------------------------------------------------------------------------------
int32_t cx_set_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, int32_t form, const double *payload) {
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_set_marginals: no graph");
    CX_REQUIRE(h, is_vmp(h) || (h->ref && h->cfg.dim == 1), CX_ERR_UNSUPPORTED,
               "cx_set_marginals: the variational families and the reference-order schedule (a wiring whose messages depend on marginals) keep settable marginals; elsewhere a marginal is the "
               "product of the messages: cx_set_messages");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && payload, CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: null argument");
    if (!is_vmp(h)) return ref_set_marginals(h, n, variable_ids, form, payload);
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
struct StatePart { int32_t id; void *dev; int64_t bytes; };      // dev == nullptr: a host section (id 7: the reference schedule's readiness shadow)
const char kStateMagic[8] = {'C', 'X', 'S', 'T', 'A', 'T', 'E', '2'};   // '2': the fingerprint covers rule parameters, v2f_stale is a bit-field
const char kStateMagicV1[8] = {'C', 'X', 'S', 'T', 'A', 'T', 'E', '1'};

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
    if (h->n_kary) { f = fnv1a(f, h->kary_slot.data(), h->kary_slot.size() * 4); f = fnv1a(f, h->kary_coef.data(), h->kary_coef.size() * 8); }
    if (h->n_kary && h->cfg.dim > 1) f = fnv1a(f, h->kary_pset.data(), h->kary_pset.size() * 4);
    if (h->user_dim) { const int32_t u = h->user_dim; f = fnv1a(f, &u, 4); }      // dim 5 .. 63 embedded in 64: the real block's size is part of the model
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
        if (h->ref) {      // CX_SCHED_REFERENCE: the segment-tree nodes are values the next call may read; the shadow decides what it computes
            parts.push_back({6, h->d_prod, (int64_t)h->prod_index.size() * 16});
            parts.push_back({7, nullptr, ref_state_bytes(h)});
            // the joint marginals of a user wiring (cx_graph_wire) are values too: the messages to the precisions read them
            if (!h->joint_index.empty()) parts.push_back({8, h->d_joint, (int64_t)h->joint_index.size() * 48});
        }
    } else {
        const int64_t nc = h->nc, ncs = h->ncs;
        parts.push_back({2, h->d_mv_f2v, ncs * slots * 8});
        parts.push_back({3, h->d_mv_f2v_alt, ncs * slots * 8});
        parts.push_back({4, h->d_mv_v2f, ncs * slots * 8});
        if (!cx::is_mfma_dim(h->cfg.dim)) parts.push_back({5, h->d_mv_marg, ncs * h->nslices * cx::kBlock * 8});
        if (h->ref) parts.push_back({7, nullptr, ref_state_bytes(h)});      // CX_SCHED_REFERENCE, dim 2 .. 4: the shadow decides what the next call computes
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
    if (h->cfg.dim > 1) { int32_t rc = mv_ensure_chain_msgs(h); if (rc != CX_OK) return rc; }
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
        if (p.bytes && p.dev) CX_HIP(h, hipMemcpy(o, p.dev, (size_t)p.bytes, hipMemcpyDeviceToHost));
        else if (p.bytes && p.id == 7) ref_state_write(h, o);
        o += p.bytes;
    }
    return CX_OK;
}

int32_t cx_state_import(cx_handle *h, const void *buf, int64_t bytes) {
    if (h) { h->chain_side_dirty = true; h->offchain_marg_dirty = true; h->pot64_fresh = false; }
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_state_import: no graph");
    if (is_vmp(h)) return cx::vmp_state_import(h, buf, bytes);
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, "cx_state_import: a cx_sweep_begin is still open");
    CX_REQUIRE(h, buf && bytes >= (int64_t)sizeof(StateHeader), CX_ERR_INVALID_ARGUMENT, "cx_state_import: blob too short");
    StateHeader hd;
    std::memcpy(&hd, buf, sizeof hd);
    CX_REQUIRE(h, std::memcmp(hd.magic, kStateMagicV1, 8) != 0, CX_ERR_INVALID_ARGUMENT,
               "cx_state_import: the blob was written by an earlier build (state format 1); this build reads format 2");
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
        if (p.id == 1) { std::memcpy(h->vinfo.data(), o, (size_t)p.bytes); h->vinfo_epoch++; }
        if (p.bytes && p.dev) CX_HIP(h, hipMemcpy(p.dev, o, (size_t)p.bytes, hipMemcpyHostToDevice));
        else if (p.bytes && p.id == 7) CX_REQUIRE(h, ref_state_read(h, o, p.bytes), CX_ERR_INVALID_ARGUMENT, "cx_state_import: the readiness section does not fit this handle's wiring");
        o += p.bytes;
    }
    h->sweeps_done = hd.sweeps_done;
    h->v2f_stale = (hd.v2f_stale & 1) != 0;
    h->offchain_marg_dirty = (hd.v2f_stale & 2) != 0;     // the marginals themselves travelled in section 5
    h->chain_msgs_stale = false; h->mvc_marg_pending = false;     // (the imported marginals are final)
    h->spdir_dirty = h->work64_dirty = h->point64_dirty = h->chains_dirty = true; h->tree_dirty = true;   // derived from the observed flags
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
