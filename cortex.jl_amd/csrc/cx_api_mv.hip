// ==================================================================================================================
// dim > 1 (cx_mv.hip): host side of the data path.  Payload rows at the ABI: MOMENT mean[d] + covariance[d*d];
// NATURAL eta[d] + Lambda[d*d]; POINT y[d].  Device: eta[d] + packed upper triangle of Lambda, component-major.
// ==================================================================================================================
#include "cx_host.h"

using namespace cxh;

namespace cxh {

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
    // fused schedule: the input buffer of the last sweep; chain scan: the one buffer there is (its messages are the fixed point)
    const double *src = (h->sweeps_done > 0 && h->cfg.schedule != CX_SCHED_CHAIN_SCAN && h->cfg.schedule != CX_SCHED_TREE) ? h->d_mv_f2v_alt : h->d_mv_f2v;
    if (cx::is_mfma_dim(h->cfg.dim)) cx::mv64_launch_v2f(h, (int)n, d_s, d_v, src);
    else cx::mv_launch_v2f(h, d_s, d_v, n, src);
    CX_HIP(h, hipGetLastError());
    CX_HIP(h, hipStreamSynchronize(h->stream));
    return CX_OK;
}

int32_t mv64_set_messages(cx_handle *h, int64_t n, const std::vector<int32_t> &idx, const std::vector<int32_t> &vars, int32_t direction,
                          int32_t form, const double *payload) {
    const int d = h->cfg.dim, nc = h->nc;
    const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16;
    int32_t rc;
    // (a partition's time block: the messages that enter it at its two ends are the only ones the fresh potentials do not depend on)
    for (int64_t i = 0; i < n && h->pot64_fresh; i++)
        h->pot64_fresh = direction == CX_TO_VARIABLE && std::find(h->pot64_end_slots, h->pot64_end_slots + 6, idx[i]) != h->pot64_end_slots + 6;
    if (form == CX_FORM_POINT) {
        rc = ensure_stage(h, bytes_idx + n * d * 8);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double *d_val = (double *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        CX_HIP(h, hipMemcpyAsync(d_val, payload, (size_t)n * d * 8, hipMemcpyHostToDevice, h->stream));
        cx::mv64_set_point(h, h->d_mv_v2f, d_idx, d_val, n);
        bool newly = false;
        for (int64_t i = 0; i < n; i++)
            if (!(h->vinfo[vars[i]] & cx::kClamped)) { h->vinfo[vars[i]] |= cx::kClamped; newly = true; }
        if (newly) {     // the work lists depend on WHICH variables are observed, not on their data
            CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream));
            h->work64_dirty = true;
            h->chains_dirty = true; h->tree_dirty = true;      // a newly observed variable leaves the chains
        }
        h->point64_dirty = true;      // the constant messages out of the observed variables are due again
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
    if (cx::is_mfma_dim(d)) {
        rc = mv64_set_messages(h, n, idx, vars, direction, form, payload);
        if (rc == CX_OK && h->ref) {      // CX_SCHED_REFERENCE: the user's set_value! on the shadow of the readiness state, in list order
            std::vector<int64_t> edges((size_t)n);
            for (int64_t i = 0; i < n; i++) edges[i] = find_edge(h, variable_ids[i], factor_ids[i]);
            ref_on_set(h, n, edges.data(), direction, 0);
        }
        return rc;
    }
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
        cx::mv_launch_scatter(h, h->d_mv_v2f, h->nslots, nc, h->ncs, d_idx, d_val, n);
        h->observed_passes_due = 2;   // a stored variable→factor message changed: observed senders are refreshed
        if (form == CX_FORM_POINT) {
            // New data for variables that were observed already leaves the structure (observed flags, rule masks, chains) as it is:
            // only the constant messages out of them are due again (observed_passes_due above).
            bool newly = false;
            for (int64_t i = 0; i < n; i++)
                if (!(h->vinfo[vars[i]] & cx::kClamped)) { h->vinfo[vars[i]] |= cx::kClamped; newly = true; }
            if (newly) {
                h->chains_dirty = true; h->tree_dirty = true;      // a newly observed variable leaves the chains
                CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream));
                h->spdir_dirty = true;
            }
        }
    } else {
        cx::mv_launch_scatter(h, h->d_mv_f2v, h->nslots, nc, h->ncs, d_idx, d_val, n);
        cx::mv_launch_scatter(h, h->d_mv_f2v_alt, h->nslots, nc, h->ncs, d_idx, d_val, n);
    }
    CX_HIP(h, hipGetLastError());
    CX_HIP(h, hipStreamSynchronize(h->stream));
    if (h->ref) {      // CX_SCHED_REFERENCE: the user's set_value! on the shadow of the readiness state, in list order
        std::vector<int64_t> edges((size_t)n);
        for (int64_t i = 0; i < n; i++) edges[i] = find_edge(h, variable_ids[i], factor_ids[i]);
        ref_on_set(h, n, edges.data(), direction, 0);
    }
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
    if (cx::is_mfma_dim(d)) {
        if (already_moment) {
            // idx = variables: computed on demand.  A marginal the kernel cannot form (an incoming message undefined, a total precision
            // that is not positive definite) is not stored: the staging rows start as UndefValue()
            CX_HIP(h, hipMemsetAsync(d_val, 0xff, (size_t)n * nc * 8, h->stream));
            cx::mv64_launch_marginals(h, (int)n, d_idx, h->d_mv_f2v, d_val);
        }
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
    cx::mv_launch_gather(h, src, stride, nc, h->ncs, d_idx, d_val, n);      // messages by slot, marginals by variable: the same pair form
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
    if ((rc = mv_ensure_chain_msgs(h)) != CX_OK) return rc;
    // (the reference-order schedule STORES its variable→factor messages — each is a signal with its own place in the order — where the
    // sweeps that keep them in registers recompute them for the reader)
    if (direction == CX_TO_FACTOR && h->cfg.schedule != CX_SCHED_REFERENCE) { rc = mv_refresh_v2f(h, idx, vars); if (rc != CX_OK) return rc; }
    return mv_get(h, direction == CX_TO_FACTOR ? h->d_mv_v2f : h->d_mv_f2v, h->nslots, idx, form, false, out);
}

int32_t mv_get_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, double *out) {
    std::vector<int32_t> idx(n);
    for (int64_t i = 0; i < n; i++) {
        int64_t v = find_var(h, variable_ids[i]);
        if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
        idx[i] = (int32_t)v;
    }
    // marginals on demand: a request for a few variables forms just those from the walks' sums and leaves the rest owed; a request for
    // an eighth of the variables or more runs the whole pass once (62 us at C3) and reads the finished array
    if (h->mvc_marg_pending && h->d_mvc_var_link && h->d_mvc_alpha && n * 8 < h->nv) {
        const int d = h->cfg.dim, nc = h->nc;
        const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16;
        int32_t rc = ensure_stage(h, bytes_idx + n * nc * 8);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double *d_val = (double *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice, h->stream));
        cx::mvc_launch_marg_gather(h, d_idx, n, d_val);
        CX_HIP(h, hipGetLastError());
        std::vector<double> val((size_t)n * nc);
        CX_HIP(h, hipMemcpyAsync(val.data(), d_val, (size_t)n * nc * 8, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < n; i++) { double *o = out + i * (d + d * d); mv_unpack(d, &val[(size_t)i * nc], o, o + d); }
        return CX_OK;
    }
    { int32_t rc = mv_ensure_marginals(h); if (rc != CX_OK) return rc; }
    return mv_get(h, h->d_mv_marg, h->nv, idx, CX_FORM_MOMENT, true, out);
}

// d = 64: which messages need the full MFMA rule, which come from observed variables (constant), which nobody reads
int32_t build_work64(cx_handle *h) {
    if (!h->work64_dirty) return CX_OK;
    std::vector<int32_t> ps, rec, pre_s, pre_v, slot_var(h->nslots, -1);
    for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
    for (int64_t e = 0; e < h->ne; e++) {
        const int32_t s = cx::slot_of_edge(h, e), p = h->partner[s], v = h->edge_var[e];
        if (p < 0) continue;
        const int32_t rvz = slot_var[p];
        if (h->vinfo[rvz] & cx::kClamped) continue;                 // a message into an observed variable has no reader
        const int32_t deg = h->var_off[v + 1] - h->var_off[v];
        if (h->vinfo[v] & cx::kClamped) { ps.push_back(s); continue; }
        // the record the rule kernel reads: sender slot, the other incoming slots in ascending neighbour order (the fold
        // order), rule-table index, destination slot, flags
        int32_t others[3] = {-1, -1, -1};
        int n_others = 0;
        const int32_t stride = cx::slot_stride(h, v);      // (a variable of degree > 8 lives in the CSR tail: consecutive slots)
        for (int32_t j = 0; j < deg; j++) {
            const int32_t sj = h->vbase[v] + j * stride;
            if (sj != s && n_others < 3) others[n_others++] = sj;
        }
        if (deg > 4) {      // more than three other messages: k_v2f64 sums them into the stored variable→factor message, the rule reads that
            pre_s.push_back(s); pre_v.push_back(v);
            rec.insert(rec.end(), {s, -1, -1, -1, h->spdir[s], p, 1, 0});
        } else
        rec.insert(rec.end(), {s, others[0], others[1], others[2], h->spdir[s], p, deg < 2 ? 1 : 0, 0});      // (flag 1, degree-1 leaf: its stored message is the input)
    }
    for (void *p : {(void *)h->d_point64_slots, (void *)h->d_rule64_rec, (void *)h->d_pre64_slots, (void *)h->d_pre64_vars}) if (p) (void)hipFree(p);
    h->d_point64_slots = h->d_rule64_rec = h->d_pre64_slots = h->d_pre64_vars = nullptr;
    h->n_rule64 = (int64_t)rec.size() / 8; h->n_point64 = (int64_t)ps.size(); h->n_pre64 = (int64_t)pre_s.size();
    int32_t rc;
    if ((rc = dev_upload(h, &h->d_pre64_slots, pre_s)) != CX_OK) return rc;
    if ((rc = dev_upload(h, &h->d_pre64_vars, pre_v)) != CX_OK) return rc;
    if ((rc = dev_upload(h, &h->d_point64_slots, ps)) != CX_OK) return rc;
    if ((rc = dev_upload(h, &h->d_rule64_rec, rec)) != CX_OK) return rc;
    CX_HIP(h, hipStreamSynchronize(h->stream));
    h->work64_dirty = false;
    h->point64_dirty = true;
    return CX_OK;
}

// mask the rules whose receiver is an observed variable (nobody reads a message into it)
static int32_t mv_refresh_spdir(cx_handle *h) {
    if (!h->spdir_dirty) return CX_OK;
    std::vector<int32_t> eff(h->spdir), slot_var(h->nslots, -1);
    for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
    for (int64_t sl = 0; sl < h->nslots; sl++) {
        const int32_t p = h->partner[sl];
        if (p >= 0 && (h->vinfo[slot_var[p]] & cx::kClamped)) eff[sl] = -1;
    }
    CX_HIP(h, hipMemcpy(h->d_spdir, eff.data(), eff.size() * 4, hipMemcpyHostToDevice));
    h->spdir_dirty = false;
    h->observed_passes_due = 2;   // the data (or the set of observed variables) changed: refresh both buffers
    return CX_OK;
}

// CX_SCHED_CHAIN_SCAN for dim 2..4 (cx_mvchain.hip): one sweep = the exact forward/backward pass along every chain.
// There is ONE message buffer (d_mv_f2v): the scan overwrites the chain messages with their exact values.
static int32_t mv_chain_sweep(cx_handle *h, int32_t n_sweeps) {
    int32_t rc = build_chains(h);
    if (rc != CX_OK) return rc;
    CX_REQUIRE(h, h->chain_covers_all, CX_ERR_UNSUPPORTED,
               "cx_sweep: the chain-scan schedule for dim > 1 needs every non-observed variable on a chain (a non-observed variable of degree 1 "
               "or a stand-in reads messages the scan does not produce): use the fused schedule for this graph");
    if (cx::is_mfma_dim(h->cfg.dim)) {
        // dim 64 (cx_mv64chain.hip): the messages out of observed variables are constants (k_point64, once per change of data or rule
        // tables); a sweep composes the blocks' potentials and walks every path in both directions, writing the exact messages into
        // the ONE message buffer.  Marginals of dim 64 are formed from the stored messages when they are read.
        if ((rc = build_work64(h)) != CX_OK) return rc;
        if (h->point64_dirty) {
            cx::mv64_launch_point(h, (int)h->n_point64, h->d_point64_slots, h->d_mv_f2v, h->d_mv_f2v_alt);
            h->point64_dirty = false;
            h->pot64_fresh = false;
        }
        for (int32_t s = 0; s < n_sweeps; s++) {
            if ((rc = cx::chain64_sweep(h)) != CX_OK) return rc;
            h->sweeps_done++;
        }
        CX_HIP(h, hipGetLastError());
        return CX_OK;
    }
    if ((rc = mv_refresh_spdir(h)) != CX_OK) return rc;
    const bool marg = h->cfg.compute_marginals_in_sweep != 0;
    for (int32_t s = 0; s < n_sweeps; s++) {
        if (h->observed_passes_due > 0 || h->chain_side_dirty) {
            // the constant messages into the chains: out of observed variables (data through the likelihood rule) and out of
            // other senders of stored messages, then the side sums — only after data, stored messages or rule tables changed
            cx::mv_launch_sweep(h, false, 3, h->d_mv_f2v);
            cx::mvc_launch_side(h, marg);
            h->observed_passes_due = 0; h->chain_side_dirty = false;
        }
        // compute_marginals_in_sweep == 2: the marginals on demand, like the chain messages (and like dim 64): the sweep leaves alpha and
        // gamma in the walks' order, the pass that turns them into marginals runs before the first reader (mv_ensure_marginals)
        const bool defer = h->cfg.compute_marginals_in_sweep == 2;
        cx::mvc_launch_scan(h, marg, false, true, defer);
        h->mvc_marg_pending = marg && defer;
        h->chain_msgs_stale = true;      // the chain messages live in the scan's own buffers until somebody asks for them
        h->sweeps_done++;
    }
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

// cx_chain_block_maps for dim 2..4: the composed forward / backward maps of the handle's ONE path (a time block of a partitioned
// chain) as functions of the message that enters the block at either end, and the side sums of its end variables.
// Map layout (ND = 2 d(d+1)/2 + d^2 + 2 d doubles): P packed upper | B row-major | C packed upper | h | c, of
//     f(eta, Lambda) = ( c + B (Lambda + P)^-1 (eta + h),  C - B (Lambda + P)^-1 B' );   sides: eta[d] | Lambda packed upper.
int32_t mv_chain_block_maps(cx_handle *h, double *fwd, double *bwd, double *side_first, double *side_last, int64_t *first_variable_id,
                            int64_t *last_variable_id, int64_t *n_links) {
    CX_REQUIRE(h, h->cfg.dim <= 4 || (h->cfg.dim == 64 && !h->user_dim), CX_ERR_UNSUPPORTED,
               "cx_chain_block_maps: dim 1 .. 4 and 64 (the maps of a block on the matrix-core path are 64 x 64 potentials: a smaller dim has to be embedded by the caller)");
    CX_REQUIRE(h, (int64_t)h->psets.size() > h->max_pset, CX_ERR_STATE, "cx_chain_block_maps: a parameter set was never set (cx_set_factor_matrices)");
    int32_t rc = build_chains(h);
    if (rc != CX_OK) return rc;
    CX_REQUIRE(h, h->chain_nlinks >= 1 && h->chain_nlinks == h->chain_npos - 1, CX_ERR_UNSUPPORTED,
               "cx_chain_block_maps: the non-observed variables of this handle must form ONE path of at least two variables (a time block of a chain)");
    if (cx::is_mfma_dim(h->cfg.dim)) {
        // dim 64 (round 4): the plan of cx_chain64_plan.h with a root — the compose launches leave the ONE potential of the block's two
        // end variables; forwards it is the map (P + side_first, B, C, h + side_first, c), backwards (C + side_last, B', P, c + side_last, h)
        // (the same convention as dim 2..4: a direction's map includes the side information of the end it is entered at)
        const int d = 64, dd = d * d, nt = d * (d + 1) / 2, msg = d + dd;
        if ((rc = build_work64(h)) != CX_OK) return rc;
        if (h->point64_dirty) {
            cx::mv64_launch_point(h, (int)h->n_point64, h->d_point64_slots, h->d_mv_f2v, h->d_mv_f2v_alt);
            h->point64_dirty = false;
        }
        std::vector<double> pot((size_t)4 * dd + 2 * d);
        int32_t sf[3], sl[3];
        bool no_root = false;
        if ((rc = cx::chain64_block_potential(h, pot.data(), sf, sl, &no_root)) != CX_OK) return rc;
        if (no_root) {      // the plan was built before the handle became a partition's: once more, with a root
            h->chains_dirty = true;
            if ((rc = build_chains(h)) != CX_OK) return rc;
            if ((rc = cx::chain64_block_potential(h, pot.data(), sf, sl, &no_root)) != CX_OK) return rc;
            CX_REQUIRE(h, !no_root, CX_ERR_STATE, "cx_chain_block_maps: the plan has no root potential");
        }
        std::vector<int32_t> idx;
        for (int k = 0; k < 3; k++) if (sf[k] >= 0) idx.push_back(sf[k]);
        const size_t nf = idx.size();
        for (int k = 0; k < 3; k++) if (sl[k] >= 0) idx.push_back(sl[k]);
        std::vector<double> sm(idx.size() * (size_t)msg), sum((size_t)2 * msg, 0.0);
        if (!idx.empty() && (rc = mv_get(h, h->d_mv_f2v, h->nslots, idx, CX_FORM_NATURAL, false, sm.data())) != CX_OK) return rc;
        for (size_t i = 0; i < idx.size(); i++)
            for (int k = 0; k < msg; k++) sum[(i < nf ? 0 : msg) + k] += sm[i * msg + k];
        // symmetric matrices travel as their upper 16 x 16 tile blocks (potentials) or whole (messages): read the upper triangle
        auto up = [&](const double *M, int r, int c) { return r <= c ? M[r * d + c] : M[c * d + r]; };
        auto pack = [&](double *o, const double *P, const double *B, const double *Cm, const double *hh, const double *cc, const double *side) {
            int k = 0;
            for (int r = 0; r < d; r++) for (int c = r; c < d; c++) o[k++] = up(P, r, c) + up(side + d, r, c);
            std::memcpy(o + nt, B, (size_t)dd * 8);
            k = nt + dd;
            for (int r = 0; r < d; r++) for (int c = r; c < d; c++) o[k++] = up(Cm, r, c);
            for (int r = 0; r < d; r++) o[2 * nt + dd + r] = hh[r] + side[r];
            std::memcpy(o + 2 * nt + dd + d, cc, (size_t)d * 8);
        };
        const double *P = pot.data(), *B = P + dd, *Bt = P + 2 * dd, *Cm = P + 3 * dd, *hh = P + 4 * dd, *cc = hh + d;
        pack(fwd, P, B, Cm, hh, cc, sum.data());
        pack(bwd, Cm, Bt, P, cc, hh, sum.data() + msg);
        for (int e = 0; e < 2; e++) {
            double *o = e == 0 ? side_first : side_last;
            const double *sd = sum.data() + (size_t)e * msg;
            std::memcpy(o, sd, (size_t)d * 8);
            int k = d;
            for (int r = 0; r < d; r++) for (int c = r; c < d; c++) o[k++] = up(sd + d, r, c);
        }
        int32_t pv[2] = {0, 0};
        CX_HIP(h, hipMemcpyAsync(&pv[0], h->d_chain_pos_var, 4, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipMemcpyAsync(&pv[1], h->d_chain_pos_var + (h->chain_npos - 1), 4, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        if (first_variable_id) *first_variable_id = h->var_ids[pv[0]];
        if (last_variable_id) *last_variable_id = h->var_ids[pv[1]];
        if (n_links) *n_links = h->chain_nlinks;
        return CX_OK;
    }
    if ((rc = mv_ensure_chain_msgs(h)) != CX_OK) return rc;      // the maps overwrite the stored prefixes of the last sweep
    if ((rc = mv_refresh_spdir(h)) != CX_OK) return rc;
    // the leaf messages and side sums from the data currently on the device (the caller has zeroed the cut messages)
    cx::mv_launch_sweep(h, false, 3, h->d_mv_f2v);
    cx::mvc_launch_side(h, false);
    cx::mvc_launch_block_maps(h);
    CX_HIP(h, hipGetLastError());
    h->observed_passes_due = 0;
    h->chain_side_dirty = true;          // the boundary messages change before the sweep proper
    const int d = h->cfg.dim, nc = h->nc, nd = 2 * (d * (d + 1) / 2) + d * d + 2 * d;
    std::vector<double> maps((size_t)2 * (nd + 1)), side((size_t)nc * h->chain_npos);
    CX_HIP(h, hipMemcpyAsync(maps.data(), h->d_mvc_block, maps.size() * 8, hipMemcpyDeviceToHost, h->stream));
    CX_HIP(h, hipMemcpyAsync(side.data(), h->d_mvc_side, side.size() * 8, hipMemcpyDeviceToHost, h->stream));     // [nc][npos]: small blocks only travel whole
    int32_t pv[2] = {0, 0};
    CX_HIP(h, hipMemcpyAsync(&pv[0], h->d_chain_pos_var, 4, hipMemcpyDeviceToHost, h->stream));
    CX_HIP(h, hipMemcpyAsync(&pv[1], h->d_chain_pos_var + (h->chain_npos - 1), 4, hipMemcpyDeviceToHost, h->stream));
    CX_HIP(h, hipStreamSynchronize(h->stream));
    std::memcpy(fwd, maps.data(), (size_t)nd * 8);
    std::memcpy(bwd, maps.data() + nd + 1, (size_t)nd * 8);
    for (int c = 0; c < nc; c++) { side_first[c] = side[(size_t)c * h->chain_npos]; side_last[c] = side[(size_t)c * h->chain_npos + h->chain_npos - 1]; }
    if (first_variable_id) *first_variable_id = h->var_ids[pv[0]];
    if (last_variable_id) *last_variable_id = h->var_ids[pv[1]];
    if (n_links) *n_links = h->chain_nlinks;
    return CX_OK;
}

// dim 2..4 under the chain-scan schedule: bring the chain messages in d_mv_f2v up to date (the tile carries of the last sweep are
// still on the device: two apply launches).  Every reader of d_mv_f2v calls this first.
int32_t mv_ensure_marginals(cx_handle *h) {
    if (!h->mvc_marg_pending) return CX_OK;
    h->mvc_marg_pending = false;
    if (h->d_mvc_alpha && h->chain_nlinks > 0) {
        cx::mvc_launch_marg_out(h);
        CX_HIP(h, hipGetLastError());
    }
    return CX_OK;
}

int32_t mv_ensure_chain_msgs(cx_handle *h) {
    // (every caller is about to read or to change state the deferred marginals depend on: they are formed first)
    { int32_t rc = mv_ensure_marginals(h); if (rc != CX_OK) return rc; }
    if (!h->chain_msgs_stale) return CX_OK;
    // side sums, alphas and tile carries of the last sweep are untouched until the next sweep (or a rebuild of the chains, which
    // calls this first): the two walks reproduce exactly the messages of that sweep
    if (h->d_mvc_alpha && h->chain_nlinks > 0) {
        cx::mvc_launch_scan(h, false, true, false);
        CX_HIP(h, hipGetLastError());
    }
    h->chain_msgs_stale = false;
    return CX_OK;
}

// cx_update_batch for dim > 1: MessageToFactor / MessageToVariable / IndividualMarginal items (process!'s dispatch,
// src/inference_engine.jl:479-509, on the three variants a linear-Gaussian model uses).  dim 2..4: one launch of k_batch_mv
// (cx_mvbatch.hip).  dim 64: the items are sorted into the three kernels the sweep is made of — variable→factor sums (k_v2f64), the
// MFMA rule on a stored variable→factor message (k_rule64w with the "stored input" flag), the point-mass rule (k_point64);
// marginals of dim 64 are computed from the stored messages when they are read (cx_get_marginals), so a marginal item only
// checks its variable.

// the ProductOfMessages table grows with the signals that name it (whole blocks of 256 entries; new entries read as UndefValue()):
// cx_update_batch's items and the segment-tree nodes of a reference-order wiring (cx_api_ref.hip: register_stores)
int32_t mv_ensure_prod_store(cx_handle *h) {
    if ((int64_t)h->prod_index.size() <= h->mv_prod_cap) return CX_OK;
    const bool d64 = cx::is_mfma_dim(h->cfg.dim);
    const int64_t cap = std::max<int64_t>(2 * h->mv_prod_cap, (((int64_t)h->prod_index.size() + 255) / 256) * 256);
    const int64_t per = d64 ? h->nc : h->ncs;
    double *bigger = nullptr;
    int32_t rc;
    if ((rc = dev_alloc(h, &bigger, cap * per)) != CX_OK) return rc;
    CX_HIP(h, hipMemsetAsync(bigger, 0xff, (size_t)(cap * per) * 8, h->stream));
    if (h->d_mv_prod) {
        CX_HIP(h, hipMemcpyAsync(bigger, h->d_mv_prod, (size_t)(h->mv_prod_cap * per) * 8, hipMemcpyDeviceToDevice, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        tree_graph_drop(h); ref_graphs_drop(h);      // captured launches hold the table's address by value
        (void)hipFree(h->d_mv_prod);
    }
    h->d_mv_prod = bigger; h->mv_prod_cap = cap;
    return CX_OK;
}

int32_t mv_update_batch(cx_handle *h, const cx_item *items, int64_t n) {
    h->pot64_fresh = false;
    CX_REQUIRE(h, (int64_t)h->psets.size() > h->max_pset, CX_ERR_STATE, "cx_update_batch: a factor names a parameter set that was never set (cx_set_factor_matrices)");
    for (int64_t i = 0; i <= h->max_pset; i++)
        CX_REQUIRE(h, !h->psets[i].empty(), CX_ERR_STATE, "cx_update_batch: parameter set " + std::to_string(i) + " was never set (cx_set_factor_matrices)");
    int32_t rc = mv_ensure_chain_msgs(h);
    if (rc != CX_OK) return rc;
    if ((rc = cx::kary_upload(h)) != CX_OK) return rc;
    const bool d64 = cx::is_mfma_dim(h->cfg.dim);
    std::vector<int32_t> rec, v2f_slots, v2f_vars, point_slots, rule_rec, slot_var, prod_rec;
    if (d64) {
        slot_var.assign(h->nslots, -1);
        for (int64_t e = 0; e < h->ne; e++) slot_var[cx::slot_of_edge(h, e)] = h->edge_var[e];
    } else {
        rec.assign(5 * n, 0);
    }
    for (int64_t i = 0; i < n; i++) {
        const cx_item &it = items[i];
        int64_t idx, var, tab = 0;
        if (it.kind == CX_ITEM_INDIVIDUAL_MARGINAL) {
            var = idx = find_var(h, it.variable_id);
            if (idx < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(it.variable_id));
            if (d64) continue;
        } else if (it.kind == CX_ITEM_MESSAGE_TO_FACTOR || it.kind == CX_ITEM_MESSAGE_TO_VARIABLE) {
            const int64_t e = find_edge(h, it.variable_id, it.factor_id);
            if (e < 0) return fail(h, CX_ERR_NOT_FOUND, "no connection between variable " + std::to_string(it.variable_id) + " and factor " + std::to_string(it.factor_id));
            idx = cx::slot_of_edge(h, e); var = h->edge_var[e];
            const int32_t p = h->partner[idx];
            if (!d64 && it.kind == CX_ITEM_MESSAGE_TO_VARIABLE && !h->slot_kary.empty() && h->slot_kary[idx] >= 0) {
                // a message out of a factor of more than two variables: from the stored messages of the factor's other edges (cx_kary_mv_core.h)
                rec[5 * i] = 32; rec[5 * i + 1] = h->slot_kary[idx]; rec[5 * i + 2] = (int32_t)var;
                continue;
            }
            if (it.kind == CX_ITEM_MESSAGE_TO_VARIABLE && p >= 0) tab = h->spdir[p];      // the rule table of the SENDING slot (unmasked: any message can be asked for)
            if (d64) {
                if (it.kind == CX_ITEM_MESSAGE_TO_FACTOR) { v2f_slots.push_back((int32_t)idx); v2f_vars.push_back((int32_t)var); continue; }
                if (p < 0) continue;                                                            // an opaque factor's message is the caller's to set
                if (h->vinfo[slot_var[p]] & cx::kClamped) {
                    if (h->vinfo[var] & cx::kClamped)
                        return fail(h, CX_ERR_UNSUPPORTED, "cx_update_batch: dim 64 does not compute a message between two observed variables (factor " + std::to_string(it.factor_id) + ")");
                    point_slots.push_back(p);
                } else {
                    rule_rec.insert(rule_rec.end(), {p, -1, -1, -1, (int32_t)tab, (int32_t)idx, 1, 0});      // flag 1: the stored variable→factor message of slot p is the input
                }
                continue;
            }
        } else if (it.kind == CX_ITEM_PRODUCT_OF_MESSAGES) {
            // ProductOfMessages(variable_id, range, ...), inference_signal.jl:62-66 (the range travels in factor_id): what the reference's
            // default resolver creates for a variable of degree > 5 (src/dependencies.jl:90-173)
            var = find_var(h, it.variable_id);
            if (var < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(it.variable_id));
            const int64_t lo = (int64_t)((uint64_t)it.factor_id >> 32), hi = (int64_t)((uint64_t)it.factor_id & 0xffffffffu);
            const int64_t deg = h->var_off[var + 1] - h->var_off[var];
            if (lo < 1 || hi < lo || hi > deg)
                return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_update_batch: ProductOfMessages range " + std::to_string(lo) + ":" + std::to_string(hi) +
                            " outside 1:" + std::to_string(deg) + " (variable " + std::to_string(it.variable_id) + ")");
            auto key = std::make_tuple((int32_t)var, (int32_t)lo, (int32_t)hi);
            auto pit = h->prod_index.find(key);
            if (pit == h->prod_index.end()) pit = h->prod_index.emplace(key, (int32_t)h->prod_index.size()).first;
            idx = pit->second; tab = lo;
            if (d64) {
                const int32_t stride = cx::slot_stride(h, (int32_t)var);
                prod_rec.insert(prod_rec.end(), {(int32_t)idx, (int32_t)(hi - lo + 1), h->vbase[var] + (int32_t)(lo - 1) * stride, stride});
                continue;
            }
            rec[5 * i + 4] = (int32_t)hi;
        } else {
            return fail(h, CX_ERR_UNSUPPORTED, "cx_update_batch: dim > 1 implements MessageToFactor, MessageToVariable, IndividualMarginal and ProductOfMessages items (kind " + std::to_string(it.kind) + ")");
        }
        rec[5 * i] = it.kind; rec[5 * i + 1] = (int32_t)idx; rec[5 * i + 2] = (int32_t)var; rec[5 * i + 3] = (int32_t)tab;
    }
    if ((rc = mv_ensure_prod_store(h)) != CX_OK) return rc;
    if (!d64 && n <= cx::kSmallBatch) {          // records in the kernel arguments, no wait (cx_api_msg.hip: cx_update_batch)
        cx::SmallBatch sb{};
        std::memcpy(sb.r, rec.data(), (size_t)(5 * n) * 4);
        cx::mv_launch_batch_small(h, sb, (int)n);
        CX_HIP(h, hipGetLastError());
        ref_on_batch(h, items, n);
        return CX_OK;
    }
    if (!d64) {
        if ((rc = ensure_stage(h, 5 * n * 4)) != CX_OK) return rc;
        CX_HIP(h, hipMemcpyAsync(h->d_stage, rec.data(), 5 * n * 4, hipMemcpyHostToDevice, h->stream));
        cx::mv_launch_batch(h, (const int32_t *)h->d_stage, n);
    } else {
        const int64_t n1 = (int64_t)v2f_slots.size(), n2 = (int64_t)point_slots.size(), n3 = (int64_t)rule_rec.size() / 8, n4 = (int64_t)prod_rec.size() / 4;
        if ((rc = ensure_stage(h, (2 * n1 + n2 + 8 * n3 + 4 * n4 + 4) * 4)) != CX_OK) return rc;
        int32_t *d = (int32_t *)h->d_stage;
        int32_t *d_s = d, *d_v = d + n1, *d_p = d + 2 * n1, *d_r = d + 2 * n1 + n2, *d_q = d + 2 * n1 + n2 + 8 * n3;
        if (n4) {
            CX_HIP(h, hipMemcpyAsync(d_q, prod_rec.data(), n4 * 16, hipMemcpyHostToDevice, h->stream));
            cx::mv64_launch_range_sums(h, (int)n4, d_q, h->d_mv_f2v, h->d_mv_prod);
        }
        if (n1) { CX_HIP(h, hipMemcpyAsync(d_s, v2f_slots.data(), n1 * 4, hipMemcpyHostToDevice, h->stream)); CX_HIP(h, hipMemcpyAsync(d_v, v2f_vars.data(), n1 * 4, hipMemcpyHostToDevice, h->stream)); }
        if (n2) CX_HIP(h, hipMemcpyAsync(d_p, point_slots.data(), n2 * 4, hipMemcpyHostToDevice, h->stream));
        if (n3) CX_HIP(h, hipMemcpyAsync(d_r, rule_rec.data(), n3 * 32, hipMemcpyHostToDevice, h->stream));
        cx::mv64_launch_v2f(h, (int)n1, d_s, d_v, h->d_mv_f2v);
        cx::mv64_launch_point(h, (int)n2, d_p, h->d_mv_f2v, h->d_mv_f2v);
        cx::mv64_launch_rule(h, (int)n3, d_r, h->d_mv_f2v, h->d_mv_f2v, CX_KERNEL_BATCH);
        h->point64_dirty = true;      // a sweep recomputes the constant messages into both of its buffers
    }
    CX_HIP(h, hipGetLastError());
    CX_HIP(h, hipStreamSynchronize(h->stream));      // synchronous: the host sets readiness bits next (signal.jl:232-253)
    ref_on_batch(h, items, n);
    return CX_OK;
}

int32_t mv_check_psets(cx_handle *h) {
    CX_REQUIRE(h, (int64_t)h->psets.size() > h->max_pset, CX_ERR_STATE, "cx_sweep: a factor names a parameter set that was never set (cx_set_factor_matrices)");
    for (int64_t i = 0; i <= h->max_pset; i++)
        CX_REQUIRE(h, !h->psets[i].empty(), CX_ERR_STATE, "cx_sweep: parameter set " + std::to_string(i) + " was never set (cx_set_factor_matrices)");
    return CX_OK;
}

int32_t mv_sweep(cx_handle *h, int32_t n_sweeps) {
    { const int32_t rp = mv_check_psets(h); if (rp != CX_OK) return rp; }
    if (h->cfg.schedule == CX_SCHED_CHAIN_SCAN) return mv_chain_sweep(h, n_sweeps);
    if (h->cfg.schedule == CX_SCHED_REFERENCE) return ref_sweep_all(h, n_sweeps);      // (dim 2 .. 4: cx_api_ref.hip)
    { const int32_t rck = cx::kary_upload(h); if (rck != CX_OK) return rck; }      // factors of more than two variables: their table and matrices
    if (h->cfg.schedule == CX_SCHED_TREE) {
        // dim 2..4 on a forest (cx_tree_plan.h): the stages' items through k_batch_mv, in place in the one message buffer; the items
        // carry the rule table of the sending slot as the plan found it (the plan is rebuilt when a variable becomes observed)
        int32_t rc = build_tree(h);
        if (rc != CX_OK) return rc;
        if (cx::is_mfma_dim(h->cfg.dim)) {       // the constant messages out of observed variables (as the other dim 64 schedules do)
            if ((rc = build_work64(h)) != CX_OK) return rc;
            if (h->point64_dirty) {
                cx::mv64_launch_point(h, (int)h->n_point64, h->d_point64_slots, h->d_mv_f2v, h->d_mv_f2v_alt);
                h->point64_dirty = false;
            }
        }
        // dim 64 over heavy paths: the plans' device records follow the base pointers (host copies: before, never inside, the capture of
        // the sweep's launches)
        if (!h->tree_c64.empty() && (rc = cx::chain64_tree_resolve(h)) != CX_OK) return rc;
        for (int32_t s = 0; s < n_sweeps; s++) { const int32_t rt = tree_sweep(h); if (rt != CX_OK) return rt; h->sweeps_done++; }
        CX_HIP(h, hipGetLastError());
        return CX_OK;
    }
    if (cx::is_mfma_dim(h->cfg.dim)) {
        int32_t rc = build_work64(h);
        if (rc != CX_OK) return rc;
        if (h->point64_dirty) {   // messages out of observed variables are constant: computed once, into both buffers
            cx::mv64_launch_point(h, (int)h->n_point64, h->d_point64_slots, h->d_mv_f2v, h->d_mv_f2v_alt);
            h->point64_dirty = false;
        }
    }
    if (!cx::is_mfma_dim(h->cfg.dim)) { int32_t rc = mv_refresh_spdir(h); if (rc != CX_OK) return rc; }
    for (int32_t s = 0; s < n_sweeps; s++) {
        if (cx::is_mfma_dim(h->cfg.dim)) {
            cx::mv64_launch_v2f(h, (int)h->n_pre64, h->d_pre64_slots, h->d_pre64_vars, h->d_mv_f2v);      // senders of degree 5 .. 8
            cx::mv64_launch_rule(h, (int)h->n_rule64, h->d_rule64_rec, h->d_mv_f2v, h->d_mv_f2v_alt, CX_KERNEL_FUSED);
            cx::mv64_launch_damp(h, (int)h->n_rule64, h->d_rule64_rec, h->d_mv_f2v, h->d_mv_f2v_alt, h->damping);
        } else {
            if (h->observed_passes_due > 0) { cx::mv_launch_sweep(h, false, 1); h->observed_passes_due--; }
            cx::mv_launch_sweep(h, h->cfg.compute_marginals_in_sweep != 0, 0);
            cx::mv_launch_big(h, h->cfg.compute_marginals_in_sweep != 0);      // variables of degree > 8 (none on most graphs: no launch)
            cx::mv_launch_kary(h);                                             // factors of more than two variables (the same)
        }
        std::swap(h->d_mv_f2v, h->d_mv_f2v_alt);
        h->sweeps_done++;
    }
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

int32_t mv_residual(cx_handle *h, double *out) {
    const int64_t n = h->ncs * h->nslots;      // every stored double (the padding of d = 2, 3 is equal in both snapshots)
    { int32_t rc = mv_ensure_chain_msgs(h); if (rc != CX_OK) return rc; }
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

}  // namespace cxh
