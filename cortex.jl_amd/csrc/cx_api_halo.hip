// cx_api_halo.hip — partitioned graphs: message halos per sweep, state halos ("deep halo"), the RCCL exchange.

#include "cx_host.h"
#include "cx_halo_plan.h"

using namespace cxh;

extern "C" {

// ---- halo -------------------------------------------------------------------------------------------------------
// doubles per message in the halo buffers: the storage form (natural): 2 | d + d(d+1)/2 (packed symmetric) | 64 + 64*64
static inline int64_t halo_doubles(const cx_handle *h) { return h->cfg.dim == 1 ? 2 : h->nc; }

int32_t cx_halo_configure(cx_handle *h, int64_t n_send, const int64_t *sv, const int64_t *sf, int64_t n_recv,
                          const int64_t *rv, const int64_t *rf) {
    CX_REQUIRE(h, !h || h->n_kary == 0, CX_ERR_UNSUPPORTED, "cx_halo_configure: per-sweep message halos are implemented for unary and pairwise factors (this graph has CX_FACTOR_GAUSS_LINEAR_N factors: use state halos, cx_halo_configure_state)");
    CX_NOT_VMP(h, "cx_halo_configure");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_halo_configure: no graph");
    CX_REQUIRE(h, h->cfg.schedule != CX_SCHED_REFERENCE, CX_ERR_UNSUPPORTED, "cx_halo_configure: the reference-order schedule is sequential by definition and is not partitioned");
    // dim 2..4 and 64 under the chain-scan schedule: the lists only name the stand-ins of a time block (cx_chain_block_maps exchanges maps,
    // not messages); every other dim > 1 partition uses state halos (cx_halo_configure_state)
    const bool mv_chain_block = ((h->cfg.dim >= 2 && h->cfg.dim <= 4) || cx::is_mfma_dim(h->cfg.dim)) && h->cfg.schedule == CX_SCHED_CHAIN_SCAN;
    CX_REQUIRE(h, h->cfg.dim == 1 || mv_chain_block, CX_ERR_UNSUPPORTED, "cx_halo_configure: message halos are implemented for dim == 1 (dim > 1: cx_halo_configure_state, or a chain-scan time block)");
    // the same exclusion as cx_set_damping's, whichever call comes first (empty lists take a halo away and are always accepted)
    CX_REQUIRE(h, h->damping == 0.0 || (n_send == 0 && n_recv == 0), CX_ERR_UNSUPPORTED,
               "cx_halo_configure: per-sweep message halos are not damped (cx_set_damping is in force: set it to 0 first, or use state halos, cx_halo_configure_state)");
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
        h->chains_dirty = true; h->tree_dirty = true;
        CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream));
        cx::ipc_destroy(h);          // the receive areas are sized by the halo lists
        for (void *p : {(void *)h->d_send_slots, (void *)h->d_recv_slots, (void *)h->d_send_vars}) if (p) (void)hipFree(p);
        if (!h->ext_halo_buffers) { if (h->d_send_buf) (void)hipFree(h->d_send_buf); if (h->d_recv_buf) (void)hipFree(h->d_recv_buf); }
        h->d_send_slots = h->d_recv_slots = h->d_send_vars = nullptr; h->d_send_buf = h->d_recv_buf = nullptr;
        h->ext_halo_buffers = false;
        if (mv_chain_block) {      // no message buffers: the stand-ins are marked, nothing else
            h->send_slots.clear(); h->recv_slots.clear();
            h->spdir_dirty = true; h->chain_partition = true; h->work64_dirty = true;
            CX_HIP(h, hipStreamSynchronize(h->stream));
            return CX_OK;
        }
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
    // (round 5) graphs with factors of more than two variables are cut like any other under STATE halos: the exchanged state is the
    // factor→variable messages of the redundant variables whatever factor sent them, and a cut factor keeps all its variables on every
    // rank that holds one of them within the halo (cortex.jl_amd/partition.py: by_assignment_deep).  The per-sweep message halo
    // (cx_halo_configure) still takes unary and pairwise factors only: its import pushes a ghost's message through ONE partner edge.
    CX_NOT_VMP(h, "cx_halo_configure_state");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_halo_configure_state: no graph");
    CX_REQUIRE(h, h->cfg.schedule != CX_SCHED_CHAIN_SCAN && h->cfg.schedule != CX_SCHED_REFERENCE, CX_ERR_UNSUPPORTED,
               "cx_halo_configure_state: fused / flooding schedules (a chain-scan partition exchanges block maps: cx_chain_block_maps; the reference-order schedule is not partitioned)");
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
            CX_HIP(h, hipMemcpyAsync(h->d_vinfo, h->vinfo.data(), (size_t)h->nv, hipMemcpyHostToDevice, h->stream));
        }
        cx::ipc_destroy(h);          // the receive areas are sized by the halo lists
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
    else if (cx::is_mfma_dim(h->cfg.dim)) cx::mv64_rows_gather(h, h->d_mv_f2v, h->d_send_slots, (double *)h->d_send_buf, n);
    else cx::mv_launch_gather(h, h->d_mv_f2v, h->nslots, h->nc, h->ncs, h->d_send_slots, (double *)h->d_send_buf, n);
}
static void state_unpack(cx_handle *h) {
    const int64_t n = (int64_t)h->recv_slots.size();
    if (h->cfg.dim == 1) cx::launch_scatter(h, h->d_f2v, h->d_recv_slots, h->d_recv_buf, n);
    else if (cx::is_mfma_dim(h->cfg.dim)) cx::mv64_rows_scatter(h, h->d_mv_f2v, h->d_recv_slots, (const double *)h->d_recv_buf, n);
    else cx::mv_launch_scatter(h, h->d_mv_f2v, h->nslots, h->nc, h->ncs, h->d_recv_slots, (const double *)h->d_recv_buf, n);
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
        cx::haloplan::layers(h, lay, depth);      // cx_halo_plan.h (GPU-free: also built and tested on the CPU under sanitizers)
        batch_graph_drop(h);                      // (a captured batch holds the old layers' slice ranges)
        h->sweeps_since_exchange = 0;
        // the quiet run of cx_halo_ipc_batch (the slices of the last sweep that may run AFTER the early push) is a function of the owned-only
        // slices set just now: an IPC block that already stands gets it recomputed, so that no stale range survives new layers
        // (cx_halo_configure_state drops the block itself, and with it the range)
        if (h->d_ipc_block) {
            h->ipc_quiet_lo = 1; h->ipc_quiet_hi = 0;
            if (h->cfg.dim == 1) cx::haloplan::quiet_run(h);
        }
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
    cx::ipc_destroy(h);              // the receive block's flags and connections are per peer entry
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

// One batch of a deep-halo partition: the state exchange, then n_sweeps sweeps.  Rounds 2 - 3 ran the exchange on a second stream
// beside the owned part of the first sweep here; it was bit-identical and measured SLOWER on every depth (13.2 against 11.1 - 11.5 us
// per sweep at depth 16: two cross-stream hand-offs and a split sweep cost more than the 9 us of sweep the exchange could hide behind;
// HISTORY.md, profiles/r03_strip.md) and was removed in round 4.  The entry point stays (ABI); the forms that do put compute between
// a push and the wait for it are cx_halo_ipc_exchange_sweep and cx_halo_ipc_batch (cx_api_ipc.hip), on ONE stream.
int32_t cx_halo_exchange_sweep(cx_handle *h, int32_t n_sweeps) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state, CX_ERR_STATE, "cx_halo_exchange_sweep: call cx_halo_configure_state first");
    CX_REQUIRE(h, n_sweeps >= 1, CX_ERR_INVALID_ARGUMENT, "cx_halo_exchange_sweep: n_sweeps < 1");
    const int32_t rc = cx_halo_state_exchange(h);
    return rc != CX_OK ? rc : cx_sweep(h, n_sweeps);
}

}  // extern "C"
