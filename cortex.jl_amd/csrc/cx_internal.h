// cx_internal.h — shared declarations of libcortex_hip.so (host side + kernel launchers).
// gfx950 only; see include/cortex_hip.h for the ABI and DESIGN.md for the data layout.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <tuple>
#include <string>
#include <vector>

#include "cortex_hip.h"
#include "cx_const.h"

namespace cx {

struct ProfileRec {
    int kernel;
    hipEvent_t start, stop;
};

}  // namespace cx

struct cx_handle {
    cx_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    bool has_graph = false;

    // ---- host copy of the flattened graph (lookup + batching) ----
    int64_t nv = 0, nf = 0, ne = 0;     // variables, factors, connections
    int64_t nslots = 0;                 // message slots (SELL region incl. padding + big-variable CSR region)
    int64_t nslices = 0;
    std::vector<int64_t> var_ids;       // ascending; index = local variable number
    std::vector<int64_t> fac_ids;       // ascending; index = local factor number
    std::vector<int32_t> fac_kind;      // by local factor number
    std::vector<double> fac_params;     // [nf][CX_NPARAM]
    std::vector<int32_t> var_off;       // [nv+1] CSR offsets into the edge table sorted by (variable id, factor id)
    std::vector<int64_t> edge_fac_id;   // [ne] factor id per CSR edge
    std::vector<int32_t> edge_var;      // [ne] local variable number per CSR edge
    std::vector<int32_t> vbase;         // [nv] slot of the variable's first message
    std::vector<uint8_t> vinfo;         // [nv] degree class + flags
    std::vector<int32_t> slice_off;     // [nslices+1] first slot of each slice
    std::vector<int32_t> partner;       // [nslots] slot of the other edge of a 2-edge factor, -1 otherwise
    std::vector<int32_t> big_vars;      // variables with degree > kSmallDeg
    std::vector<int32_t> big_slots;     // all slots of big variables (the fused schedule pushes them separately)
    int64_t n_messages_per_sweep = 0;
    int64_t sweeps_done = 0;
    bool any_linear = false;

    // ---- device buffers ----
    int32_t *d_slice_off = nullptr, *d_partner = nullptr, *d_vbase = nullptr, *d_var_deg = nullptr;
    int32_t *d_big = nullptr, *d_big_slots = nullptr, *d_big_slot_var = nullptr;      // (dim 2..4: the variable of every slot of the CSR tail)
    double2 *d_big_tmp = nullptr;   // prefix scratch of the big-variable kernel, one entry per big slot
    int32_t big_start = 0;          // first slot of the big-variable CSR tail
    uint8_t *d_vinfo = nullptr;
    double *d_q = nullptr, *d_a = nullptr, *d_b = nullptr;     // per RECEIVING slot: effective rule parameters
    double *d_sq = nullptr, *d_sa = nullptr, *d_sb = nullptr;  // the same, indexed by the SENDING slot (push)
    // chain scan, set by an owner that drives this handle (cx_vmp.hip's inner handle; never owned here):
    const int32_t *d_q_gamma = nullptr;   // per slot: index into d_q_gmean of the precision variable whose mean sets this factor's variance
    const double *d_q_gmean = nullptr;    //   q = 1 / d_q_gmean[d_q_gamma[slot]] read in place of d_q[slot] (no per-slot table to rewrite per call)
    double *d_split_mean = nullptr, *d_split_prec = nullptr;   // the scan writes (mean, precision) of the chain variables here instead of d_marg
    bool split_marg_written = false;      // ... and did so in the last sweep
    bool pot64_fresh = false;             // dim 64, a partition's time block: the potentials of the composition tree are those of the data on the device
                                          // (cx_chain_block_maps just composed them and nothing they read has changed): the next sweep starts at its walks
    int32_t pot64_end_slots[6] = {-1, -1, -1, -1, -1, -1};      // side slots of the path's two END positions: no composition reads them
    bool chain_msgs_unread = false;       // the owner reads the chain links' variable→factor messages and marginals only: the scan need not store
                                          // the factor→variable messages of the links (nothing of this handle is asked for them)
    double2 *d_f2v = nullptr, *d_v2f = nullptr, *d_marg = nullptr;  // natural-form messages, moment-form marginals
    double2 *d_f2v_alt = nullptr;   // second factor→variable buffer (Jacobi double buffering of the fused sweep)
    double2 *d_prev = nullptr;      // snapshot for cx_residual
    double *d_scratch = nullptr;    // small reduction scratch
    int64_t device_bytes = 0;

    // factors with more than two edges (cx_kary.hip, CX_FACTOR_GAUSS_LINEAR_N): entry = 8 * row + edge position (OUT first, then IN by
    // ascending variable id); coefficient c_e = +1 (OUT) / -a_i (IN); their slots have partner -1 (no pairwise rule touches them)
    int64_t n_kary = 0;
    std::vector<int32_t> kary_slot, slot_kary;      // [8 n_kary] slot per entry (-1 padding); [nslots] entry of a slot, -1 otherwise
    std::vector<int32_t> kary_pset;                 // dim 2..4 (cx_kary_mv.hip): [8 n_kary] parameter set of the entry — an IN entry's A, the OUT entry's Q
    int32_t *d_kary_pset = nullptr, *d_kary_v2f_slots = nullptr, *d_kary_v2f_vars = nullptr;
    double *d_kary_aq = nullptr;                    // [sets][2][d * d]: A | Q raw
    int64_t n_kary_v2f = 0, kary_aq_sets = 0;
    std::vector<double> kary_coef, kary_qb;         // [8 n_kary] c_e; [2 n_kary] q, b
    int32_t *d_kary_slot = nullptr, *d_slot_kary = nullptr;
    double *d_kary_coef = nullptr, *d_kary_qb = nullptr;
    bool kary_dirty = true;

    // multivariate path (cx_mv.hip), dim in {2,3,4}: SoA component-major buffers [nc][nslots], packed symmetric Lambda
    int nc = 2;                                    // doubles per message (eta + packed Lambda; 64 + 64*64 for dim 64)
    int ncs = 2;                                   // STORED doubles per message slot: dim 2..4 pad nc to whole 16-byte pairs (cx_mv_core.h)
    std::vector<std::vector<double>> psets;        // per parameter set: A (d*d) then Q (d*d)
    int32_t *d_spdir = nullptr;                    // per SENDING slot: 2*pset + direction of the receiving edge; -1: receiver observed
    std::vector<int32_t> spdir;                    // host copy without the observed-receiver mask
    bool spdir_dirty = true;
    int observed_passes_due = 2;                   // sweeps that still have to write the messages out of observed variables
    double *d_ptab = nullptr;                      // [2*npsets][3][d*d]: (P, B, C) triples
    double *d_zero_msg = nullptr;                  // d = 64: one message of zeros (what an absent source reads)
    // dim 16 / 32: the marginal read-out's scratch (sums of the listed variables' messages + their rule records) and identity table
    double *d_marg64_sums = nullptr, *d_marg64_tab = nullptr;
    int32_t *d_marg64_rec = nullptr;
    int64_t marg64_cap = 0;
    double *d_ptab_bt = nullptr;                   // d = 64: [2*npsets][d*d], the transposes of the B tables (cx_mv64w.hip)
    int64_t ptab_sets = 0, ptab_bt_sets = 0, max_pset = -1;     // parameter sets the device tables have room for (rewritten in place while that holds)
    double *d_mv_f2v = nullptr, *d_mv_f2v_alt = nullptr, *d_mv_v2f = nullptr, *d_mv_marg = nullptr, *d_mv_prev = nullptr;
    // d = 64 work lists (built lazily: they depend on which variables are observed)
    bool work64_dirty = true, point64_dirty = true;
    int64_t n_rule64 = 0, n_point64 = 0;
    int32_t *d_point64_slots = nullptr;
    int32_t *d_rule64_rec = nullptr;               // 8 words per work item (see k_rule64)
    // senders of degree 5 .. 8 (a rule sums at most three sources): their variable→factor messages are summed first (k_v2f64) and the
    // rule reads the stored message; the same per stage of the tree schedule
    int64_t n_pre64 = 0;
    int32_t *d_pre64_slots = nullptr, *d_pre64_vars = nullptr, *d_tree_pre_slots = nullptr, *d_tree_pre_vars = nullptr;
    std::vector<int64_t> tree_pre_off;

    // chain-scan schedule (cx_chain.hip): paths of free variables, built lazily by build_chains()
    // cfg.dim of 5 .. 63 as the caller gave it (0 otherwise): such a handle runs as dim 64 with every message, datum and rule matrix
    // embedded block-diagonally (cx_api.hip: pad_* helpers); cfg.dim holds 64
    int user_dim = 0;
    int sweep_max_w = 0;                 // the widest SELL slice (0: not yet computed): picks the register footprint of the fused sweep
    int16_t *d_partner16 = nullptr;      // partner[s] - s where every difference fits (kNoPartner16: none); null otherwise (cx_kernels.hip: PACK)
    bool chains_dirty = true;
    // CX_SCHED_TREE (cx_tree_plan.h): the stages' items and k-ary entries on the device, their offsets on the host
    bool tree_dirty = true;
    int32_t *d_tree_rec = nullptr, *d_tree_kary = nullptr;
    int64_t *d_tree_stage_off = nullptr;      // the stage table on the device (runs of thin stages go out as one launch)
    std::vector<int64_t> tree_stage_off, tree_kary_off;
    hipGraphExec_t tree_exec = nullptr;    // the stages of one sweep as ONE graph launch (hundreds of small launches otherwise: the sweep was bound by
    hipStream_t tree_capture_stream = nullptr;      //  the host's launch rate); captured on a stream of the handle's own, launched on the caller's
    bool tree_graph_failed = false;        // capture or instantiation refused once: plain launches from then on
    int64_t tree_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // depth, stages, items, k-ary entries, components, up, down, marginals
    // the same sweep over heavy paths (cx_tree_plan.h: build_hp), chosen when it takes fewer launches: the paths' arrays live in the chain
    // fields (d_chain_pos_var .. d_chain_totals; d_chain_skip1 is the way up's), item stages in d_tree_rec as before
    bool tree_hp = false;
    int32_t *d_tree_skip1_down = nullptr;
    std::vector<int32_t> tree_hp_steps;            // pairs (kind, index): 0 item stage, 1 scan of a light depth on the way up, 2 its final scan
    std::vector<int64_t> tree_hp_pos_off, tree_hp_link_off;
    int32_t tree_hp_marginal_stage = -1;
    int64_t tree_hp_stats[4] = {0, 0, 0, 0};       // light depths, paths of two or more, variables on no such path, launches per sweep
    // heavy paths through factors with more than two edges: their pairwise parameters per receiving slot are written every sweep
    // (cx_kary.hip: k_kary_link_params) into d_q and into d_a / d_b — or, on a graph without pairwise linear factors, into these
    double *d_tree_a = nullptr, *d_tree_b = nullptr;
    int64_t tree_hp_kary_links = 0;
    // dim 2 .. 4 over heavy paths: the scans of cx_mvchain.hip on a light depth's range; links per thread chosen per depth, the
    // interleaved buffers (d_mvc_*) sized for the largest depth and shared by all of them
    std::vector<int32_t> tree_hp_K;
    int64_t tree_hp_npos = 0;
    // dim 64 over heavy paths: a plan of compositions and walks (cx_chain64_plan.h) per light depth and direction of travel
    std::vector<void *> tree_c64;
    std::vector<int32_t> tree_c64_up, tree_c64_final;      // per light depth: index into tree_c64, -1 = no such scan
    int64_t chain_npos = 0, chain_nlinks = 0;
    int64_t chain_npos_linked = 0;   // dim > 1: positions [0, this) belong to paths with links; the isolated ones follow
    bool chain_side_dirty = true;    // the leaf messages / side sums of the chain positions must be recomputed (data or rule parameters changed)
    bool chain_partition = false;    // the handle holds a time block of a partitioned chain (cx_chain_block_maps was called)
    bool chain_covers_all = false;   // every variable that reads messages is a chain position: the scan's side pass produces all leaf messages
    int32_t *d_chain_pos_var = nullptr, *d_chain_skip0 = nullptr, *d_chain_skip1 = nullptr;
    int32_t *d_chain_link_pos = nullptr, *d_chain_from = nullptr, *d_chain_to = nullptr;
    uint8_t *d_chain_head_fwd = nullptr, *d_chain_head_bwd = nullptr;
    double2 *d_chain_side = nullptr;
    void *d_chain_totals = nullptr;
    // the sweeps between two exchanges of a deep-halo partition as ONE graph launch (cx_api_sweep.hip: cx_sweep): the same (first sweep after
    // the exchange, sweeps, buffers) seen a second time is captured, from then on replayed; two slots (an odd batch alternates its buffers)
    struct BatchGraph { uint64_t key = 0; hipGraphExec_t exec = nullptr; int seen = 0; bool failed = false; };
    BatchGraph batch_graph[2];
    uint64_t batch_epoch = 1;        // moved on by whatever a captured batch bakes in (layers, damping, the graph itself): cxh::batch_graph_drop
    int64_t batch_graph_launches = 0;
    // the chain scan as ONE launch (cx_chain.hip: k_chain_onepass): tile totals + flags on the device, the word in mapped host memory that a
    // workgroup raises when a wait of it times out (checked by every CX_HIP of the host: the call that finds it fails, the handle goes back to two launches)
    void *d_chain_onepass = nullptr, *d_chain_abort = nullptr;
    volatile unsigned *chain_abort_host = nullptr;
    // the chain links' rule parameters in link order (cx_chain.hip: k_chain_linkpar); chain_pos0 >= 0: link l's left end is position chain_pos0 + l
    void *d_chain_linkpar = nullptr; int64_t chain_linkpar_cap = 0; bool chain_linkpar_dirty = true; const void *chain_linkpar_qg = nullptr; int chain_pos0 = -1;
    int chain_onepass_state = 0, chain_onepass_cus = 0;      // 0 not prepared, 1 ready, -1 off (CX_CHAIN_ONEPASS=0, no memory, or a wait once timed out)
    int64_t chain_onepass_launches = 0;
    // dim 2..4 (cx_mvchain.hip): rule-table index of each link's two messages, side sums [nc][npos], tile totals of the map scan
    int32_t *d_chain_tab_fwd = nullptr, *d_chain_tab_bwd = nullptr;
    double *d_mvc_side = nullptr, *d_mvc_totals = nullptr;
    int32_t *d_mvc_var_link = nullptr;     // per variable: the chain link whose RIGHT end it is, -1 otherwise (dim 2..4; marginals on demand)
    bool mvc_marg_pending = false;      // dim 2..4 chain scan, compute_marginals_in_sweep == 2: the last sweep left alpha and gamma, the marginals are formed when read
    double *d_mvc_side_l = nullptr, *d_mvc_alpha = nullptr, *d_mvc_gamma = nullptr, *d_mvc_prefix = nullptr, *d_mvc_wave_carry = nullptr, *d_mvc_block = nullptr;   // thread-interleaved by link, [nc][ntiles * 256 * K]
    int mvc_K = 4;                   // links per thread of the scan (fixed when the chains are built)
    void *chain64 = nullptr;         // dim 64 chain scan (cx_mv64chain.hip): the plan's records, potential and entry arenas on the device
    bool chain_msgs_stale = false;   // dim 2..4 chain scan: the chain messages in d_mv_f2v are older than the last sweep (refreshed on demand)

    // halo
    std::vector<int32_t> send_slots, recv_slots;
    int32_t *d_send_slots = nullptr, *d_recv_slots = nullptr, *d_send_vars = nullptr;
    double2 *d_send_buf = nullptr, *d_recv_buf = nullptr;
    bool ext_halo_buffers = false;
    bool halo_state = false;         // halo lists name factor→variable messages of redundant variables (deep halo)
    // deep halo: redundant layer of each variable (cx_halo_set_layers) -> slice range that sweep j after an exchange has to run
    int halo_depth = 0, sweeps_since_exchange = 0;
    std::vector<int32_t> trim_lo, trim_hi;   // [depth + 1]: first / last slice holding a variable of layer <= L
    int run_slice0 = 0, run_nslices = 0;     // what launch_fused covers (0 slices: everything)
    int run_excl_lo = 1, run_excl_hi = 0;    // ... minus this slice range (empty by default)
    int own_slice_lo = 1, own_slice_hi = 0;  // deep halo: the run of slices that hold OWNED variables only (cx_halo_set_layers); empty: none
    // RCCL exchange issued by the library (cx_comm.hip)
    struct Peer { int rank; int64_t send_off, send_count, recv_off, recv_count; };
    std::vector<Peer> peers;
    void *comm = nullptr;            // ncclComm_t
    int comm_world = 0, comm_rank = -1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_recv = nullptr, ev_swept = nullptr;
    // deep-halo exchange through IPC-mapped receive areas and epoch flags (cx_api_ipc.hip)
    struct IpcConn { void *opened = nullptr; void *mapped = nullptr; char handle[64] = {0}; unsigned long long *flag = nullptr; double2 *area[2] = {nullptr, nullptr}; bool connected = false; };
    std::vector<IpcConn> ipc_conn;   // one per peer entry: where this rank pushes
    void *d_ipc_block = nullptr;     // this rank's flags + two receive areas (fine-grained, exported)
    void *d_ipc_local = nullptr;     // push completion counter, error word
    int64_t ipc_area_bytes = 0, ipc_epoch = 0, ipc_pushed = 0;   // epochs unpacked / pushed
    double ipc_timeout_s = 20.0;
    bool ipc_fused = false;          // cx_halo_ipc_set_fused: push and unpack of an exchange as ONE launch (every neighbour pushes from another device)
    int ipc_quiet_lo = 1, ipc_quiet_hi = 0;   // the longest run of owned-only slices none of whose variables WRITES a message of the send list
                                              // (cx_halo_ipc_batch: that run of the last sweep is computed after the push); empty: none
    double damping = 0.0;            // cx_set_damping: new = (1 - damping) rule + damping old (fused and flooding sweeps)
    // cx_set_messages of a long list the caller repeats (an iteration re-sets its priors before every call): ids -> slots / variables / edges,
    // kept for the last lists (the ids themselves are kept and compared: a hash alone would be trusted with the device's memory)
    struct SetMemo { uint64_t key = 0, used = 0; int32_t direction = 0; std::vector<int64_t> var_ids, fac_ids, edges; std::vector<int32_t> idx, vars; };
    std::vector<SetMemo> set_memos;
    uint64_t set_memo_tick = 0;
    uint64_t vinfo_epoch = 0;        // bumped whenever the observed flags of vinfo change (a cached "every free variable" request is then stale)
    bool in_sweep = false;
    bool v2f_stale = false;          // fused schedule without materialisation: v2f must be recomputed before use
    int mv_max_deg = 0;              // dim 2..4: widest slice of the graph (0: not computed yet)
    bool chain_v2f_from_scan = false; // the scan also stores the variable→factor messages of the chain links (set by cx_vmp.hip on its inner handle)
    bool offchain_marg_dirty = true; // chain scan: marginals of variables OFF the chains (observed, stand-ins) are due — they depend on
                                     // stored factor→variable messages only, so a full variable phase runs after those were set

    // stores of the batched API's intermediates: ProductOfMessages nodes (variable, lo, hi) and JointMarginal nodes (factor)
    std::map<std::tuple<int32_t, int32_t, int32_t>, int32_t> prod_index;
    std::map<int32_t, int32_t> joint_index;
    std::vector<uint8_t> lin_out_is_second;   // per factor (GAUSS_LINEAR): the OUT edge is the edge of the higher variable id
    std::vector<int8_t> np_role;              // per CSR edge: the role on a CX_FACTOR_NORMAL_PRECISION factor, -1 elsewhere (empty: no such factor)
    std::vector<uint8_t> var_gamma;           // per variable: 1 = the precision of such factors, Gamma-distributed (marginal stored as (shape, scale))
    std::vector<int32_t> fac_edges;           // [2 nf] CSR edges of each (≤ 2-edge) factor, built on first use
    double2 *d_prod = nullptr;
    double *d_joint = nullptr;
    int64_t prod_cap = 0, joint_cap = 0;
    double *d_mv_prod = nullptr;         // dim > 1: the ProductOfMessages table (natural form; dim 2..4 in the messages' pair form, dim 64 one row of 4,160 doubles each)
    int64_t mv_prod_cap = 0;             // entries it holds (a multiple of 256)


    // the XCD-resident cluster (cx_batch.hip: k_ref_cluster; cx_api_ref.hip: cluster_prepare / cluster_run): stage plans of wide stages in ONE launch
    void *d_cluster_ctl = nullptr;   // 512 B the launch scribbles on (cx_batch.hip: ClusterCtl)
    int cluster_cu = 0;              // compute units = workgroups of a cluster launch
    int cluster_state = 0;           // 0 not prepared, 1 ready, -1 off (CX_REF_CLUSTER=0, no memory, another architecture, or a barrier once timed out)
    int64_t cluster_recoveries = 0;  // calls whose cluster gave up at a barrier and that were finished on plain launches (cx_cluster_stats)
    std::string cluster_note;        // what happened, for cx_last_error's reader
    int64_t cluster_max_items = 16384, cluster_min_items = 128;      // a stage wider than max (one pass of the members) is a launch of its own on the whole chip; plans of fewer than min items per stage are chains

    // CX_SCHED_REFERENCE (cx_refsched.h, cx_api_ref.hip): wiring + shadow readiness state + plans (opaque); the list of the plan being launched
    void *ref = nullptr;
    int32_t *d_ref_list = nullptr;

    // variational families (cx_vmp.hip): opaque state
    void *vmp = nullptr;

    // staging for set/get/batch
    void *d_stage = nullptr;
    int64_t stage_bytes = 0;

    // profiling
    bool profiling = false, prof_armed = false;
    hipStream_t prof_stream = nullptr;
    int prof_stride = 1;
    int64_t prof_count[CX_KERNEL_COUNT] = {0};
    std::vector<cx::ProfileRec> recs;
};

namespace cx {

// slot of CSR edge e (host)
// the distance between two consecutive slots of variable v: 1 in the CSR tail (degree > 8), 256 in a slice
inline int32_t slot_stride(const cx_handle *h, int32_t v) { return ((h->vinfo[v] & kDegMask) == kBigDeg) ? 1 : kBlock; }
inline int32_t slot_of_edge(const cx_handle *h, int64_t e) {
    const int32_t v = h->edge_var[e];
    const int32_t k = (int32_t)(e - h->var_off[v]);
    return ((h->vinfo[v] & kDegMask) == kBigDeg) ? h->vbase[v] + k : h->vbase[v] + k * kBlock;
}

// kernel launchers (cx_kernels.hip)
void launch_fused(cx_handle *h, const double2 *f2v_in, double2 *f2v_out, bool write_marg, bool store_v2f, bool skip_ghosts);
void launch_var_to_factor(cx_handle *h, const double2 *f2v, bool write_marg);
void launch_big_var_to_factor(cx_handle *h, const double2 *f2v, bool write_marg);
void launch_factor_to_var(cx_handle *h, const double2 *v2f, double2 *f2v);
void launch_push_slots(cx_handle *h, const int32_t *d_slots, int64_t n, double2 *f2v_out, int kernel_id);
void launch_halo_export(cx_handle *h, const double2 *f2v, hipStream_t stream);
void launch_halo_import(cx_handle *h, double2 *f2v_out, bool push);
void launch_batch(cx_handle *h, const int32_t *d_rec, int64_t n);
void launch_flat_run(cx_handle *h, const int32_t *d_flat, const int32_t *d_rec, const int64_t *d_stage_off, int s0, int s1);
int flat_run_max();      // stages of at most this many records   // launch_batch_run on flat records
void launch_ref_cluster(cx_handle *h, void *d_ctl, int n_workgroups, const int32_t *d_flat, const int32_t *d_rec, const int64_t *d_stage_off, int n_stages);   // all stages behind single-XCD barriers; ctl[4] != 0 afterwards: a wait timed out
void launch_wide_sum(cx_handle *h, const int32_t *d_rec, int64_t n, void *d_partial);   // reference plans: list items of more than cx::refsched::kWideList sources, a workgroup each
void launch_batch_run(cx_handle *h, const int32_t *d_rec, const int64_t *d_stage_off, int s0, int s1);   // consecutive thin stages (<= 1024 items each) in one launch
constexpr int kSmallBatch = 48;                        // items whose records (5 int32 each) ride in the kernel arguments
struct SmallBatch { int32_t r[5 * kSmallBatch]; };
void launch_batch_small(cx_handle *h, const SmallBatch &recs, int n);   // 5 int32 per item: kind, index, var, lo, hi
void launch_scatter(cx_handle *h, double2 *dst, const int32_t *d_idx, const double2 *d_val, int64_t n);
void launch_gather(cx_handle *h, const double2 *src, const int32_t *d_idx, double2 *d_val, int64_t n);
void launch_seed(cx_handle *h, double2 *buf, int64_t n, double2 value, const int32_t *partner);
// factors with more than two edges (cx_kary.hip)
int32_t kary_upload(cx_handle *h);
void kary_free(cx_handle *h);
void launch_kary(cx_handle *h, const double2 *v2f, double2 *f2v_out);
void launch_kary_items(cx_handle *h, const int32_t *d_entries, int64_t n);
void launch_kary_link_params(cx_handle *h, int64_t link_lo, int64_t nlinks, double *a, double *b);
// the same factors for dim 2..4 (cx_kary_mv.hip)
int32_t kary_mv_upload(cx_handle *h);
void kary_mv_free(cx_handle *h);
void mv_launch_kary(cx_handle *h, double *f2v_out = nullptr);
void launch_residual(cx_handle *h, const double2 *cur, const double2 *prev, int64_t n, double *d_out);
void launch_chain_scan(cx_handle *h, double2 *f2v, bool fused_leaves, int marg_form, bool chain_v2f);
void launch_chain_totals(cx_handle *h, double2 *f2v, bool fused_leaves, int64_t *ntiles_out);
void launch_chain_scan_range(cx_handle *h, double2 *f2v, int64_t pos_lo, int64_t npos, int64_t link_lo, int64_t nlinks, const int32_t *skip1, bool final);
// multivariate (cx_mv.hip)
void mv_launch_sweep(cx_handle *h, bool write_marg, int only, double *f2v_out = nullptr);   // only: 0 regular, 1 observed variables, 2 other fixed senders (degree 1, stand-ins)
void mv_launch_v2f(cx_handle *h, const int32_t *d_slots, const int32_t *d_vars, int64_t n, const double *f2v);
void mv_launch_big(cx_handle *h, bool write_marg, double *f2v_out = nullptr);     // the variables of degree > 8 of a fused sweep (dim 2..4)
void mv_launch_scatter(cx_handle *h, double *dst, int64_t stride, int nc, int ncs, const int32_t *d_idx, const double *d_val, int64_t n);
void mv_launch_gather(cx_handle *h, const double *src, int64_t stride, int nc, int ncs, const int32_t *d_idx, double *d_val, int64_t n);
void mv_launch_seed(cx_handle *h, double *buf, double eta, double lam);
void mv_launch_batch_small(cx_handle *h, const SmallBatch &recs, int n);
void mv_launch_batch_run(cx_handle *h, const int32_t *d_rec, const int64_t *d_stage_off, int s0, int s1);   // stages [s0, s1), each at most mv_run_block() items, one workgroup
int mv_run_block();
void mv_launch_batch(cx_handle *h, const int32_t *d_rec, int64_t n);   // cx_mvbatch.hip: 5 int32 per item (kind, index, variable, rule table, 0)
void mv_launch_residual(cx_handle *h, const double *cur, const double *prev, int64_t n, double *d_out);
bool spd_inverse(int d, const double *S, double *out);
// d = 64 (cx_mv64.hip): message-major layout, MFMA rule kernel
void mv64_launch_rule(cx_handle *h, int nwork, const int32_t *d_rec, const double *f2v_in, double *f2v_out, int kernel_id);
void mv64w_launch_rule(cx_handle *h, int nwork, const int32_t *d_rec, const double *f2v_in, double *f2v_out);   // cx_mv64w.hip
void mv64_launch_marginals(cx_handle *h, int n, const int32_t *d_vars, const double *f2v, double *out);
void mv64w_launch_marginal_rule(cx_handle *h, int n, const int32_t *d_rec, const double *ident_tab, const double *ident_bt, const double *sums, double *out);
void mv64_launch_point(cx_handle *h, int nwork, const int32_t *d_slots, double *out_a, double *out_b);
void mv64_launch_v2f(cx_handle *h, int n, const int32_t *d_slots, const int32_t *d_vars, const double *f2v);
void mv64_launch_damp(cx_handle *h, int n, const int32_t *d_rec, const double *old, double *out, double lam);
void mv64_launch_range_sums(cx_handle *h, int n, const int32_t *d_rec4, const double *f2v, double *out);      // rec: destination row, n sources, first slot, slot stride
void mv64_launch_seed(cx_handle *h, double *buf, double eta, double lam);
void mv64_rows_scatter(cx_handle *h, double *dst, const int32_t *d_idx, const double *d_val, int64_t n);
void mv64_rows_gather(cx_handle *h, const double *src, const int32_t *d_idx, double *d_val, int64_t n);
void mv64_set_point(cx_handle *h, double *dst, const int32_t *d_idx, const double *d_y, int64_t n);
bool mv_rule_tables(int d, const double *A, const double *Q, double *out);
size_t chain_total_bytes(int64_t nlinks);
void chain_onepass_free(cx_handle *h);
// chains inside reference-order plans as scans (cx_planscan.hip)
int64_t plan_scan_scratch_bytes(int64_t nlinks);
int64_t plan_scan_max_links();
void launch_plan_scan(cx_handle *h, const int32_t *lead_dst, const int32_t *lead_var, const int32_t *fol_dst, const int32_t *prec, const int32_t *src_off,
                      const int32_t *src, const uint8_t *head, int64_t lo, int64_t hi, void *scratch);
// chain scan for dim 2..4 (cx_mvchain.hip)
int mvc_links_per_thread(int64_t nlinks);
size_t mvc_prefix_doubles(int dim, int64_t nlinks, int K);
size_t mvc_wave_carry_doubles(int dim, int64_t nlinks, int K);
int64_t mvc_ntiles(int64_t nlinks, int K);
size_t mvc_totals_doubles(int dim, int64_t nlinks, int K);
void mvc_launch_side(cx_handle *h, bool write_marg);
void mvc_launch_scan(cx_handle *h, bool write_marg, bool store_msgs, bool scan, bool defer_marg = false);
void mvc_launch_scan_range(cx_handle *h, int64_t npos_total, int64_t pos_hi, int64_t link_lo, int64_t nlinks, int K, const int32_t *skip1, bool final);
void mvc_launch_marg_out(cx_handle *h);
void mvc_launch_marg_gather(cx_handle *h, const int32_t *d_vars, int64_t n, double *d_val);   // a few marginals from alpha + gamma (rows of nc doubles, moment form, packed)      // alpha + gamma of the last sweep -> the marginals (what a sweep with defer_marg left undone)
void mvc_launch_block_maps(cx_handle *h);
// chain scan for dim 64 (cx_mv64chain.hip; the plan: cx_chain64_plan.h)
int32_t chain64_build(cx_handle *h, const std::vector<int32_t> &pos_var, const std::vector<int32_t> &skip0, const std::vector<int32_t> &skip1,
                      const std::vector<int32_t> &link_pos, const std::vector<int32_t> &from, const std::vector<int32_t> &to,
                      const std::vector<uint8_t> &head_fwd, const std::vector<uint8_t> &head_bwd, const std::vector<int32_t> &tab_fwd,
                      const std::vector<int32_t> &tab_bwd);
int32_t chain64_sweep(cx_handle *h);
void chain64_free(cx_handle *h);
int32_t chain64_tree_build(cx_handle *h, int *index, const std::vector<int32_t> &pos_var, const std::vector<int32_t> &skip0, const std::vector<int32_t> &skip1,
                           const std::vector<int32_t> &link_pos, const std::vector<int32_t> &from, const std::vector<int32_t> &to,
                           const std::vector<uint8_t> &head_fwd, const std::vector<uint8_t> &head_bwd, const std::vector<int32_t> &tab_fwd,
                           const std::vector<int32_t> &tab_bwd);
int32_t chain64_tree_resolve(cx_handle *h);
int64_t chain64_tree_launches(const cx_handle *h, int index);
int32_t chain64_tree_sweep(cx_handle *h, int index);
void chain64_tree_free(cx_handle *h);
void chain64_stats(const cx_handle *h, int64_t *out8);
int32_t chain64_block_potential(cx_handle *h, double *pot, int32_t *side_first3, int32_t *side_last3, bool *no_root);
// variational families (cx_vmp.hip)
int32_t vmp_graph_create(cx_handle *h, int64_t ne, const int64_t *edge_var, const int64_t *edge_fac, const int32_t *edge_role,
                         int64_t nf, const int64_t *factor_ids, const int32_t *factor_kind);
int32_t vmp_set_marginals(cx_handle *h, int64_t n, const int64_t *ids, int32_t form, const double *payload);
int32_t vmp_get_marginals(cx_handle *h, int64_t n, const int64_t *ids, double *out);
int32_t vmp_update_marginals(cx_handle *h, int64_t n, const int64_t *ids);
int32_t vmp_set_stream(cx_handle *h);
int32_t vmp_state_bytes(cx_handle *h, int64_t *bytes);
int32_t vmp_state_export(cx_handle *h, void *buf, int64_t bytes);
int32_t vmp_state_import(cx_handle *h, const void *buf, int64_t bytes);
void vmp_free(cx_handle *h);
// RCCL halo exchange (cx_comm.hip)
bool comm_unique_id(void *out128, std::string &err);
bool comm_init(cx_handle *h, int world, int rank, const void *id128, std::string &err);
void comm_destroy(cx_handle *h);
bool comm_exchange(cx_handle *h, std::string &err, bool packed_on_comm_stream);
bool comm_exchange_on(cx_handle *h, hipStream_t stream, std::string &err);
void ipc_destroy(cx_handle *h);  // cx_api_ipc.hip

}  // namespace cx
