// cx_internal.h — shared declarations of libcortex_hip.so (host side + kernel launchers).
// gfx950 only; see include/cortex_hip.h for the ABI and DESIGN.md for the data layout.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "cortex_hip.h"

namespace cx {

constexpr int kBlock = 256;        // threads per workgroup = 4 wave64
constexpr int kCapEdges = 1280;    // edges staged per workgroup: 1280 * 16 B = 20 KiB LDS -> 8 workgroups / CU
constexpr int kSmallDeg = 8;       // variables up to this degree take the in-register leave-one-out path

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
    int64_t nv = 0, nf = 0, ne = 0;
    std::vector<int64_t> var_ids;       // ascending; index = local variable number
    std::vector<int64_t> fac_ids;       // ascending; index = local factor number
    std::vector<int32_t> fac_kind;      // by local factor number
    std::vector<double> fac_params;     // [nf][CX_NPARAM]
    std::vector<int32_t> var_off;       // [nv+1] into the edge table
    std::vector<int64_t> edge_fac_id;   // [ne] factor id per edge (edges sorted by variable id, factor id)
    std::vector<int32_t> edge_var;      // [ne] local variable number per edge
    std::vector<uint8_t> var_flags;     // [nv] 1 = observed (clamped) variable
    std::vector<int32_t> partner;       // [ne] the other edge of a 2-edge factor, -1 otherwise
    std::vector<int32_t> blk;           // [nblk+1] variable ranges of the small-degree workgroups
    std::vector<int32_t> big_vars;      // variables with degree > kSmallDeg
    std::vector<int32_t> big_tmp_off;   // per big variable: offset of its prefix scratch
    std::vector<int32_t> big_edges;     // all edges of big variables (fused schedule pushes them separately)
    int64_t n_messages_per_sweep = 0;
    int64_t sweeps_done = 0;
    bool any_linear = false;

    // ---- device buffers ----
    int32_t *d_var_off = nullptr, *d_partner = nullptr, *d_edge_var = nullptr, *d_blk = nullptr;
    int32_t *d_big = nullptr, *d_big_tmp_off = nullptr, *d_big_edges = nullptr;
    double2 *d_big_tmp = nullptr;
    uint8_t *d_var_flags = nullptr;
    double *d_q = nullptr, *d_a = nullptr, *d_b = nullptr;  // per receiving edge: effective rule parameters
    double *d_sq = nullptr, *d_sa = nullptr, *d_sb = nullptr;  // the same, indexed by the SENDING edge (fused push)
    double2 *d_f2v = nullptr, *d_v2f = nullptr, *d_marg = nullptr;  // natural-form messages, moment-form marginals
    double2 *d_f2v_alt = nullptr;   // second buffer of the fused schedule
    double2 *d_prev = nullptr;      // snapshot for cx_residual
    double *d_scratch = nullptr;    // small reduction scratch
    int64_t device_bytes = 0;

    // halo
    std::vector<int32_t> send_edges, recv_edges;
    int32_t *d_send_edges = nullptr, *d_recv_edges = nullptr;
    double2 *d_send_buf = nullptr, *d_recv_buf = nullptr;

    // staging for set/get/batch
    void *d_stage = nullptr;
    int64_t stage_bytes = 0;

    // profiling
    bool profiling = false;
    std::vector<cx::ProfileRec> recs;
};

namespace cx {

// kernel launchers (cx_kernels.hip)
void launch_var_to_factor(cx_handle *h, const double2 *f2v, double2 *v2f, bool write_marg);
void launch_big_var_to_factor(cx_handle *h, const double2 *f2v, double2 *v2f, bool write_marg);
void launch_factor_to_var(cx_handle *h, const double2 *v2f, double2 *f2v);
void launch_fused(cx_handle *h, const double2 *f2v_in, double2 *f2v_out, double2 *v2f, bool write_marg, bool store_v2f);
void launch_push_edges(cx_handle *h, const int32_t *d_edges, int64_t n, const double2 *v2f, double2 *f2v_out);
void launch_batch(cx_handle *h, const int32_t *d_kind, const int32_t *d_index, int64_t n);
void launch_scatter(cx_handle *h, double2 *dst, const int32_t *d_idx, const double2 *d_val, int64_t n);
void launch_gather(cx_handle *h, const double2 *src, const int32_t *d_idx, double2 *d_val, int64_t n);
void launch_seed(cx_handle *h, double2 *buf, int64_t n, double2 value, const int32_t *skip_if_partner_negative);
void launch_residual(cx_handle *h, const double2 *cur, const double2 *prev, int64_t n, double *d_out);

}  // namespace cx
