// cx_kary.hip — linear-Gaussian factors with MORE THAN TWO edges (CX_FACTOR_GAUSS_LINEAR_N, scalar messages):
//     x_out = a_1 x_1 + ... + a_k x_k + b + N(0, q),      2 <= k <= 6 inputs.
// The reference wires every factor→variable message of a factor to ALL the other variable→factor messages of that factor
// (/root/reference/src/dependencies.jl:17-31: msg_to_variable(v1, f) depends on msg_to_factor(v2, f) for every v2 != v1) and
// leaves the rule to the user (compute_message_to_variable!, src/inference_engine.jl:351-361; the only Gaussian instance in the
// reference is the pairwise one of test/inference_engine_tests.jl:415-432).  This is the k-ary member of that family: with the
// factor written as  sum_e c_e x_e = b + eps,  c_out = 1, c_i = -a_i,  eps ~ N(0, q),  the message to edge j is, in moment form,
//     mean_j = (b - sum_{e != j} c_e m_e) / c_j,        variance_j = (q + sum_{e != j} c_e^2 v_e) / c_j^2
// — the same expression for the OUT edge and for every IN edge.
//
// Device form (BASELINE.json north_star: "incoming messages staged per factor and reduced with wavefront shuffles"): EIGHT LANES
// per factor, lane e = the factor's e-th edge (OUT first, then IN by ascending variable id).  A lane loads its edge's
// variable→factor message, turns it into the term (c m, c^2 v), and collects the OTHER lanes' terms by seven rotations inside its
// group of eight (no subtraction of its own term from a total: a vague message next to sharp ones would cancel) — the
// leave-one-out sums of all k + 1 outgoing messages in 14 shuffles per lane, eight factors per wave.
// Such a factor cannot be pushed through from the variable side (the fused sweep's trick for pairwise factors): its inputs are
// computed by different threads.  A graph that has one stores its variable→factor messages every sweep and runs this kernel
// after the variable phase, into the same Jacobi buffer the pairwise messages go to.
//
// Parity: the CPU checker's statement of the same phase (moment form, direct sums) per sweep; trees against the dense solve of the
// joint Gaussian (tests/test_gpu_kary.py, tests/test_kary_checker.py).

#include "cx_host.h"
#include "cx_kary_core.h"

namespace cx {

namespace {

__global__ __launch_bounds__(kBlock) void k_factor_kary(int nrows, const int32_t *__restrict__ kslot, const double *__restrict__ kcoef,
                                                        const double *__restrict__ kqb, const double2 *__restrict__ v2f, double2 *__restrict__ f2v,
                                                        const double2 *__restrict__ prev, double lam) {
    const int t = blockIdx.x * kBlock + threadIdx.x, row = t >> 3, e = t & 7;
    if (row >= nrows) return;                                   // whole groups of eight leave together
    const int slot = kslot[t];
    const double c = kcoef[t];
    double tm = 0.0, tv = 0.0;
    if (slot >= 0) {
        const double2 in = kary_moment(v2f[slot]);              // undefined input: NaN terms, every OTHER edge's message stays as it is
        tm = c * in.x;
        tv = c * c * in.y;
    }
    const int lane = threadIdx.x & 63, g0 = lane & ~7;
    double sm = 0.0, sv = 0.0;
#pragma unroll
    for (int r = 1; r < 8; r++) {
        sm += __shfl(tm, g0 | ((e + r) & 7), 64);
        sv += __shfl(tv, g0 | ((e + r) & 7), 64);
    }
    if (slot < 0) return;
    const double mean = (kqb[2 * row + 1] - sm) / c, var = (kqb[2 * row] + sv) / (c * c);
    if (__builtin_isnan(mean) || __builtin_isnan(var)) return;  // a dependency is undefined: the signal is not pending
    double2 r = kary_natural(mean, var);
    if (lam != 0.0) {                                           // cx_set_damping: against the message this one replaces
        const double2 old = prev[slot];
        if (!__builtin_isnan(old.y)) r = make_double2((1.0 - lam) * r.x + lam * old.x, (1.0 - lam) * r.y + lam * old.y);
    }
    f2v[slot] = r;
}

// cx_update_batch: MessageToVariable items of such factors, one thread per item (entry = 8 * row + edge position)
__global__ void k_kary_items(int n, const int32_t *__restrict__ entries, const int32_t *__restrict__ kslot, const double *__restrict__ kcoef,
                             const double *__restrict__ kqb, const double2 *__restrict__ v2f, double2 *__restrict__ f2v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kary_item(entries[i], kslot, kcoef, kqb, v2f, f2v);
}

// CX_SCHED_TREE over heavy paths (cx_tree_plan.h): a link of a path that runs through a factor of this table.  With the messages of the
// factor's OTHER variables fixed (its light children — final once everything below the path is — and its observed variables) the
// factor is a pairwise rule between the link's two variables: from  sum_e c_e x_e = b + eps  and the others' moments (m_j, v_j),
//     x_r = (-c_s / c_r) x_s + (b - sum_j c_j m_j) / c_r + N(0, (q + sum_j c_j^2 v_j) / c_r^2)
// for receiver r and sender s — written as the (a, b, q) of the RECEIVING slot, the form cx_chain.hip's scans read, for both directions.
// One thread per link; links through two-edge factors keep the parameters of the graph.  A vague input (variance inf) makes the
// rule's variance the largest finite number instead of inf (inf x 0 in the scans' maps); an undefined input leaves NaN parameters:
// the messages that depend on the link stay what they were, like everything that reads an undefined signal.
__global__ void k_kary_link_params(int n, const int32_t *__restrict__ from, const int32_t *__restrict__ to, const int32_t *__restrict__ slot_kary,
                                   const int32_t *__restrict__ kslot, const double *__restrict__ kcoef, const double *__restrict__ kqb,
                                   const double2 *__restrict__ v2f, double *__restrict__ q, double *__restrict__ a, double *__restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sf = from[i], st = to[i], ef = slot_kary[sf];
    if (ef < 0) return;
    const int et = slot_kary[st], row = ef >> 3;
    double sm = 0.0, sv = 0.0;
    for (int r = 0; r < 8; r++) {
        const int o = 8 * row + r, s = kslot[o];
        if (s < 0 || o == ef || o == et) continue;
        const double2 in = kary_moment(v2f[s]);
        sm += kcoef[o] * in.x;
        sv += kcoef[o] * kcoef[o] * in.y;
    }
    const double cf = kcoef[ef], ct = kcoef[et], rest = kqb[2 * row + 1] - sm, var = fmin(kqb[2 * row] + sv, 1.7976931348623157e308);
    a[st] = -cf / ct; b[st] = rest / ct; q[st] = fmin(var / (ct * ct), 1.7976931348623157e308);
    a[sf] = -ct / cf; b[sf] = rest / cf; q[sf] = fmin(var / (cf * cf), 1.7976931348623157e308);
}

}  // namespace

int32_t kary_upload(cx_handle *h) {
    if (h->cfg.dim > 1) return kary_mv_upload(h);
    if (h->n_kary == 0 || !h->kary_dirty) return CX_OK;
    using namespace cxh;
    int32_t rc;
    if (!h->d_kary_slot) {
        if ((rc = dev_alloc(h, &h->d_kary_slot, (int64_t)h->kary_slot.size())) != CX_OK) return rc;
        if ((rc = dev_alloc(h, &h->d_kary_coef, (int64_t)h->kary_coef.size())) != CX_OK) return rc;
        if ((rc = dev_alloc(h, &h->d_kary_qb, (int64_t)h->kary_qb.size())) != CX_OK) return rc;
        if ((rc = dev_alloc(h, &h->d_slot_kary, (int64_t)h->slot_kary.size())) != CX_OK) return rc;
        CX_HIP(h, hipMemcpy(h->d_kary_slot, h->kary_slot.data(), h->kary_slot.size() * 4, hipMemcpyHostToDevice));
        CX_HIP(h, hipMemcpy(h->d_slot_kary, h->slot_kary.data(), h->slot_kary.size() * 4, hipMemcpyHostToDevice));
        CX_HIP(h, hipMemcpy(h->d_kary_qb, h->kary_qb.data(), h->kary_qb.size() * 8, hipMemcpyHostToDevice));
    }
    CX_HIP(h, hipStreamSynchronize(h->stream));
    CX_HIP(h, hipMemcpy(h->d_kary_coef, h->kary_coef.data(), h->kary_coef.size() * 8, hipMemcpyHostToDevice));
    h->kary_dirty = false;
    return CX_OK;
}

void kary_free(cx_handle *h) {
    kary_mv_free(h);
    for (void *p : {(void *)h->d_kary_slot, (void *)h->d_kary_coef, (void *)h->d_kary_qb, (void *)h->d_slot_kary}) if (p) (void)hipFree(p);
    h->d_kary_slot = h->d_slot_kary = nullptr; h->d_kary_coef = h->d_kary_qb = nullptr;
    h->n_kary = 0; h->kary_slot.clear(); h->kary_coef.clear(); h->kary_qb.clear(); h->slot_kary.clear(); h->kary_pset.clear(); h->kary_dirty = true;
}

// all factor→variable messages of the k-ary factors from the stored variable→factor messages
void launch_kary(cx_handle *h, const double2 *v2f, double2 *f2v_out) {
    if (h->n_kary == 0) return;
    const int64_t threads = 8 * h->n_kary;
    hipLaunchKernelGGL(k_factor_kary, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, (int)h->n_kary, h->d_kary_slot,
                       h->d_kary_coef, h->d_kary_qb, v2f, f2v_out, (const double2 *)h->d_f2v, h->damping);      // d_f2v: the sweep's input buffer (flooding: in place)
}

void launch_kary_link_params(cx_handle *h, int64_t link_lo, int64_t nlinks, double *a, double *b) {
    if (nlinks <= 0 || h->n_kary == 0) return;
    hipLaunchKernelGGL(k_kary_link_params, dim3((unsigned)((nlinks + 255) / 256)), dim3(256), 0, h->stream, (int)nlinks, h->d_chain_from + link_lo,
                       h->d_chain_to + link_lo, h->d_slot_kary, h->d_kary_slot, h->d_kary_coef, h->d_kary_qb, (const double2 *)h->d_v2f, h->d_q, a, b);
}

void launch_kary_items(cx_handle *h, const int32_t *d_entries, int64_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_kary_items, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (int)n, d_entries, h->d_kary_slot, h->d_kary_coef,
                       h->d_kary_qb, (const double2 *)h->d_v2f, h->d_f2v);
}

}  // namespace cx

extern "C" {

// a_i of the ROLE_IN edges of CX_FACTOR_GAUSS_LINEAR_N factors (default 1: a plain sum); any time before a sweep
int32_t cx_set_factor_coefficients(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, const double *a) {
    using namespace cxh;
    CX_NOT_VMP(h, "cx_set_factor_coefficients");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_set_factor_coefficients: no graph");
    if (n == 0) return CX_OK;
    CX_REQUIRE(h, n > 0 && variable_ids && factor_ids && a, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_coefficients: null argument");
    for (int64_t i = 0; i < n; i++) {
        const int64_t e = find_edge(h, variable_ids[i], factor_ids[i]);
        if (e < 0) return fail(h, CX_ERR_NOT_FOUND, "no connection between variable " + std::to_string(variable_ids[i]) + " and factor " + std::to_string(factor_ids[i]));
        const int32_t en = h->slot_kary.empty() ? -1 : h->slot_kary[cx::slot_of_edge(h, e)];
        if (en < 0 || (en & 7) == 0)
            return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_coefficients: (variable " + std::to_string(variable_ids[i]) + ", factor " + std::to_string(factor_ids[i]) +
                        ") is not a ROLE_IN edge of a CX_FACTOR_GAUSS_LINEAR_N factor");
        if (!(a[i] != 0.0) || !std::isfinite(a[i])) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_factor_coefficients: a coefficient must be finite and non-zero");
        h->kary_coef[en] = -a[i];
    }
    h->kary_dirty = true;
    h->chain_side_dirty = true; h->offchain_marg_dirty = true;
    return CX_OK;
}

}  // extern "C"
