// cx_mvbatch.hip — cx_update_batch for d-dimensional messages (d = 2, 3, 4): the per-signal / per-wavefront granularity of the
// plug-in boundary.  One work item is one process!(processor, engine, variable_id, signal) of the reference
// (src/inference_engine.jl:479-509) on a signal whose variant is
//   MessageToFactor      compute_message_to_factor!     (:381-389): the product of the variable's OTHER incoming factor→variable
//                        messages, left fold in neighbour order (ascending factor id) — a sum in natural form
//   MessageToVariable    compute_message_to_variable!   (:351-361): the linear-Gaussian factor rule on the message the factor's other
//                        variable sent (cx_mv.hip: Cholesky + two triangular solves; a point-mass datum gives N(A y, Q) / its backward form)
//   IndividualMarginal   compute_individual_marginal!   (:409-419): the product of ALL incoming messages, stored in moment form
//   ProductOfMessages    compute_product_of_messages!   (:439-449): the product of a RANGE of the variable's incoming messages — the
//                        intermediate signals the reference's default resolver creates for a variable of degree > 5
//                        (src/dependencies.jl:90-173); stored (natural form) in a table of its own, read back by cx_get_products.  The
//                        other items never read it: they fold the messages themselves, as the scalar items do
// ... and, internal to the reference-order plans (cx_refsched.h: kItemSumTo*), the signals of a variable of degree > 5 under the default
// resolver, whose dependencies are segment-tree nodes: kinds 64 (MessageToFactor), 65 (ProductOfMessages), 66 (IndividualMarginal) are the sum
// of `hi` sources list[tab ..) in the reference's order — an entry >= 0 a factor→variable slot, ~entry a node of the product table.
// A batch holds mutually independent signals (the host's scheduler guarantees it: one wavefront of pending signals), so the items
// of a launch never read what another item of the same launch writes.  A result with an undefined dependency (NaN) is not stored:
// the signal was not pending.
// d = 64 goes through the kernels of cx_mv64.hip / cx_mv64w.hip instead (cx_api_mv.hip: mv_update_batch).

#include "cx_internal.h"
#include "cx_mv_core.h"
#include "cx_kary_mv_core.h"

namespace cx {

// rec: 5 int32 per item — kind, index (slot of the signal's edge | local variable | place in the product table), local variable, rule table
// of the sending slot (ProductOfMessages: first message of the range, 1-based), 0 (ProductOfMessages: last message of the range)
template <int D>
__device__ __forceinline__ void batch_item_mv(int kind, int idx, int v, int tab, int hi, const int32_t *__restrict__ vbase, const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ vdeg,
                                              const int32_t *__restrict__ partner, const double *__restrict__ ptab, double *__restrict__ f2v,
                                              double *__restrict__ v2f, double *__restrict__ marg, double *__restrict__ prod, const KaryMvTab kt, const int32_t *__restrict__ list) {
    if (kind == 32) { kary_item_mv<D>(idx, kt, v2f, f2v, nullptr, 0.0); return; }      // a message out of a factor of more than two variables (index = entry of its table)
    if (kind >= 64 && kind <= 66) {      // (a node may lag behind its leaves on a graph with loops, and the reference reads the node)
        Msg<D> acc = msg_zero<D>();
        for (int j = 0; j < hi; j++) { const int s = list[tab + j]; msg_add<D>(acc, s >= 0 ? slot_load<D>(f2v, s) : slot_load<D>(prod, ~s)); }
        const bool ok = hi > 0 && !__builtin_isnan(acc.lam[0]);
        if (kind == 66) slot_store<D>(marg, v, ok ? mv_to_moment<D>(acc) : msg_all_nan<D>());
        else if (ok) slot_store<D>(kind == 64 ? v2f : prod, idx, acc);
        return;
    }
    // (a variable of degree > 8 lives in the CSR tail: consecutive slots; the others in their slice, a slot every 256)
    const int info = vinfo[v], deg = vdeg[v], b = vbase[v], st = (info & kDegMask) == kBigDeg ? 1 : kBlock;
    if (kind == CX_ITEM_MESSAGE_TO_FACTOR) {
        // variables of degree 1, observed variables and stand-ins have no dependencies: their message is what the caller stored
        if (deg < 2 || (info & (kClamped | kGhost))) return;
        Msg<D> o = msg_zero<D>();
        for (int j = 0; j < deg; j++)
            if (b + j * st != idx) msg_add<D>(o, slot_load<D>(f2v, b + j * st));
        if (!__builtin_isnan(o.lam[0])) slot_store<D>(v2f, idx, o);
    } else if (kind == CX_ITEM_MESSAGE_TO_VARIABLE) {
        const int p = hi > 0 ? hi - 1 : partner[idx];       // (a reference-order plan's record names the sending slot: cx_refsched.h)
        if (p < 0) return;                                  // an opaque factor's message is the caller's to set
        const Msg<D> in = slot_load<D>(v2f, p);
        if (__builtin_isnan(in.lam[0])) return;
        const Msg<D> r = mv_rule<D>(in, ptab + (int64_t)tab * 3 * D * D);
        if (!__builtin_isnan(r.lam[0])) slot_store<D>(f2v, idx, r);
    } else if (kind == CX_ITEM_INDIVIDUAL_MARGINAL) {
        Msg<D> total = msg_zero<D>();
        for (int j = 0; j < deg; j++) msg_add<D>(total, slot_load<D>(f2v, b + j * st));
        const bool ok = deg > 0 && !__builtin_isnan(total.lam[0]);
        slot_store<D>(marg, v, ok ? mv_to_moment<D>(total) : msg_all_nan<D>());
    } else if (kind == CX_ITEM_PRODUCT_OF_MESSAGES) {
        Msg<D> acc = msg_zero<D>();
        for (int j = tab - 1; j < hi; j++) msg_add<D>(acc, slot_load<D>(f2v, b + j * st));
        if (!__builtin_isnan(acc.lam[0])) slot_store<D>(prod, idx, acc);      // a dependency is undefined: not pending, the stored value stays
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_batch_mv(int64_t n, const int32_t *__restrict__ rec, int64_t nslots, int nv, const int32_t *__restrict__ vbase,
                                                     const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ vdeg, const int32_t *__restrict__ partner,
                                                     const double *__restrict__ ptab, double *__restrict__ f2v, double *__restrict__ v2f,
                                                     double *__restrict__ marg, double *__restrict__ prod, const KaryMvTab kt, const int32_t *__restrict__ list) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    batch_item_mv<D>(rec[5 * i], rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], vbase, vinfo, vdeg, partner, ptab, f2v, v2f, marg, prod, kt, list);
}

// at most kSmallBatch items: the records are the first kernel argument (cx_batch.hip: k_batch_small)
template <int D>
__global__ __launch_bounds__(64) void k_batch_mv_small(SmallBatch recs, int n, const int32_t *__restrict__ vbase, const uint8_t *__restrict__ vinfo,
                                                       const int32_t *__restrict__ vdeg, const int32_t *__restrict__ partner, const double *__restrict__ ptab, double *__restrict__ f2v,
                                                       double *__restrict__ v2f, double *__restrict__ marg, double *__restrict__ prod, const KaryMvTab kt, const int32_t *__restrict__ list) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const __attribute__((address_space(4))) int32_t *rec = (const __attribute__((address_space(4))) int32_t *)__builtin_amdgcn_kernarg_segment_ptr();
    batch_item_mv<D>(rec[5 * i], rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], vbase, vinfo, vdeg, partner, ptab, f2v, v2f, marg, prod, kt, list);
    (void)recs;
}

// A run of consecutive thin stages of a plan (each at most kMvRunBlock items) in ONE launch of one workgroup, a barrier between the stages
// instead of a kernel boundary (cx_batch.hip: k_batch_run, for d-dimensional messages): the stages of a reference-order plan on a chain
// or a small loopy graph are a few items each, and a launch apiece is 6 us of latency per stage.
constexpr int kMvRunBlock = 256;
template <int D>
__global__ __launch_bounds__(kMvRunBlock) void k_batch_mv_run(const int64_t *__restrict__ stage_off, int s0, int s1, const int32_t *__restrict__ rec, const int32_t *__restrict__ vbase,
                                                              const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ vdeg, const int32_t *__restrict__ partner,
                                                              const double *__restrict__ ptab, double *__restrict__ f2v, double *__restrict__ v2f, double *__restrict__ marg,
                                                              double *__restrict__ prod, const KaryMvTab kt, const int32_t *__restrict__ list) {
    for (int st = s0; st < s1; st++) {
        for (int64_t i = stage_off[st] + threadIdx.x; i < stage_off[st + 1]; i += kMvRunBlock)
            batch_item_mv<D>(rec[5 * i], rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], vbase, vinfo, vdeg, partner, ptab, f2v, v2f, marg, prod, kt, list);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

int mv_run_block() { return kMvRunBlock; }

void mv_launch_batch_run(cx_handle *h, const int32_t *d_rec, const int64_t *d_stage_off, int s0, int s1) {
    if (s1 <= s0) return;
#define CX_MVB(DD) hipLaunchKernelGGL((k_batch_mv_run<DD>), dim3(1), dim3(kMvRunBlock), 0, h->stream, d_stage_off, s0, s1, d_rec, h->d_vbase, h->d_vinfo, h->d_var_deg, h->d_partner, \
                                      h->d_ptab, h->d_mv_f2v, h->d_mv_v2f, h->d_mv_marg, h->d_mv_prod, kt, h->d_ref_list)
    const KaryMvTab kt{h->d_kary_slot, h->d_kary_pset, h->d_kary_aq};
    if (h->cfg.dim == 2) CX_MVB(2);
    else if (h->cfg.dim == 3) CX_MVB(3);
    else CX_MVB(4);
#undef CX_MVB
}

void mv_launch_batch(cx_handle *h, const int32_t *d_rec, int64_t n) {
    if (n == 0) return;
    const dim3 g((unsigned)((n + kBlock - 1) / kBlock)), b(kBlock);
#define CX_MVB(DD) hipLaunchKernelGGL((k_batch_mv<DD>), g, b, 0, h->stream, n, d_rec, h->nslots, (int)h->nv, h->d_vbase, h->d_vinfo, h->d_var_deg, h->d_partner, \
                                      h->d_ptab, h->d_mv_f2v, h->d_mv_v2f, h->d_mv_marg, h->d_mv_prod, kt, h->d_ref_list)
    const KaryMvTab kt{h->d_kary_slot, h->d_kary_pset, h->d_kary_aq};
    if (h->cfg.dim == 2) CX_MVB(2);
    else if (h->cfg.dim == 3) CX_MVB(3);
    else CX_MVB(4);
#undef CX_MVB
}

void mv_launch_batch_small(cx_handle *h, const SmallBatch &recs, int n) {
    if (n == 0) return;
#define CX_MVB(DD) hipLaunchKernelGGL((k_batch_mv_small<DD>), dim3(1), dim3(64), 0, h->stream, recs, n, h->d_vbase, h->d_vinfo, h->d_var_deg, h->d_partner, h->d_ptab, \
                                      h->d_mv_f2v, h->d_mv_v2f, h->d_mv_marg, h->d_mv_prod, kt, h->d_ref_list)
    const KaryMvTab kt{h->d_kary_slot, h->d_kary_pset, h->d_kary_aq};
    if (h->cfg.dim == 2) CX_MVB(2);
    else if (h->cfg.dim == 3) CX_MVB(3);
    else CX_MVB(4);
#undef CX_MVB
}

}  // namespace cx
