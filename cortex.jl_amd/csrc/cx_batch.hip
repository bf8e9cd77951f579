// cx_batch.hip — scalar (dim 1) work as ITEMS: the batched boundary of the plug-in (cx_update_batch: one thread per enqueued signal), the
// stage plans of the tree and reference-order schedules (a launch per stage, runs of thin stages in one workgroup), the XCD-resident
// cluster that takes a whole reference-order plan in one launch, and the workgroup sums of long dependency lists.  The sweeps are
// cx_kernels.hip; what both share is cx_scalar_core.h.
#include <algorithm>
#include <cstdlib>

#include "cx_scalar_core.h"

namespace cx {

// ------------------------------------------------------------------------------------------------
// Batched mode: one thread per enqueued signal (the processor's `process!` override flushes a batch of
// mutually independent pending signals; inference_engine.jl:528-537 is the precedent for collecting).
// ------------------------------------------------------------------------------------------------
// Item record (5 int32): kind, index, var, lo, hi.
//   MESSAGE_TO_FACTOR / MESSAGE_TO_VARIABLE: index = slot, var = local variable
//   INDIVIDUAL_MARGINAL:                     index = var = local variable
//   PRODUCT_OF_MESSAGES (compute_product_of_messages!, inference_engine.jl:439-449; the segment-tree intermediates of
//                        dependencies.jl:128-173): index = node in the product store, var = local variable, [lo, hi] = 1-based
//                        inclusive range over the variable's neighbours (ascending factor id); value = product of those
//                        factor→variable messages (natural form: their sum, left to right like the reference's fold)
//   JOINT_MARGINAL (compute_joint_marginal!, :469-477): index = node in the joint store, var = slot of the factor's edge whose
//                        rule parameters describe x_out = a x_in + b + N(0, q) (the OUT edge; either edge of an additive
//                        factor), lo = slot of the other edge, hi = 1 when the OUT edge's variable has the LOWER id (output
//                        order is ascending variable id).  Value: the 2-d Gaussian proportional to factor x the two
//                        variable→factor messages, as mean[2] + covariance[4].
// a message out of a factor with more than two edges as a batch item (cx_kary_core.h): not a kind of the public interface —
// cx_update_batch refuses kinds it does not know and routes such messages itself
constexpr int kItemKaryEntry = 32;
struct KaryTab { const int32_t *slot; const double *coef, *qb; const int32_t *list; };
// items of the reference-order plans (cx_refsched.h) for the signals of variables of degree > 5, whose dependencies are segment-tree
// nodes (dependencies.jl:90-173): the value is the sum — the reference's `reduce(product, get_value.(deps))` in natural form — of the
// `hi` sources list[lo ..): an entry >= 0 is a factor→variable slot, ~entry a node of the product store.  What the reference's rule
// call read, node by node — a node may lag behind its leaves on a graph with loops, and the reference reads the node.
constexpr int kItemSumToFactor = 64, kItemSumToProduct = 65, kItemSumToMarginal = 66;
// a record that LEADS is followed in its stage's list by a record that FOLLOWS: one thread computes the first, waits for its store, and
// computes the second — a MessageToFactor and the MessageToVariable that reads it, levelled as one item (cx_refsched.h: kRecLeads)
constexpr int kRecLeads = 0x40000000, kRecFollows = 0x20000000, kRecKindMask = 0x0fffffff;
// ... and the variational rules of a CX_FACTOR_NORMAL_PRECISION factor (out ~ N(in, 1 / precision), precision ~ Gamma) that a user wiring
// selects (cx_refsched.h: kRule*; the reference's test rules, test/inference_engine_tests.jl:647-689, 939-1030).  Marginals are read from
// the marginal store: (mean, variance) of a Normal variable ((datum, 0) when observed), (shape, scale) of a precision — 72 stores the
// latter from the natural-parameter sum (shape - 1, rate).  A message to a precision is Gamma(3/2, 2 / spread) = natural (1/2, spread / 2).
constexpr int kItemMfNormal = 67, kItemMfGamma = 68, kItemStNormal = 69, kItemVmpJoint = 70, kItemStGamma = 71, kItemSumToGammaMarginal = 72;
template <bool COH>
__device__ __forceinline__ void vmp_item(int k, int idx, int s0, int s1, int s2, double2 *__restrict__ f2v, const double2 *__restrict__ v2f,
                                         const double2 *__restrict__ marg, double *__restrict__ joint) {      // (s0 .. s2: the record's list entries)
    const double inf = __builtin_inf();
    if (k == kItemMfNormal) {                 // N(E[other], E[precision])                                                        (:654-664)
        const double2 a = ld2<COH>(marg, s0), g = ld2<COH>(marg, s1);
        const double eg = g.x * g.y;
        if (!__builtin_isnan(a.x) && !__builtin_isnan(eg)) f2v[idx] = make_double2(a.x * eg, eg);
    } else if (k == kItemMfGamma) {           // Gamma(3/2, 2 / (var a + var b + (E a - E b)^2))                                  (:666-684)
        const double2 a = ld2<COH>(marg, s0), b = ld2<COH>(marg, s1);
        const double d = a.x - b.x, spread = a.y + b.y + d * d;
        if (!__builtin_isnan(spread)) f2v[idx] = make_double2(0.5, 0.5 * spread);
    } else if (k == kItemStNormal) {          // N(mean m, 1 / (var m + 1 / E[precision])), m the other Normal variable's message    (:1004-1010)
        const double2 m = ld2<COH>(v2f, s0), g = ld2<COH>(marg, s1);
        const double eg = g.x * g.y;
        if (__builtin_isnan(m.y) || __builtin_isnan(m.x) || __builtin_isnan(eg)) return;
        const double mean = m.y == inf ? m.x : m.x / m.y, var = m.y == inf ? 0.0 : 1.0 / m.y;
        const double w = 1.0 / (var + 1.0 / eg);
        f2v[idx] = make_double2(mean * w, w);
    } else if (k == kItemVmpJoint) {          // the 2-d Gaussian with precision [[w1 + E, -E], [-E, w2 + E]] and potential (xi1, xi2)  (:939-967)
        const double2 m1 = ld2<COH>(v2f, s0), m2 = ld2<COH>(v2f, s1), g = ld2<COH>(marg, s2);
        const double eg = g.x * g.y;
        if (__builtin_isnan(m1.y) || __builtin_isnan(m1.x) || __builtin_isnan(m2.y) || __builtin_isnan(m2.x) || __builtin_isnan(eg)) return;
        double mu1, mu2, v11, v12, v22;
        if (m1.y == inf && m2.y == inf) { mu1 = m1.x; mu2 = m2.x; v11 = v12 = v22 = 0.0; }
        else if (m1.y == inf) { mu1 = m1.x; v11 = v12 = 0.0; v22 = 1.0 / (m2.y + eg); mu2 = v22 * (m2.x + eg * mu1); }
        else if (m2.y == inf) { mu2 = m2.x; v22 = v12 = 0.0; v11 = 1.0 / (m1.y + eg); mu1 = v11 * (m1.x + eg * mu2); }
        else {
            const double a = m1.y + eg, c = m2.y + eg, idet = 1.0 / (a * c - eg * eg);
            v11 = c * idet; v12 = eg * idet; v22 = a * idet;
            mu1 = v11 * m1.x + v12 * m2.x; mu2 = v12 * m1.x + v22 * m2.x;
        }
        double *o = joint + 6 * (int64_t)idx;
        o[0] = mu1; o[1] = mu2; o[2] = v11; o[3] = v12; o[4] = v12; o[5] = v22;
    } else {                                  // kItemStGamma: Gamma(3/2, 2 / (V11 - 2 V12 + V22 + (m1 - m2)^2)) from the joint          (:1011-1016)
        const int jb = 6 * s0;
        const double o0 = ld1<COH>(joint, jb), o1 = ld1<COH>(joint, jb + 1), o2 = ld1<COH>(joint, jb + 2), o3 = ld1<COH>(joint, jb + 3), o4 = ld1<COH>(joint, jb + 4), o5 = ld1<COH>(joint, jb + 5);
        const double d = o0 - o1, spread = o2 - o3 - o4 + o5 + d * d;
        if (!__builtin_isnan(spread)) f2v[idx] = make_double2(0.5, 0.5 * spread);
    }
}
template <int MODE, bool COH = false>
__device__ __forceinline__ void batch_item(int k, int idx, int v, int lo, int hi, const int32_t *__restrict__ vbase,
                                           const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                           const int32_t *__restrict__ partner, const double *__restrict__ q,
                                           const double *__restrict__ pa, const double *__restrict__ pb,
                                           double2 *__restrict__ f2v, double2 *__restrict__ v2f, double2 *__restrict__ marg, int nat_marg,
                                           double2 *__restrict__ prod, double *__restrict__ joint, const KaryTab kt, double2 *fwd = nullptr) {
    if (k == kItemKaryEntry) {            // internal (the tree schedule's stage lists): index = entry of the k-ary table
        kary_item<COH>(idx, kt.slot, kt.coef, kt.qb, v2f, f2v);
    } else if (k >= kItemMfNormal && k <= kItemStGamma) {
        vmp_item<COH>(k, idx, kt.list[lo], hi > 1 ? kt.list[lo + 1] : 0, hi > 2 ? kt.list[lo + 2] : 0, f2v, v2f, marg, joint);
    } else if (k >= kItemSumToFactor) {   // internal (reference-order plans)
        double2 acc = zero2();
        for (int j = 0; j < hi; j++) { const int s = kt.list[lo + j]; acc = add2(acc, s >= 0 ? ld2<COH>(f2v, s) : ld2<COH>(prod, ~s)); }
        if (k == kItemSumToMarginal) marg[v] = nat_marg ? acc : to_moment(acc);
        else if (k == kItemSumToGammaMarginal) marg[v] = make_double2(acc.x + 1.0, 1.0 / acc.y);
        else if (!__builtin_isnan(acc.y)) { if (k == kItemSumToFactor) { v2f[idx] = acc; if (fwd) *fwd = acc; } else prod[idx] = acc; }
    } else if (k == CX_ITEM_MESSAGE_TO_FACTOR) {
        m2f_one<COH>(idx, v, vbase, vdeg, vinfo, f2v, v2f, fwd);
    } else if (k == CX_ITEM_MESSAGE_TO_VARIABLE) {
        const int p = partner[idx];
        if (p < 0) return;
        const double2 m = ld2<COH>(v2f, p);
        if (__builtin_isnan(m.y)) return;
        const double2 r = factor_rule<MODE>(m, q[idx], MODE == kRuleLinear ? pa[idx] : 1.0, MODE == kRuleLinear ? pb[idx] : 0.0);
        if (MODE != kRuleBernoulli || !__builtin_isnan(r.y)) f2v[idx] = r;
    } else if (k == CX_ITEM_INDIVIDUAL_MARGINAL) {
        const int deg = vdeg[v];
        const int stride = ((vinfo[v] & kDegMask) == kBigDeg) ? 1 : kBlock;
        const int b = vbase[v];
        double2 acc = zero2();
        for (int j = 0; j < deg; j++) acc = add2(acc, ld2<COH>(f2v, b + j * stride));
        marg[v] = (deg > 0) ? (nat_marg ? acc : to_moment(acc)) : nan2();
    } else if (k == CX_ITEM_PRODUCT_OF_MESSAGES) {
        const int stride = ((vinfo[v] & kDegMask) == kBigDeg) ? 1 : kBlock;
        const int b = vbase[v];
        double2 acc = zero2();
        for (int j = lo - 1; j < hi; j++) acc = add2(acc, ld2<COH>(f2v, b + j * stride));
        if (!__builtin_isnan(acc.y)) prod[idx] = acc;     // a dependency is undefined: not pending, keep the stored value
    } else if (k == CX_ITEM_JOINT_MARGINAL) {
        if (MODE == kRuleBernoulli) return;
        const int s_out = v, s_in = lo;
        const double2 m_in = ld2<COH>(v2f, s_in), m_out = ld2<COH>(v2f, s_out);
        double *o = joint + 6 * (int64_t)idx;
        if (__builtin_isnan(m_in.y) || __builtin_isnan(m_out.y)) return;
        const double a = MODE == kRuleLinear ? pa[s_out] : 1.0, b = MODE == kRuleLinear ? pb[s_out] : 0.0, iq = 1.0 / q[s_out];
        double mi, mo, cii, cio, coo;
        const double inf = __builtin_inf();
        if (m_in.y == inf && m_out.y == inf) { mi = m_in.x; mo = m_out.x; cii = cio = coo = 0.0; }
        else if (m_in.y == inf) {          // x_in observed: x_out | x_in
            mi = m_in.x; cii = cio = 0.0;
            coo = 1.0 / (m_out.y + iq); mo = coo * (m_out.x + (a * mi + b) * iq);
        } else if (m_out.y == inf) {       // x_out observed
            mo = m_out.x; coo = cio = 0.0;
            cii = 1.0 / (m_in.y + a * a * iq); mi = cii * (m_in.x + a * (mo - b) * iq);
        } else {
            // precision [[w_in + a²/q, -a/q], [-a/q, w_out + 1/q]], potential [xi_in - a b/q, xi_out + b/q]
            const double l11 = m_in.y + a * a * iq, l12 = -a * iq, l22 = m_out.y + iq;
            const double e1 = m_in.x - a * b * iq, e2 = m_out.x + b * iq;
            const double idet = 1.0 / (l11 * l22 - l12 * l12);
            cii = l22 * idet; cio = -l12 * idet; coo = l11 * idet;
            mi = cii * e1 + cio * e2; mo = cio * e1 + coo * e2;
        }
        if (hi) { o[0] = mo; o[1] = mi; o[2] = coo; o[3] = cio; o[4] = cio; o[5] = cii; }
        else    { o[0] = mi; o[1] = mo; o[2] = cii; o[3] = cio; o[4] = cio; o[5] = coo; }
    }
}


// a record that leads and the record behind it, by one thread (cx_refsched.h: kRecLeads).  A MessageToFactor and the pairwise rule that reads
// it pass the message on in a register; any other pair stores first (the follower then loads what the leader has written through)
template <int MODE>
__device__ __forceinline__ void batch_pair(const int32_t *__restrict__ lead, const int32_t *__restrict__ fol, const int32_t *__restrict__ vbase,
                                           const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner,
                                           const double *__restrict__ q, const double *__restrict__ pa, const double *__restrict__ pb, double2 *__restrict__ f2v,
                                           double2 *__restrict__ v2f, double2 *__restrict__ marg, int nat_marg, double2 *__restrict__ prod, double *__restrict__ joint,
                                           const KaryTab kt) {
    const int kl = lead[0] & kRecKindMask, kf = fol[0] & kRecKindMask, fidx = fol[1];
    if ((kl == CX_ITEM_MESSAGE_TO_FACTOR || kl == kItemSumToFactor) && kf == CX_ITEM_MESSAGE_TO_VARIABLE && partner[fidx] == lead[1]) {
        const double qq = q[fidx], a = MODE == kRuleLinear ? pa[fidx] : 1.0, b = MODE == kRuleLinear ? pb[fidx] : 0.0;
        double2 m = nan2();
        batch_item<MODE>(kl, lead[1], lead[2], lead[3], lead[4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, &m);
        if (__builtin_isnan(m.y)) m = v2f[lead[1]];      // the leader stored nothing (an observed variable, an undefined input): what is stored there
        if (__builtin_isnan(m.y)) return;
        const double2 r = factor_rule<MODE>(m, qq, a, b);
        if (MODE != kRuleBernoulli || !__builtin_isnan(r.y)) f2v[fidx] = r;
        return;
    }
    if ((kl == CX_ITEM_MESSAGE_TO_FACTOR || kl == kItemSumToFactor) && kf == kItemStNormal && kt.list[fol[3]] == lead[1]) {
        // the structured variational rule behind the message it reads (a chain of a wired model): N(mean m, 1 / (var m + 1 / E[precision]))
        const double2 g = marg[kt.list[fol[3] + 1]];
        double2 m = nan2();
        batch_item<MODE>(kl, lead[1], lead[2], lead[3], lead[4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, &m);
        if (__builtin_isnan(m.y)) m = v2f[lead[1]];
        const double eg = g.x * g.y;
        if (__builtin_isnan(m.y) || __builtin_isnan(m.x) || __builtin_isnan(eg)) return;
        const double inf = __builtin_inf();
        const double mean = m.y == inf ? m.x : m.x / m.y, var = m.y == inf ? 0.0 : 1.0 / m.y;
        const double w = 1.0 / (var + 1.0 / eg);
        f2v[fidx] = make_double2(mean * w, w);
        return;
    }
    batch_item<MODE>(kl, lead[1], lead[2], lead[3], lead[4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the leader's store has been written through before the follower loads it
    batch_item<MODE>(kf, fol[1], fol[2], fol[3], fol[4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_batch(int64_t n, const int32_t *__restrict__ rec, const int32_t *__restrict__ vbase,
                                                  const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                                  const int32_t *__restrict__ partner, const double *__restrict__ q,
                                                  const double *__restrict__ pa, const double *__restrict__ pb,
                                                  double2 *__restrict__ f2v, double2 *__restrict__ v2f, double2 *__restrict__ marg, int nat_marg,
                                                  double2 *__restrict__ prod, double *__restrict__ joint, const KaryTab kt) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int k0 = rec[5 * i];
    if (k0 & kRecFollows) return;
    if (k0 & kRecLeads) batch_pair<MODE>(rec + 5 * i, rec + 5 * i + 5, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
    else batch_item<MODE>(k0, rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
}

// A RUN of consecutive thin stages of the tree schedule (each at most kRunBlock items) in ONE launch of ONE workgroup: the stages of a
// run depend on each other, so the workgroup takes them one after the other with a barrier (and a workgroup-scope fence: the threads of
// a workgroup share their compute unit's vector cache) in between — instead of a launch of ≈ 5 us per stage of a few items.  The thin
// stages are the levels next to the roots, on the way up and again on the way down.
constexpr int kRunBlock = 1024;
template <int MODE>
__global__ __launch_bounds__(kRunBlock) void k_batch_run(const int64_t *__restrict__ stage_off, int s0, int s1, const int32_t *__restrict__ rec,
                                                         const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                                         const int32_t *__restrict__ partner, const double *__restrict__ q, const double *__restrict__ pa,
                                                         const double *__restrict__ pb, double2 *__restrict__ f2v, double2 *__restrict__ v2f,
                                                         double2 *__restrict__ marg, int nat_marg, double2 *__restrict__ prod, double *__restrict__ joint,
                                                         const KaryTab kt, unsigned long long *ticks) {
    // (ticks: CX_RUN_TIME=1, thread 0's clock in 10 ns — items, the release fence, the barrier — added up over every launch of the process.
    // Measured on the wired variational chain: items 2.4 – 2.6 us a stage, fence 0.04, barrier 0.03; a wavefront that touched the records and
    // list entries three stages ahead, and records loaded whole instead of kind first: 2.35 – 2.4, i.e. nothing — a stage is its chain of
    // value loads, arithmetic and stores, not its plan constants.  Both taken out again.)
    unsigned long long tk[3] = {0, 0, 0}, t0 = ticks && threadIdx.x == 0 ? wall_clock64() : 0;
    const bool timed = ticks && threadIdx.x == 0;
    for (int st = s0; st < s1; st++) {
        for (int64_t i = stage_off[st] + threadIdx.x; i < stage_off[st + 1]; i += kRunBlock) {      // (the tree schedule folds stages of at most kRunBlock items: one trip)
            const int k0 = rec[5 * i];
            if (k0 & kRecFollows) continue;
            if (k0 & kRecLeads) batch_pair<MODE>(rec + 5 * i, rec + 5 * i + 5, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
            else batch_item<MODE>(k0, rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
        }
        if (timed) { const unsigned long long t = wall_clock64(); tk[0] += t - t0; t0 = t; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (timed) { const unsigned long long t = wall_clock64(); tk[1] += t - t0; t0 = t; }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (timed) { const unsigned long long t = wall_clock64(); tk[2] += t - t0; t0 = t; }
    }
    if (timed) { for (int j = 0; j < 3; j++) atomicAdd(ticks + j, tk[j]); atomicAdd(ticks + 3, (unsigned long long)(s1 - s0)); }
}

// ---- an XCD-resident cluster: the stages of a reference-order plan behind barriers that never leave one L2 -------------------------------
// A reference-order call on a loopy graph is thousands of DEPENDENT stages of a few thousand items (C4: 5,659 stages, 18 M items).  As
// launches a stage costs ≈ 9 us (launch latency + three dependent memory round trips); a device-wide barrier costs 19 us and more, because
// the eight XCDs keep separate L2s and an agent-scope release / acquire writes them back and invalidates them (profiles/r04_grid_barrier.txt).
// The workgroups of ONE XCD share one L2: a store is written through the compute unit's vector cache to it, a load that bypasses the vector
// cache (ld2<true>: a 16-byte buffer load with scope bit sc1) reads from it, and nothing is written back or invalidated in between — a barrier among them is one
// counter in that L2: 0.75 us bare, 1.5 us with every thread passing a value to a thread of another workgroup (tools/lab/xcd_barrier.hip,
// profiles/r05_xcd_barrier.txt: 32 workgroups x 1,024 threads, no wrong value in 2,000 rounds).  So ONE launch of (compute units) workgroups:
// those that find themselves on XCD 0 (hardware XCC_ID) form the cluster — 32 x 1,024 threads, an eighth of the chip, which is more than
// a stage is wide — the others leave at once; the members take the plan's stages one after the other, items dealt in runs of 1,024, with
// that barrier in between.  Values are loaded coherently (batch_item<MODE, true>), plan records and graph constants as always.  Every wait
// is bounded: a member that gives up raises a flag that all members see and the host checks (cx_api_ref.hip: the call then fails loudly).
// 512 B, zeroed before every launch: the barrier's line; `progress` (the stage the members are in, what the helpers poll — on a line of its own,
// so that they do not queue behind the barrier's atomics); `ticks` (CX_REF_CLUSTER_TIME=1: where member 0 spent the call, in 10 ns)
struct ClusterCtl { unsigned registered, members, rank_next, arrive, abort_, xcd_plus_1, pad[26]; unsigned progress, pad2[31]; unsigned long long ticks[8], wave_ticks[16]; unsigned pad3[16]; };
static_assert(sizeof(ClusterCtl) == 512, "cx_api_ref.hip allocates and reads back 512 bytes");

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_ref_cluster: hwreg(20) is XCC_ID, s_waitcnt vmcnt(0) acknowledges stores and one XCD's L2 serves sc1 loads of plain stores on gfx942 / gfx950 only"
#endif
__device__ __forceinline__ unsigned hw_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(v));      // HW_REG_XCC_ID
    return v & 7u;
}
// Every wait is bounded by TIME, not by a poll count: `limit` ticks of the constant 100 MHz clock (wall_clock64; cx_api_ref.hip passes 1.5 s,
// CX_REF_CLUSTER_TIMEOUT_MS changes it) — a bound that is known to sit under the driver's compute-lockup timeout whatever the contention.
// The clock is read every 64th poll (the first reading starts the wait's own clock: a wait that is satisfied at once never reads it).
__device__ __forceinline__ bool cluster_wait(unsigned *p, unsigned target, unsigned *abort_, unsigned long long limit) {
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; spins++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if ((spins & 63u) == 63u) {
            if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            const unsigned long long now = wall_clock64();
            if (!t0) t0 = now;
            else if (now - t0 > limit) { __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
}
constexpr int kClusterBlock = 1024;
// What a member does per stage is a chain of dependent round trips to the L2: stage table -> record -> graph tables (degree, base slot,
// partner) -> values -> store -> barrier.  The first three are known when the plan is made, so the cluster runs FLAT records (cx_api_ref.hip:
// flat_records, 8 ints): kind | n << 8, destination, variable, five sources resolved to slots (>= 0: a factor→variable slot — the
// variable→factor slot for a rule —, ~index: a node of the product store).  All sources of an item are loaded together (a source that is
// not there reads past the end of the buffer: zero, no traffic), and a thread fetches its record of the NEXT stage before it waits at the
// barrier: value loads -> store -> barrier is what is left on the chain.  kFlatGeneric: the item's ordinary record (five ints at index
// `destination`) through batch_item — rules of factors with more than two edges, the variational rules, sums of more than five sources.
constexpr int kFlatSumToFactor = 1, kFlatSumToMarginal = 2, kFlatSumToGamma = 3, kFlatSumToProduct = 4, kFlatRule = 5, kFlatGeneric = 6;
constexpr int kFlatVmp = 7;      // a variational rule (kItemMfNormal + n, n in the count field), its list entries in s[0 .. 2]
constexpr int kFlatCheckObserved = 0x80;      // MessageToFactor of the compact form: not recomputed for an observed / stand-in variable (m2f_one)
struct FlatRec { int32_t k, dst, v, s[5]; };

// fwd (may be NULL): where a kFlatSumToFactor item leaves the message it stored (fwd->y NaN: it stored nothing) — its follower's input
template <int MODE>
__device__ __forceinline__ void flat_item(const FlatRec r, const int32_t *__restrict__ rec, const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg,
                                          const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner, const double *__restrict__ q,
                                          const double *__restrict__ pa, const double *__restrict__ pb, double2 *f2v, double2 *v2f, double2 *marg, int nat_marg,
                                          double2 *prod, double *joint, const KaryTab kt, double2 *fwd = nullptr) {
    const int kind = r.k & 0x7f, n = (r.k >> 8) & 0xff;
    if (kind == kFlatGeneric) {
        const int32_t *g = rec + 5 * (int64_t)r.dst;
        batch_item<MODE, true>(g[0] & kRecKindMask, g[1], g[2], g[3], g[4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
        return;
    }
    if (kind == kFlatVmp) { vmp_item<true>(kItemMfNormal + n, r.dst, r.s[0], r.s[1], r.s[2], f2v, v2f, marg, joint); return; }
    if (kind == kFlatRule) {
        const double2 m = ld2<true>(v2f, r.s[0]);
        const double qq = q[r.dst], a = MODE == kRuleLinear ? pa[r.dst] : 1.0, b = MODE == kRuleLinear ? pb[r.dst] : 0.0;
        if (__builtin_isnan(m.y)) return;
        const double2 o = factor_rule<MODE>(m, qq, a, b);
        if (MODE != kRuleBernoulli || !__builtin_isnan(o.y)) f2v[r.dst] = o;
        return;
    }
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)f2v, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)prod, 0, 0x7fffffff, 0x00020000);
    // (a buffer load like the values', not a branch on a byte that has to arrive first: one round trip to the L2 for all of them)
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)vinfo, 0, 0x7fffffff, 0x00020000);
    const int info = __builtin_amdgcn_raw_buffer_load_b8(ri, (r.k & kFlatCheckObserved) ? r.v : -1, 0, 0);
    cx_d2v val[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int sj = r.s[j];
        const bool on = j < n;
        // (an offset of -1 is past the end of the 2 GiB window: the load returns zero and moves nothing)
        const cx_d2v a = __builtin_bit_cast(cx_d2v, __builtin_amdgcn_raw_buffer_load_b128(rf, (on && sj >= 0) ? sj * 16 : -1, 0, 16));
        const cx_d2v b = __builtin_bit_cast(cx_d2v, __builtin_amdgcn_raw_buffer_load_b128(rp, (on && sj < 0) ? (~sj) * 16 : -1, 0, 16));
        val[j] = a + b;
    }
    double2 acc = zero2();      // left to right, like the reference's fold
#pragma unroll
    for (int j = 0; j < 5; j++) if (j < n) acc = make_double2(acc.x + val[j][0], acc.y + val[j][1]);
    if (kind == kFlatSumToMarginal) marg[r.dst] = nat_marg ? acc : to_moment(acc);
    else if (kind == kFlatSumToGamma) marg[r.dst] = make_double2(acc.x + 1.0, 1.0 / acc.y);
    else if (!__builtin_isnan(acc.y) && !(info & (kClamped | kGhost))) {
        if (kind == kFlatSumToFactor) { v2f[r.dst] = acc; if (fwd) *fwd = acc; }
        else prod[r.dst] = acc;
    }
}

// a leader and the record behind it (cx_refsched.h: kRecLeads).  The common pair — a MessageToFactor sum and the rule of the message that reads
// it — passes the message on in a register: the follower does not wait for the leader's store.  Any other pair: the store first, then the follower.
template <int MODE>
__device__ __forceinline__ void flat_pair(const FlatRec lead, const FlatRec fol, const int32_t *__restrict__ rec, const int32_t *__restrict__ vbase,
                                          const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner,
                                          const double *__restrict__ q, const double *__restrict__ pa, const double *__restrict__ pb, double2 *f2v, double2 *v2f,
                                          double2 *marg, int nat_marg, double2 *prod, double *joint, const KaryTab kt, bool forward) {
    if (forward && (lead.k & 0x7f) == kFlatSumToFactor && (fol.k & 0x7f) == kFlatRule && fol.s[0] == lead.dst) {
        const double qq = q[fol.dst], a = MODE == kRuleLinear ? pa[fol.dst] : 1.0, b = MODE == kRuleLinear ? pb[fol.dst] : 0.0;
        double2 m = nan2();
        flat_item<MODE>(lead, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, &m);
        if (__builtin_isnan(m.y)) m = ld2<true>(v2f, fol.s[0]);      // the leader stored nothing (an observed variable, an undefined input): what is stored there
        if (__builtin_isnan(m.y)) return;
        const double2 o = factor_rule<MODE>(m, qq, a, b);
        if (MODE != kRuleBernoulli || !__builtin_isnan(o.y)) f2v[fol.dst] = o;
        return;
    }
    if (forward && (lead.k & 0x7f) == kFlatSumToFactor && (fol.k & 0x7f) == kFlatVmp && ((fol.k >> 8) & 0xff) == kItemStNormal - kItemMfNormal && fol.s[0] == lead.dst) {
        // the structured variational rule behind the message it reads (a chain of a wired model): N(mean m, 1 / (var m + 1 / E[precision]))
        const double2 g = ld2<true>(marg, fol.s[1]);
        double2 m = nan2();
        flat_item<MODE>(lead, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, &m);
        if (__builtin_isnan(m.y)) m = ld2<true>(v2f, fol.s[0]);
        const double eg = g.x * g.y;
        if (__builtin_isnan(m.y) || __builtin_isnan(m.x) || __builtin_isnan(eg)) return;
        const double inf = __builtin_inf();
        const double mean = m.y == inf ? m.x : m.x / m.y, var = m.y == inf ? 0.0 : 1.0 / m.y;
        const double w = 1.0 / (var + 1.0 / eg);
        f2v[fol.dst] = make_double2(mean * w, w);
        return;
    }
    flat_item<MODE>(lead, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flat_item<MODE>(fol, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
}

__device__ __forceinline__ FlatRec flat_load(const int32_t *__restrict__ flat, int64_t i) {
    const int4 a = *(const int4 *)(flat + 8 * i), b = *(const int4 *)(flat + 8 * i + 4);
    FlatRec r;
    r.k = a.x; r.dst = a.y; r.v = a.z; r.s[0] = a.w; r.s[1] = b.x; r.s[2] = b.y; r.s[3] = b.z; r.s[4] = b.w;
    return r;
}

// A helper's pass over one stage (help = 3): what a member will wait for, touched ahead of it so that it waits for the XCD's L2 and not for
// memory — the records, the value lines of every source, a rule's constants, an observed-flag byte.  Nothing here depends on anything but the
// record, so a thread takes four records at a time and has all their lines in flight together (one word of each line is enough; a load
// that is not wanted points past the end of its buffer: zero, no traffic).
template <int MODE>
__device__ __forceinline__ unsigned help_stage(const int32_t *__restrict__ flat, int64_t lo, int64_t hi, const double2 *f2v, const double2 *v2f, const double2 *prod,
                                               const double *q, const double *pa, const double *pb, const uint8_t *vinfo) {
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)f2v, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)v2f, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)prod, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)q, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(MODE == kRuleLinear ? pa : q), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(MODE == kRuleLinear ? pb : q), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)vinfo, 0, 0x7fffffff, 0x00020000);
    constexpr int U = 4;
    unsigned sink = 0;
    for (int64_t base = lo + threadIdx.x; base < hi; base += U * kClusterBlock) {
        FlatRec r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + (int64_t)u * kClusterBlock;
            r[u] = flat_load(flat, i < hi ? i : base);
            if (i >= hi) r[u].k = 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kind = r[u].k & 0x7f, n = (r[u].k >> 8) & 0xff;
            const bool rule = kind == kFlatRule, sum = kind >= kFlatSumToFactor && kind <= kFlatSumToProduct;
            // (a follower's message comes to it in a register: cx_batch.hip flat_pair)
            sink ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rv, rule && !(r[u].k & kRecFollows) ? r[u].s[0] * 16 + 8 : -1, 0, 16);
            sink ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rq, rule ? r[u].dst * 8 : -1, 0, 16);
            if (MODE == kRuleLinear) {
                sink ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, rule ? r[u].dst * 8 : -1, 0, 16);
                sink ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rb, rule ? r[u].dst * 8 : -1, 0, 16);
            }
            sink ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ri, (r[u].k & kFlatCheckObserved) && kind != kFlatGeneric ? (r[u].v & ~3) : -1, 0, 16);
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int sj = r[u].s[j];
                const bool on = sum && j < n;
                sink ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rf, (on && sj >= 0) ? sj * 16 + 8 : -1, 0, 16);
                sink ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rp, (on && sj < 0) ? (~sj) * 16 + 8 : -1, 0, 16);
            }
        }
    }
    return sink;
}

template <int MODE>
__global__ __launch_bounds__(kClusterBlock) void k_ref_cluster(ClusterCtl *c, unsigned G, const int64_t *__restrict__ stage_off, int n_stages,
                                                               const int32_t *__restrict__ flat, const int32_t *__restrict__ rec, const int32_t *__restrict__ vbase,
                                                               const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner,
                                                               const double *__restrict__ q, const double *__restrict__ pa, const double *__restrict__ pb, double2 *f2v,
                                                               double2 *v2f, double2 *marg, int nat_marg, double2 *prod, double *joint, const KaryTab kt, int dry,
                                                               unsigned long long wait_limit, int fault_stage) {
    __shared__ unsigned rank_s, members_s, ok_s, mine_s;
    if (threadIdx.x == 0) {      // the cluster's XCD is the one of the first workgroup to ask (whatever the partition mode numbers it)
        const unsigned me = hw_xcc_id() + 1u;
        unsigned expected = 0u;
        const bool won = __hip_atomic_compare_exchange_strong(&c->xcd_plus_1, &expected, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mine_s = (won || expected == me) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = mine_s != 0;
    if (threadIdx.x == 0) {
        rank_s = mine ? __hip_atomic_fetch_add(&c->rank_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (mine) __hip_atomic_fetch_add(&c->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&c->registered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok_s = 1u;
        if (mine) {      // once every workgroup of the launch has said where it runs, the membership is final
            ok_s = cluster_wait(&c->registered, G, &c->abort_, wait_limit) ? 1u : 0u;
            members_s = __hip_atomic_load(&c->members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!mine || !ok_s) return;
    // Roles.  A stage is rarely wider than a few thousand items, and what a member waits for most is memory it has never touched: the plan's
    // records (hundreds of MB, read once) and values last written many stages ago.  So three eighths of the cluster's workgroups are HELPERS: they
    // take no part in the barriers and store nothing; helper j runs ahead of the members through the stages s = j (mod H), loads their
    // records (help >= 1) and the lines of their sources (help >= 2) — into the L2 the members read from — and never lets anybody wait.
    const int help = (dry >> 1) & 3, ahead_arg = (dry >> 8) & 0xff, members_arg = (dry >> 16) & 0xff;
    const int64_t all = members_s, P = help && all >= 4 ? (members_arg && members_arg < all ? members_arg : all * 5 / 8) : all, H = all - P;
    // (what the host needs to find the first incomplete stage after a wait that timed out: arrive / P, cx_api_ref.hip: cluster_run)
    if (rank_s == 0 && threadIdx.x == 0) __hip_atomic_store(&c->pad[1], (unsigned)P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (20 of 32: 7.9 – 8.1 against 8.0 – 8.4 ms for 16 in four pairs of runs; 24 and more: the helpers fall behind)
    if ((int64_t)rank_s >= P) {
        const int64_t hj = (int64_t)rank_s - P;
        const int ahead = ahead_arg ? ahead_arg : (help >= 3 ? 10 : help >= 2 ? 4 : 12);      // stages: the L2 is 4 MB, a stage's sources up to 1 MB of lines, its records 32 B an item
        unsigned sink = 0;
        for (int64_t s = 2 + hj; s < n_stages; s += H) {
            if (threadIdx.x == 0) {
                unsigned okh = 1u;
                unsigned long long t0 = 0;
                for (unsigned spins = 0;; spins++) {
                    const int64_t cur = __hip_atomic_load(&c->progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (s <= cur + ahead) break;
                    if ((spins & 63u) == 63u) {      // (a helper never raises the flag: it only stops helping when the members have stopped or its own wait is too long)
                        if (__hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { okh = 0u; break; }
                        const unsigned long long now = wall_clock64();
                        if (!t0) t0 = now; else if (now - t0 > wait_limit) { okh = 0u; break; }
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                ok_s = okh;
            }
            __syncthreads();
            if (!ok_s) return;
            if (help >= 3) { sink ^= help_stage<MODE>(flat, stage_off[s], stage_off[s + 1], f2v, v2f, prod, q, pa, pb, vinfo); __syncthreads(); continue; }
            for (int64_t i = stage_off[s] + threadIdx.x; i < stage_off[s + 1]; i += kClusterBlock) {
                const FlatRec r = flat_load(flat, i);
                sink ^= (unsigned)r.k ^ (unsigned)r.dst;
                if (help >= 2) {
                    const int kind = r.k & 0x7f, n = (r.k >> 8) & 0xff;
                    if (kind == kFlatRule) sink ^= (unsigned)__double_as_longlong(ld2<true>(v2f, r.s[0]).y);
                    else if (kind != kFlatGeneric && kind != kFlatVmp)
                        for (int j = 0; j < 5; j++) if (j < n) sink ^= (unsigned)__double_as_longlong((r.s[j] >= 0 ? ld2<true>(f2v, r.s[j]) : ld2<true>(prod, ~r.s[j])).y);
                }
            }
            __syncthreads();
        }
        if (sink == 0x9e3779b9u) c->pad[0] = sink;      // (keeps the loads)
        return;
    }
    // records are dealt a wavefront at a time, the members' FIRST wavefronts before anybody's second: a stage is rarely as wide as the cluster, and
    // the wavefronts that share a SIMD take turns at issuing (measured: each 0.3 us behind the one before it)
    const int64_t first = ((int64_t)(threadIdx.x >> 6) * P + rank_s) * 64 + (threadIdx.x & 63), step = P * kClusterBlock;
    int64_t lo = stage_off[0], hi = stage_off[1];
    FlatRec cur{}, cur2{};      // this thread's record of the stage and the one behind it (a leader's follower)
    bool have = lo + first < hi;
    if (have) cur = flat_load(flat, lo + first);
    if (lo + first + 1 < hi) cur2 = flat_load(flat, lo + first + 1);
    int64_t nhi = n_stages > 1 ? stage_off[2] : hi;
    const bool timed = (dry & 32) && rank_s == 0 && threadIdx.x == 0;
    const bool wtimed = (dry & 32) && rank_s == 0 && (threadIdx.x & 63) == 0;      // each wavefront of member 0: from the barrier's release to its stores' acknowledgement
    unsigned long long tk[5] = {0, 0, 0, 0, 0}, t0 = timed ? wall_clock64() : 0, wt = 0, w0 = wtimed ? wall_clock64() : 0;
    for (int st = 0; st < n_stages; st++) {
        // the next stage's bounds (fetched a stage earlier still); this thread's first record of it is asked for once this stage's items are done:
        // plan constants, on their way while the stores are acknowledged and the others arrive
        const int64_t nlo = hi, nnhi = st + 2 < n_stages ? stage_off[st + 3] : nhi;
        const bool nhave = st + 1 < n_stages && nlo + first < nhi;
        if (!(dry & 1)) {      // (CX_REF_CLUSTER_DRY=1: the plan's skeleton — records and barriers, no item — for timing what a stage costs before it computes)
            if (have && !(cur.k & kRecFollows)) {
                if (cur.k & kRecLeads) flat_pair<MODE>(cur, cur2, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, !(dry & 16));
                else flat_item<MODE>(cur, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
            }
            for (int64_t i = lo + first + step; i < hi; i += step) {      // (a stage wider than the cluster)
                const FlatRec r = flat_load(flat, i);
                if (r.k & kRecFollows) continue;
                if (r.k & kRecLeads) flat_pair<MODE>(r, flat_load(flat, i + 1), rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, !(dry & 16));
                else flat_item<MODE>(r, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
            }
        }
        if (st + 1 == n_stages) break;
        // (after the items, not before: the hardware counts loads in order, and a record asked for first would be waited for first —
        // a round trip to the L2 before the stage's values are even asked for)
        asm volatile("" ::: "memory");
        FlatRec nxt{}, nxt2{};
        if (nhave && !(dry & 8)) nxt = flat_load(flat, nlo + first);      // (bit 3, CX_REF_CLUSTER_DRY=2: not even the records — the bare barriers)
        if (nlo + first + 1 < nhi && !(dry & 8)) nxt2 = flat_load(flat, nlo + first + 1);
        if (timed) { const unsigned long long t = wall_clock64(); tk[0] += t - t0; t0 = t; }      // items issued (and what the compiler made them wait for)
        // the barrier: this thread's stores have reached the L2, the workgroup has arrived, one thread reports and waits for the others
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (timed) { const unsigned long long t = wall_clock64(); tk[1] += t - t0; t0 = t; }      // loads returned, stores acknowledged
        if (wtimed) wt += wall_clock64() - w0;
        __syncthreads();
        if (timed) { const unsigned long long t = wall_clock64(); tk[2] += t - t0; t0 = t; }      // the workgroup's other wavefronts
        // (fault injection, CX_REF_CLUSTER_FAULT=<stage>: the member of rank 1 never arrives at this stage's barrier — what a workgroup
        // that was never resident, or died, looks like to the others; tests/test_gpu_cluster.py)
        if (fault_stage > 0 && st + 1 == fault_stage && rank_s == 1) return;
        if (threadIdx.x == 0) {
            // the last to arrive writes a word on a line of its own and the others poll that word, not the counter the arrivals are still
            // adding to (8.2 – 8.4 against 8.3 – 8.8 ms per call at C4, three pairs of runs; CX_REF_CLUSTER_RELEASE=0: everybody polls the counter)
            if ((dry >> 27) & 1) {
                const unsigned before = __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1u == (unsigned)((st + 1) * P)) { __hip_atomic_store(&c->pad3[0], (unsigned)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok_s = 1u; }
                else ok_s = cluster_wait(&c->pad3[0], (unsigned)(st + 1), &c->abort_, wait_limit) ? 1u : 0u;
            } else {
                __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok_s = cluster_wait(&c->arrive, (unsigned)((st + 1) * P), &c->abort_, wait_limit) ? 1u : 0u;
            }
            if (rank_s == 0) __hip_atomic_store(&c->progress, (unsigned)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (timed) { const unsigned long long t = wall_clock64(); tk[3] += t - t0; t0 = t; }      // the other workgroups
        __syncthreads();
        if (!ok_s) return;
        if (timed) { const unsigned long long t = wall_clock64(); tk[4] += t - t0; t0 = t; }
        if (wtimed) w0 = wall_clock64();
        lo = nlo; hi = nhi; nhi = nnhi; have = nhave; cur = nxt; cur2 = nxt2;
    }
    if (timed) for (int j = 0; j < 5; j++) c->ticks[j] = tk[j];
    if (wtimed) c->wave_ticks[threadIdx.x >> 6] = wt;
}

// A run of consecutive thin stages of a reference-order plan in ONE launch of ONE workgroup (k_batch_run's job) on FLAT records.  What a
// stage of such a run costs is a chain of dependent loads — stage table, record, list entries, graph tables, values — and a hand-off
// through memory inside one workgroup is only 0.16 – 0.19 us (tools/lab/wg_handoff.hip: store, barrier, load; 0.15 more for every index
// that has to come from memory first), so the plan's constants are taken off the chain: a thread holds its flat record (sources resolved) and
// the one behind it, and asks for the next stage's pair before this stage's barrier.  Stages of at most kRunBlock items.
// (Measured on top and not kept: the next record asked for BEFORE the items, so that a stage does not begin by waiting for the last one's
// stores to be acknowledged — loads and stores come back in issue order — 1.51 against 1.37 ms on the 1,001-stage wired chain; the last
// wavefront touching the records six stages ahead: nothing, 1.37.)
// The records travel through LDS: loads and stores come back in the order they were issued, so a thread that asked for its next record
// after its stores began the next stage by waiting for those stores' acknowledgement (0.3 us), and one that asked before its items made its
// values wait for the record.  The LAST wavefront stores nothing: it fetches the next stage's records into the other half of a double buffer
// while the fifteen others work, and after the barrier everybody reads its record from LDS (another counter).  Per stage on the 1,001-stage
// wired chain: 2.06 us (k_batch_run) -> 1.63 (flat records) -> 1.31 (one item per wavefront) -> see DESIGN.md §4c for this form.
// (Measured on top and not kept: even stages on wavefronts 0 .. 6 and odd stages on 7 .. 13, so that a stage's loads do not queue behind the
// last one's store acknowledgements: 1.22 against 1.14 ms, 158 against 119 – 138 ms at n = 1e5.)
constexpr int kRunItemWaves = kRunBlock / 64 - 1, kFlatRunMax = kRunItemWaves * 64;      // 15 wavefronts of items: stages of at most 960 records
// (cx_api_ref.hip: flat_records) the first record of a group of one kind in a stage of at most 64 records, which are sorted by kind
constexpr int kFlatGroupStart = 0x10000000;
template <int MODE>
__global__ __launch_bounds__(kRunBlock) void k_flat_run(const int64_t *__restrict__ stage_off, int s0, int s1, const int32_t *__restrict__ flat, const int32_t *__restrict__ rec,
                                                        const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                                        const int32_t *__restrict__ partner, const double *__restrict__ q, const double *__restrict__ pa,
                                                        const double *__restrict__ pb, double2 *f2v, double2 *v2f, double2 *marg, int nat_marg, double2 *prod, double *joint,
                                                        const KaryTab kt, int by_kind) {
    __shared__ int4 rb[2][kFlatRunMax][2];      // 60 KB: the records of this stage and of the next, where their threads look for them
    __shared__ int grouped_s[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool loader = wave == kRunItemWaves;
    const int4 *flat4 = (const int4 *)flat;
    // WHERE a record goes decides who runs it: thread t takes the record in slot t.  The records of a stage are sorted by kind when the plan is
    // made.  At most 64 of them: one wavefront per kind (a wavefront takes the branches of its lanes one after the other — a round of loads
    // each — and every wavefront that loads at all costs its load instructions: tools/lab/wg_handoff.hip), a pair's follower next to its
    // leader.  More: consecutive runs of ceil(W / 15) records, a run per wavefront (mostly of one kind, all fifteen busy).
    auto place = [&](int p, int64_t from, int64_t W) {
        const int4 zero = make_int4(0, 0, 0, 0);
        for (int c = 0; c < kRunItemWaves; c++) rb[p][c * 64 + lane][0] = zero;      // (kind 0: nothing to do)
        if (W <= 64) {
            int4 a = zero, b = zero;
            if (lane < W) { a = flat4[2 * (from + lane)]; b = flat4[2 * (from + lane) + 1]; }
            const unsigned long long starts = __ballot(lane < W && (a.x & kFlatGroupStart));
            const bool ok = by_kind && (starts & 1ull) && __popcll(starts) <= kRunItemWaves;
            const int c = by_kind ? (int)((W + kRunItemWaves - 1) / kRunItemWaves) : 0;      // (0: the old dealing, record j in lane j / 15 of wavefront j mod 15)
            int slot = c ? (lane / c) * 64 + lane % c : (lane % kRunItemWaves) * 64 + lane / kRunItemWaves;
            if (ok) {
                const unsigned long long below = starts & ((2ull << lane) - 1ull);
                slot = (__popcll(below) - 1) * 64 + (lane - (63 - __clzll((long long)below)));
            }
            if (lane < W) { rb[p][slot][0] = a; rb[p][slot][1] = b; }
            if (lane == 0) grouped_s[p] = ok ? 1 : c << 1;
        } else {
            const int c = by_kind ? (int)((W + kRunItemWaves - 1) / kRunItemWaves) : 0;
            for (int64_t j = lane; j < W; j += 64) {
                const int slot = c ? (int)(j / c) * 64 + (int)(j % c) : (int)(j % kRunItemWaves) * 64 + (int)(j / kRunItemWaves);
                rb[p][slot][0] = flat4[2 * (from + j)]; rb[p][slot][1] = flat4[2 * (from + j) + 1];
            }
            if (lane == 0) grouped_s[p] = c << 1;
        }
    };
    int64_t lo = stage_off[s0], hi = stage_off[s0 + 1], nhi = s0 + 2 <= s1 ? stage_off[s0 + 2] : hi;
    if (loader) place(0, lo, hi - lo);
    __syncthreads();
    auto record = [&](int p, int slot) {
        const int4 a = rb[p][slot][0], b = rb[p][slot][1];
        FlatRec r;
        r.k = a.x; r.dst = a.y; r.v = a.z; r.s[0] = a.w; r.s[1] = b.x; r.s[2] = b.y; r.s[3] = b.z; r.s[4] = b.w;
        return r;
    };
    for (int st = s0; st < s1; st++) {
        const int p = (st - s0) & 1;
        const int64_t nlo = hi, nnhi = st + 3 <= s1 ? stage_off[st + 3] : nhi;
        if (loader) {
            if (st + 1 < s1) place(p ^ 1, nlo, nhi - nlo);
        } else {
            const FlatRec cur = record(p, threadIdx.x);
            if ((cur.k & 0x7f) && !(cur.k & kRecFollows)) {
                if (cur.k & kRecLeads) {
                    // the follower: the next slot of the same wavefront (a wavefront per kind), or where record j + 1 was put
                    const int g = grouped_s[p], c = g >> 1;
                    const int j1 = c ? wave * c + lane + 1 : lane * kRunItemWaves + wave + 1;
                    const int fs = g == 1 ? (int)threadIdx.x + 1 : c ? (j1 / c) * 64 + j1 % c : (j1 % kRunItemWaves) * 64 + j1 / kRunItemWaves;
                    flat_pair<MODE>(cur, record(p, fs), rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, true);
                } else flat_item<MODE>(cur, rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        lo = nlo; hi = nhi; nhi = nnhi;
    }
}

int flat_run_max() { return kFlatRunMax; }

// A list item of thousands of sources (cx_refsched.h: kWideList — the flat product a mean-field wiring makes of a precision's marginal):
// a strided partial sum per thread, the partials folded in a fixed tree (wavefront shuffles, then shared memory), the workgroups' partials
// in order: deterministic, and a different association than the reference's left fold — a rounding-level difference.
// grid (items, kWideParts): part p of an item sums the sources p, p + kWideParts, ... of its list in strides of the workgroup — a list of 10^6
// sources is 64 workgroups' work, not one's (0.83 ms for one workgroup) — and leaves ONE partial; k_wide_finish folds an item's partials
// in order and stores the result
constexpr int kWideParts = 64;
__global__ __launch_bounds__(1024) void k_wide_sum(const int32_t *__restrict__ rec, const int32_t *__restrict__ list, const double2 *__restrict__ f2v,
                                                   const double2 *__restrict__ prod, double2 *__restrict__ partial) {
    __shared__ double2 part[16];
    const int32_t *r = rec + 5 * (int64_t)blockIdx.x;
    const int lo = r[3], hi = r[4];
    double2 acc = zero2();
    for (int j = blockIdx.y * 1024 + threadIdx.x; j < hi; j += 1024 * kWideParts) { const int s = list[lo + j]; acc = add2(acc, s >= 0 ? f2v[s] : prod[~s]); }
    for (int off = 32; off > 0; off >>= 1) { acc.x += __shfl_down(acc.x, off, 64); acc.y += __shfl_down(acc.y, off, 64); }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 t = part[0];
        for (int w = 1; w < 16; w++) t = add2(t, part[w]);
        partial[(int64_t)blockIdx.x * kWideParts + blockIdx.y] = t;
    }
}
__global__ __launch_bounds__(64) void k_wide_finish(const int32_t *__restrict__ rec, int n, const double2 *__restrict__ partial, double2 *__restrict__ v2f,
                                                    double2 *__restrict__ marg, int nat_marg, double2 *__restrict__ prod) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int32_t *r = rec + 5 * (int64_t)i;
    const int k = r[0], idx = r[1], v = r[2];
    double2 t = partial[(int64_t)i * kWideParts];
    for (int p = 1; p < kWideParts; p++) t = add2(t, partial[(int64_t)i * kWideParts + p]);
    if (k == kItemSumToMarginal) marg[v] = nat_marg ? t : to_moment(t);
    else if (k == kItemSumToGammaMarginal) marg[v] = make_double2(t.x + 1.0, 1.0 / t.y);
    else if (!__builtin_isnan(t.y)) { if (k == kItemSumToFactor) v2f[idx] = t; else prod[idx] = t; }
}

// A batch of at most kSmallBatch items travels IN the kernel arguments: no staging copy, nothing for the host to wait for before
// it reuses its buffer — the launch is all a per-signal `process!` or a wavefront of a few signals costs.  The records are the
// first parameter, i.e. the start of the kernarg segment, which every thread reads like any other constant memory.
template <int MODE>
__global__ __launch_bounds__(64) void k_batch_small(SmallBatch recs, int n, const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg,
                                                    const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner,
                                                    const double *__restrict__ q, const double *__restrict__ pa, const double *__restrict__ pb,
                                                    double2 *__restrict__ f2v, double2 *__restrict__ v2f, double2 *__restrict__ marg, int nat_marg,
                                                    double2 *__restrict__ prod, double *__restrict__ joint) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const __attribute__((address_space(4))) int32_t *rec = (const __attribute__((address_space(4))) int32_t *)__builtin_amdgcn_kernarg_segment_ptr();
    batch_item<MODE>(rec[5 * i], rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg,
                     nat_marg, prod, joint, KaryTab{nullptr, nullptr, nullptr, nullptr});
    (void)recs;
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_batch(cx_handle *h, const int32_t *d_rec, int64_t n) {
    if (n == 0) return;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    prof_begin(h, CX_KERNEL_BATCH);
    const int mode = rule_mode(h), nat = h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0;
    const KaryTab kt{h->d_kary_slot, h->d_kary_coef, h->d_kary_qb, h->d_ref_list};
#define CX_B(M, PA, PB) hipLaunchKernelGGL(k_batch<M>, dim3(nb), dim3(kBlock), 0, h->stream, n, d_rec, h->d_vbase, h->d_var_deg, h->d_vinfo, \
                                           h->d_partner, h->d_q, PA, PB, h->d_f2v, h->d_v2f, h->d_marg, nat, h->d_prod, h->d_joint, kt)
    if (mode == kRuleLinear) CX_B(kRuleLinear, h->d_a, h->d_b);
    else if (mode == kRuleBernoulli) CX_B(kRuleBernoulli, (const double *)nullptr, (const double *)nullptr);
    else CX_B(kRuleAdditive, (const double *)nullptr, (const double *)nullptr);
#undef CX_B
    prof_end(h);
}

// stages [s0, s1) of a device-resident stage table, each of at most kRunBlock items, in one launch (cx_api_sweep.hip: the tree schedule)
void launch_batch_run(cx_handle *h, const int32_t *d_rec, const int64_t *d_stage_off, int s0, int s1) {
    if (s1 <= s0) return;
    const int mode = rule_mode(h), nat = h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0;
    const KaryTab kt{h->d_kary_slot, h->d_kary_coef, h->d_kary_qb, h->d_ref_list};
    static unsigned long long *d_ticks = [] {      // CX_RUN_TIME=1: printed when the process ends
        unsigned long long *p = nullptr;
        const char *e = std::getenv("CX_RUN_TIME");
        if (e && e[0] == '1' && hipMalloc(&p, 64) == hipSuccess) {
            (void)hipMemset(p, 0, 64);
            static unsigned long long *keep; keep = p;
            std::atexit([] {
                unsigned long long t[4] = {0, 0, 0, 0};
                if (hipMemcpy(t, keep, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess && t[3])
                    std::fprintf(stderr, "[k_batch_run %llu stages] us per stage: items %.3f release %.3f barrier %.3f\n", t[3], t[0] * 0.01 / t[3], t[1] * 0.01 / t[3], t[2] * 0.01 / t[3]);
            });
        }
        return p;
    }();
#define CX_B(M, PA, PB) hipLaunchKernelGGL(k_batch_run<M>, dim3(1), dim3(kRunBlock), 0, h->stream, d_stage_off, s0, s1, d_rec, h->d_vbase, h->d_var_deg, h->d_vinfo, \
                                           h->d_partner, h->d_q, PA, PB, h->d_f2v, h->d_v2f, h->d_marg, nat, h->d_prod, h->d_joint, kt, d_ticks)
    if (mode == kRuleLinear) CX_B(kRuleLinear, h->d_a, h->d_b);
    else if (mode == kRuleBernoulli) CX_B(kRuleBernoulli, (const double *)nullptr, (const double *)nullptr);
    else CX_B(kRuleAdditive, (const double *)nullptr, (const double *)nullptr);
#undef CX_B
}

// the same run on the plan's flat records (cx_api_ref.hip: flat_records)
void launch_flat_run(cx_handle *h, const int32_t *d_flat, const int32_t *d_rec, const int64_t *d_stage_off, int s0, int s1) {
    if (s1 <= s0) return;
    static const int by_kind = [] { const char *e = std::getenv("CX_FLAT_RUN_BY_KIND"); return e && e[0] == '0' ? 0 : 1; }();      // (A/B: 0 = every stage dealt)
    const int mode = rule_mode(h), nat = h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0;
    const KaryTab kt{h->d_kary_slot, h->d_kary_coef, h->d_kary_qb, h->d_ref_list};
#define CX_B(M, PA, PB) hipLaunchKernelGGL(k_flat_run<M>, dim3(1), dim3(kRunBlock), 0, h->stream, d_stage_off, s0, s1, d_flat, d_rec, h->d_vbase, h->d_var_deg, h->d_vinfo, \
                                           h->d_partner, h->d_q, PA, PB, h->d_f2v, h->d_v2f, h->d_marg, nat, h->d_prod, h->d_joint, kt, by_kind)
    if (mode == kRuleLinear) CX_B(kRuleLinear, h->d_a, h->d_b);
    else if (mode == kRuleBernoulli) CX_B(kRuleBernoulli, (const double *)nullptr, (const double *)nullptr);
    else CX_B(kRuleAdditive, (const double *)nullptr, (const double *)nullptr);
#undef CX_B
}

// every stage of a reference-order plan in ONE launch of an XCD-resident cluster; d_ctl: 512 bytes the launch may scribble on (zeroed here)
void launch_ref_cluster(cx_handle *h, void *d_ctl, int n_workgroups, const int32_t *d_flat, const int32_t *d_rec, const int64_t *d_stage_off, int n_stages) {
    if (n_stages <= 0) return;
    (void)hipMemsetAsync(d_ctl, 0, sizeof(ClusterCtl), h->stream);
    const int mode = rule_mode(h), nat = h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0;
    const KaryTab kt{h->d_kary_slot, h->d_kary_coef, h->d_kary_qb, h->d_ref_list};
    // bit 0: CX_REF_CLUSTER_DRY=1; bits 1..: CX_REF_CLUSTER_HELP = 0 no helpers, 1 helpers load records, 2 records and source lines, 3 (default) those and the rules' constants, four records in flight per thread
    static const int dry = [] {
        const char *e = std::getenv("CX_REF_CLUSTER_DRY"), *hp = std::getenv("CX_REF_CLUSTER_HELP");
        const char *ah = std::getenv("CX_REF_CLUSTER_AHEAD"), *mb = std::getenv("CX_REF_CLUSTER_MEMBERS");      // A/B: stages the helpers run ahead, member workgroups
        const char *fw = std::getenv("CX_REF_PAIR_FWD"), *tm = std::getenv("CX_REF_CLUSTER_TIME");
        const char *rl = std::getenv("CX_REF_CLUSTER_RELEASE");      // 0: a follower always waits for its leader's store (A/B); 1: member 0's clock
        return ((e && (e[0] == '1' || e[0] == '2')) ? 1 : 0) | ((hp ? std::max(0, std::min(3, std::atoi(hp))) : 3) << 1) | ((e && e[0] == '2') ? 8 : 0) | ((fw && fw[0] == '0') ? 16 : 0) | ((tm && tm[0] == '1') ? 32 : 0) |
               ((ah ? std::max(0, std::min(255, std::atoi(ah))) : 0) << 8) | ((mb ? std::max(0, std::min(255, std::atoi(mb))) : 0) << 16) | ((rl && rl[0] == '0') ? 0 : (1 << 27));
    }();
    // the bound of every wait in ticks of the 100 MHz clock (read per launch: a test shortens it), and the fault-injection stage (1-based, 0: none)
    double ms = 1500.0;
    if (const char *e = std::getenv("CX_REF_CLUSTER_TIMEOUT_MS")) ms = std::max(1.0, std::min(20000.0, std::atof(e)));
    const unsigned long long wait_limit = (unsigned long long)(ms * 1e5);
    int fault_stage = 0;
    if (const char *e = std::getenv("CX_REF_CLUSTER_FAULT")) fault_stage = std::max(0, std::atoi(e));
#define CX_B(M, PA, PB) hipLaunchKernelGGL(k_ref_cluster<M>, dim3(n_workgroups), dim3(kClusterBlock), 0, h->stream, (ClusterCtl *)d_ctl, (unsigned)n_workgroups, d_stage_off, n_stages, \
                                           d_flat, d_rec, h->d_vbase, h->d_var_deg, h->d_vinfo, h->d_partner, h->d_q, PA, PB, h->d_f2v, h->d_v2f, h->d_marg, nat, h->d_prod, h->d_joint, kt, dry, \
                                           wait_limit, fault_stage)
    if (mode == kRuleLinear) CX_B(kRuleLinear, h->d_a, h->d_b);
    else if (mode == kRuleBernoulli) CX_B(kRuleBernoulli, (const double *)nullptr, (const double *)nullptr);
    else CX_B(kRuleAdditive, (const double *)nullptr, (const double *)nullptr);
#undef CX_B
}

// d_partial: n x 64 pairs of scratch (cx_api_ref.hip keeps one per plan that has wide items)
void launch_wide_sum(cx_handle *h, const int32_t *d_rec, int64_t n, void *d_partial) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_wide_sum, dim3((unsigned)n, kWideParts), dim3(1024), 0, h->stream, d_rec, h->d_ref_list, h->d_f2v, h->d_prod, (double2 *)d_partial);
    hipLaunchKernelGGL(k_wide_finish, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, d_rec, (int)n, (const double2 *)d_partial, h->d_v2f, h->d_marg,
                       h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0, h->d_prod);
}

void launch_batch_small(cx_handle *h, const SmallBatch &recs, int n) {
    if (n == 0) return;
    prof_begin(h, CX_KERNEL_BATCH);
    const int mode = rule_mode(h), nat = h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0;
#define CX_B(M, PA, PB) hipLaunchKernelGGL(k_batch_small<M>, dim3(1), dim3(64), 0, h->stream, recs, n, h->d_vbase, h->d_var_deg, h->d_vinfo, \
                                           h->d_partner, h->d_q, PA, PB, h->d_f2v, h->d_v2f, h->d_marg, nat, h->d_prod, h->d_joint)
    if (mode == kRuleLinear) CX_B(kRuleLinear, h->d_a, h->d_b);
    else if (mode == kRuleBernoulli) CX_B(kRuleBernoulli, (const double *)nullptr, (const double *)nullptr);
    else CX_B(kRuleAdditive, (const double *)nullptr, (const double *)nullptr);
#undef CX_B
    prof_end(h);
}

}  // namespace cx
