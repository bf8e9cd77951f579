// cx_kernels.hip — gfx950 kernels of the scalar-Gaussian sum-product sweep.
//
// What they replace (reference = Cortex.jl v0.3.0): one `process!` → `compute!` → user rule →
// `set_value!` round trip per directed message (src/inference_engine.jl:479-509, src/signal.jl:392-410)
// with the Gaussian rules of test/inference_engine_tests.jl:385-432 and the product of
// test/runtests.jl:40-46.  Here one launch updates every message of the graph.
//
// Storage.  Messages live in NATURAL form m = (xi, w) = (mean/variance, 1/variance), one 16-byte double2
// per directed message.  In natural form the reference's `product` is a plain sum and its additive-Gaussian
// factor rule N(m, v + q) is  s = 1/(1 + q w);  (xi, w) <- (xi s, w s): one reciprocal per message instead of
// the five divisions of every moment-form `product`.  UndefValue() is NaN and propagates by itself: an output
// is defined iff every dependency is (the "pending" criterion of src/signal.jl:668-730 on first computation);
// NaN outputs are never stored, so a signal that is not pending keeps its value as in the reference.
// A point-mass datum y (the `Real` branch, test/inference_engine_tests.jl:424) is stored as (y, +inf).
//
// Layout.  Sliced ELLPACK (SELL-256): variables in ascending id order, 256 per slice = one workgroup; the
// k-th incoming message (k = rank of the factor id among the variable's neighbours, ascending) of variable v
// in slice s sits at slot  slice_off[s] + k*256 + (v & 255).  Lane <-> variable, so every load and store of
// the sweep is a unit-stride 16 B/lane access (1 KiB per wave instruction), with no index needed to find a
// variable's messages; the only index is partner[slot] (where the factor's other edge lives).  On grids and
// chains neighbouring variables have neighbouring partners, so the partner scatter is unit-stride too.
// Variables of degree > 8 live in a CSR tail and are handled by wave-per-variable scans.
//
// Roofline: HBM bytes (≈12 flop per 32 payload bytes).

#include <cstdlib>

#include "cx_internal.h"
#include "cx_kary_core.h"

namespace cx {

// blockIdx -> slice.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, "Workgroup
// dispatch"), each with a private 4 MiB L2.  XCD x takes the x-th contiguous run of slices, so the lines a
// workgroup scatters into (its grid row ± 1) are lines its own XCD's L2 is streaming.  Bijective for any
// grid size; speed only, never correctness.  (Measured on C4 after the SELL layout: within ±1 % of the
// identity mapping — the scatter targets are already close; kept because it never loses.)
__device__ __forceinline__ int xcd_slab(int b, int nb) {
    int xcd = b & 7, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Streaming accesses.  Every factor→variable message is read exactly once per sweep and the marginals are written once:
// issued as nontemporal (`nt`) they do not displace the index arrays (partner, q: 120 MB on the 10M-edge grid) from the
// 256 MiB Infinity Cache nor the lines of the partner scatter from L2.  Measured on C4 (interleaved A/B, same process
// flags): 86.3 -> 67.5 us per sweep with nt message loads, -> 62.5 us with nt marginal stores as well; nt on the scatter
// stores (+2 us), on q (+5 us) or on partner (+4 us) is slower — those lines are re-touched.
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_stream(const double2 *p) {
    const d2v v = __builtin_nontemporal_load((const d2v *)p);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void store_stream(double2 *p, double2 v) {
    d2v t;
    t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, (d2v *)p);
}

__device__ __forceinline__ double2 add2(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 zero2() { return make_double2(0.0, 0.0); }
__device__ __forceinline__ double2 nan2() { return make_double2(__builtin_nan(""), __builtin_nan("")); }

// factor→variable rule with the RECEIVING edge's effective parameters (a, b, q):
//   moment form:  N(a m + b, a² v + q)        [a=1, b=0: test/inference_engine_tests.jl:426-427]
//   natural form: s = 1/(a² + q w);  w' = w s;  xi' = (a xi + b w) s
//   point mass y: N(a y + b, q)               [:424-425]
//   MODE 2 (CX_FAMILY_NATURAL2, CX_FACTOR_BERNOULLI): the other edge carries an observed Bool r as a point mass; the message
//   is Beta(1 + r, 2 - r) (test/inference_engine_tests.jl:256-258) = natural parameters (r, 1 - r).  Without a datum the
//   reference's rule is error("Unreachable reached"): the output stays undefined.
constexpr int kRuleAdditive = 0, kRuleLinear = 1, kRuleBernoulli = 2;
template <int MODE>
__device__ __forceinline__ double2 factor_rule(double2 m, double q, double a, double b) {
    constexpr bool LINEAR = MODE == kRuleLinear;
    double2 o;
    if (MODE == kRuleBernoulli) {
        if (m.y == __builtin_inf()) { o.x = m.x; o.y = 1.0 - m.x; }
        else o = make_double2(__builtin_nan(""), __builtin_nan(""));
        return o;
    }
    if (m.y == __builtin_inf()) {
        double mean = LINEAR ? (a * m.x + b) : m.x;
        o.y = 1.0 / q;
        o.x = mean * o.y;
    } else {
        double s = 1.0 / ((LINEAR ? a * a : 1.0) + q * m.y);
        o.y = m.y * s;
        o.x = (LINEAR ? (a * m.x + b * m.y) : m.x) * s;
    }
    return o;
}

// damping (cx_set_damping): new = (1 - lambda) rule + lambda old, in natural form; an old value that is undefined does not damp
__device__ __forceinline__ double2 damped(double2 r, double2 old, double lam) {
    if (__builtin_isnan(old.y)) return r;
    return make_double2((1.0 - lam) * r.x + lam * old.x, (1.0 - lam) * r.y + lam * old.y);
}

__device__ __forceinline__ double2 to_moment(double2 nat) {
    double var = 1.0 / nat.y;
    return make_double2(nat.x * var, var);
}

// store one fresh variable→factor message and/or push it through its factor to the partner slot
// PACK (the fused sweep of graphs that allow it; 104 of the 427 MB a C4 sweep moves were indices and parameters):
//   bit 0  the partner comes as a 16-bit difference to the slot (kNoPartner16: none) — graphs whose partners are all within 32 k slots
//   bit 1  q is read at the LOWER slot of the factor's two: additive factors have one q for both directions, so only the lines of
//          the lower slots are ever fetched (a line holds the k-th neighbour of 16 consecutive variables: all lower or all higher on
//          grid-like graphs)
constexpr int kPackPartner16 = 1, kPackQLow = 2;
constexpr int16_t kNoPartner16 = -32768;
template <int MODE, bool STORE_V2F, bool PUSH, int PACK = 0, bool DAMP = false>
__device__ __forceinline__ void emit(int slot, double2 o, const int32_t *__restrict__ partner, const double *__restrict__ sq,
                                     const double *__restrict__ sa, const double *__restrict__ sb, double2 *__restrict__ f2v_out,
                                     double2 *__restrict__ v2f, int nt_out = 0, const int16_t *__restrict__ partner16 = nullptr,
                                     const double2 *__restrict__ prev = nullptr, double lam = 0.0) {
    if (__builtin_isnan(o.y)) return;  // a dependency is undefined: the signal is not pending, keep stored values
    if (STORE_V2F) v2f[slot] = o;
    if (PUSH) {
        int p;
        if (PACK & kPackPartner16) { const int dlt = partner16[slot]; p = dlt == kNoPartner16 ? -1 : slot + dlt; }
        else p = partner[slot];
        if (p >= 0) {
            double2 r = factor_rule<MODE>(o, sq[(PACK & kPackQLow) ? (p < slot ? p : slot) : slot], MODE == kRuleLinear ? sa[slot] : 1.0, MODE == kRuleLinear ? sb[slot] : 0.0);
            if (DAMP && !__builtin_isnan(r.y)) r = damped(r, prev[p], lam);      // the message this one replaces: the receiving slot in the sweep's input buffer
            if (MODE != kRuleBernoulli || !__builtin_isnan(r.y)) { if (nt_out) store_stream(&f2v_out[p], r); else f2v_out[p] = r; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The sweep kernel.  One thread = one variable of degree <= 8.
//   variable→factor  compute_message_to_factor! = reduce(product, others)   test/inference_engine_tests.jl:405-413
//                    dependency set: dependencies.jl:60-88 ("all other factors' messages")
//   marginal         compute_individual_marginal! = reduce(product, all)     test/inference_engine_tests.jl:385-393
//   factor→variable  compute_message_to_variable!                            test/inference_engine_tests.jl:415-432
//                    dependency: dependencies.jl:17-31 (the other variable's message to the factor)
// leave-one-out sums: out[k] = (in[0] + … + in[k-1]) + (in[deg-1] + … + in[k+1]); all register indices are
// compile-time constants (runtime-indexed arrays would go to scratch).
// PUSH: send each fresh variable→factor message straight through its factor into the partner's slot of the
// OTHER factor→variable buffer (Jacobi double buffering): the whole sweep is this one launch.
// !PUSH: phase A of the two-phase flooding schedule (stores variable→factor messages only).
// ------------------------------------------------------------------------------------------------
// MAXW: the widest slice of the graph rounded up to 5 or 8 (a 2-D grid with one observation per variable has degree 5): the incoming and
// the outgoing messages of a variable live in 2 x MAXW register pairs, and with five instead of eight the kernel fits 64 registers —
// eight waves per SIMD instead of six.
// nt: bit 0 nontemporal scatter stores, bit 1 nontemporal message loads, bit 2 nontemporal marginal stores (chosen by footprint: launchers)
// (Round 5, measured on the 1/8 strip of C4 and removed again: workgroups of 128 / 64 threads — a quarter-slice each, to spread a launch
// of 4.5 workgroups per compute unit more evenly: 10.15 / 10.98 us against 10.02; the partner differences and q loaded at the top of
// the kernel beside the messages, one or two round trips to memory fewer per wave: 11.1 - 11.6 us against 10.0, and 63 - 66 against 56
// on the whole grid.  profiles/r05_strip.md.)
constexpr int kNtOut = 1, kNtIn = 2, kNtMarg = 4;
template <int MODE, bool STORE_V2F, bool PUSH, int PACK = 0, int MAXW = kSmallDeg, bool DAMP = false>
__global__ __launch_bounds__(kBlock) void k_sweep(int nv, const int32_t *__restrict__ slice_off, const uint8_t *__restrict__ vinfo,
                                                  const int32_t *__restrict__ partner, const double *__restrict__ sq,
                                                  const double *__restrict__ sa, const double *__restrict__ sb,
                                                  const double2 *__restrict__ f2v_in, double2 *__restrict__ f2v_out,
                                                  double2 *__restrict__ v2f, double2 *__restrict__ marg, int write_marg,
                                                  int skip_ghosts, int nt, int slice_lo, int slice_hi, int excl_lo, int excl_hi,
                                                  const int16_t *__restrict__ partner16, double lam) {
    // the slice -> XCD mapping stays the same from sweep to sweep (a strip's messages largely live in the L2s between sweeps:
    // launching only the active slice range re-deals the slices over the XCDs and measured 10 % SLOWER); idle slices exit here.
    // [excl_lo, excl_hi]: slices another launch of the same sweep covers (the owned interior, run beside the halo exchange)
    const int s = xcd_slab(blockIdx.x, gridDim.x);
    if (s < slice_lo || s > slice_hi || (s >= excl_lo && s <= excl_hi)) return;
    const int tid = threadIdx.x;
    const int nt_out = nt & kNtOut;
    const int v = (s << kSliceShift) + tid;
    const int off = slice_off[s];
    const int W = (slice_off[s + 1] - off) >> kSliceShift;  // slice width: workgroup-uniform
    if (v >= nv) return;
    const int info = vinfo[v];
    const int deg = info & kDegMask;
    if (deg == kBigDeg) return;                       // wave-per-variable kernels own it
    if (skip_ghosts && (info & kGhost)) return;       // pushed by cx_sweep_end once the halo has arrived
    const int base = off + tid;

    double2 in[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        in[k] = zero2();
        if (k < W) {  // uniform branch
            double2 x = (nt & kNtIn) ? load_stream(&f2v_in[base + k * kBlock]) : f2v_in[base + k * kBlock];
            if (k < deg) in[k] = x;
        }
    }
    double2 out[MAXW];
    double2 acc = zero2();
#pragma unroll
    for (int k = 0; k < MAXW; k++) { out[k] = acc; acc = add2(acc, in[k]); }
    const double2 total = acc;
    acc = zero2();
#pragma unroll
    for (int k = MAXW - 1; k >= 0; k--) { out[k] = add2(out[k], acc); acc = add2(acc, in[k]); }

    if (write_marg) {                                    // 2: natural-parameter marginals
        const double2 mg = write_marg == 2 ? total : to_moment(total);
        if (nt & kNtMarg) store_stream(&marg[v], mg); else marg[v] = mg;
    }

    // a variable with <2 factors has no dependencies on its message to the factor (dependencies.jl:48-55): never
    // computed; observed variables keep the data the caller set.  Their stored message still feeds the factor.
    const bool fixed = (deg < 2) || (info & (kClamped | kGhost));
    if (!fixed) {
#pragma unroll
        for (int k = 0; k < MAXW; k++)
            if (k < deg) emit<MODE, STORE_V2F, PUSH, PACK, DAMP>(base + k * kBlock, out[k], partner, sq, sa, sb, f2v_out, v2f, nt_out, partner16, f2v_in, lam);
    } else if (PUSH) {
        // separate path (not a select on the message) so that out[] never has its address taken
#pragma unroll
        for (int k = 0; k < MAXW; k++)
            if (k < deg) emit<MODE, false, true, PACK, DAMP>(base + k * kBlock, v2f[base + k * kBlock], partner, sq, sa, sb, f2v_out, v2f, 0, partner16, f2v_in, lam);
    }
}

// ------------------------------------------------------------------------------------------------
// Phase B of the two-phase flooding schedule: factor → variable by pulling from the partner slot.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_factor_to_var(int nslots, const int32_t *__restrict__ partner,
                                                          const double *__restrict__ q, const double *__restrict__ pa,
                                                          const double *__restrict__ pb, const double2 *__restrict__ v2f,
                                                          double2 *__restrict__ f2v, double lam) {
    const int e = xcd_slab(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
    if (e >= nslots) return;
    const int p = partner[e];
    if (p < 0) return;
    const double2 m = v2f[p];
    if (__builtin_isnan(m.y)) return;  // dependency not computed: not pending, keep the old value
    double2 r = factor_rule<MODE>(m, q[e], MODE == kRuleLinear ? pa[e] : 1.0, MODE == kRuleLinear ? pb[e] : 0.0);
    if (lam != 0.0 && !__builtin_isnan(r.y)) r = damped(r, f2v[e], lam);
    if (MODE != kRuleBernoulli || !__builtin_isnan(r.y)) f2v[e] = r;
}

// ------------------------------------------------------------------------------------------------
// Big variables (degree > 8): one wave per variable, exclusive prefix + exclusive suffix by wave scans — the
// device analogue of the reference's segment tree of ProductOfMessages intermediates (dependencies.jl:90-173):
// "product of all but me" in O(deg) work.  Their slots are contiguous (CSR tail of the slot space).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 wave_inclusive_scan(double2 x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double ox = __shfl_up(x.x, d, 64), oy = __shfl_up(x.y, d, 64);
        if (lane >= d) { x.x += ox; x.y += oy; }
    }
    return x;
}

__global__ __launch_bounds__(kBlock) void k_big_var_to_factor(const int32_t *__restrict__ big, int nbig,
                                                              const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg,
                                                              const uint8_t *__restrict__ vinfo, const double2 *__restrict__ f2v,
                                                              double2 *__restrict__ v2f, double2 *__restrict__ tmp, int big_start,
                                                              double2 *__restrict__ marg, int write_marg) {
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (w >= nbig) return;
    const int v = big[w];
    const int s = vbase[v], t = s + vdeg[v];
    double2 *pre = tmp - big_start;  // prefix scratch, indexed by slot (a NaN result must not clobber v2f)
    // forward: exclusive prefix of every element
    double2 carry = zero2();
    for (int base = s; base < t; base += 64) {
        const int i = base + lane;
        double2 x = (i < t) ? f2v[i] : zero2();
        double2 inc = wave_inclusive_scan(x, lane);
        double ex = __shfl_up(inc.x, 1, 64), ey = __shfl_up(inc.y, 1, 64);
        double2 exc = (lane == 0) ? zero2() : make_double2(ex, ey);
        if (i < t) pre[i] = add2(carry, exc);
        carry = add2(carry, make_double2(__shfl(inc.x, 63, 64), __shfl(inc.y, 63, 64)));
    }
    if (write_marg && lane == 0) marg[v] = write_marg == 2 ? carry : to_moment(carry);
    if (vinfo[v] & (kClamped | kGhost)) return;
    // backward: exclusive suffix; lanes walk each chunk from its end
    const int nchunk = (t - s + 63) >> 6;
    carry = zero2();
    for (int c = nchunk - 1; c >= 0; c--) {
        const int i = s + c * 64 + (63 - lane);
        double2 x = (i < t) ? f2v[i] : zero2();
        double2 inc = wave_inclusive_scan(x, lane);
        double ex = __shfl_up(inc.x, 1, 64), ey = __shfl_up(inc.y, 1, 64);
        double2 exc = (lane == 0) ? zero2() : make_double2(ex, ey);
        if (i < t) {
            double2 o = add2(pre[i], add2(carry, exc));
            if (!__builtin_isnan(o.y)) v2f[i] = o;
        }
        carry = add2(carry, make_double2(__shfl(inc.x, 63, 64), __shfl(inc.y, 63, 64)));
    }
}

// push the stored variable→factor messages of a list of slots through their factors
// (big variables in the fused schedule; ghost variables once the halo has arrived)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_push_slots(const int32_t *__restrict__ slots, int64_t n, const int32_t *__restrict__ partner,
                                                       const double *__restrict__ sq, const double *__restrict__ sa,
                                                       const double *__restrict__ sb, const double2 *__restrict__ v2f,
                                                       double2 *__restrict__ f2v_out, const double2 *__restrict__ prev, double lam) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int e = slots[i], p = partner[e];
    if (p < 0) return;
    const double2 o = v2f[e];
    if (__builtin_isnan(o.y)) return;
    double2 r = factor_rule<MODE>(o, sq[e], MODE == kRuleLinear ? sa[e] : 1.0, MODE == kRuleLinear ? sb[e] : 0.0);
    if (lam != 0.0 && prev && !__builtin_isnan(r.y)) r = damped(r, prev[p], lam);
    if (MODE != kRuleBernoulli || !__builtin_isnan(r.y)) f2v_out[p] = r;
}

// variable→factor for one slot of variable v (sequential sums in the order of the sweep kernel)
template <bool COH = false>
__device__ __forceinline__ void m2f_one(int slot, int v, const int32_t *vbase, const int32_t *vdeg, const uint8_t *vinfo,
                                        const double2 *f2v, double2 *v2f, double2 *fwd = nullptr) {
    const int info = vinfo[v];
    const int deg = vdeg[v];
    if (deg < 2 || (info & (kClamped | kGhost))) return;
    const int stride = ((info & kDegMask) == kBigDeg) ? 1 : kBlock;
    const int b = vbase[v];
    const int k = (slot - b) / stride;
    double2 pre = zero2(), suf = zero2();
    for (int j = 0; j < k; j++) pre = add2(pre, ld2<COH>(f2v, b + j * stride));
    for (int j = deg - 1; j > k; j--) suf = add2(suf, ld2<COH>(f2v, b + j * stride));
    const double2 o = add2(pre, suf);
    if (!__builtin_isnan(o.y)) { v2f[slot] = o; if (fwd) *fwd = o; }
}

// halo export: variable→factor of the exported slots, computed straight into the send buffer (and into v2f)
__global__ __launch_bounds__(kBlock) void k_halo_export(const int32_t *__restrict__ slots, const int32_t *__restrict__ vars, int64_t n,
                                                        const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg,
                                                        const uint8_t *__restrict__ vinfo, const double2 *__restrict__ f2v,
                                                        double2 *__restrict__ v2f, double2 *__restrict__ send) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i];
    m2f_one(slot, vars[i], vbase, vdeg, vinfo, f2v, v2f);
    send[i] = v2f[slot];   // same thread wrote it (or it keeps its stored value: fixed / not yet defined)
}

// halo import: received messages become the ghost variables' variable→factor messages and go through the cut factors
template <int MODE, bool PUSH>
__global__ __launch_bounds__(kBlock) void k_halo_import(const int32_t *__restrict__ slots, int64_t n, const double2 *__restrict__ recv,
                                                        const int32_t *__restrict__ partner, const double *__restrict__ sq,
                                                        const double *__restrict__ sa, const double *__restrict__ sb,
                                                        double2 *__restrict__ v2f, double2 *__restrict__ f2v_out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int e = slots[i];
    const double2 o = recv[i];
    v2f[e] = o;
    if (!PUSH) return;
    const int p = partner[e];
    if (p < 0 || __builtin_isnan(o.y)) return;
    const double2 r = factor_rule<MODE>(o, sq[e], MODE == kRuleLinear ? sa[e] : 1.0, MODE == kRuleLinear ? sb[e] : 0.0);
    if (MODE != kRuleBernoulli || !__builtin_isnan(r.y)) f2v_out[p] = r;
}

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
__device__ __forceinline__ void vmp_item(int k, int idx, int lo, const int32_t *__restrict__ list, double2 *__restrict__ f2v, const double2 *__restrict__ v2f,
                                         const double2 *__restrict__ marg, double *__restrict__ joint) {
    const double inf = __builtin_inf();
    if (k == kItemMfNormal) {                 // N(E[other], E[precision])                                                        (:654-664)
        const double2 a = ld2<COH>(marg, list[lo]), g = ld2<COH>(marg, list[lo + 1]);
        const double eg = g.x * g.y;
        if (!__builtin_isnan(a.x) && !__builtin_isnan(eg)) f2v[idx] = make_double2(a.x * eg, eg);
    } else if (k == kItemMfGamma) {           // Gamma(3/2, 2 / (var a + var b + (E a - E b)^2))                                  (:666-684)
        const double2 a = ld2<COH>(marg, list[lo]), b = ld2<COH>(marg, list[lo + 1]);
        const double d = a.x - b.x, spread = a.y + b.y + d * d;
        if (!__builtin_isnan(spread)) f2v[idx] = make_double2(0.5, 0.5 * spread);
    } else if (k == kItemStNormal) {          // N(mean m, 1 / (var m + 1 / E[precision])), m the other Normal variable's message    (:1004-1010)
        const double2 m = ld2<COH>(v2f, list[lo]), g = ld2<COH>(marg, list[lo + 1]);
        const double eg = g.x * g.y;
        if (__builtin_isnan(m.y) || __builtin_isnan(m.x) || __builtin_isnan(eg)) return;
        const double mean = m.y == inf ? m.x : m.x / m.y, var = m.y == inf ? 0.0 : 1.0 / m.y;
        const double w = 1.0 / (var + 1.0 / eg);
        f2v[idx] = make_double2(mean * w, w);
    } else if (k == kItemVmpJoint) {          // the 2-d Gaussian with precision [[w1 + E, -E], [-E, w2 + E]] and potential (xi1, xi2)  (:939-967)
        const double2 m1 = ld2<COH>(v2f, list[lo]), m2 = ld2<COH>(v2f, list[lo + 1]), g = ld2<COH>(marg, list[lo + 2]);
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
        const int jb = 6 * list[lo];
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
        vmp_item<COH>(k, idx, lo, kt.list, f2v, v2f, marg, joint);
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
                                                         const KaryTab kt) {
    for (int st = s0; st < s1; st++) {
        for (int64_t i = stage_off[st] + threadIdx.x; i < stage_off[st + 1]; i += kRunBlock) {      // (the tree schedule folds stages of at most kRunBlock items: one trip)
            const int k0 = rec[5 * i];
            if (k0 & kRecFollows) continue;
            if (k0 & kRecLeads) batch_pair<MODE>(rec + 5 * i, rec + 5 * i + 5, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
            else batch_item<MODE>(k0, rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
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
struct ClusterCtl { unsigned registered, members, rank_next, arrive, abort_, xcd_plus_1, pad[10]; };      // 64 B, zeroed before every launch

__device__ __forceinline__ unsigned hw_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(v));      // HW_REG_XCC_ID
    return v & 7u;
}
__device__ __forceinline__ bool cluster_wait(unsigned *p, unsigned target, unsigned *abort_) {
    for (unsigned spins = 0;; spins++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (spins > (1u << 24)) { __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }      // seconds
        if ((spins & 63u) == 63u && __hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
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
    const int info = (r.k & kFlatCheckObserved) ? vinfo[r.v] : 0;
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

template <int MODE>
__global__ __launch_bounds__(kClusterBlock) void k_ref_cluster(ClusterCtl *c, unsigned G, const int64_t *__restrict__ stage_off, int n_stages,
                                                               const int32_t *__restrict__ flat, const int32_t *__restrict__ rec, const int32_t *__restrict__ vbase,
                                                               const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner,
                                                               const double *__restrict__ q, const double *__restrict__ pa, const double *__restrict__ pb, double2 *f2v,
                                                               double2 *v2f, double2 *marg, int nat_marg, double2 *prod, double *joint, const KaryTab kt, int dry) {
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
            ok_s = cluster_wait(&c->registered, G, &c->abort_) ? 1u : 0u;
            members_s = __hip_atomic_load(&c->members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!mine || !ok_s) return;
    // Roles.  A stage is rarely wider than a few thousand items, and what a member waits for most is memory it has never touched: the plan's
    // records (hundreds of MB, read once) and values last written many stages ago.  So half of the cluster's workgroups are HELPERS: they
    // take no part in the barriers and store nothing; helper j runs ahead of the members through the stages s = j (mod H), loads their
    // records (help >= 1) and the lines of their sources (help >= 2) — into the L2 the members read from — and never lets anybody wait.
    const int help = (dry >> 1) & 3, ahead_arg = (dry >> 8) & 0xff, members_arg = (dry >> 16) & 0xff;
    const int64_t all = members_s, P = help && all >= 4 ? (members_arg && members_arg < all ? members_arg : all / 2) : all, H = all - P;
    if ((int64_t)rank_s >= P) {
        const int64_t hj = (int64_t)rank_s - P;
        const int ahead = ahead_arg ? ahead_arg : (help >= 2 ? 4 : 12);      // stages: the L2 is 4 MB, a stage's sources up to 1 MB of lines, its records 32 B an item
        unsigned sink = 0;
        for (int64_t s = 2 + hj; s < n_stages; s += H) {
            if (threadIdx.x == 0) {
                unsigned okh = 1u;
                for (unsigned spins = 0;; spins++) {
                    const int64_t cur = __hip_atomic_load(&c->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (unsigned)P;
                    if (s <= cur + ahead) break;
                    if (spins > (1u << 24) || ((spins & 63u) == 63u && __hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { okh = 0u; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                ok_s = okh;
            }
            __syncthreads();
            if (!ok_s) return;
            for (int64_t i = stage_off[s] + threadIdx.x; i < stage_off[s + 1]; i += kClusterBlock) {
                const FlatRec r = flat_load(flat, i);
                sink ^= (unsigned)r.k ^ (unsigned)r.dst;
                if (help >= 2) {
                    const int kind = r.k & 0x7f, n = (r.k >> 8) & 0xff;
                    if (kind == kFlatRule) sink ^= (unsigned)__double_as_longlong(ld2<true>(v2f, r.s[0]).y);
                    else if (kind != kFlatGeneric)
                        for (int j = 0; j < 5; j++) if (j < n) sink ^= (unsigned)__double_as_longlong((r.s[j] >= 0 ? ld2<true>(f2v, r.s[j]) : ld2<true>(prod, ~r.s[j])).y);
                }
            }
            __syncthreads();
        }
        if (sink == 0x9e3779b9u) c->pad[0] = sink;      // (keeps the loads)
        return;
    }
    const int64_t first = (int64_t)rank_s * kClusterBlock + threadIdx.x, step = P * kClusterBlock;
    int64_t lo = stage_off[0], hi = stage_off[1];
    FlatRec cur{};
    bool have = lo + first < hi;
    if (have) cur = flat_load(flat, lo + first);
    for (int st = 0; st < n_stages; st++) {
        // the next stage's bounds and this thread's first record of it: plan constants, on their way while this stage's values are loaded
        const int64_t nlo = hi, nhi = st + 1 < n_stages ? stage_off[st + 2] : hi;
        const bool nhave = st + 1 < n_stages && nlo + first < nhi;
        FlatRec nxt{};
        if (nhave && !(dry & 8)) nxt = flat_load(flat, nlo + first);      // (bit 3, CX_REF_CLUSTER_DRY=2: not even the records — the bare barriers)
        if (!(dry & 1)) {      // (CX_REF_CLUSTER_DRY=1: the plan's skeleton — records and barriers, no item — for timing what a stage costs before it computes)
            if (have && !(cur.k & kRecFollows)) {
                if (cur.k & kRecLeads) flat_pair<MODE>(cur, flat_load(flat, lo + first + 1), rec, vbase, vdeg, vinfo, partner, q, pa, pb, f2v, v2f, marg, nat_marg, prod, joint, kt, !(dry & 16));
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
        // the barrier: this thread's stores have reached the L2, the workgroup has arrived, one thread reports and waits for the others
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok_s = cluster_wait(&c->arrive, (unsigned)((st + 1) * P), &c->abort_) ? 1u : 0u;
        }
        __syncthreads();
        if (!ok_s) return;
        lo = nlo; hi = nhi; have = nhave; cur = nxt;
    }
}

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

// ------------------------------------------------------------------------------------------------ utilities
__global__ void k_scatter(double2 *__restrict__ dst, const int32_t *__restrict__ idx, const double2 *__restrict__ val, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val[i];
}

__global__ void k_gather(const double2 *__restrict__ src, const int32_t *__restrict__ idx, double2 *__restrict__ val, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) val[i] = src[idx[i]];
}

__global__ void k_seed(double2 *__restrict__ buf, int64_t n, double2 value, const int32_t *__restrict__ partner, const int32_t *__restrict__ slot_kary) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (partner[i] < 0 && !(slot_kary && slot_kary[i] >= 0)) return;  // padding slots and messages nobody computes are the caller's to set
    if (__builtin_isnan(buf[i].y)) buf[i] = value;
}

// max |Δmean|, |Δvariance| per workgroup (moment form), reduced on the host from 1024 partials
__global__ __launch_bounds__(kBlock) void k_residual(const double2 *__restrict__ cur, const double2 *__restrict__ prev, int64_t n,
                                                     double *__restrict__ out, int natural) {
    __shared__ double red[kBlock / 64];
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double2 a = cur[i], b = prev[i];
        const bool da = !__builtin_isnan(a.y), db = !__builtin_isnan(b.y);
        if (da != db) m = __builtin_inf();
        else if (da) {
            // Two snapshots of one message that are bitwise equal have not moved, whatever they hold — a zero-precision message
            // (0, 0) (moment form 0 * inf), a Beta(1 + r, 2 - r) with a zero natural parameter: both legitimate and constant.
            if (a.x == b.x && a.y == b.y) continue;
            double d;
            if (natural) {
                // any 2-parameter family in natural coordinates (CX_FAMILY_NATURAL2): the moment form below is Gaussian-only
                d = fmax(fabs(a.x - b.x), fabs(a.y - b.y));
            } else {
                const double2 ma = to_moment(a), mb = to_moment(b);
                const double dm = (ma.x == mb.x) ? 0.0 : fabs(ma.x - mb.x), dv = (ma.y == mb.y) ? 0.0 : fabs(ma.y - mb.y);   // equal infinities: no change
                d = fmax(dm, dv);
                if (dm != dm || dv != dv) d = __builtin_inf();      // a difference that is NaN (inf - inf, 0 * inf on one side only): never converged
            }
            m = (d != d) ? __builtin_inf() : fmax(m, d);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; k++) m = fmax(m, red[k]);
        out[blockIdx.x] = m;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// hipEvent pair around a launch.  Every event is a barrier packet on the queue, so with a stride > 1 only every
// stride-th launch of a kernel is bracketed and the others dispatch back to back.
static inline void prof_begin(cx_handle *h, int kernel, hipStream_t stream = nullptr) {
    if (!stream) stream = h->stream;
    h->prof_stream = stream;
    h->prof_armed = false;
    if (!h->profiling) return;
    if ((h->prof_count[kernel]++ % h->prof_stride) != 0) return;
    h->prof_armed = true;
    ProfileRec r;
    r.kernel = kernel;
    (void)hipEventCreate(&r.start);
    (void)hipEventCreate(&r.stop);
    (void)hipEventRecord(r.start, stream);
    h->recs.push_back(r);
}
static inline void prof_end(cx_handle *h) {
    if (!h->profiling || !h->prof_armed) return;
    (void)hipEventRecord(h->recs.back().stop, h->prof_stream);
}

// Scatter stores of the fused sweep.  On the 10M-edge grid plain stores win (nt: +2 us, the lines are re-touched by the
// neighbouring rows' stores).  On a graph whose written bytes FIT the L2s (a 1/8 strip: 22 MB against 8 x 4 MB) plain stores
// stay dirty in L2 until the kernel ends and are written back in its tail; CX_NT_SCATTER=1/0 forces either form, default: by size.
static inline int nt_scatter(const cx_handle *h) {
    static const int forced = [] { const char *e = std::getenv("CX_NT_SCATTER"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    if (forced >= 0) return forced;
    // measured on MI355X: 1/8 strip of the C4 grid (25 MB of message slots) 10.97 -> 9.75 us per sweep with nt stores; the whole grid
    // (160 MB) 63.4 -> 64.3 us.  The L2s hold 32 MB in all.
    return (h->nslots * (int64_t)sizeof(double2) <= (int64_t)40 << 20) ? 1 : 0;
}

// load / store policies of the fused sweep by footprint (CX_NT_FLAGS=<bits> forces: 1 scatter stores, 2 message loads, 4 marginal stores)
static inline int nt_flags(const cx_handle *h) {
    static const int forced = [] { const char *e = std::getenv("CX_NT_FLAGS"); return e ? std::atoi(e) : -1; }();
    if (forced >= 0) return forced & 7;
    // message loads: nontemporal always (C4: 86.3 -> 67.5 us; the 1/8 strip: 11.9 -> 10.0).  Marginal stores: nontemporal on the whole grid
    // (C4: 60.5 -> 55.8), plain on a strip whose buffers the caches hold (9.68 - 9.79 against 9.99 - 10.02, three runs each): the
    // inverse of the scatter stores' rule
    const int out = nt_scatter(h);
    return out | kNtIn | (out ? 0 : kNtMarg);
}

static inline int rule_mode(const cx_handle *h) {
    return h->cfg.family == CX_FAMILY_NATURAL2 ? kRuleBernoulli : (h->any_linear ? kRuleLinear : kRuleAdditive);
}

template <int LINEAR, bool STORE, bool PUSH>
static void launch_sweep_t(cx_handle *h, const double2 *f2v_in, double2 *f2v_out, bool write_marg, bool skip_ghosts) {
    const double *sq = h->any_linear ? h->d_sq : h->d_q;  // additive factors: q is symmetric in the two edges
    // deep-halo partitions: k sweeps after an exchange only the redundant layers that can still be valid are run (slice range)
    const int lo = (PUSH && h->run_nslices > 0) ? h->run_slice0 : 0, hi = (PUSH && h->run_nslices > 0) ? h->run_slice0 + h->run_nslices - 1 : (int)h->nslices;
    const int xlo = PUSH ? h->run_excl_lo : 1, xhi = PUSH ? h->run_excl_hi : 0;
    // CX_PACK=0 in the environment: the unpacked kernel (A/B); otherwise the packed form wherever the graph allows it (additive
    // Gaussian factors, every partner within 32 k slots) — bit-identical results: the same values travel, fewer bytes
    static const bool pack_on = [] { const char *e = std::getenv("CX_PACK"); return !(e && e[0] == '0'); }();
    const int nt = nt_flags(h);
#define CX_SWEEP_TAIL 0, h->stream, (int)h->nv, h->d_slice_off, h->d_vinfo, h->d_partner, sq, h->d_sa, \
                      h->d_sb, f2v_in, f2v_out, h->d_v2f, h->d_marg, write_marg ? (h->cfg.family == CX_FAMILY_NATURAL2 ? 2 : 1) : 0,             \
                      skip_ghosts ? 1 : 0, nt, lo, hi, xlo, xhi, h->d_partner16, h->damping
#define CX_SWEEP_ARGS dim3((unsigned)h->nslices), dim3(kBlock), CX_SWEEP_TAIL
    // the widest slice, once per graph (CX_MAXW8=1: the eight-message instance for every graph, A/B)
    if (h->sweep_max_w == 0) {
        int w = 1;
        for (int64_t sl = 0; sl < h->nslices; sl++) w = std::max<int>(w, (h->slice_off[sl + 1] - h->slice_off[sl]) >> kSliceShift);
        h->sweep_max_w = w;
    }
    static const bool force8 = [] { const char *e = std::getenv("CX_MAXW8"); return e && e[0] == '1'; }();
    const bool w5 = h->sweep_max_w <= 5 && !force8;
    if (PUSH && h->damping != 0.0) {      // damped sweeps: the general instance, with one more gather (the message each result replaces)
        hipLaunchKernelGGL((k_sweep<LINEAR, STORE, PUSH, 0, kSmallDeg, true>), CX_SWEEP_ARGS);
    } else if (PUSH && LINEAR == kRuleAdditive && pack_on && h->d_partner16) {
        if (w5) hipLaunchKernelGGL((k_sweep<LINEAR, STORE, PUSH, kPackPartner16 | kPackQLow, 5>), CX_SWEEP_ARGS);
        else hipLaunchKernelGGL((k_sweep<LINEAR, STORE, PUSH, kPackPartner16 | kPackQLow, kSmallDeg>), CX_SWEEP_ARGS);
    } else
        hipLaunchKernelGGL((k_sweep<LINEAR, STORE, PUSH, 0, kSmallDeg>), CX_SWEEP_ARGS);
#undef CX_SWEEP_ARGS
#undef CX_SWEEP_TAIL
}

void launch_fused(cx_handle *h, const double2 *f2v_in, double2 *f2v_out, bool write_marg, bool store_v2f, bool skip_ghosts) {
    if (h->nslices == 0) return;
    prof_begin(h, CX_KERNEL_FUSED);
    const int mode = rule_mode(h);
#define CX_F(M) do { if (store_v2f) launch_sweep_t<M, true, true>(h, f2v_in, f2v_out, write_marg, skip_ghosts); \
                     else launch_sweep_t<M, false, true>(h, f2v_in, f2v_out, write_marg, skip_ghosts); } while (0)
    if (mode == kRuleLinear) CX_F(kRuleLinear); else if (mode == kRuleBernoulli) CX_F(kRuleBernoulli); else CX_F(kRuleAdditive);
#undef CX_F
    prof_end(h);
}

void launch_var_to_factor(cx_handle *h, const double2 *f2v, bool write_marg) {
    if (h->nslices == 0) return;
    prof_begin(h, CX_KERNEL_VAR_TO_FACTOR);
    launch_sweep_t<kRuleAdditive, true, false>(h, f2v, nullptr, write_marg, false);   // no factor rule in this phase
    prof_end(h);
}

void launch_big_var_to_factor(cx_handle *h, const double2 *f2v, bool write_marg) {
    const int nbig = (int)h->big_vars.size();
    if (nbig == 0) return;
    prof_begin(h, CX_KERNEL_BIG_VAR);
    const int waves_per_block = kBlock / 64;
    const int nb = (nbig + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL(k_big_var_to_factor, dim3(nb), dim3(kBlock), 0, h->stream, h->d_big, nbig, h->d_vbase, h->d_var_deg,
                       h->d_vinfo, f2v, h->d_v2f, h->d_big_tmp, h->big_start, h->d_marg, write_marg ? (h->cfg.family == CX_FAMILY_NATURAL2 ? 2 : 1) : 0);
    prof_end(h);
}

void launch_factor_to_var(cx_handle *h, const double2 *v2f, double2 *f2v) {
    const int n = (int)h->nslots;
    if (n == 0) return;
    const int nb = (n + kBlock - 1) / kBlock;
    prof_begin(h, CX_KERNEL_FACTOR_TO_VAR);
    const int mode = rule_mode(h);
    if (mode == kRuleLinear)
        hipLaunchKernelGGL(k_factor_to_var<kRuleLinear>, dim3(nb), dim3(kBlock), 0, h->stream, n, h->d_partner, h->d_q, h->d_a, h->d_b, v2f, f2v, h->damping);
    else if (mode == kRuleBernoulli)
        hipLaunchKernelGGL(k_factor_to_var<kRuleBernoulli>, dim3(nb), dim3(kBlock), 0, h->stream, n, h->d_partner, h->d_q,
                           (const double *)nullptr, (const double *)nullptr, v2f, f2v, h->damping);
    else
        hipLaunchKernelGGL(k_factor_to_var<kRuleAdditive>, dim3(nb), dim3(kBlock), 0, h->stream, n, h->d_partner, h->d_q,
                           (const double *)nullptr, (const double *)nullptr, v2f, f2v, h->damping);
    prof_end(h);
}

void launch_push_slots(cx_handle *h, const int32_t *d_slots, int64_t n, double2 *f2v_out, int kernel_id) {
    // the big variables' part of a fused sweep (f2v_out is the sweep's output buffer, d_f2v its input): damped like the rest of it;
    // the halo's pushes are not sweeps of their own (message halos refuse damping)
    const double2 *prev = (kernel_id == CX_KERNEL_BIG_VAR && h->damping != 0.0) ? h->d_f2v : nullptr;
    if (n == 0) return;
    const double *sq = h->any_linear ? h->d_sq : h->d_q;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    prof_begin(h, kernel_id);
    const int mode = rule_mode(h);
    if (mode == kRuleLinear)
        hipLaunchKernelGGL(k_push_slots<kRuleLinear>, dim3(nb), dim3(kBlock), 0, h->stream, d_slots, n, h->d_partner, sq, h->d_sa, h->d_sb,
                           h->d_v2f, f2v_out, prev, h->damping);
    else if (mode == kRuleBernoulli)
        hipLaunchKernelGGL(k_push_slots<kRuleBernoulli>, dim3(nb), dim3(kBlock), 0, h->stream, d_slots, n, h->d_partner, sq,
                           (const double *)nullptr, (const double *)nullptr, h->d_v2f, f2v_out, prev, h->damping);
    else
        hipLaunchKernelGGL(k_push_slots<kRuleAdditive>, dim3(nb), dim3(kBlock), 0, h->stream, d_slots, n, h->d_partner, sq,
                           (const double *)nullptr, (const double *)nullptr, h->d_v2f, f2v_out, prev, h->damping);
    prof_end(h);
}

void launch_halo_export(cx_handle *h, const double2 *f2v, hipStream_t stream) {
    const int64_t n = (int64_t)h->send_slots.size();
    if (n == 0) return;
    prof_begin(h, CX_KERNEL_HALO_BEGIN, stream);
    hipLaunchKernelGGL(k_halo_export, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, h->d_send_slots, h->d_send_vars, n,
                       h->d_vbase, h->d_var_deg, h->d_vinfo, f2v, h->d_v2f, h->d_send_buf);
    prof_end(h);
}

void launch_halo_import(cx_handle *h, double2 *f2v_out, bool push) {
    const int64_t n = (int64_t)h->recv_slots.size();
    if (n == 0) return;
    const double *sq = h->any_linear ? h->d_sq : h->d_q;
    const dim3 g((unsigned)((n + kBlock - 1) / kBlock)), b(kBlock);
    prof_begin(h, CX_KERNEL_HALO_END);
#define CX_IMP(LIN, PU) hipLaunchKernelGGL((k_halo_import<LIN, PU>), g, b, 0, h->stream, h->d_recv_slots, n, h->d_recv_buf, h->d_partner, sq, h->d_sa, h->d_sb, h->d_v2f, f2v_out)
    if (h->any_linear) { if (push) CX_IMP(kRuleLinear, true); else CX_IMP(kRuleLinear, false); }
    else { if (push) CX_IMP(kRuleAdditive, true); else CX_IMP(kRuleAdditive, false); }
#undef CX_IMP
    prof_end(h);
}

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
#define CX_B(M, PA, PB) hipLaunchKernelGGL(k_batch_run<M>, dim3(1), dim3(kRunBlock), 0, h->stream, d_stage_off, s0, s1, d_rec, h->d_vbase, h->d_var_deg, h->d_vinfo, \
                                           h->d_partner, h->d_q, PA, PB, h->d_f2v, h->d_v2f, h->d_marg, nat, h->d_prod, h->d_joint, kt)
    if (mode == kRuleLinear) CX_B(kRuleLinear, h->d_a, h->d_b);
    else if (mode == kRuleBernoulli) CX_B(kRuleBernoulli, (const double *)nullptr, (const double *)nullptr);
    else CX_B(kRuleAdditive, (const double *)nullptr, (const double *)nullptr);
#undef CX_B
}

// every stage of a reference-order plan in ONE launch of an XCD-resident cluster; d_ctl: 64 bytes the launch may scribble on (zeroed here)
void launch_ref_cluster(cx_handle *h, void *d_ctl, int n_workgroups, const int32_t *d_flat, const int32_t *d_rec, const int64_t *d_stage_off, int n_stages) {
    if (n_stages <= 0) return;
    (void)hipMemsetAsync(d_ctl, 0, sizeof(ClusterCtl), h->stream);
    const int mode = rule_mode(h), nat = h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0;
    const KaryTab kt{h->d_kary_slot, h->d_kary_coef, h->d_kary_qb, h->d_ref_list};
    // bit 0: CX_REF_CLUSTER_DRY=1; bits 1..: CX_REF_CLUSTER_HELP = 0 no helpers, 1 helpers load records, 2 (default) records and source lines
    static const int dry = [] {
        const char *e = std::getenv("CX_REF_CLUSTER_DRY"), *hp = std::getenv("CX_REF_CLUSTER_HELP");
        const char *ah = std::getenv("CX_REF_CLUSTER_AHEAD"), *mb = std::getenv("CX_REF_CLUSTER_MEMBERS");      // A/B: stages the helpers run ahead, member workgroups
        const char *fw = std::getenv("CX_REF_PAIR_FWD");      // 0: a follower always waits for its leader's store (A/B)
        return ((e && (e[0] == '1' || e[0] == '2')) ? 1 : 0) | ((hp ? std::max(0, std::min(2, std::atoi(hp))) : 2) << 1) | ((e && e[0] == '2') ? 8 : 0) | ((fw && fw[0] == '0') ? 16 : 0) |
               ((ah ? std::max(0, std::min(255, std::atoi(ah))) : 0) << 8) | ((mb ? std::max(0, std::min(255, std::atoi(mb))) : 0) << 16);
    }();
#define CX_B(M, PA, PB) hipLaunchKernelGGL(k_ref_cluster<M>, dim3(n_workgroups), dim3(kClusterBlock), 0, h->stream, (ClusterCtl *)d_ctl, (unsigned)n_workgroups, d_stage_off, n_stages, \
                                           d_flat, d_rec, h->d_vbase, h->d_var_deg, h->d_vinfo, h->d_partner, h->d_q, PA, PB, h->d_f2v, h->d_v2f, h->d_marg, nat, h->d_prod, h->d_joint, kt, dry)
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

void launch_scatter(cx_handle *h, double2 *dst, const int32_t *d_idx, const double2 *d_val, int64_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, dst, d_idx, d_val, n);
}

void launch_gather(cx_handle *h, const double2 *src, const int32_t *d_idx, double2 *d_val, int64_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, d_idx, d_val, n);
}

void launch_seed(cx_handle *h, double2 *buf, int64_t n, double2 value, const int32_t *partner) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, buf, n, value, partner, (const int32_t *)h->d_slot_kary);
}

void launch_residual(cx_handle *h, const double2 *cur, const double2 *prev, int64_t n, double *d_out) {
    hipLaunchKernelGGL(k_residual, dim3(1024), dim3(kBlock), 0, h->stream, cur, prev, n, d_out, h->cfg.family == CX_FAMILY_NATURAL2 ? 1 : 0);
}

}  // namespace cx
