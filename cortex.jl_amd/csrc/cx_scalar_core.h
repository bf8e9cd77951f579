// cx_scalar_core.h — what the scalar (dim 1) kernels share: natural-form pairs, the factor→variable rules, one variable→factor message,
// and the host-side helpers every launcher of cx_kernels.hip (the sweeps) and cx_batch.hip (batched items, stage plans) uses.
#pragma once
#include <cstdlib>

#include "cx_internal.h"
#include "cx_kary_core.h"

namespace cx {

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

// ------------------------------------------------------------------------------------------------ host side
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

static inline int rule_mode(const cx_handle *h) {
    return h->cfg.family == CX_FAMILY_NATURAL2 ? kRuleBernoulli : (h->any_linear ? kRuleLinear : kRuleAdditive);
}

}  // namespace cx
