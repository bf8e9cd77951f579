// cx_lin.h — the map algebra of the scalar chain scans (cx_chain.hip: CX_SCHED_CHAIN_SCAN, heavy paths; cx_planscan.hip: the chains
// inside reference-order plans).  With messages in natural form m = (xi, w), adding side information u and passing through a factor rule
// is a projective-linear map on (xi, w, 1); such maps compose, every product rescaled to D = 1 (cx_chain.hip's header derives them).
#pragma once
#include <hip/hip_runtime.h>

namespace cx {

struct Lin {  // projective-linear map with D normalised to 1; seg = 1 marks "starts a new path" (scan does not cross)
    double e, f, g, A, B, C;
    int seg;
};

__device__ __forceinline__ Lin lin_identity() { return Lin{1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0}; }

// (second ∘ first), segmented: if `second` starts a segment the result is `second` alone
__device__ __forceinline__ Lin lin_compose(const Lin &first, const Lin &second) {
    if (second.seg) return second;
    Lin r;
    const double D = second.C * first.B + 1.0;
    const double inv = 1.0 / D;
    r.A = (second.A * first.A + second.B * first.C) * inv;
    r.B = (second.A * first.B + second.B) * inv;
    r.C = (second.C * first.A + first.C) * inv;
    r.e = (second.e * first.e) * inv;
    r.f = (second.e * first.f + second.f * first.A + second.g * first.C) * inv;
    r.g = (second.e * first.g + second.f * first.B + second.g) * inv;
    r.seg = first.seg;
    return r;
}

__device__ __forceinline__ Lin lin_shfl_up(const Lin &x, int d) {
    Lin r;
    r.e = __shfl_up(x.e, d, 64); r.f = __shfl_up(x.f, d, 64); r.g = __shfl_up(x.g, d, 64);
    r.A = __shfl_up(x.A, d, 64); r.B = __shfl_up(x.B, d, 64); r.C = __shfl_up(x.C, d, 64);
    r.seg = __shfl_up(x.seg, d, 64);
    return r;
}

// the map of one link: add side information u, then the factor rule with the receiving slot's (a, b, q)
__device__ __forceinline__ Lin lin_of_link(double2 u, double q, double a, double b, int seg) {
    const double D = a * a + q * u.y;
    const double inv = 1.0 / D;
    return Lin{a * inv, b * inv, (a * u.x + b * u.y) * inv, inv, u.y * inv, q * inv, seg};
}

// the factor→variable rule of cx_kernels.hip (receiving edge's parameters); a = 1, b = 0 for additive factors
__device__ __forceinline__ double2 chain_factor_rule(double2 m, double q, double a, double b) {
    double2 o;
    if (m.y == __builtin_inf()) {
        o.y = 1.0 / q;
        o.x = (a * m.x + b) * o.y;
    } else {
        const double s = 1.0 / (a * a + q * m.y);
        o.y = m.y * s;
        o.x = (a * m.x + b * m.y) * s;
    }
    return o;
}

// a map applied to a message (D = 1); a map that starts a path ignores what comes in
__device__ __forceinline__ double2 lin_apply(const Lin &p, double2 m) {
    if (p.seg) return make_double2(p.g, p.B);
    const double inv = 1.0 / (p.C * m.y + 1.0);
    return make_double2((p.e * m.x + p.f * m.y + p.g) * inv, (p.A * m.y + p.B) * inv);
}


}  // namespace cx
