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

// ---- the same maps WITHOUT the division (round 6: the one-launch scan, where a composition is a link of a latency chain) -----------------
// w' = (A w + B) / (C w + D), xi' = (e xi + f w + g) / (C w + D): a 3 x 3 matrix up to scale.  A product is 15 multiply-adds two deep;
// instead of dividing by D the seven entries are scaled by the power of two that puts D into [0.5, 1) (exact), so a million compositions
// neither overflow nor lose anything the normalised form keeps.  The division happens once, where a map meets a message (linp_apply).
struct LinP {
    double e, f, g, A, B, C, D;
    int seg;
};

__device__ __forceinline__ LinP linp_identity() { return LinP{1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0}; }

__device__ __forceinline__ LinP linp_compose(const LinP &first, const LinP &second) {
    if (second.seg) return second;
    LinP r;
    r.A = second.A * first.A + second.B * first.C;
    r.B = second.A * first.B + second.B * first.D;
    r.C = second.C * first.A + second.D * first.C;
    r.D = second.C * first.B + second.D * first.D;
    r.e = second.e * first.e;
    r.f = second.e * first.f + second.f * first.A + second.g * first.C;
    r.g = second.e * first.g + second.f * first.B + second.g * first.D;
    const int k = -__builtin_amdgcn_frexp_exp(r.D);      // (0 for D = 0, infinite or NaN)
    r.A = __builtin_ldexp(r.A, k); r.B = __builtin_ldexp(r.B, k); r.C = __builtin_ldexp(r.C, k); r.D = __builtin_ldexp(r.D, k);
    r.e = __builtin_ldexp(r.e, k); r.f = __builtin_ldexp(r.f, k); r.g = __builtin_ldexp(r.g, k);
    r.seg = first.seg;
    return r;
}

__device__ __forceinline__ LinP linp_shfl_up(const LinP &x, int d) {
    LinP r;
    r.e = __shfl_up(x.e, d, 64); r.f = __shfl_up(x.f, d, 64); r.g = __shfl_up(x.g, d, 64);
    r.A = __shfl_up(x.A, d, 64); r.B = __shfl_up(x.B, d, 64); r.C = __shfl_up(x.C, d, 64); r.D = __shfl_up(x.D, d, 64);
    r.seg = __shfl_up(x.seg, d, 64);
    return r;
}

// lin_of_link before its division: v = m + u, out = ((a v.x + b v.y), v.y) / (a a + q v.y)
__device__ __forceinline__ LinP linp_of_link(double2 u, double q, double a, double b, int seg) {
    return LinP{a, b, a * u.x + b * u.y, 1.0, u.y, q, a * a + q * u.y, seg};
}

// the shuffles of a wave scan as DPP moves (row shifts, then the row totals broadcast to the rows behind them): no trip through the LDS
// crossbar (ds_bpermute) between two compositions
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ LinP linp_dpp(const LinP &x) {
    LinP r;
    r.e = dpp_f64<CTRL, ROW_MASK>(x.e); r.f = dpp_f64<CTRL, ROW_MASK>(x.f); r.g = dpp_f64<CTRL, ROW_MASK>(x.g);
    r.A = dpp_f64<CTRL, ROW_MASK>(x.A); r.B = dpp_f64<CTRL, ROW_MASK>(x.B); r.C = dpp_f64<CTRL, ROW_MASK>(x.C); r.D = dpp_f64<CTRL, ROW_MASK>(x.D);
    r.seg = __builtin_amdgcn_update_dpp(x.seg, x.seg, CTRL, ROW_MASK, 0xf, false);
    return r;
}
// inclusive scan over the 64 lanes of a wave, earlier lanes first (every lane of the wave has to be active)
__device__ __forceinline__ LinP linp_wave_scan(LinP t, int lane) {
    const int r = lane & 15;
    { const LinP o = linp_dpp<0x111, 0xf>(t); if (r >= 1) t = linp_compose(o, t); }      // row_shr:1
    { const LinP o = linp_dpp<0x112, 0xf>(t); if (r >= 2) t = linp_compose(o, t); }      // row_shr:2
    { const LinP o = linp_dpp<0x114, 0xf>(t); if (r >= 4) t = linp_compose(o, t); }      // row_shr:4
    { const LinP o = linp_dpp<0x118, 0xf>(t); if (r >= 8) t = linp_compose(o, t); }      // row_shr:8
    { const LinP o = linp_dpp<0x142, 0xa>(t); if (lane & 16) t = linp_compose(o, t); }   // row_bcast:15 into rows 1 and 3
    { const LinP o = linp_dpp<0x143, 0xc>(t); if (lane & 32) t = linp_compose(o, t); }   // row_bcast:31 into rows 2 and 3
    return t;
}
// the map of the lane before (lane 0: the identity)
__device__ __forceinline__ LinP linp_wave_prev(const LinP &t, int lane) {
    LinP ex = linp_dpp<0x138, 0xf>(t);      // wave_shr:1
    if (lane == 0) ex = linp_identity();
    return ex;
}

// a message kept as (x, w) / d, and a map applied to it without the division
struct MsgP { double x, w, d; };
__device__ __forceinline__ MsgP linp_apply_p(const LinP &p, const MsgP &m) {
    if (p.seg) return MsgP{p.g, p.B, p.D};
    return MsgP{p.e * m.x + p.f * m.w + p.g * m.d, p.A * m.w + p.B * m.d, p.C * m.w + p.D * m.d};
}

__device__ __forceinline__ MsgP msgp_rescale(const MsgP &m) {
    const int k = -__builtin_amdgcn_frexp_exp(m.d);
    return MsgP{__builtin_ldexp(m.x, k), __builtin_ldexp(m.w, k), __builtin_ldexp(m.d, k)};
}

__device__ __forceinline__ double2 linp_apply(const LinP &p, double2 m) {
    if (p.seg) { const double inv = 1.0 / p.D; return make_double2(p.g * inv, p.B * inv); }
    const double inv = 1.0 / (p.C * m.y + p.D);
    return make_double2((p.e * m.x + p.f * m.y + p.g) * inv, (p.A * m.y + p.B) * inv);
}

}  // namespace cx
