// cx_mvchain.hip — CX_SCHED_CHAIN_SCAN for d-dimensional messages (d = 2, 3, 4): ONE cx_sweep on a state-space chain is the
// exact forward/backward smoother, i.e. what ONE update_marginals! of the reference computes on such a graph
// (src/inference_engine.jl:575-608: a forward pass and a reverse pass of process! calls, 5T-4 message computations; SURVEY.md
// §3.3 hand trace; the SSM of test/inference_engine_tests.jl:436-487) — here by parallel prefix scans over per-link maps
// instead of a walk that is sequential in t.  (The flooding sweep of cx_mv.hip moves information one link per sweep.)
//
// The map of a link.  With messages in natural form m = (eta, Lambda), "add the side information u of the sending variable
// (the sum of its non-chain incoming messages: likelihoods, priors), then apply the factor rule with the receiving edge's triple
// (P, B, C)" (cx_mv.hip) is
//     f(eta, Lambda) = ( c + B (Lambda + P~)^-1 (eta + h),   C - B (Lambda + P~)^-1 B' ),    P~ = P + U,  h = u_eta,  c = 0.
// Maps of this form are closed under composition (Woodbury): for f2 after f1, with S = (C1 + P2)^-1 and g = c1 + h2,
//     P12 = P1 - B1' S B1,   B12 = B2 S B1,   C12 = C2 - B2 S B2',   h12 = h1 + B1' S g,   c12 = c2 + B2 S g
// — one d x d Cholesky and five small products per combine, all in registers.  (The scalar scan of cx_chain.hip is the d = 1
// case written as a projective 3 x 3 matrix; the (P, B, C, h, c) form is the information-form element of the parallel Kalman
// smoother.)  The first link of a path receives nothing from the chain: its map is the CONSTANT map "rule applied to u alone"
// (B = 0), so every prefix of a path is a constant map and yields the message (c, C) directly; a constant second operand also
// ends a segment of the segmented scan.
//
// Work decomposition (three launches per sweep; two when the marginals are formed on demand).  A thread owns K consecutive links.
//   k_mvc_totals      composes the K maps of every thread (thread total), scans the thread totals over the workgroup (wave shuffles
//                     + LDS across waves) and stores each thread's exclusive prefix within its tile, and the tile total;
//                     forward and backward direction in one grid
//   k_mvc_apply       the tile carry (composed by the workgroup's first wave from the tile totals before it: no scan launch in between),
//                     tile carry ∘ thread prefix = the message entering the thread's first link; then the thread WALKS its K links
//                     with the ordinary rule of the flooding sweep (one Cholesky per message) — map composition is paid once per
//                     thread and scan step, not per link.  The forward walk leaves alpha (into the right end of every link), the
//                     backward walk gamma (what the right end hears from everybody but the link); both directions in one grid
//   k_mvc_marg_out    marginal of a link's right variable = alpha + gamma, to moment form, transposed through LDS into the marginals (pair form by variable)
// Results are re-associated relative to the sequential schedule: they agree with it to rounding, not bitwise.
//
// Memory layout.  "Thread owns K consecutive links" would make every access to a chain-ordered array a stride-K access.  The
// arrays the sweep streams — the side sums of a link's two ends, alpha and gamma — are therefore kept in a thread-interleaved
// order IL(l) = tile * 256 K + (l mod K) * 256 + thread: what the 256 threads of a workgroup touch in step k of their walks is
// one contiguous run of 2 KB per component.  Messages go to the SELL slots of cx_mv.hip only on demand (cx_get_messages, a
// checkpoint, a residual: mvc_launch_scan(..., store_msgs)), like the variable→factor messages of the fused sweep; the
// marginals are written every sweep.

#include <algorithm>
#include <cstdlib>

#include "cx_internal.h"
#include "cx_mv_core.h"

namespace cx {

constexpr int kMapSeg = 1;     // constant map: the first link of a path (the scan does not cross it)
constexpr int kMapIdent = 2;   // identity (padding, exclusive prefix of the first element)

template <int D>
struct CMap {
    static constexpr int NT = D * (D + 1) / 2;
    static constexpr int oB = NT, oC = NT + D * D, oH = oC + NT, oS = oH + D, ND = oS + D;   // P | B | C | h | c
    double v[ND];
    int flags;
};

template <int D>
__device__ __forceinline__ CMap<D> cmap_identity() {
    CMap<D> r;
#pragma unroll
    for (int i = 0; i < CMap<D>::ND; i++) r.v[i] = 0.0;
    r.flags = kMapIdent;
    return r;
}

// second ∘ first
template <int D>
__device__ __forceinline__ CMap<D> cmap_compose(const CMap<D> &f, const CMap<D> &s) {
    using M = CMap<D>;
    if (s.flags & (kMapSeg | kMapIdent)) return (s.flags & kMapSeg) ? s : f;
    if (f.flags & kMapIdent) return s;
    double Kp[M::NT];
#pragma unroll
    for (int i = 0; i < M::NT; i++) Kp[i] = f.v[M::oC + i] + s.v[i];
    double Lm[D][D], ri[D];
    chol<D>(Kp, nullptr, Lm, ri);
    double X1[D][D], X2[D][D];      // X1 = L^-1 B1,  X2 = L^-1 B2'
#pragma unroll
    for (int j = 0; j < D; j++) {
        double col[D];
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = f.v[M::oB + k * D + j];
        fwd_solve<D>(Lm, ri, col);
#pragma unroll
        for (int k = 0; k < D; k++) X1[k][j] = col[k];
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = s.v[M::oB + j * D + k];
        fwd_solve<D>(Lm, ri, col);
#pragma unroll
        for (int k = 0; k < D; k++) X2[k][j] = col[k];
    }
    double g[D];
#pragma unroll
    for (int k = 0; k < D; k++) g[k] = f.v[M::oS + k] + s.v[M::oH + k];
    fwd_solve<D>(Lm, ri, g);
    M r;
#pragma unroll
    for (int i = 0; i < D; i++) {
        double hh = f.v[M::oH + i], cc = s.v[M::oS + i];
#pragma unroll
        for (int k = 0; k < D; k++) { hh += X1[k][i] * g[k]; cc += X2[k][i] * g[k]; }
        r.v[M::oH + i] = hh; r.v[M::oS + i] = cc;
#pragma unroll
        for (int j = 0; j < D; j++) {
            double b = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) b += X2[k][i] * X1[k][j];
            r.v[M::oB + i * D + j] = b;
        }
#pragma unroll
        for (int j = i; j < D; j++) {
            double p = f.v[tri<D>(i, j)], c = s.v[M::oC + tri<D>(i, j)];
#pragma unroll
            for (int k = 0; k < D; k++) { p -= X1[k][i] * X1[k][j]; c -= X2[k][i] * X2[k][j]; }
            r.v[tri<D>(i, j)] = p; r.v[M::oC + tri<D>(i, j)] = c;
        }
    }
    r.flags = f.flags;
    return r;
}

template <int D>
__device__ __forceinline__ CMap<D> cmap_shfl(const CMap<D> &x, int src) {
    CMap<D> r;
    src = src < 0 ? 0 : (src > 63 ? 63 : src);
#pragma unroll
    for (int i = 0; i < CMap<D>::ND; i++) r.v[i] = __shfl(x.v[i], src, 64);
    r.flags = __shfl(x.flags, src, 64);
    return r;
}

template <int D>
__device__ __forceinline__ void cmap_store(double *p, const CMap<D> &x) {      // ND + 1 doubles
#pragma unroll
    for (int i = 0; i < CMap<D>::ND; i++) p[i] = x.v[i];
    p[CMap<D>::ND] = (double)x.flags;
}
template <int D>
__device__ __forceinline__ CMap<D> cmap_load(const double *p) {
    CMap<D> r;
#pragma unroll
    for (int i = 0; i < CMap<D>::ND; i++) r.v[i] = p[i];
    r.flags = (int)p[CMap<D>::ND];
    return r;
}

// the map of one link from the side information u of its sending variable and the receiving edge's (P | B | C) tables
template <int D>
__device__ __forceinline__ CMap<D> cmap_of_link(const Msg<D> &u, const double *__restrict__ tab, bool head) {
    using M = CMap<D>;
    M r;
    if (head) {
        const Msg<D> o = mv_rule<D, false>(u, tab);
#pragma unroll
        for (int i = 0; i < D; i++) {
            r.v[M::oH + i] = 0.0; r.v[M::oS + i] = o.eta[i];
#pragma unroll
            for (int j = 0; j < D; j++) r.v[M::oB + i * D + j] = 0.0;
#pragma unroll
            for (int j = i; j < D; j++) { r.v[tri<D>(i, j)] = i == j ? 1.0 : 0.0; r.v[M::oC + tri<D>(i, j)] = o.lam[tri<D>(i, j)]; }
        }
        r.flags = kMapSeg;
        return r;
    }
    const double *P = tab, *B = tab + D * D, *C = tab + 2 * D * D;
#pragma unroll
    for (int i = 0; i < D; i++) {
        r.v[M::oH + i] = u.eta[i]; r.v[M::oS + i] = 0.0;
#pragma unroll
        for (int j = 0; j < D; j++) r.v[M::oB + i * D + j] = B[i * D + j];
#pragma unroll
        for (int j = i; j < D; j++) { r.v[tri<D>(i, j)] = P[i * D + j] + u.lam[tri<D>(i, j)]; r.v[M::oC + tri<D>(i, j)] = C[i * D + j]; }
    }
    r.flags = 0;
    return r;
}

// ---- in-place forms (round 3, second pass): the generic cmap_compose above keeps both operands and the result in registers —
// 360 of them for d = 4, one wave per SIMD.  The forms below overwrite the accumulator field by field and fetch the other
// operand's fields when they are consumed (from registers, LDS / global memory, the link's tables, or another lane): ≈ 250
// registers, two waves per SIMD for the kernel that is bound by these compositions' dependent chains (k_mvc_totals).

// operand sources: P(a, b) / C(a, b) for a <= b, B(a, b), h(a), c(a), flags()
template <int D>
struct SrcMem {            // ND + 1 doubles at p (cmap_store's form)
    using M = CMap<D>;
    const double *p;
    __device__ __forceinline__ double P(int a, int b) const { return p[tri<D>(a, b)]; }
    __device__ __forceinline__ double B(int a, int b) const { return p[M::oB + a * D + b]; }
    __device__ __forceinline__ double C(int a, int b) const { return p[M::oC + tri<D>(a, b)]; }
    __device__ __forceinline__ double h(int a) const { return p[M::oH + a]; }
    __device__ __forceinline__ double c(int a) const { return p[M::oS + a]; }
    __device__ __forceinline__ int flags() const { return (int)p[M::ND]; }
};
template <int D>
struct SrcStrided {        // component i of the map at p[i * stride] (prefix_store's form)
    using M = CMap<D>;
    const double *p; int64_t stride;
    __device__ __forceinline__ double at(int i) const { return p[(int64_t)i * stride]; }
    __device__ __forceinline__ double P(int a, int b) const { return at(tri<D>(a, b)); }
    __device__ __forceinline__ double B(int a, int b) const { return at(M::oB + a * D + b); }
    __device__ __forceinline__ double C(int a, int b) const { return at(M::oC + tri<D>(a, b)); }
    __device__ __forceinline__ double h(int a) const { return at(M::oH + a); }
    __device__ __forceinline__ double c(int a) const { return at(M::oS + a); }
    __device__ __forceinline__ int flags() const { return (int)at(M::ND); }
};
template <int D>
struct SrcLink {           // the (non-constant) map of a link: side information u, the receiving edge's (P | B | C) tables
    const Msg<D> &u; const double *tab;
    __device__ __forceinline__ double P(int a, int b) const { return tab[a * D + b] + u.lam[tri<D>(a, b)]; }
    __device__ __forceinline__ double B(int a, int b) const { return tab[D * D + a * D + b]; }
    __device__ __forceinline__ double C(int a, int b) const { return tab[2 * D * D + a * D + b]; }
    __device__ __forceinline__ double h(int a) const { return u.eta[a]; }
    __device__ __forceinline__ double c(int) const { return 0.0; }
    __device__ __forceinline__ int flags() const { return 0; }
};

template <int D, class S>
__device__ __forceinline__ void cmap_assign(CMap<D> &f, const S &s) {
    using M = CMap<D>;
#pragma unroll
    for (int a = 0; a < D; a++) {
        f.v[M::oH + a] = s.h(a); f.v[M::oS + a] = s.c(a);
#pragma unroll
        for (int b = 0; b < D; b++) f.v[M::oB + a * D + b] = s.B(a, b);
#pragma unroll
        for (int b = a; b < D; b++) { f.v[tri<D>(a, b)] = s.P(a, b); f.v[M::oC + tri<D>(a, b)] = s.C(a, b); }
    }
    f.flags = s.flags();
}

// f <- s ∘ f, both general (no flag logic); f is overwritten field by field
template <int D, class S>
__device__ __forceinline__ void cmap_append(CMap<D> &f, const S &s) {
    using M = CMap<D>;
    double Lm[D][D], ri[D];
    {
        double Kp[M::NT];
#pragma unroll
        for (int a = 0; a < D; a++)
#pragma unroll
            for (int b = a; b < D; b++) Kp[tri<D>(a, b)] = f.v[M::oC + tri<D>(a, b)] + s.P(a, b);
        chol<D>(Kp, nullptr, Lm, ri);
    }
    double X1[D][D], X2[D][D], g[D];      // X1 = L^-1 B1,  X2 = L^-1 B2'
#pragma unroll
    for (int j = 0; j < D; j++) {
        double col[D];
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = f.v[M::oB + k * D + j];
        fwd_solve<D>(Lm, ri, col);
#pragma unroll
        for (int k = 0; k < D; k++) X1[k][j] = col[k];
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = s.B(j, k);
        fwd_solve<D>(Lm, ri, col);
#pragma unroll
        for (int k = 0; k < D; k++) X2[k][j] = col[k];
    }
#pragma unroll
    for (int k = 0; k < D; k++) g[k] = f.v[M::oS + k] + s.h(k);
    fwd_solve<D>(Lm, ri, g);
#pragma unroll
    for (int i = 0; i < D; i++) {
        double hh = f.v[M::oH + i], cc = s.c(i);
#pragma unroll
        for (int k = 0; k < D; k++) { hh += X1[k][i] * g[k]; cc += X2[k][i] * g[k]; }
        f.v[M::oH + i] = hh; f.v[M::oS + i] = cc;
#pragma unroll
        for (int j = i; j < D; j++) {
            double p = f.v[tri<D>(i, j)], c = s.C(i, j);
#pragma unroll
            for (int k = 0; k < D; k++) { p -= X1[k][i] * X1[k][j]; c -= X2[k][i] * X2[k][j]; }
            f.v[tri<D>(i, j)] = p; f.v[M::oC + tri<D>(i, j)] = c;
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = 0; j < D; j++) {
            double b = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) b += X2[k][i] * X1[k][j];
            f.v[M::oB + i * D + j] = b;
        }
}

// f <- s ∘ f with the segment / identity rules of cmap_compose
template <int D, class S>
__device__ __forceinline__ void cmap_append_any(CMap<D> &f, const S &s) {
    const int sf = s.flags();
    if (sf & kMapIdent) return;
    if ((sf & kMapSeg) || (f.flags & kMapIdent)) { cmap_assign<D>(f, s); return; }
    cmap_append<D>(f, s);
}

// One step of the wave scan, in place: t <- t ∘ (the map of lane src) where `active`, t unchanged elsewhere.  Every field of the
// other lane's map is fetched exactly once, BEFORE this lane (and, in lock step, every lane) overwrites its own copy of that
// field; the lane's own old fields are consumed before they are overwritten.
template <int D>
__device__ __forceinline__ void cmap_scan_step(CMap<D> &t, int src, bool active) {
    using M = CMap<D>;
    src = src < 0 ? 0 : (src > 63 ? 63 : src);
    const int ff = __shfl(t.flags, src, 64);
    const bool keep = !active || (t.flags & kMapSeg) || (ff & kMapIdent);
    const bool copy = !keep && (t.flags & kMapIdent);          // own map is the identity: the result is the other lane's map
    const bool gen = !keep && !copy;
    double Lm[D][D], ri[D], X2[D][D], g[D];
    {
        double fC[M::NT], Kp[M::NT];
#pragma unroll
        for (int a = 0; a < D; a++)
#pragma unroll
            for (int b = a; b < D; b++) {
                const int i = tri<D>(a, b);
                fC[i] = __shfl(t.v[M::oC + i], src, 64);
                Kp[i] = gen ? fC[i] + t.v[i] : (a == b ? 1.0 : 0.0);
            }
        chol<D>(Kp, nullptr, Lm, ri);          // lanes that do not compose factor the identity: no NaN, no exception, result unused
#pragma unroll
        for (int j = 0; j < D; j++) {
            double col[D];
#pragma unroll
            for (int k = 0; k < D; k++) col[k] = t.v[M::oB + j * D + k];
            fwd_solve<D>(Lm, ri, col);
#pragma unroll
            for (int k = 0; k < D; k++) X2[k][j] = col[k];
        }
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = i; j < D; j++) {
                double c = t.v[M::oC + tri<D>(i, j)];
#pragma unroll
                for (int k = 0; k < D; k++) c -= X2[k][i] * X2[k][j];
                t.v[M::oC + tri<D>(i, j)] = copy ? fC[tri<D>(i, j)] : (gen ? c : t.v[M::oC + tri<D>(i, j)]);
            }
    }
    {
        double fcv[D];
#pragma unroll
        for (int k = 0; k < D; k++) { fcv[k] = __shfl(t.v[M::oS + k], src, 64); g[k] = fcv[k] + t.v[M::oH + k]; }
        fwd_solve<D>(Lm, ri, g);
#pragma unroll
        for (int i = 0; i < D; i++) {
            double cc = t.v[M::oS + i];
#pragma unroll
            for (int k = 0; k < D; k++) cc += X2[k][i] * g[k];
            t.v[M::oS + i] = copy ? fcv[i] : (gen ? cc : t.v[M::oS + i]);
        }
    }
    double X1[D][D];
    {
        double fB[D * D];
#pragma unroll
        for (int i = 0; i < D * D; i++) fB[i] = __shfl(t.v[M::oB + i], src, 64);
#pragma unroll
        for (int j = 0; j < D; j++) {
            double col[D];
#pragma unroll
            for (int k = 0; k < D; k++) col[k] = fB[k * D + j];
            fwd_solve<D>(Lm, ri, col);
#pragma unroll
            for (int k = 0; k < D; k++) X1[k][j] = col[k];
        }
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = 0; j < D; j++) {
                double b = 0.0;
#pragma unroll
                for (int k = 0; k < D; k++) b += X2[k][i] * X1[k][j];
                t.v[M::oB + i * D + j] = copy ? fB[i * D + j] : (gen ? b : t.v[M::oB + i * D + j]);
            }
    }
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = i; j < D; j++) {
            const double fP = __shfl(t.v[tri<D>(i, j)], src, 64);
            double p = fP;
#pragma unroll
            for (int k = 0; k < D; k++) p -= X1[k][i] * X1[k][j];
            t.v[tri<D>(i, j)] = copy ? fP : (gen ? p : t.v[tri<D>(i, j)]);
        }
#pragma unroll
    for (int i = 0; i < D; i++) {
        const double fh = __shfl(t.v[M::oH + i], src, 64);
        double hh = fh;
#pragma unroll
        for (int k = 0; k < D; k++) hh += X1[k][i] * g[k];
        t.v[M::oH + i] = copy ? fh : (gen ? hh : t.v[M::oH + i]);
    }
    t.flags = keep ? t.flags : ff;
}

struct MvcArgs {
    int nlinks, npos, nv, ntab;
    int64_t nslots;
    const int32_t *link_pos;             // chain position of the link's left variable
    const int32_t *from_slot, *to_slot;  // slot (left variable, factor), slot (right variable, factor)
    const int32_t *tab_fwd, *tab_bwd;    // rule-table index of the message left → right / right → left
    const uint8_t *head_fwd, *head_bwd;  // first / last link of its path
    const int32_t *pos_var;
    const double *side;                  // [nc][npos]: sum of the non-chain messages into each chain variable, by position
    const double *side_l;                // the same for the LEFT variable of link l, at IL(l); block-major pairs like the message buffers
                                         // (cx_mv_core.h: slot_load): a block = the 256 threads of a workgroup at one step k.  The right
                                         // variable of a link is the left variable of the next one: mvc_side_right
    double *alpha, *gamma;               // at IL(l), same form: the forward message link l produces; what its right variable hears
                                         // from everybody but the link (side + the backward message of the next link)
    double *prefix;                      // [2][ND + 1][nthreads]: every thread's inclusive prefix within its WAVE, per direction
    double *wave_carry;                  // [2][ntiles][4][ND + 1]: the carry into each wave of each tile
    int64_t il_stride;                   // ntiles * 256 * K = K * nthreads
    int collapse_heads;                  // 1 (a sweep): the first link of a path is the constant map "rule applied to the side information";
                                         // 0 (cx_chain_block_maps): it stays a general map of the message that enters the block
    const double *ptab;                  // [ntab][3][D*D]
};

// What the RIGHT variable of link l hears from off the chain.  Inside a path that is the side sum of the next link's left variable —
// IL(l + 1): the same thread's next step, or the first step of the next thread (of the next tile) — and for the last link of a path
// the by-position entry of the path's last variable.  (A copy at IL(l) used to be kept: a third of what the side pass wrote.)
template <int D>
__device__ __forceinline__ Msg<D> mvc_side_right(const MvcArgs &A, int l, int64_t il, int K) {
    int64_t iln = il + kBlock;
    if ((l + 1) % K == 0) { const int64_t g = (int64_t)(l + 1) / K; iln = (g / kBlock) * (int64_t)kBlock * K + (g % kBlock); }
    // the load goes out before the path-end flag is known (a dependent round trip per step of the backward walk otherwise); the
    // last link of the last tile would point one past the array
    Msg<D> u = slot_load<D, true>(A.side_l, (int)(iln < A.il_stride ? iln : il));
    if (A.head_bwd[l]) u = msg_load<D>(A.side, A.npos, A.link_pos[l] + 1);
    return u;
}

template <int D>
__device__ __forceinline__ Msg<D> msg_nan() {
    Msg<D> m;
#pragma unroll
    for (int i = 0; i < D; i++) m.eta[i] = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) m.lam[i] = __builtin_nan("");
    return m;
}

constexpr int kMvcTabLds = 16;           // rule tables (parameter set x direction) kept in LDS; graphs with more read them from memory

template <int D, bool GT>
__device__ __forceinline__ const double *mvc_tab(const MvcArgs &A, const double *tab_s, int t) {
    return (GT ? A.ptab : tab_s) + (size_t)t * 3 * D * D;
}

// the composed map of a thread's K consecutive links, in the direction's order, accumulated in place: the link's own map is
// never materialised (its B and C are the LDS tables, its P and h the tables plus the side information)
template <int D, bool GT>
__device__ __forceinline__ void mvc_thread_total(CMap<D> &tot, const MvcArgs &A, const double *tab_s, int l0, int64_t il0, int K, int dir) {
    using M = CMap<D>;
    tot = cmap_identity<D>();
    // (the next link's loads in flight while this link's map is appended — what the walks do — measured: no gain here, 0.243 vs 0.241 ms)
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        const int kk = dir > 0 ? k : K - 1 - k, l = l0 + kk;
        if (l >= A.nlinks) continue;
        const bool head = (dir > 0 ? A.head_fwd[l] : A.head_bwd[l]) != 0;
        const double *tab = mvc_tab<D, GT>(A, tab_s, dir > 0 ? A.tab_fwd[l] : A.tab_bwd[l]);
        const Msg<D> u = dir > 0 ? slot_load<D, true>(A.side_l, (int)(il0 + (int64_t)kk * kBlock)) : mvc_side_right<D>(A, l, il0 + (int64_t)kk * kBlock, K);
        if (head && A.collapse_heads) {      // the first link of a path: the constant map "rule applied to u alone"
            const Msg<D> o = mv_rule<D, false>(u, tab);
#pragma unroll
            for (int i = 0; i < D; i++) {
                tot.v[M::oH + i] = 0.0; tot.v[M::oS + i] = o.eta[i];
#pragma unroll
                for (int j = 0; j < D; j++) tot.v[M::oB + i * D + j] = 0.0;
#pragma unroll
                for (int j = i; j < D; j++) { tot.v[tri<D>(i, j)] = i == j ? 1.0 : 0.0; tot.v[M::oC + tri<D>(i, j)] = o.lam[tri<D>(i, j)]; }
            }
            tot.flags = kMapSeg;
        } else {
            const SrcLink<D> link{u, tab};
            if (tot.flags & kMapIdent) cmap_assign<D>(tot, link);
            else cmap_append<D>(tot, link);
        }
    }
}

// Workgroup scan in the direction's logical order (dir = -1: thread 255 first).  In: every thread's total.  Out: t = the
// exclusive prefix of the thread within the workgroup; total (WANT_TOTAL) = the workgroup's total, valid in the logically
// LAST thread only.  wt: LDS, kBlock/64 elements of ND + 1 doubles.
template <int D, bool WANT_TOTAL>
__device__ __forceinline__ void mvc_wg_scan(CMap<D> &t, CMap<D> &total, double *wt, int tid, int dir) {
    constexpr int E = CMap<D>::ND + 1, NW = kBlock / 64;
    const int lane = tid & 63, wid = tid >> 6;
    const int li = dir > 0 ? lane : 63 - lane, wl = dir > 0 ? wid : NW - 1 - wid;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
        const CMap<D> o = cmap_shfl<D>(t, dir > 0 ? lane - d : lane + d);
        if (li >= d) t = cmap_compose<D>(o, t);
    }
    if (li == 63) cmap_store<D>(wt + wid * E, t);
    __syncthreads();
    CMap<D> ex = cmap_shfl<D>(t, dir > 0 ? lane - 1 : lane + 1);
    if (li == 0) ex = cmap_identity<D>();
    CMap<D> carry = cmap_identity<D>();
#pragma unroll 1
    for (int w = 0; w < wl; w++) carry = cmap_compose<D>(carry, cmap_load<D>(wt + (dir > 0 ? w : NW - 1 - w) * E));
    if (WANT_TOTAL && wl == NW - 1 && li == 63) total = cmap_compose<D>(carry, t);
    t = cmap_compose<D>(carry, ex);
    __syncthreads();
}

template <int D>
__device__ __forceinline__ void mvc_load_tabs(const MvcArgs &A, double *tab_s, int tid) {
    const int n = (A.ntab < kMvcTabLds ? A.ntab : kMvcTabLds) * 3 * D * D;
    for (int i = tid; i < n; i += kBlock) tab_s[i] = A.ptab[i];
    __syncthreads();
}

// a thread's map to / from the prefix array: component-major over the threads of the grid, so that a wave's access is contiguous
template <int D>
__device__ __forceinline__ void prefix_store(double *__restrict__ p, int64_t nthreads, int64_t gid, const CMap<D> &x) {
#pragma unroll
    for (int i = 0; i < CMap<D>::ND; i++) p[(int64_t)i * nthreads + gid] = x.v[i];
    p[(int64_t)CMap<D>::ND * nthreads + gid] = (double)x.flags;
}
template <int D>
__device__ __forceinline__ CMap<D> prefix_load(const double *__restrict__ p, int64_t nthreads, int64_t gid) {
    CMap<D> r;
#pragma unroll
    for (int i = 0; i < CMap<D>::ND; i++) r.v[i] = p[(int64_t)i * nthreads + gid];
    r.flags = (int)p[(int64_t)CMap<D>::ND * nthreads + gid];
    return r;
}

// grid (ntiles, 2): blockIdx.y = 0 forward, 1 backward.  Per thread: the composed map of its K links, then the INCLUSIVE scan
// over its wave (in place, by shuffles) — stored as the thread's prefix.  Per wave: the carry into it (the waves logically before
// it in the workgroup, composed) — stored per wave; the logically last wave also stores the tile total = its carry ∘ its total,
// at totals[dir][pos], pos in the direction's scan order (backward: tile ntiles - 1 first).  The walks put the pieces together
// (k_mvc_apply): tile carry ∘ wave carry ∘ prefix of the previous lane.
template <int D, bool GT>
__global__ __launch_bounds__(kBlock, 2) void k_mvc_totals(MvcArgs A, int K, double *__restrict__ totals) {
    constexpr int E = CMap<D>::ND + 1, NW = kBlock / 64;
    __shared__ double tab_s[GT ? 1 : kMvcTabLds * 3 * D * D];
    __shared__ double wt[NW * E];
    const int tid = threadIdx.x, dir = blockIdx.y ? -1 : 1, ntiles = gridDim.x;
    if (!GT) mvc_load_tabs<D>(A, tab_s, tid);
    const int64_t gid = (int64_t)blockIdx.x * kBlock + tid, nthreads = (int64_t)ntiles * kBlock;
    const int lane = tid & 63, wid = tid >> 6;
    const int li = dir > 0 ? lane : 63 - lane, wl = dir > 0 ? wid : NW - 1 - wid;
    CMap<D> t;
    mvc_thread_total<D, GT>(t, A, tab_s, (int)gid * K, (int64_t)blockIdx.x * kBlock * K + tid, K, dir);
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) cmap_scan_step<D>(t, dir > 0 ? lane - d : lane + d, li >= d);
    prefix_store<D>(A.prefix + (size_t)blockIdx.y * E * nthreads, nthreads, gid, t);
    if (li == 63) cmap_store<D>(wt + wid * E, t);
    __syncthreads();
    // wave-uniform from here on: every lane of a wave composes the same carry
    CMap<D> carry = cmap_identity<D>();
#pragma unroll 1
    for (int w = 0; w < wl; w++) cmap_append_any<D>(carry, SrcMem<D>{wt + (dir > 0 ? w : NW - 1 - w) * E});
    if (li == 0) cmap_store<D>(A.wave_carry + (((size_t)blockIdx.y * ntiles + blockIdx.x) * NW + wid) * E, carry);
    if (wl == NW - 1) {
        cmap_append_any<D>(carry, SrcMem<D>{wt + wid * E});
        const int pos = dir > 0 ? blockIdx.x : ntiles - 1 - blockIdx.x;
        if (li == 0) cmap_store<D>(totals + ((size_t)blockIdx.y * ntiles + pos) * E, carry);
    }
}

// exclusive scan of the tile totals, in place: one workgroup per direction, chunks of kBlock tiles
template <int D>
__global__ __launch_bounds__(kBlock) void k_mvc_scan_totals(int ntiles, double *__restrict__ totals, double *__restrict__ block_total) {
    constexpr int E = CMap<D>::ND + 1;
    __shared__ double wt[(kBlock / 64) * E];
    __shared__ double carry_s[E];
    const int tid = threadIdx.x;
    totals += (size_t)blockIdx.x * ntiles * E;
    if (tid == 0) cmap_store<D>(carry_s, cmap_identity<D>());
    __syncthreads();
#pragma unroll 1
    for (int chunk = 0; chunk < ntiles; chunk += kBlock) {
        const int j = chunk + tid;
        CMap<D> t = j < ntiles ? cmap_load<D>(totals + (size_t)j * E) : cmap_identity<D>();
        CMap<D> total = cmap_identity<D>();
        mvc_wg_scan<D, true>(t, total, wt, tid, 1);
        const CMap<D> carry = cmap_load<D>(carry_s);
        if (j < ntiles) cmap_store<D>(totals + (size_t)j * E, cmap_compose<D>(carry, t));
        __syncthreads();
        if (tid == kBlock - 1) cmap_store<D>(carry_s, cmap_compose<D>(carry, total));
        __syncthreads();
    }
    // the composed map of the whole direction (what a partition exchanges: cx_chain_block_maps)
    if (block_total && tid < E) block_total[(size_t)blockIdx.x * E + tid] = carry_s[tid];
}

template <int D>
__device__ __forceinline__ void marg_store(double *__restrict__ marg, int64_t, int64_t v, const Msg<D> &nat) {
    slot_store_nt<D>(marg, (int)v, mv_to_moment<D>(nat));      // marginals: the messages' pair form, indexed by the variable
}

// grid (ntiles, 2): the forward and the backward walks are independent of each other.
// flags & 1: marginals are wanted (the backward walk writes those of the first variable of every path itself; all others come
// from alpha + gamma in k_mvc_marg_out).  flags & 2: alpha_l / beta_l also go to their SELL slots f2v[to_slot[l]] / f2v[from_slot[l]].
// PF (round 4): the walks fetch the NEXT step's side information, head flag and table index while the current step's rule runs — and
// the first step's before the tile carry is composed; a step of the plain form is a memory round trip, then the rule, then the
// stores, at the two waves per SIMD a chain of 1e6 links fills.  (The plain form, PF = false, is kept for CX_MVC_PREFETCH=0 A/B runs;
// until round 5 it was held to 128 registers — four waves per SIMD — and carried 112 B of scratch for d = 4; a chain of 1e6 links
// fills two waves per SIMD whatever the register count, so both forms now take what they need: 202 / 168 registers, no scratch.)
template <int D, bool GT, bool PF>
__global__ __launch_bounds__(kBlock, 2) void k_mvc_apply(MvcArgs A, int K, const double *__restrict__ excl, double *__restrict__ f2v,
                                                               double *__restrict__ marg, int flags) {
    constexpr int E = CMap<D>::ND + 1;
    using M = CMap<D>;
    __shared__ double tab_s[GT ? 1 : kMvcTabLds * 3 * D * D];
    __shared__ double red_s[64 * E];          // the tile carry's reduction: one partial product per lane of the first wave
    const int tid = threadIdx.x, ntiles = gridDim.x, dir = blockIdx.y ? -1 : 1;
    if (!GT) mvc_load_tabs<D>(A, tab_s, tid);
    const int64_t gid = (int64_t)blockIdx.x * kBlock + tid, nthreads = (int64_t)ntiles * kBlock;
    const int l0 = (int)gid * K;
    const int64_t il0 = (int64_t)blockIdx.x * kBlock * K + tid;
    const int pos = dir > 0 ? blockIdx.x : ntiles - 1 - blockIdx.x;
    // PF: what the first step of the walk reads, issued before the carry
    Msg<D> nx = msg_nan<D>();
    int nx_head = 0, nx_tab = 0;
    if (PF) {
        const int k = dir > 0 ? 0 : K - 1, l = l0 + k;
        // (backward: a thread whose run is ragged starts at its last existing link — found below; prefetch only the common case)
        if (l < A.nlinks) {
            const int64_t il = il0 + (int64_t)k * kBlock;
            nx = dir > 0 ? slot_load<D, true>(A.side_l, (int)il) : mvc_side_right<D>(A, l, il, K);
            nx_head = dir > 0 ? A.head_fwd[l] : A.head_bwd[l];
            nx_tab = dir > 0 ? A.tab_fwd[l] : A.tab_bwd[l];
        }
    }
    // ---- the tile carry: the composition of the tile totals before this tile in the direction's scan order, by the workgroup's FIRST
    // WAVE (a launch of one workgroup per direction used to scan them, 25 us of pure latency): every lane
    // composes its run of consecutive totals straight from memory, then the 64 partial products are reduced in order through LDS —
    // the operand of a composition is always read from memory, so nothing but the accumulator lives in registers
    if (tid < 64) {
        const double *T = excl + (size_t)blockIdx.y * ntiles * E;
        const int per = (pos + 63) / 64, j0 = tid * per, j1 = min(pos, j0 + per);
        CMap<D> acc = cmap_identity<D>();
#pragma unroll 1
        for (int j = j0; j < j1; j++) cmap_append_any<D>(acc, SrcMem<D>{T + (size_t)j * E});
        cmap_store<D>(red_s + tid * E, acc);
#pragma unroll 1
        for (int st = 1; st < 64; st <<= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS accesses complete in order; hipcc must not move them)
            if ((tid & (2 * st - 1)) == 0 && tid + st < 64) {
                cmap_append_any<D>(acc, SrcMem<D>{red_s + (tid + st) * E});
                cmap_store<D>(red_s + tid * E, acc);
            }
        }
    }
    __syncthreads();
    if (l0 >= A.nlinks) return;
    // the map from the start of the scan to this thread's first link: tile carry, then the carry into the thread's wave, then the
    // inclusive prefix of the logically previous lane of the wave (none for the wave's first lane)
    constexpr int NW = kBlock / 64;
    const int lane = tid & 63, wid = tid >> 6, li = dir > 0 ? lane : 63 - lane;
    CMap<D> inc = cmap_load<D>(red_s);
    cmap_append_any<D>(inc, SrcMem<D>{A.wave_carry + (((size_t)blockIdx.y * ntiles + blockIdx.x) * NW + wid) * E});
    if (li > 0) cmap_append_any<D>(inc, SrcStrided<D>{A.prefix + (size_t)blockIdx.y * E * nthreads + (gid - dir), nthreads});
    // every prefix that reaches back to the first link of a path is a constant map: the message it produces from nothing
    Msg<D> cur = msg_nan<D>();
    if (inc.flags & kMapSeg) {
#pragma unroll
        for (int i = 0; i < D; i++) cur.eta[i] = inc.v[M::oS + i];
#pragma unroll
        for (int i = 0; i < Msg<D>::NT; i++) cur.lam[i] = inc.v[M::oC + i];
    }
    const bool store_msgs = (flags & 2) != 0, write_marg = (flags & 1) != 0;
    if (dir > 0) {
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            const int l = l0 + k;
            if (l >= A.nlinks) break;
            const int64_t il = il0 + (int64_t)k * kBlock;
            Msg<D> in;
            int head, tab;
            if (PF) {
                in = nx; head = nx_head; tab = nx_tab;
                if (k + 1 < K && l + 1 < A.nlinks) {
                    nx = slot_load<D, true>(A.side_l, (int)(il + kBlock));
                    nx_head = A.head_fwd[l + 1]; nx_tab = A.tab_fwd[l + 1];
                }
            } else {
                in = slot_load<D, true>(A.side_l, (int)il);
                head = A.head_fwd[l]; tab = A.tab_fwd[l];
            }
            if (!head) msg_add<D>(in, cur);
            cur = mv_rule<D, false>(in, mvc_tab<D, GT>(A, tab_s, tab));
            slot_store<D>(A.alpha, (int)il, cur);
            if (store_msgs && !__builtin_isnan(cur.lam[0])) slot_store<D>(f2v, A.to_slot[l], cur);
        }
    } else {
        bool have = PF && l0 + K - 1 < A.nlinks;        // the prefetch at the top was for step K - 1
#pragma unroll 1
        for (int k = K - 1; k >= 0; k--) {
            const int l = l0 + k;
            if (l >= A.nlinks) continue;
            const int64_t il = il0 + (int64_t)k * kBlock;
            Msg<D> in;                                               // what the right variable hears from everybody but this link
            int head, tab;
            if (PF) {
                if (!have) { nx = mvc_side_right<D>(A, l, il, K); nx_head = A.head_bwd[l]; nx_tab = A.tab_bwd[l]; }
                in = nx; head = nx_head; tab = nx_tab;
                have = k > 0;
                if (have) { nx = mvc_side_right<D>(A, l - 1, il - kBlock, K); nx_head = A.head_bwd[l - 1]; nx_tab = A.tab_bwd[l - 1]; }
            } else {
                in = mvc_side_right<D>(A, l, il, K);
                head = A.head_bwd[l]; tab = A.tab_bwd[l];
            }
            if (!head) msg_add<D>(in, cur);
            slot_store<D>(A.gamma, (int)il, in);
            cur = mv_rule<D, false>(in, mvc_tab<D, GT>(A, tab_s, tab));
            if (store_msgs && !__builtin_isnan(cur.lam[0])) slot_store<D>(f2v, A.from_slot[l], cur);
            if (write_marg && A.head_fwd[l]) {                        // the first variable of a path hears no alpha
                const int p = A.link_pos[l];
                Msg<D> tot = msg_load<D>(A.side, A.npos, p);
                msg_add<D>(tot, cur);
                marg_store<D>(marg, A.nv, A.pos_var[p], tot);
            }
        }
    }
}

// Marginal of every link's right variable: (alpha + gamma) to moment form, then from the interleaved order of the walks to
// the marginals' place (pair form by variable) through LDS.  A tile goes through in slabs of W = 512 / K threads (all K steps of each: <= 512 links, 57 KB
// for d = 4): the reads are runs of W doubles per step and component, the writes runs of W K variables per component.
__host__ __device__ inline int mvc_slab_threads(int K) { return K >= 512 ? 1 : (512 / K > kBlock ? kBlock : 512 / K); }
// k_mvc_side_links takes half-size slabs: 30 KB of LDS per workgroup instead of 60 — five workgroups per compute unit in flight
__host__ __device__ inline int mvc_side_slab_threads(int K) { return K >= 256 ? 1 : 256 / K; }
__host__ __device__ inline int mvc_slab_pitch(int K) { return mvc_slab_threads(K) + (((K & (K - 1)) == 0 && K <= 32) ? 32 / K : 1); }   // conflict-free column reads for K | 32

// One workgroup per SLAB of a tile: the W threads t0 .. t0 + W - 1 of the tile (W * K <= 512 links, contiguous in chain order) —
// grid = tiles x slabs, so the pass has thousands of small workgroups in flight instead of one per tile looping over its slabs
// behind barriers (244 workgroups on 256 compute units: 83 us for 348 MB).
template <int D>
__global__ __launch_bounds__(kBlock) void k_mvc_marg_out(int nlinks, int K, int64_t il_stride, int nv, const int32_t *__restrict__ link_pos,
                                                         const int32_t *__restrict__ pos_var, const double *__restrict__ alpha,
                                                         const double *__restrict__ gamma, double *__restrict__ marg) {
    constexpr int NC = Msg<D>::NC;
    extern __shared__ double buf[];          // [NC][K][Wp]
    const int tid = threadIdx.x, W = mvc_slab_threads(K), Wp = mvc_slab_pitch(K);
    const int slabs = (kBlock + W - 1) / W, tile = blockIdx.x / slabs, t0 = (blockIdx.x - tile * slabs) * W;
    const int64_t base = (int64_t)tile * kBlock * K;      // first link of the tile == its first interleaved index
    const int wn = min(W, kBlock - t0), nitems = wn * K;
    for (int i = tid; i < nitems; i += kBlock) {
        const int k = i / wn, tt = i - k * wn;
        if (base + (int64_t)(t0 + tt) * K + k >= nlinks) continue;
        const int64_t il = base + (int64_t)k * kBlock + t0 + tt;
        Msg<D> tot = slot_load<D, true>(alpha, (int)il);
        msg_add<D>(tot, slot_load<D, true>(gamma, (int)il));
        const Msg<D> mo = mv_to_moment<D>(tot);
#pragma unroll
        for (int c = 0; c < D; c++) buf[(c * K + k) * Wp + tt] = mo.eta[c];
#pragma unroll
        for (int c = 0; c < Msg<D>::NT; c++) buf[((D + c) * K + k) * Wp + tt] = mo.lam[c];
    }
    __syncthreads();
    for (int e = tid; e < nitems; e += kBlock) {
        const int64_t l = base + (int64_t)t0 * K + e;       // links in chain order: thread t0 + e / K, step e % K
        if (l >= nlinks) break;
        const int64_t v = pos_var[link_pos[l] + 1];
        const int src = (e % K) * Wp + e / K;
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v *dst = reinterpret_cast<d2v *>(marg + slot_offset<D>((int)v));
#pragma unroll
        for (int q = 0; q < MsgStore<D>::NCP; q++) {
            d2v t2;
            t2.x = buf[(2 * q) * K * Wp + src];
            t2.y = (2 * q + 1 < NC) ? buf[(2 * q + 1) * K * Wp + src] : 0.0;
            __builtin_nontemporal_store(t2, dst + q * kBlock);
        }
    }
}

// Marginals ON DEMAND for a few variables (compute_marginals_in_sweep == 2, cx_get_marginals of less than the whole chain): the
// marginal of the right variable of link l is alpha_l + gamma_l, both still in the walks' order; a variable that is not the right end
// of a link (the first of a path, an isolated one) had its marginal written by the sweep itself.  Rows come out as k_mv_gather's.
template <int D>
__global__ __launch_bounds__(kBlock) void k_mvc_marg_gather(int64_t n, const int32_t *__restrict__ vars, const int32_t *__restrict__ var_link, int K,
                                                            const double *__restrict__ alpha, const double *__restrict__ gamma,
                                                            const double *__restrict__ marg, int64_t marg_stride, int ncs, double *__restrict__ val) {
    constexpr int NC = Msg<D>::NC;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int v = vars[i], l = var_link[v];
    (void)marg_stride; (void)ncs;
    Msg<D> mo;
    if (l < 0) {
        mo = slot_load<D, false>(marg, v);          // written by the sweep itself, in the marginals' pair form by variable
    } else {
        const int64_t g = (int64_t)l / K, il = (g / kBlock) * (int64_t)kBlock * K + (int64_t)(l % K) * kBlock + (g % kBlock);
        Msg<D> tot = slot_load<D, true>(alpha, (int)il);
        msg_add<D>(tot, slot_load<D, true>(gamma, (int)il));
        mo = mv_to_moment<D>(tot);
    }
#pragma unroll
    for (int c = 0; c < D; c++) val[i * NC + c] = mo.eta[c];
#pragma unroll
    for (int c = 0; c < Msg<D>::NT; c++) val[i * NC + D + c] = mo.lam[c];
}

// side information of every chain position (runs after data, stored messages or rule parameters changed); a position without
// links (an isolated non-observed variable) gets its marginal here: the product of everything it hears
template <int D>
__device__ __forceinline__ Msg<D> mvc_side_sum(int v, int s0, int s1, const int32_t *__restrict__ vbase, const uint8_t *__restrict__ vinfo,
                                               const double *__restrict__ f2v) {
    const int b = vbase[v], deg = vinfo[v] & kDegMask;
    Msg<D> acc = msg_zero<D>();
    for (int k = 0; k < deg; k++) {
        const int slot = b + k * kBlock;
        if (slot == s0 || slot == s1) continue;
        msg_add<D>(acc, slot_load<D>(f2v, slot));
    }
    return acc;
}

// positions first .. npos - 1 (the positions WITHOUT links — isolated non-observed variables — sit behind the paths' positions;
// those of the paths get their side sums from k_mvc_side_links)
template <int D>
__global__ __launch_bounds__(kBlock) void k_mvc_side(int first, int npos, int64_t nslots, int nv, const int32_t *__restrict__ pos_var,
                                                     const int32_t *__restrict__ skip0, const int32_t *__restrict__ skip1,
                                                     const int32_t *__restrict__ vbase, const uint8_t *__restrict__ vinfo,
                                                     const double *__restrict__ f2v, double *__restrict__ side,
                                                     double *__restrict__ marg, int write_marg) {
    const int i = first + blockIdx.x * kBlock + threadIdx.x;
    if (i >= npos) return;
    const int v = pos_var[i], s0 = skip0[i], s1 = skip1[i];
    const Msg<D> acc = mvc_side_sum<D>(v, s0, s1, vbase, vinfo, f2v);
    msg_store<D>(side, npos, i, acc);
    if (write_marg && s0 < 0 && s1 < 0) marg_store<D>(marg, nv, v, acc);
}

// The side sums of the paths' positions — what a chain variable hears from everybody who is not on the chain — computed, stored by
// position (ends of paths), and stored as each link's LEFT end in the interleaved order the scan kernels read.  One workgroup per
// (tile, slab): a slab is W = 256 / K threads with all K links of each; its links are consecutive and link_pos grows by one per link
// (by two across a path boundary), so its positions are one range.  Phase 1 sums the range's positions into LDS (and, for the ends
// of the paths, into the by-position array); phase 2 reads the left end of every link from LDS and writes contiguous runs
// of the interleaved array.  A slab with so many path boundaries that its range overflows the buffer sums the overflow on the spot.
// (Until the end of round 3 the sums were a kernel of their own and this one re-read them: 40 + 100 us for a 1M-state chain.)
template <int D>
__global__ __launch_bounds__(kBlock) void k_mvc_side_links(int nlinks, int npos, int npos_linked, int K, const int32_t *__restrict__ link_pos,
                                                           const int32_t *__restrict__ pos_var, const int32_t *__restrict__ skip0,
                                                           const int32_t *__restrict__ skip1, const int32_t *__restrict__ vbase,
                                                           const uint8_t *__restrict__ vinfo, const double *__restrict__ f2v,
                                                           double *__restrict__ side, double *__restrict__ side_l) {
    constexpr int NC = Msg<D>::NC;
    extern __shared__ double buf[];          // [NC][span + span / 32 + 1]: position j of the slab at j + j / 32 (stride-K reads spread over the banks)
    const int tid = threadIdx.x, W = mvc_side_slab_threads(K);
    const int span = W * K + 8, pitch = span + span / 32 + 1;
    const int nslab = (kBlock + W - 1) / W;                 // grid: one workgroup per (tile, slab) — a tile alone would leave the chip under one wave per SIMD
    const int64_t base = (int64_t)(blockIdx.x / nslab) * kBlock * K;
    const int t0 = (blockIdx.x % nslab) * W;
    const int wn = min(W, kBlock - t0), nitems = wn * K;
    const int64_t L0 = base + (int64_t)t0 * K;
    if (L0 >= nlinks) return;
    const int p0 = link_pos[L0];
    for (int j = tid; j < span; j += kBlock) {
        const int p = p0 + j;
        Msg<D> acc = msg_zero<D>();
        if (p < npos_linked) {
            const int s0 = skip0[p], s1 = skip1[p];
            acc = mvc_side_sum<D>(pos_var[p], s0, s1, vbase, vinfo, f2v);
            // by position only where somebody reads it that way: the ends of the paths (the marginal of a path's first variable in
            // k_mvc_apply, the end variables' side sums of cx_chain_block_maps) — a position inside a path lacks no chain neighbour
            if (s0 < 0 || s1 < 0) msg_store<D>(side, npos, p, acc);
        }
        const int jj = j + j / 32;
#pragma unroll
        for (int c = 0; c < D; c++) buf[c * pitch + jj] = acc.eta[c];
#pragma unroll
        for (int c = 0; c < Msg<D>::NT; c++) buf[(D + c) * pitch + jj] = acc.lam[c];
    }
    __syncthreads();
    for (int i = tid; i < nitems; i += kBlock) {
        const int k = i / wn, tt = i - k * wn;
        const int64_t l = L0 + (int64_t)tt * K + k;
        if (l >= nlinks) continue;
        const int p = link_pos[l], q = p - p0;
        Msg<D> a;
        if (q < span) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const double va = buf[c * pitch + q + q / 32];
                if (c < D) a.eta[c] = va; else a.lam[c - D] = va;
            }
        } else {                                            // beyond the buffer: nobody else is sure to cover this position
            a = mvc_side_sum<D>(pos_var[p], skip0[p], skip1[p], vbase, vinfo, f2v);
            if (skip0[p] < 0 || skip1[p] < 0) msg_store<D>(side, npos, p, a);      // an end of a path (the heavy-path plan keeps a head's next slot in skip0, -1 in skip1)
        }
        if (q + 1 >= span && skip1[p + 1] < 0)              // ... nor the last position of a path that ends out there
            msg_store<D>(side, npos, p + 1, mvc_side_sum<D>(pos_var[p + 1], skip0[p + 1], skip1[p + 1], vbase, vinfo, f2v));
        const int il = (int)(base + (int64_t)k * kBlock + t0 + tt);
        slot_store<D>(side_l, il, a);
    }
}

// ------------------------------------------------------------------------------------------------ host side
// links per thread of the scan; CX_MVC_K overrides.  Chosen when the chains are (re)built: the interleaved buffers are laid out for it.
// Long chains: 16 (map compositions per link fall as 1 + 9 / K; the walks are one rule per link whatever K); short chains fewer,
// so that the grid still has a few workgroups per compute unit.
int mvc_links_per_thread(int64_t nlinks) {
    const char *e = getenv("CX_MVC_K");
    const int v = e ? atoi(e) : 0;
    if (v >= 1 && v <= 32) return v;
    int64_t k = 1;
    while (k < 16 && 2 * k <= nlinks / ((int64_t)kBlock * 128)) k *= 2;      // a power of two (the marginal transpose's LDS reads are conflict-free then)
    return (int)k;
}

int64_t mvc_ntiles(int64_t nlinks, int K) {
    const int64_t per = (int64_t)kBlock * K;
    return std::max<int64_t>((nlinks + per - 1) / per, 1);
}

static int mvc_map_doubles(int dim) { return 2 * (dim * (dim + 1) / 2) + dim * dim + 2 * dim + 1; }

size_t mvc_totals_doubles(int dim, int64_t nlinks, int K) { return (size_t)2 * (size_t)mvc_ntiles(nlinks, K) * mvc_map_doubles(dim); }
size_t mvc_prefix_doubles(int dim, int64_t nlinks, int K) { return (size_t)2 * (size_t)mvc_ntiles(nlinks, K) * kBlock * mvc_map_doubles(dim); }
size_t mvc_wave_carry_doubles(int dim, int64_t nlinks, int K) { return (size_t)2 * (size_t)mvc_ntiles(nlinks, K) * (kBlock / 64) * mvc_map_doubles(dim); }

// side sums by position, then by link in the interleaved order (after data, stored messages or rule tables changed)
void mvc_launch_side(cx_handle *h, bool write_marg) {
    const int npos = (int)h->chain_npos, nlinks = (int)h->chain_nlinks, K = h->mvc_K;
    if (npos == 0) return;
    const int linked = nlinks ? (int)h->chain_npos_linked : 0, alone = npos - linked;
    const dim3 g((unsigned)std::max((alone + kBlock - 1) / kBlock, 1)), b(kBlock);
    const dim3 gl((unsigned)(mvc_ntiles(nlinks, K) * ((kBlock + mvc_side_slab_threads(K) - 1) / mvc_side_slab_threads(K))));
    const int span = mvc_side_slab_threads(K) * K + 8;
    const size_t lds = (size_t)h->nc * (span + span / 32 + 1) * sizeof(double);
#define CX_MVC(DD)                                                                                                                           \
    do {                                                                                                                                     \
        if (alone > 0)                                                                                                                       \
            hipLaunchKernelGGL((k_mvc_side<DD>), g, b, 0, h->stream, linked, npos, h->nslots, (int)h->nv, h->d_chain_pos_var,                \
                               h->d_chain_skip0, h->d_chain_skip1, h->d_vbase, h->d_vinfo, h->d_mv_f2v, h->d_mvc_side, h->d_mv_marg,         \
                               write_marg ? 1 : 0);                                                                                          \
        if (nlinks) hipLaunchKernelGGL((k_mvc_side_links<DD>), gl, b, lds, h->stream, nlinks, npos, linked, K, h->d_chain_link_pos,          \
                                       h->d_chain_pos_var, h->d_chain_skip0, h->d_chain_skip1, h->d_vbase, h->d_vinfo, h->d_mv_f2v,          \
                                       h->d_mvc_side, h->d_mvc_side_l);                                                    \
    } while (0)
    if (h->cfg.dim == 2) CX_MVC(2);
    else if (h->cfg.dim == 3) CX_MVC(3);
    else CX_MVC(4);
#undef CX_MVC
}

template <int D>
static void mvc_marg_out_t(cx_handle *h, const MvcArgs &A, int K) {
    const int ntiles = (int)mvc_ntiles(A.nlinks, K);
    hipLaunchKernelGGL((k_mvc_marg_out<D>), dim3((unsigned)(ntiles * ((kBlock + mvc_slab_threads(K) - 1) / mvc_slab_threads(K)))), dim3(kBlock), (size_t)Msg<D>::NC * K * mvc_slab_pitch(K) * sizeof(double), h->stream, A.nlinks, K,
                       A.il_stride, A.nv, A.link_pos, A.pos_var, A.alpha, A.gamma, h->d_mv_marg);
}

template <int D, bool GT>
static void mvc_launch_t(cx_handle *h, const MvcArgs &A, int K, int flags, bool scan) {
    const int ntiles = (int)mvc_ntiles(A.nlinks, K);
    // (the tile totals stay as k_mvc_totals left them: the walks' workgroups compose their own carry from them — also when only the
    // walks run again, scan == false)
    if (scan) hipLaunchKernelGGL((k_mvc_totals<D, GT>), dim3(ntiles, 2), dim3(kBlock), 0, h->stream, A, K, h->d_mvc_totals);
    // CX_MVC_PREFETCH=0: the walks without the next step's loads in flight (A/B)
    static const bool pf = [] { const char *e = getenv("CX_MVC_PREFETCH"); return !(e && e[0] == '0'); }();
    if (pf) hipLaunchKernelGGL((k_mvc_apply<D, GT, true>), dim3(ntiles, 2), dim3(kBlock), 0, h->stream, A, K, h->d_mvc_totals, h->d_mv_f2v, h->d_mv_marg, flags);
    else hipLaunchKernelGGL((k_mvc_apply<D, GT, false>), dim3(ntiles, 2), dim3(kBlock), 0, h->stream, A, K, h->d_mvc_totals, h->d_mv_f2v, h->d_mv_marg, flags);
    if ((flags & 1) && !(flags & 4)) mvc_marg_out_t<D>(h, A, K);
}

// One sweep: all forward and backward chain messages and, with write_marg, the chain variables' marginals.
// store_msgs: the messages also go to their slots of d_mv_f2v.  scan = false: the thread prefixes and tile carries of the last sweep
// are still valid (nothing changed since): only the walks run — how the messages are materialised on demand.
// defer_marg (with write_marg): the walks leave alpha and gamma — and write the marginal of every path's first variable, which hears no
// alpha — but the pass that adds them up, converts to moment form and moves them to the marginals' place is left to mvc_launch_marg_out.
static MvcArgs mvc_args(cx_handle *h, int collapse_heads) {
    const int K = h->mvc_K;
    return MvcArgs{(int)h->chain_nlinks, (int)h->chain_npos, (int)h->nv, (int)(2 * h->ptab_sets), h->nslots, h->d_chain_link_pos, h->d_chain_from,
                   h->d_chain_to, h->d_chain_tab_fwd, h->d_chain_tab_bwd, h->d_chain_head_fwd, h->d_chain_head_bwd, h->d_chain_pos_var,
                   h->d_mvc_side, h->d_mvc_side_l, h->d_mvc_alpha, h->d_mvc_gamma, h->d_mvc_prefix, h->d_mvc_wave_carry,
                   mvc_ntiles(h->chain_nlinks, K) * kBlock * K, collapse_heads, h->d_ptab};
}

void mvc_launch_marg_gather(cx_handle *h, const int32_t *d_vars, int64_t n, double *d_val) {
    if (n == 0) return;
    const dim3 g((unsigned)((n + kBlock - 1) / kBlock)), b(kBlock);
    const int64_t stride = h->nslices * kBlock;
#define CX_MG(DD) hipLaunchKernelGGL((k_mvc_marg_gather<DD>), g, b, 0, h->stream, n, d_vars, h->d_mvc_var_link, h->mvc_K, h->d_mvc_alpha, h->d_mvc_gamma, \
                                     h->d_mv_marg, stride, (int)h->ncs, d_val)
    if (h->cfg.dim == 2) CX_MG(2);
    else if (h->cfg.dim == 3) CX_MG(3);
    else CX_MG(4);
#undef CX_MG
}

void mvc_launch_marg_out(cx_handle *h) {
    if (h->chain_nlinks == 0) return;
    const MvcArgs A = mvc_args(h, 1);
    if (h->cfg.dim == 2) mvc_marg_out_t<2>(h, A, h->mvc_K);
    else if (h->cfg.dim == 3) mvc_marg_out_t<3>(h, A, h->mvc_K);
    else mvc_marg_out_t<4>(h, A, h->mvc_K);
}

void mvc_launch_scan(cx_handle *h, bool write_marg, bool store_msgs, bool scan, bool defer_marg) {
    if (h->chain_nlinks == 0) return;
    const int K = h->mvc_K;
    MvcArgs A{(int)h->chain_nlinks, (int)h->chain_npos, (int)h->nv, (int)(2 * h->ptab_sets), h->nslots, h->d_chain_link_pos, h->d_chain_from,
              h->d_chain_to, h->d_chain_tab_fwd, h->d_chain_tab_bwd, h->d_chain_head_fwd, h->d_chain_head_bwd, h->d_chain_pos_var,
              h->d_mvc_side, h->d_mvc_side_l, h->d_mvc_alpha, h->d_mvc_gamma, h->d_mvc_prefix, h->d_mvc_wave_carry,
              mvc_ntiles(h->chain_nlinks, K) * kBlock * K, 1, h->d_ptab};
    const bool gt = A.ntab > kMvcTabLds;
    const int flags = (write_marg ? 1 : 0) | (store_msgs ? 2 : 0) | (defer_marg ? 4 : 0);
#define CX_MVC(DD) do { if (gt) mvc_launch_t<DD, true>(h, A, K, flags, scan); else mvc_launch_t<DD, false>(h, A, K, flags, scan); } while (0)
    if (h->cfg.dim == 2) CX_MVC(2);
    else if (h->cfg.dim == 3) CX_MVC(3);
    else CX_MVC(4);
#undef CX_MVC
}

// CX_SCHED_TREE over heavy paths, dim 2 .. 4 (cx_tree_plan.h: build_hp): the scan of ONE light depth — links [link_lo, link_lo + nlinks)
// of the plan's arrays (positions stay global: link_pos, the by-position side sums and pos_var are indexed by them; pos_hi = one past the
// depth's last position).  `skip1`: the second skipped slot of every position (on the way up a head also skips its slot towards its
// parent).  final: both directions are exact — the messages go to their slots and the marginals of the paths' variables are written;
// otherwise the messages only (the way up reads those towards the heads).  The interleaved buffers are scratch shared by all depths.
void mvc_launch_scan_range(cx_handle *h, int64_t npos_total, int64_t pos_hi, int64_t link_lo, int64_t nlinks, int K, const int32_t *skip1, bool final) {
    if (nlinks <= 0) return;
    const dim3 b(kBlock);
    const dim3 gl((unsigned)(mvc_ntiles(nlinks, K) * ((kBlock + mvc_side_slab_threads(K) - 1) / mvc_side_slab_threads(K))));
    const int span = mvc_side_slab_threads(K) * K + 8;
    const size_t lds = (size_t)h->nc * (span + span / 32 + 1) * sizeof(double);
#define CX_MVC(DD) hipLaunchKernelGGL((k_mvc_side_links<DD>), gl, b, lds, h->stream, (int)nlinks, (int)npos_total, (int)pos_hi, K, h->d_chain_link_pos + link_lo, \
                                      h->d_chain_pos_var, h->d_chain_skip0, skip1, h->d_vbase, h->d_vinfo, h->d_mv_f2v, h->d_mvc_side, h->d_mvc_side_l)
    if (h->cfg.dim == 2) CX_MVC(2);
    else if (h->cfg.dim == 3) CX_MVC(3);
    else CX_MVC(4);
#undef CX_MVC
    MvcArgs A{(int)nlinks, (int)npos_total, (int)h->nv, (int)(2 * h->ptab_sets), h->nslots, h->d_chain_link_pos + link_lo, h->d_chain_from + link_lo,
              h->d_chain_to + link_lo, h->d_chain_tab_fwd + link_lo, h->d_chain_tab_bwd + link_lo, h->d_chain_head_fwd + link_lo, h->d_chain_head_bwd + link_lo,
              h->d_chain_pos_var, h->d_mvc_side, h->d_mvc_side_l, h->d_mvc_alpha, h->d_mvc_gamma, h->d_mvc_prefix, h->d_mvc_wave_carry,
              mvc_ntiles(nlinks, K) * kBlock * K, 1, h->d_ptab};
    const bool gt = A.ntab > kMvcTabLds;
    const int flags = final ? 3 : 2;
#define CX_MVC(DD) do { if (gt) mvc_launch_t<DD, true>(h, A, K, flags, true); else mvc_launch_t<DD, false>(h, A, K, flags, true); } while (0)
    if (h->cfg.dim == 2) CX_MVC(2);
    else if (h->cfg.dim == 3) CX_MVC(3);
    else CX_MVC(4);
#undef CX_MVC
}

// The composed forward and backward maps of the handle's one path, heads NOT collapsed: totals + scan of the totals only, the
// two maps end up in h->d_mvc_block (2 x (ND + 1) doubles).  The stored thread prefixes are overwritten: the caller materialises
// the last sweep's messages first (mv_ensure_chain_msgs) and marks the side sums dirty.
template <int D, bool GT>
static void mvc_block_maps_t(cx_handle *h, MvcArgs A, int K) {
    A.collapse_heads = 0;
    const int ntiles = (int)mvc_ntiles(A.nlinks, K);
    hipLaunchKernelGGL((k_mvc_totals<D, GT>), dim3(ntiles, 2), dim3(kBlock), 0, h->stream, A, K, h->d_mvc_totals);
    hipLaunchKernelGGL((k_mvc_scan_totals<D>), dim3(2), dim3(kBlock), 0, h->stream, ntiles, h->d_mvc_totals, h->d_mvc_block);
}

void mvc_launch_block_maps(cx_handle *h) {
    if (h->chain_nlinks == 0) return;
    const int K = h->mvc_K;
    MvcArgs A{(int)h->chain_nlinks, (int)h->chain_npos, (int)h->nv, (int)(2 * h->ptab_sets), h->nslots, h->d_chain_link_pos, h->d_chain_from,
              h->d_chain_to, h->d_chain_tab_fwd, h->d_chain_tab_bwd, h->d_chain_head_fwd, h->d_chain_head_bwd, h->d_chain_pos_var,
              h->d_mvc_side, h->d_mvc_side_l, h->d_mvc_alpha, h->d_mvc_gamma, h->d_mvc_prefix, h->d_mvc_wave_carry,
              mvc_ntiles(h->chain_nlinks, K) * kBlock * K, 0, h->d_ptab};
    const bool gt = A.ntab > kMvcTabLds;
#define CX_MVC(DD) do { if (gt) mvc_block_maps_t<DD, true>(h, A, K); else mvc_block_maps_t<DD, false>(h, A, K); } while (0)
    if (h->cfg.dim == 2) CX_MVC(2);
    else if (h->cfg.dim == 3) CX_MVC(3);
    else CX_MVC(4);
#undef CX_MVC
}

}  // namespace cx
