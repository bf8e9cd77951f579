// cx_mv.hip — multivariate (d-dimensional) linear-Gaussian messages, small d (2, 4): "batched d×d updates in
// registers, no MFMA" (BASELINE.json config 3: d = 4 state-space chain).
//
// The reference has no d-dimensional rule anywhere (closest: the 2×2 MvNormalMeanPrecision joint of
// test/inference_engine_tests.jl:949-979); the rules below are the d-dimensional form of its scalar test rules
// (test/inference_engine_tests.jl:385-432): products of Gaussians and a linear-Gaussian factor
// x_out = A x_in + N(0, Q).  Parity for dim > 1 is therefore against exact block-tridiagonal solves and the numpy
// restatement kept with the tests ("parity unpinned" by the reference itself — DESIGN.md §3).
//
// Storage: natural form (eta = Lambda mu, Lambda = Sigma^-1), symmetric Lambda packed (upper triangle), over the same SELL-256
// slots as the scalar path, block-major in 16-byte pairs (cx_mv_core.h: slot_load / slot_store): lane <-> variable accesses are
// one contiguous kilobyte per wave and plane.  d = 4: 14 doubles = 112 B per message (the dense 4 + 16 layout would be 160 B).
//
// Factor rule with the receiving edge's precomputed triple (P, B, C) — host side, from (A, Q), Qi = Q^-1:
//   receiver = out (forward):  P = A' Qi A,  B = Qi A,   C = Qi
//   receiver = in  (backward): P = Qi,       B = A' Qi,  C = A' Qi A
//   M = Lambda_in + P = L L';   Y = L^-1 B';   Lambda_out = C - Y' Y;   eta_out = Y' (L^-1 eta_in)
//   point-mass input y (observed variable):  Lambda_out = C,  eta_out = B y.
// One Cholesky + two triangular solves per message, all in registers, every loop unrolled on the template D.

#include <cmath>
#include <limits>
#include <vector>

#include "cx_internal.h"
#include "cx_mv_core.h"

namespace cx {

// Msg<D>, chol, fwd_solve, mv_rule, mv_to_moment: cx_mv_core.h


constexpr int kMvDeg = 4;  // variables of higher degree are refused at graph creation for dim > 1 (this round)

// The fused sweep for small d: thread = variable.
// Latency: a wave first issues EVERY load it will need — all incoming messages, the partner and rule-selector words of all
// its edges — under workgroup-uniform branches (k < slice width), and only then waits; the (P, B, C) rule tables of up to
// kTabLds parameter-set/direction pairs sit in LDS.  The earlier form loaded partner[], then spdir[], then the 48 table
// doubles, one dependent round trip after the other and per edge: ≈10 serialised memory latencies per wave (C3: 0.19 ms
// per sweep; PMC: VALU busy 6 %, waves waiting on memory 65 % of their cycles).
// The leave-one-out sums add the messages in ascending neighbour order, the reference's left fold order.
constexpr int kTabLds = 8;   // parameter-set/direction pairs kept in LDS (3 d*d matrices each); more fall back to global memory

// DEG: the largest variable degree of the graph (3, 4, or 8 for anything wider: eight messages in registers is one wave per SIMD and some
// scratch for d = 4 — the instance exists so that such graphs run, not to be fast).  A state-space chain has degree 3: three incoming messages instead of four
// are 28 registers less for d = 4 — 168 VGPRs, THREE waves per SIMD instead of two (no scratch), which is what this kernel's
// load / compute lock-step was short of.
// DAMP (cx_set_damping): a result is mixed with the message it replaces — the receiving slot in the sweep's input buffer — in natural
// form, (1 - lambda) rule + lambda old: a convex combination of positive definite precisions is positive definite.  Instances of their
// own (one more gather and its registers), launched only while a damping factor is set.
template <int D, int DEG, bool NT_LOADS, bool DAMP = false>
__global__ __launch_bounds__(kBlock, DAMP ? 1 : DEG == 3 ? 3 : DEG == 4 ? 2 : 1) void k_sweep_mv(int nv, int64_t nslots, const int32_t *__restrict__ slice_off,
                                                     const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner,
                                                     const int32_t *__restrict__ spdir, const double *__restrict__ ptab, int ntab,
                                                     const double *__restrict__ f2v_in, double *__restrict__ f2v_out,
                                                     const double *__restrict__ v2f, double *__restrict__ marg, int write_marg,
                                                     int observed_only, double lam) {
    __shared__ double tab_s[kTabLds * 3 * D * D];
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int v = (s << kSliceShift) + tid;
    const int off = slice_off[s];
    const int W = (slice_off[s + 1] - off) >> kSliceShift;   // slice width: workgroup-uniform
    const int nt = ntab < kTabLds ? ntab : kTabLds;
    for (int i = tid; i < nt * 3 * D * D; i += kBlock) tab_s[i] = ptab[i];
    const int info = v < nv ? vinfo[v] : 0, deg = info & kDegMask;
    // Messages out of observed variables are constants of the data: the regular sweep skips those variables (on a
    // state-space chain a third of all rule evaluations and stores); the host runs an `observed_only` pass into the output
    // buffer for the first two sweeps after the data changed, which leaves them in both buffers of the Jacobi pair.
    // observed_only == 2: the other senders of constant messages — free variables of degree 1 and stand-ins, whose stored
    // variable→factor message goes through the rule (the chain scan's leaf pass; the regular sweep handles them itself)
    const bool is_fixed = (info & kGhost) || deg < 2;
    // observed_only == 3: both kinds of constant senders in one pass (what the chain scan's leaf pass launches)
    const bool active = v < nv && deg != kBigDeg && (observed_only == 3 ? ((info & kClamped) || (is_fixed && deg > 0))
                                   : observed_only == 2 ? (!(info & kClamped) && is_fixed && deg > 0)
                                                        : (((info & kClamped) != 0) == (observed_only != 0)));
    const int base = off + tid;
    Msg<D> in[DEG];
    int pk[DEG], sd[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        in[k] = msg_zero<D>();
        pk[k] = -1; sd[k] = -1;
        if (k < W && active) {   // k < W is uniform; lanes with deg <= k read a padding slot of their own slice and drop it
            const Msg<D> x = NT_LOADS ? slot_load<D, true>(f2v_in, base + k * kBlock) : slot_load<D, false>(f2v_in, base + k * kBlock);
            const int p = partner[base + k * kBlock], d = spdir[base + k * kBlock];
            if (k < deg) { in[k] = x; pk[k] = p; sd[k] = d; }
        }
    }
    __syncthreads();   // rule tables are in LDS (every thread of the workgroup reaches this point)
    if (!active) return;
    if (write_marg) {
        Msg<D> total = msg_zero<D>();
#pragma unroll
        for (int k = 0; k < DEG; k++)
            if (k < deg) msg_add<D>(total, in[k]);
        const Msg<D> mo = (deg > 0) ? mv_to_moment<D>(total) : total;
        // marginals are written once and not re-read by the sweep: nontemporal 16-byte stores, the pair form of the messages
        slot_store_nt<D>(marg, v, (deg > 0) ? mo : msg_all_nan<D>());
    }
    const bool fixed = (deg < 2) || (info & (kClamped | kGhost));
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        if (k >= deg) continue;
        const int slot = base + k * kBlock;
        const int p = pk[k], pd = sd[k];
        // p < 0: nobody listens to this variable→factor message; pd < 0: the receiver is an observed variable, nobody reads
        // the message into it (lazy, like the reference)
        if (p < 0 || pd < 0) continue;
        Msg<D> o;
        if (fixed) {
            o = slot_load<D>(v2f, slot);
        } else {
            o = msg_zero<D>();
#pragma unroll
            for (int j = 0; j < DEG; j++)
                if (j < deg && j != k) msg_add<D>(o, in[j]);
        }
        if (__builtin_isnan(o.lam[0])) continue;
        Msg<D> r = pd < nt ? mv_rule<D>(o, tab_s + pd * 3 * D * D) : mv_rule<D>(o, ptab + (int64_t)pd * 3 * D * D);
        if (DAMP && !__builtin_isnan(r.lam[0])) {
            const Msg<D> old = slot_load<D>(f2v_in, p);
            if (!__builtin_isnan(old.lam[0])) {
#pragma unroll
                for (int i = 0; i < D; i++) r.eta[i] = (1.0 - lam) * r.eta[i] + lam * old.eta[i];
#pragma unroll
                for (int i = 0; i < D * (D + 1) / 2; i++) r.lam[i] = (1.0 - lam) * r.lam[i] + lam * old.lam[i];
            }
        }
        if (!__builtin_isnan(r.lam[0])) slot_store<D>(f2v_out, p, r);
    }
}

// variable→factor messages of a list of slots, on demand (cx_get_messages): leave-one-out of the retained input buffer
template <int D>
__global__ __launch_bounds__(kBlock) void k_v2f_mv(int64_t n, int64_t nslots, const int32_t *__restrict__ slots,
                                                   const int32_t *__restrict__ vars, const int32_t *__restrict__ vbase,
                                                   const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ vdeg, const double *__restrict__ f2v,
                                                   double *__restrict__ v2f) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i], v = vars[i];
    const int info = vinfo[v], deg = vdeg[v], stride = (info & kDegMask) == kBigDeg ? 1 : kBlock;
    if (deg < 2 || (info & (kClamped | kGhost))) return;
    const int b = vbase[v];
    Msg<D> o = msg_zero<D>();
    for (int j = 0; j < deg; j++)
        if (b + j * stride != slot) msg_add<D>(o, slot_load<D>(f2v, b + j * stride));
    if (!__builtin_isnan(o.lam[0])) slot_store<D>(v2f, slot, o);
}

// Variables of degree > 8 (the CSR tail of the slot space; round 5): one thread per slot of such a variable — the variable→factor message
// of the slot (the sum of the variable's other incoming messages, in slot order), through the factor rule into the partner's slot of
// the sweep's output buffer; the thread of a variable's first slot also writes its marginal.  The reference wires such a variable
// through a segment tree of ProductOfMessages nodes (src/dependencies.jl:90-173: O(log degree) inputs per message); here every slot
// sums the others itself — O(degree^2) loads per hub and sweep out of the caches, which is what a hub of some hundreds of neighbours
// costs without a second kernel and a scratch array (the scalar path has the wave scans: k_big_var_to_factor).
template <int D>
__global__ __launch_bounds__(kBlock) void k_big_mv(int64_t n, const int32_t *__restrict__ slots, const int32_t *__restrict__ slot_var,
                                                   const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                                   const int32_t *__restrict__ partner, const int32_t *__restrict__ spdir, const double *__restrict__ ptab,
                                                   const double *__restrict__ f2v_in, double *__restrict__ f2v_out, const double *__restrict__ v2f,
                                                   double *__restrict__ marg, int write_marg, double lam) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int slot = slots[i], v = slot_var[i], info = vinfo[v], deg = vdeg[v], b = vbase[v];
    Msg<D> o = msg_zero<D>();
    for (int j = 0; j < deg; j++)
        if (b + j != slot) msg_add<D>(o, slot_load<D>(f2v_in, b + j));
    if (write_marg && slot == b) {
        Msg<D> total = o;
        msg_add<D>(total, slot_load<D>(f2v_in, slot));
        slot_store_nt<D>(marg, v, __builtin_isnan(total.lam[0]) ? msg_all_nan<D>() : mv_to_moment<D>(total));
    }
    if (info & (kClamped | kGhost)) o = slot_load<D>(v2f, slot);      // an observed variable sends its data
    const int p = partner[slot], pd = spdir[slot];
    if (p < 0 || pd < 0 || __builtin_isnan(o.lam[0])) return;
    Msg<D> r = mv_rule<D>(o, ptab + (int64_t)pd * 3 * D * D);
    if (__builtin_isnan(r.lam[0])) return;
    if (lam != 0.0) {
        const Msg<D> old = slot_load<D>(f2v_in, p);
        if (!__builtin_isnan(old.lam[0])) {
#pragma unroll
            for (int c = 0; c < D; c++) r.eta[c] = (1.0 - lam) * r.eta[c] + lam * old.eta[c];
#pragma unroll
            for (int c = 0; c < D * (D + 1) / 2; c++) r.lam[c] = (1.0 - lam) * r.lam[c] + lam * old.lam[c];
        }
    }
    slot_store<D>(f2v_out, p, r);
}

// host staging <-> device.  ncs > 0: a MESSAGE buffer (block-major pairs, ncs stored doubles per slot: cx_mv_core.h);
// ncs == 0: a plain component-major array of `stride` entries (the marginals).  Staging rows are nc doubles, compact.
__device__ __forceinline__ int64_t mv_addr(int64_t stride, int ncs, int c, int idx) {
    return ncs ? (int64_t)(idx >> kSliceShift) * kBlock * ncs + (int64_t)(c >> 1) * 2 * kBlock + (int64_t)(idx & (kBlock - 1)) * 2 + (c & 1)
               : (int64_t)c * stride + idx;
}

__global__ void k_mv_scatter(double *__restrict__ dst, int64_t stride, int nc, int ncs, const int32_t *__restrict__ idx,
                             const double *__restrict__ val, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < nc; c++) dst[mv_addr(stride, ncs, c, idx[i])] = val[i * nc + c];
    for (int c = nc; c < ncs; c++) dst[mv_addr(stride, ncs, c, idx[i])] = 0.0;      // padding of the last pair
}

__global__ void k_mv_gather(const double *__restrict__ src, int64_t stride, int nc, int ncs, const int32_t *__restrict__ idx,
                            double *__restrict__ val, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < nc; c++) val[i * nc + c] = src[mv_addr(stride, ncs, c, idx[i])];
}

__global__ void k_mv_seed(double *__restrict__ buf, int64_t nslots, int dim, int nc, int ncs, double eta, double lam,
                          const int32_t *__restrict__ partner) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots || partner[s] < 0) return;
    if (!__builtin_isnan(buf[mv_addr(nslots, ncs, dim, (int)s)])) return;
    for (int i = 0; i < dim; i++) buf[mv_addr(nslots, ncs, i, (int)s)] = eta;
    int c = dim;
    for (int i = 0; i < dim; i++)
        for (int j = i; j < dim; j++) buf[mv_addr(nslots, ncs, c++, (int)s)] = (i == j) ? lam : 0.0;
    for (; c < ncs; c++) buf[mv_addr(nslots, ncs, c, (int)s)] = 0.0;
}

__global__ __launch_bounds__(kBlock) void k_mv_residual(const double *__restrict__ cur, const double *__restrict__ prev, int64_t n,
                                                        double *__restrict__ out) {
    __shared__ double red[kBlock / 64];
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double a = cur[i], b = prev[i];
        const bool da = !__builtin_isnan(a), db = !__builtin_isnan(b);
        if (da != db) m = __builtin_inf();
        else if (da) m = fmax(m, fabs(a - b));   // a, b defined: the difference is NaN only for inf - inf, caught below
        if (da && db && (a - b) != (a - b) && a != b) m = __builtin_inf();
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

// ------------------------------------------------------------------------------------------------ host side
static void prof_b(cx_handle *h, int kernel) {
    h->prof_armed = false;
    if (!h->profiling) return;
    if ((h->prof_count[kernel]++ % h->prof_stride) != 0) return;
    h->prof_armed = true;
    ProfileRec r;
    r.kernel = kernel;
    (void)hipEventCreate(&r.start);
    (void)hipEventCreate(&r.stop);
    (void)hipEventRecord(r.start, h->stream);
    h->recs.push_back(r);
}
static void prof_e(cx_handle *h) {
    if (h->profiling && h->prof_armed) (void)hipEventRecord(h->recs.back().stop, h->stream);
}

void mv_launch_sweep(cx_handle *h, bool write_marg, int observed_only, double *f2v_out) {
    if (!f2v_out) f2v_out = h->d_mv_f2v_alt;
    if (h->nslices == 0) return;
    if (!observed_only) prof_b(h, CX_KERNEL_FUSED);
    const dim3 g((unsigned)h->nslices), b(kBlock);
    // widest slice = largest variable degree (computed once per graph); CX_MV_DEG4=1 forces the four-message instantiation (A/B)
    if (h->mv_max_deg == 0) {
        int w = 0;
        for (int64_t sl = 0; sl < h->nslices; sl++) w = std::max<int>(w, (h->slice_off[sl + 1] - h->slice_off[sl]) >> kSliceShift);
        h->mv_max_deg = std::max(w, 1);
    }
    static const bool force4 = [] { const char *e = getenv("CX_MV_DEG4"); return e && e[0] == '1'; }();
    const bool deg3 = h->mv_max_deg <= 3 && !force4, deg8 = h->mv_max_deg > 4;
#define CX_MV_ARGS g, b, 0, h->stream, (int)h->nv, h->nslots, h->d_slice_off, h->d_vinfo, h->d_partner, h->d_spdir, h->d_ptab, (int)(2 * h->ptab_sets), \
                   h->d_mv_f2v, f2v_out, h->d_mv_v2f, h->d_mv_marg, (write_marg && !observed_only) ? 1 : 0, observed_only, h->damping
    // CX_MV_NT=0/1: nontemporal loads of the incoming messages off / on (A/B; default on: every message is read once per sweep)
    static const bool nt = [] { const char *e = getenv("CX_MV_NT"); return !(e && e[0] == '0'); }();
    // messages out of observed variables and other constant senders (observed_only passes) are functions of the data alone: never damped
    const bool damp = h->damping != 0.0 && !observed_only;
#define CX_MV(DD)                                                                      \
    if (damp && deg8) hipLaunchKernelGGL((k_sweep_mv<DD, 8, true, true>), CX_MV_ARGS); \
    else if (damp) hipLaunchKernelGGL((k_sweep_mv<DD, 4, true, true>), CX_MV_ARGS);    \
    else if (deg8) hipLaunchKernelGGL((k_sweep_mv<DD, 8, true>), CX_MV_ARGS);          \
    else if (deg3 && nt) hipLaunchKernelGGL((k_sweep_mv<DD, 3, true>), CX_MV_ARGS);    \
    else if (deg3) hipLaunchKernelGGL((k_sweep_mv<DD, 3, false>), CX_MV_ARGS);         \
    else if (nt) hipLaunchKernelGGL((k_sweep_mv<DD, 4, true>), CX_MV_ARGS);            \
    else hipLaunchKernelGGL((k_sweep_mv<DD, 4, false>), CX_MV_ARGS)
    if (h->cfg.dim == 2) CX_MV(2);
    else if (h->cfg.dim == 3) CX_MV(3);
    else CX_MV(4);
#undef CX_MV_ARGS
#undef CX_MV
    if (!observed_only) prof_e(h);
}

// the variables of degree > 8 of a fused sweep (after the sliced part: same input and output buffers)
void mv_launch_big(cx_handle *h, bool write_marg, double *f2v_out) {
    const int64_t n = (int64_t)h->big_slots.size();
    if (n == 0) return;
    if (!f2v_out) f2v_out = h->d_mv_f2v_alt;
    const dim3 g((unsigned)((n + kBlock - 1) / kBlock)), b(kBlock);
#define CX_MV(DD) hipLaunchKernelGGL((k_big_mv<DD>), g, b, 0, h->stream, n, h->d_big_slots, h->d_big_slot_var, h->d_vbase, h->d_var_deg, h->d_vinfo, h->d_partner, \
                                     h->d_spdir, h->d_ptab, h->d_mv_f2v, f2v_out, h->d_mv_v2f, h->d_mv_marg, write_marg ? 1 : 0, h->damping)
    if (h->cfg.dim == 2) CX_MV(2);
    else if (h->cfg.dim == 3) CX_MV(3);
    else CX_MV(4);
#undef CX_MV
}

void mv_launch_v2f(cx_handle *h, const int32_t *d_slots, const int32_t *d_vars, int64_t n, const double *f2v) {
    if (n == 0) return;
    const dim3 g((unsigned)((n + kBlock - 1) / kBlock)), b(kBlock);
#define CX_MV(DD) hipLaunchKernelGGL((k_v2f_mv<DD>), g, b, 0, h->stream, n, h->nslots, d_slots, d_vars, h->d_vbase, h->d_vinfo, h->d_var_deg, f2v, h->d_mv_v2f)
    if (h->cfg.dim == 2) CX_MV(2);
    else if (h->cfg.dim == 3) CX_MV(3);
    else CX_MV(4);
#undef CX_MV
}

// ncs: h->ncs for a message buffer (f2v, v2f), 0 for the component-major marginals
void mv_launch_scatter(cx_handle *h, double *dst, int64_t stride, int nc, int ncs, const int32_t *d_idx, const double *d_val, int64_t n) {
    if (n) hipLaunchKernelGGL(k_mv_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, dst, stride, nc, ncs, d_idx, d_val, n);
}
void mv_launch_gather(cx_handle *h, const double *src, int64_t stride, int nc, int ncs, const int32_t *d_idx, double *d_val, int64_t n) {
    if (n) hipLaunchKernelGGL(k_mv_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, stride, nc, ncs, d_idx, d_val, n);
}
void mv_launch_seed(cx_handle *h, double *buf, double eta, double lam) {
    hipLaunchKernelGGL(k_mv_seed, dim3((unsigned)((h->nslots + 255) / 256)), dim3(256), 0, h->stream, buf, h->nslots, h->cfg.dim, h->nc, h->ncs, eta,
                       lam, h->d_partner);
}
void mv_launch_residual(cx_handle *h, const double *cur, const double *prev, int64_t n, double *d_out) {
    hipLaunchKernelGGL(k_mv_residual, dim3(1024), dim3(kBlock), 0, h->stream, cur, prev, n, d_out);
}

// ---- small dense helpers (host) -------------------------------------------------------------------------------
// in-place inverse of an SPD matrix (row-major d x d) by Cholesky; returns false if not positive definite
bool spd_inverse(int d, const double *S, double *out) {
    std::vector<double> L((size_t)d * d, 0.0), Li((size_t)d * d, 0.0);
    for (int j = 0; j < d; j++) {
        double s = S[j * d + j];
        for (int k = 0; k < j; k++) s -= L[j * d + k] * L[j * d + k];
        if (!(s > 0.0)) return false;
        L[j * d + j] = std::sqrt(s);
        for (int i = j + 1; i < d; i++) {
            double t = S[i * d + j];
            for (int k = 0; k < j; k++) t -= L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / L[j * d + j];
        }
    }
    for (int c = 0; c < d; c++) {
        for (int i = 0; i < d; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= L[i * d + k] * Li[k * d + c];
            Li[i * d + c] = s / L[i * d + i];
        }
    }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int k = 0; k < d; k++) s += Li[k * d + i] * Li[k * d + j];
            out[i * d + j] = s;
        }
    return true;
}

// (A, Q) -> the two (P, B, C) triples: [forward | backward], each 3*d*d doubles
bool mv_rule_tables(int d, const double *A, const double *Q, double *out) {
    std::vector<double> Qi((size_t)d * d), QiA((size_t)d * d), AtQi((size_t)d * d), AtQiA((size_t)d * d);
    if (!spd_inverse(d, Q, Qi.data())) return false;
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            double s = 0.0, t = 0.0;
            for (int k = 0; k < d; k++) { s += Qi[i * d + k] * A[k * d + j]; t += A[k * d + i] * Qi[k * d + j]; }
            QiA[i * d + j] = s; AtQi[i * d + j] = t;
        }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int k = 0; k < d; k++) s += AtQi[i * d + k] * A[k * d + j];
            AtQiA[i * d + j] = s;
        }
    for (int i = 0; i < d; i++)      // exactly symmetric P and C tables
        for (int j = i + 1; j < d; j++) {
            const double q = 0.5 * (Qi[i * d + j] + Qi[j * d + i]), t = 0.5 * (AtQiA[i * d + j] + AtQiA[j * d + i]);
            Qi[i * d + j] = Qi[j * d + i] = q; AtQiA[i * d + j] = AtQiA[j * d + i] = t;
        }
    const size_t n = (size_t)d * d;
    double *fw = out, *bw = out + 3 * n;
    for (size_t i = 0; i < n; i++) {
        fw[i] = AtQiA[i]; fw[n + i] = QiA[i]; fw[2 * n + i] = Qi[i];
        bw[i] = Qi[i];    bw[n + i] = AtQi[i]; bw[2 * n + i] = AtQiA[i];
    }
    return true;
}

}  // namespace cx
