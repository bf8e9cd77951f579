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

#include "cx_scalar_core.h"

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
