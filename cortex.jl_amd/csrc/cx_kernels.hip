// cx_kernels.hip — gfx950 kernels of the scalar-Gaussian sum-product sweep.
//
// What they replace (reference = Cortex.jl v0.3.0): one `process!` → `compute!` → user rule →
// `set_value!` round trip per directed message (src/inference_engine.jl:479-509, src/signal.jl:392-410)
// with the Gaussian rules of test/inference_engine_tests.jl:385-432 and the product of
// test/runtests.jl:40-46.  Here one launch updates every message of one kind.
//
// Storage form.  Messages live in NATURAL form m = (xi, w) = (mean/variance, 1/variance), one 16-byte
// double2 per directed message, edge table sorted by (variable, factor): the incoming messages of a
// variable are contiguous, so a workgroup stages a run of variables with fully coalesced 16 B/lane loads.
// In natural form the reference's `product` is a plain sum, and its additive-Gaussian factor rule
// N(m, v + q) becomes  s = 1/(1 + q w);  (xi, w) <- (xi s, w s): one reciprocal per message instead of the
// five divisions per `product` call of the moment form.  UndefValue() is NaN and propagates by itself:
// an output is defined iff every dependency is (the reference's "pending" criterion, src/signal.jl:668-730);
// NaN outputs are never stored, so a signal that is not pending keeps its value as in the reference.
// A point-mass datum y (the `Real` branch, test/inference_engine_tests.jl:424) is stored as (y, +inf).
//
// All kernels are HBM-streaming: the roofline is bytes, not flops (≈12 flop per 32 payload bytes).

#include "cx_internal.h"

namespace cx {

// blockIdx -> work chunk.  Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md,
// "Workgroup dispatch"), each XCD with a private 4 MiB L2.  Give XCD x the x-th contiguous slab of
// chunks so that the partner gathers of grid-like graphs (±one grid row away in the edge table) are
// served by the L2 that streamed those rows.  Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_slab(int b, int nb) {
    int xcd = b & 7, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ double2 add2(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }

// factor→variable rule for the receiving edge's effective parameters (a, b, q):
//   moment form:  N(a m + b, a² v + q)        [a=1, b=0: test/inference_engine_tests.jl:426-427]
//   natural form: s = 1/(a² + q w);  w' = w s;  xi' = (a xi + b w) s
//   point mass y: N(a y + b, q)               [:424-425]
template <bool LINEAR>
__device__ __forceinline__ double2 factor_rule(double2 m, double q, double a, double b) {
    double2 o;
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

// leave-one-out sums of up to kSmallDeg natural-form messages held in registers:
// out[k] = (in[0] + … + in[k-1]) + (in[deg-1] + … + in[k+1]), total = in[0] + … + in[deg-1].
// Everything is unrolled with compile-time indices (runtime-indexed arrays would go to scratch).
struct Loo {
    double2 out[kSmallDeg];
    double2 total;
};

__device__ __forceinline__ void leave_one_out(const double2 (&in)[kSmallDeg], Loo &r) {
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < kSmallDeg; k++) {
        r.out[k] = acc;
        acc = add2(acc, in[k]);
    }
    r.total = acc;
    acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = kSmallDeg - 1; k >= 0; k--) {
        r.out[k] = add2(r.out[k], acc);
        acc = add2(acc, in[k]);
    }
}

__device__ __forceinline__ double2 to_moment(double2 nat) {
    double var = 1.0 / nat.y;
    return make_double2(nat.x * var, var);
}

// ------------------------------------------------------------------------------------------------
// K1: variable → factor for every small-degree variable (+ optional marginals).
// compute_message_to_factor! = reduce(product, others)   test/inference_engine_tests.jl:405-413
// compute_individual_marginal! = reduce(product, all)    test/inference_engine_tests.jl:385-393
// dependency sets: dependencies.jl:60-88.
// One workgroup = a run of ≤256 consecutive variables with ≤ kCapEdges edges (host partition `blk`).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_var_to_factor(const int32_t *__restrict__ blk, const int32_t *__restrict__ var_off,
                                                          const uint8_t *__restrict__ var_flags,
                                                          const double2 *__restrict__ f2v, double2 *__restrict__ v2f,
                                                          double2 *__restrict__ marg, int write_marg) {
    __shared__ double2 lds[kCapEdges];
    const int b = xcd_slab(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int v0 = blk[b], v1 = blk[b + 1];
    const int e0 = var_off[v0], n = var_off[v1] - e0;
    for (int i = tid; i < n; i += kBlock) lds[i] = f2v[e0 + i];
    __syncthreads();
    const int v = v0 + tid;
    if (v < v1) {
        const int s = var_off[v] - e0, deg = var_off[v + 1] - e0 - s;
        double2 in[kSmallDeg];
#pragma unroll
        for (int k = 0; k < kSmallDeg; k++) in[k] = (k < deg) ? lds[s + k] : make_double2(0.0, 0.0);
        Loo r;
        leave_one_out(in, r);
        // a variable with <2 factors has no dependencies on its message to the factor (dependencies.jl:48-55):
        // never computed.  Clamped (observed) variables keep the data the caller set.
        const bool skip = (deg < 2) || (var_flags[v] != 0);
        const double nan = __builtin_nan("");
#pragma unroll
        for (int k = 0; k < kSmallDeg; k++)
            if (k < deg) lds[s + k] = skip ? make_double2(nan, nan) : r.out[k];
        if (write_marg) marg[v] = to_moment(r.total);
    }
    __syncthreads();
    for (int i = tid; i < n; i += kBlock) {
        double2 o = lds[i];
        if (!__builtin_isnan(o.y)) v2f[e0 + i] = o;
    }
}

// K1-big: one wave per variable of degree > kSmallDeg: exclusive prefix + exclusive suffix by wave scans.
// This is the device analogue of the reference's segment tree of ProductOfMessages intermediates
// (dependencies.jl:90-173): "product of all but me" without O(deg²) work.
__device__ __forceinline__ double2 wave_inclusive_scan(double2 x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double ox = __shfl_up(x.x, d, 64), oy = __shfl_up(x.y, d, 64);
        if (lane >= d) { x.x += ox; x.y += oy; }
    }
    return x;
}

__global__ __launch_bounds__(kBlock) void k_big_var_to_factor(const int32_t *__restrict__ big, int nbig,
                                                              const int32_t *__restrict__ var_off,
                                                              const uint8_t *__restrict__ var_flags,
                                                              const double2 *__restrict__ f2v, double2 *__restrict__ v2f,
                                                              double2 *__restrict__ tmp, const int32_t *__restrict__ tmp_off,
                                                              double2 *__restrict__ marg, int write_marg) {
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (w >= nbig) return;
    const int v = big[w];
    const int s = var_off[v], t = var_off[v + 1];
    double2 *pre = tmp + tmp_off[w];
    double2 carry = make_double2(0.0, 0.0);
    for (int base = s; base < t; base += 64) {
        int i = base + lane;
        double2 x = (i < t) ? f2v[i] : make_double2(0.0, 0.0);
        double2 inc = wave_inclusive_scan(x, lane);
        double ex = __shfl_up(inc.x, 1, 64), ey = __shfl_up(inc.y, 1, 64);
        double2 exc = (lane == 0) ? make_double2(0.0, 0.0) : make_double2(ex, ey);
        if (i < t) pre[i - s] = add2(carry, exc);
        carry = add2(carry, make_double2(__shfl(inc.x, 63, 64), __shfl(inc.y, 63, 64)));
    }
    if (write_marg && lane == 0) marg[v] = to_moment(carry);
    if (var_flags[v] != 0) return;
    const int nchunk = (t - s + 63) >> 6;
    carry = make_double2(0.0, 0.0);
    for (int c = nchunk - 1; c >= 0; c--) {
        int i = s + c * 64 + (63 - lane);  // lanes walk the chunk backwards
        double2 x = (i < t) ? f2v[i] : make_double2(0.0, 0.0);
        double2 inc = wave_inclusive_scan(x, lane);
        double ex = __shfl_up(inc.x, 1, 64), ey = __shfl_up(inc.y, 1, 64);
        double2 exc = (lane == 0) ? make_double2(0.0, 0.0) : make_double2(ex, ey);
        if (i < t) {
            double2 o = add2(pre[i - s], add2(carry, exc));
            if (!__builtin_isnan(o.y)) v2f[i] = o;
        }
        carry = add2(carry, make_double2(__shfl(inc.x, 63, 64), __shfl(inc.y, 63, 64)));
    }
}

// ------------------------------------------------------------------------------------------------
// K2: factor → variable for every edge of a 2-edge Gaussian factor (gather from the partner edge).
// compute_message_to_variable!   test/inference_engine_tests.jl:415-432;  dependency: dependencies.jl:17-31.
// ------------------------------------------------------------------------------------------------
template <bool LINEAR>
__global__ __launch_bounds__(kBlock) void k_factor_to_var(int ne, const int32_t *__restrict__ partner,
                                                          const double *__restrict__ q, const double *__restrict__ pa,
                                                          const double *__restrict__ pb, const double2 *__restrict__ v2f,
                                                          double2 *__restrict__ f2v) {
    const int e = xcd_slab(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
    if (e >= ne) return;
    const int p = partner[e];
    if (p < 0) return;
    const double2 m = v2f[p];
    if (__builtin_isnan(m.y)) return;  // dependency not computed: not pending, keep the old value
    f2v[e] = factor_rule<LINEAR>(m, q[e], LINEAR ? pa[e] : 1.0, LINEAR ? pb[e] : 0.0);
}

// ------------------------------------------------------------------------------------------------
// Fused sweep: K1 then K2 of the same flooding map in one launch, pushing each fresh variable→factor
// message straight through its factor into the partner's slot of the *other* factor→variable buffer.
// Same fixed-point map as K1;K2 (Jacobi on double-buffered messages), half the HBM traffic.
// sq/sa/sb are the rule parameters of the RECEIVING edge, indexed by the SENDING edge.
// ------------------------------------------------------------------------------------------------
template <bool LINEAR, bool STORE_V2F>
__global__ __launch_bounds__(kBlock) void k_fused(const int32_t *__restrict__ blk, const int32_t *__restrict__ var_off,
                                                  const uint8_t *__restrict__ var_flags, const int32_t *__restrict__ partner,
                                                  const double *__restrict__ sq, const double *__restrict__ sa,
                                                  const double *__restrict__ sb, const double2 *__restrict__ f2v_in,
                                                  double2 *__restrict__ f2v_out, double2 *__restrict__ v2f,
                                                  double2 *__restrict__ marg, int write_marg) {
    __shared__ double2 lds[kCapEdges];
    const int b = xcd_slab(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int v0 = blk[b], v1 = blk[b + 1];
    const int e0 = var_off[v0], n = var_off[v1] - e0;
    for (int i = tid; i < n; i += kBlock) lds[i] = f2v_in[e0 + i];
    __syncthreads();
    const int v = v0 + tid;
    if (v < v1) {
        const int s = var_off[v] - e0, deg = var_off[v + 1] - e0 - s;
        double2 in[kSmallDeg];
#pragma unroll
        for (int k = 0; k < kSmallDeg; k++) in[k] = (k < deg) ? lds[s + k] : make_double2(0.0, 0.0);
        Loo r;
        leave_one_out(in, r);
        const bool fixed = (deg < 2) || (var_flags[v] != 0);
#pragma unroll
        for (int k = 0; k < kSmallDeg; k++)
            if (k < deg) lds[s + k] = fixed ? v2f[e0 + s + k] : r.out[k];  // fixed: the caller's data / halo message
        if (write_marg) marg[v] = to_moment(r.total);
        // tag fixed entries so the store loop below leaves v2f alone: nothing to do, storing the same value back
        // is harmless, but skip it to save the write
    }
    __syncthreads();
    for (int i = tid; i < n; i += kBlock) {
        const int e = e0 + i;
        const double2 o = lds[i];
        if (__builtin_isnan(o.y)) continue;
        if (STORE_V2F) v2f[e] = o;
        const int p = partner[e];
        if (p >= 0) f2v_out[p] = factor_rule<LINEAR>(o, sq[e], LINEAR ? sa[e] : 1.0, LINEAR ? sb[e] : 0.0);
    }
}

// fused schedule, big variables: push the variable→factor messages of a listed set of edges through their factors
template <bool LINEAR>
__global__ __launch_bounds__(kBlock) void k_push_edges(const int32_t *__restrict__ edges, int64_t n, const int32_t *__restrict__ partner,
                                                       const double *__restrict__ sq, const double *__restrict__ sa,
                                                       const double *__restrict__ sb, const double2 *__restrict__ v2f,
                                                       double2 *__restrict__ f2v_out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int e = edges[i], p = partner[e];
    if (p < 0) return;
    const double2 o = v2f[e];
    if (__builtin_isnan(o.y)) return;
    f2v_out[p] = factor_rule<LINEAR>(o, sq[e], LINEAR ? sa[e] : 1.0, LINEAR ? sb[e] : 0.0);
}

// ------------------------------------------------------------------------------------------------
// Batched mode: one thread per enqueued signal (the processor's `process!` override flushes a batch of
// mutually independent pending signals; inference_engine.jl:528-537 is the precedent for collecting).
// ------------------------------------------------------------------------------------------------
template <bool LINEAR>
__global__ __launch_bounds__(kBlock) void k_batch(int64_t n, const int32_t *__restrict__ kind, const int32_t *__restrict__ index,
                                                  const int32_t *__restrict__ var_off, const int32_t *__restrict__ edge_var,
                                                  const uint8_t *__restrict__ var_flags, const int32_t *__restrict__ partner,
                                                  const double *__restrict__ q, const double *__restrict__ pa,
                                                  const double *__restrict__ pb, double2 *__restrict__ f2v,
                                                  double2 *__restrict__ v2f, double2 *__restrict__ marg) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int k = kind[i], idx = index[i];
    if (k == CX_ITEM_MESSAGE_TO_FACTOR) {
        const int v = edge_var[idx];
        const int s = var_off[v], t = var_off[v + 1];
        if (t - s < 2 || var_flags[v] != 0) return;
        double2 pre = make_double2(0.0, 0.0), suf = make_double2(0.0, 0.0);
        for (int j = s; j < idx; j++) pre = add2(pre, f2v[j]);
        for (int j = t - 1; j > idx; j--) suf = add2(suf, f2v[j]);
        double2 o = add2(pre, suf);
        if (!__builtin_isnan(o.y)) v2f[idx] = o;
    } else if (k == CX_ITEM_MESSAGE_TO_VARIABLE) {
        const int p = partner[idx];
        if (p < 0) return;
        const double2 m = v2f[p];
        if (__builtin_isnan(m.y)) return;
        f2v[idx] = factor_rule<LINEAR>(m, q[idx], LINEAR ? pa[idx] : 1.0, LINEAR ? pb[idx] : 0.0);
    } else if (k == CX_ITEM_INDIVIDUAL_MARGINAL) {
        const int s = var_off[idx], t = var_off[idx + 1];
        double2 acc = make_double2(0.0, 0.0);
        for (int j = s; j < t; j++) acc = add2(acc, f2v[j]);
        marg[idx] = (t > s) ? to_moment(acc) : make_double2(__builtin_nan(""), __builtin_nan(""));
    }
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

__global__ void k_seed(double2 *__restrict__ buf, int64_t n, double2 value, const int32_t *__restrict__ partner) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (partner && partner[i] < 0) return;  // messages nobody computes are the caller's to set
    if (__builtin_isnan(buf[i].y)) buf[i] = value;
}

// max |Δmean|, |Δvariance| per workgroup (moment form), reduced on the host from a few thousand partials
__global__ __launch_bounds__(kBlock) void k_residual(const double2 *__restrict__ cur, const double2 *__restrict__ prev, int64_t n,
                                                     double *__restrict__ out) {
    __shared__ double red[kBlock / 64];
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double2 a = cur[i], b = prev[i];
        const bool da = !__builtin_isnan(a.y), db = !__builtin_isnan(b.y);
        if (da != db) m = __builtin_inf();
        else if (da) {
            double2 ma = to_moment(a), mb = to_moment(b);
            m = fmax(m, fmax(fabs(ma.x - mb.x), fabs(ma.y - mb.y)));
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
static inline void prof_begin(cx_handle *h, int kernel) {
    if (!h->profiling) return;
    ProfileRec r;
    r.kernel = kernel;
    hipEventCreate(&r.start);
    hipEventCreate(&r.stop);
    hipEventRecord(r.start, h->stream);
    h->recs.push_back(r);
}
static inline void prof_end(cx_handle *h) {
    if (!h->profiling) return;
    hipEventRecord(h->recs.back().stop, h->stream);
}

void launch_var_to_factor(cx_handle *h, const double2 *f2v, double2 *v2f, bool write_marg) {
    const int nblk = (int)h->blk.size() - 1;
    if (nblk <= 0) return;
    prof_begin(h, CX_KERNEL_VAR_TO_FACTOR);
    hipLaunchKernelGGL(k_var_to_factor, dim3(nblk), dim3(kBlock), 0, h->stream, h->d_blk, h->d_var_off, h->d_var_flags, f2v,
                       v2f, h->d_marg, write_marg ? 1 : 0);
    prof_end(h);
}

void launch_big_var_to_factor(cx_handle *h, const double2 *f2v, double2 *v2f, bool write_marg) {
    const int nbig = (int)h->big_vars.size();
    if (nbig == 0) return;
    prof_begin(h, CX_KERNEL_BIG_VAR);
    const int waves_per_block = kBlock / 64;
    const int nb = (nbig + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL(k_big_var_to_factor, dim3(nb), dim3(kBlock), 0, h->stream, h->d_big, nbig, h->d_var_off,
                       h->d_var_flags, f2v, v2f, h->d_big_tmp, h->d_big_tmp_off, h->d_marg, write_marg ? 1 : 0);
    prof_end(h);
}

void launch_factor_to_var(cx_handle *h, const double2 *v2f, double2 *f2v) {
    const int ne = (int)h->ne;
    if (ne == 0) return;
    const int nb = (ne + kBlock - 1) / kBlock;
    prof_begin(h, CX_KERNEL_FACTOR_TO_VAR);
    if (h->any_linear)
        hipLaunchKernelGGL(k_factor_to_var<true>, dim3(nb), dim3(kBlock), 0, h->stream, ne, h->d_partner, h->d_q, h->d_a,
                           h->d_b, v2f, f2v);
    else
        hipLaunchKernelGGL(k_factor_to_var<false>, dim3(nb), dim3(kBlock), 0, h->stream, ne, h->d_partner, h->d_q,
                           (const double *)nullptr, (const double *)nullptr, v2f, f2v);
    prof_end(h);
}

void launch_fused(cx_handle *h, const double2 *f2v_in, double2 *f2v_out, double2 *v2f, bool write_marg, bool store_v2f) {
    const int nblk = (int)h->blk.size() - 1;
    if (nblk <= 0) return;
    prof_begin(h, CX_KERNEL_FUSED);
    const double *sq = h->any_linear ? h->d_sq : h->d_q;  // additive factors: q is symmetric in the two edges
#define CX_FUSED(LIN, ST)                                                                                              \
    hipLaunchKernelGGL((k_fused<LIN, ST>), dim3(nblk), dim3(kBlock), 0, h->stream, h->d_blk, h->d_var_off,             \
                       h->d_var_flags, h->d_partner, sq, h->d_sa, h->d_sb, f2v_in, f2v_out, v2f, h->d_marg,            \
                       write_marg ? 1 : 0)
    if (h->any_linear) { if (store_v2f) CX_FUSED(true, true); else CX_FUSED(true, false); }
    else { if (store_v2f) CX_FUSED(false, true); else CX_FUSED(false, false); }
#undef CX_FUSED
    prof_end(h);
}

void launch_push_edges(cx_handle *h, const int32_t *d_edges, int64_t n, const double2 *v2f, double2 *f2v_out) {
    if (n == 0) return;
    const double *sq = h->any_linear ? h->d_sq : h->d_q;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    if (h->any_linear)
        hipLaunchKernelGGL(k_push_edges<true>, dim3(nb), dim3(kBlock), 0, h->stream, d_edges, n, h->d_partner, sq, h->d_sa,
                           h->d_sb, v2f, f2v_out);
    else
        hipLaunchKernelGGL(k_push_edges<false>, dim3(nb), dim3(kBlock), 0, h->stream, d_edges, n, h->d_partner, sq,
                           (const double *)nullptr, (const double *)nullptr, v2f, f2v_out);
}

void launch_batch(cx_handle *h, const int32_t *d_kind, const int32_t *d_index, int64_t n) {
    if (n == 0) return;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    prof_begin(h, CX_KERNEL_BATCH);
    if (h->any_linear)
        hipLaunchKernelGGL(k_batch<true>, dim3(nb), dim3(kBlock), 0, h->stream, n, d_kind, d_index, h->d_var_off,
                           h->d_edge_var, h->d_var_flags, h->d_partner, h->d_q, h->d_a, h->d_b, h->d_f2v, h->d_v2f, h->d_marg);
    else
        hipLaunchKernelGGL(k_batch<false>, dim3(nb), dim3(kBlock), 0, h->stream, n, d_kind, d_index, h->d_var_off,
                           h->d_edge_var, h->d_var_flags, h->d_partner, h->d_q, (const double *)nullptr,
                           (const double *)nullptr, h->d_f2v, h->d_v2f, h->d_marg);
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
    hipLaunchKernelGGL(k_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, buf, n, value, partner);
}

void launch_residual(cx_handle *h, const double2 *cur, const double2 *prev, int64_t n, double *d_out) {
    hipLaunchKernelGGL(k_residual, dim3(1024), dim3(kBlock), 0, h->stream, cur, prev, n, d_out);
}

}  // namespace cx
