// cx_health.hip — cx_message_health: the numerical guards of the boundary as counters ("Errors": numerical guards —
// non-finite values, variance <= 0, matrices that are not positive definite — are reported via status and counters, never by an abort).
// The reference has no such thing to mirror: its values are Julia objects and a rule that divides by zero throws in the user's code.
// Here an undefined value is NaN and propagates by itself, a rule whose input is not positive definite leaves its
// output undefined or unchanged (tests/test_gpu_mv_conditioning.py); this call says how many of the stored factor→variable messages
// INTO NON-OBSERVED VARIABLES (the messages somebody reads) are in which state, without moving them to the host.
#include "cx_host.h"
#include "cx_mv_core.h"

namespace cx {
namespace {


// counters: 0 defined, 1 undefined (UndefValue), 2 a negative precision (dim > 1: a negative diagonal entry of Lambda), 3 non-finite
__device__ __forceinline__ void tally(unsigned long long *c, bool undef, bool neg, bool nonfin) {
    atomicAdd(c + (undef ? 1 : 0), 1ull);
    if (!undef && neg) atomicAdd(c + 2, 1ull);
    if (!undef && nonfin) atomicAdd(c + 3, 1ull);
}

__global__ void k_health1(int64_t n, const int32_t *__restrict__ slots, const double2 *__restrict__ f2v, unsigned long long *__restrict__ c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 m = f2v[slots[i]];
    const bool undef = __builtin_isnan(m.y) || __builtin_isnan(m.x);
    // a point mass (y, +inf) and the flat message (0, 0) are values like any other
    tally(c, undef, m.y < 0.0, __builtin_isinf(m.x) || m.y == -__builtin_inf());
}

template <int D>
__global__ void k_health_mv(int64_t n, const int32_t *__restrict__ slots, const double *__restrict__ f2v, unsigned long long *__restrict__ c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Msg<D> m = slot_load<D, false>(f2v, slots[i]);
    bool undef = false, neg = false, nonfin = false;
#pragma unroll
    for (int k = 0; k < D; k++) { undef = undef || __builtin_isnan(m.eta[k]); nonfin = nonfin || __builtin_isinf(m.eta[k]); }
#pragma unroll
    for (int k = 0; k < Msg<D>::NT; k++) { undef = undef || __builtin_isnan(m.lam[k]); nonfin = nonfin || __builtin_isinf(m.lam[k]); }
#pragma unroll
    for (int k = 0; k < D; k++) neg = neg || m.lam[tri<D>(k, k)] < 0.0;
    tally(c, undef, neg, nonfin);
}

// the matrix-core dims (16, 32, 64): one wave per message record eta[kd] | Lambda[kd][kd]
__global__ __launch_bounds__(64) void k_health64(int kd, int64_t n, const int32_t *__restrict__ slots, const double *__restrict__ f2v, unsigned long long *__restrict__ c) {
    const int64_t w = blockIdx.x;
    if (w >= n) return;
    const int km = kd + kd * kd;
    const double *m = f2v + (int64_t)slots[w] * km;
    int undef = 0, neg = 0, nonfin = 0;
    for (int e = threadIdx.x; e < km; e += 64) {
        const double x = m[e];
        undef |= __builtin_isnan(x);
        nonfin |= __builtin_isinf(x);
        if (e >= kd && (e - kd) / kd == (e - kd) % kd) neg |= x < 0.0;
    }
    undef = __any(undef); neg = __any(neg); nonfin = __any(nonfin);
    if (threadIdx.x == 0) tally(c, undef != 0, neg != 0, nonfin != 0);
}

}  // namespace
}  // namespace cx

using namespace cxh;

extern "C" int32_t cx_message_health(cx_handle *h, int64_t *out4) {
    CX_NOT_VMP(h, "cx_message_health");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_message_health: no graph");
    CX_REQUIRE(h, out4, CX_ERR_INVALID_ARGUMENT, "cx_message_health: null argument");
    try {
        // the messages somebody reads: every edge into a variable that is neither observed nor a stand-in
        std::vector<int32_t> slots;
        slots.reserve((size_t)h->ne);
        for (int64_t e = 0; e < h->ne; e++)
            if (!(h->vinfo[h->edge_var[e]] & (cx::kClamped | cx::kGhost))) slots.push_back(cx::slot_of_edge(h, e));
        const int64_t n = (int64_t)slots.size();
        for (int k = 0; k < 4; k++) out4[k] = 0;
        if (n == 0) return CX_OK;
        CX_HIP(h, hipSetDevice(h->cfg.device));
        if (h->cfg.dim > 1) { int32_t rc = mv_ensure_chain_msgs(h); if (rc != CX_OK) return rc; }      // (chain scan, dim 2..4: the messages go to their slots on demand)
        int32_t rc = ensure_stage(h, n * 4 + 64);
        if (rc != CX_OK) return rc;
        unsigned long long *d_c = (unsigned long long *)h->d_stage;
        int32_t *d_s = (int32_t *)((char *)h->d_stage + 64);
        CX_HIP(h, hipMemsetAsync(d_c, 0, 32, h->stream));
        CX_HIP(h, hipMemcpyAsync(d_s, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        const dim3 g((unsigned)((n + 255) / 256)), b(256);
        if (h->cfg.dim == 1) hipLaunchKernelGGL(cx::k_health1, g, b, 0, h->stream, n, d_s, (const double2 *)h->d_f2v, d_c);
        else if (h->cfg.dim == 2) hipLaunchKernelGGL(cx::k_health_mv<2>, g, b, 0, h->stream, n, d_s, (const double *)h->d_mv_f2v, d_c);
        else if (h->cfg.dim == 3) hipLaunchKernelGGL(cx::k_health_mv<3>, g, b, 0, h->stream, n, d_s, (const double *)h->d_mv_f2v, d_c);
        else if (h->cfg.dim == 4) hipLaunchKernelGGL(cx::k_health_mv<4>, g, b, 0, h->stream, n, d_s, (const double *)h->d_mv_f2v, d_c);
        else hipLaunchKernelGGL(cx::k_health64, dim3((unsigned)n), dim3(64), 0, h->stream, h->cfg.dim, n, d_s, (const double *)h->d_mv_f2v, d_c);
        unsigned long long c[4];
        CX_HIP(h, hipMemcpyAsync(c, d_c, 32, hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        for (int k = 0; k < 4; k++) out4[k] = (int64_t)c[k];
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_message_health: host allocation failed"); }
}
