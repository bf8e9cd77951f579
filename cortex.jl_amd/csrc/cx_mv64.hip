// cx_mv64.hip — d = 64 linear-Gaussian messages: the batched 64x64 update on the f64 matrix cores
// (BASELINE.json config 5: d = 64 factors, 1e5 nodes, "batched 64x64 MFMA update path").
//
// One workgroup (4 wave64) computes one factor→variable message
//     M = Lambda_in + P = L L',   Yt = B L^-T,   Lambda_out = C - Yt Yt',   eta_out = Yt (L^-1 eta_in)
// (same rule as cx_mv.hip; P, B, C are the receiving edge's precomputed 64x64 tables).  Everything that is a dense
// contraction runs on v_mfma_f64_16x16x4_f64: the Cholesky panel and trailing updates, the blocked triangular solve for
// Yt and the final Yt Yt' (≈ 450 MFMA per message); only the four 16x16 diagonal blocks are factored and inverted by
// one wave with v_readlane broadcasts.  The reference has no such rule (parity unpinned; checked against a
// numpy restatement and the exact block-tridiagonal smoother).
//
// Operand layout.  For v_mfma_f64_16x16x4_f64 lane l supplies A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15]
// and receives D[row = (l >> 4) + 4 r][col = l & 15], r = 0..3 (cdna_hip_programming.md §3: the f64 map differs from
// every other dtype).  Every product here is written as X Z' with X and Z stored row-major in LDS with leading
// dimension 66 doubles: both operands are then read as "row = l & 15, k = l >> 4", and (row * 66 + k) mod 32 takes 32
// distinct values over a 32-lane group, so each ds_read_b64 is bank-conflict free (the 16x16 inverses use ld = 18, same
// property).  Messages are message-major in HBM (eta[64] | Lambda[64][64], 33,280 B per slot): a workgroup streams whole
// messages with unit-stride 8 B/lane loads; LDS per workgroup 78 KiB -> 2 workgroups per CU, so one workgroup's serial
// diagonal-block phase overlaps the other's MFMA phase.

#include "cx_internal.h"

namespace cx {

constexpr int kD = 64;
constexpr int kLd = 66;                 // leading dimension of the 64x64 LDS matrices (doubles)
constexpr int kLdw = 18;                // leading dimension of the 16x16 inverse blocks
constexpr int kMsg = kD + kD * kD;      // doubles per message slot

using d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// acc += X[xr0 .. xr0+15][kx0 .. kx0+4*ksteps) * Z[zr0 .. zr0+15][kz0 .. kz0+4*ksteps)'   (16x16 tile of X Z')
__device__ __forceinline__ d4 mfma_xzt(const double *__restrict__ X, int ldx, int xr0, int kx0, const double *__restrict__ Z, int ldz,
                                       int zr0, int kz0, int ksteps, d4 acc, int lane) {
    const int r = lane & 15, kk = lane >> 4;
    const double *xp = X + (xr0 + r) * ldx + kx0 + kk;
    const double *zp = Z + (zr0 + r) * ldz + kz0 + kk;
    for (int s = 0; s < ksteps; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[4 * s], zp[4 * s], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ void tile_store(double *__restrict__ T, int ld, int r0, int c0, d4 acc, int lane) {
    const int c = lane & 15, rb = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) T[(r0 + rb + 4 * r) * ld + c0 + c] = acc[r];
}

__device__ __forceinline__ d4 tile_load(const double *__restrict__ T, int ld, int r0, int c0, int lane) {
    const int c = lane & 15, rb = lane >> 4;
    d4 a;
#pragma unroll
    for (int r = 0; r < 4; r++) a[r] = T[(r0 + rb + 4 * r) * ld + c0 + c];
    return a;
}

// Cholesky of the 16x16 diagonal block at (o, o) of Ms and its inverse into W (row-major, ld kLdw); one wave, lanes 0..15
// hold one row each, the pivot row is broadcast with v_readlane (its lane index is a compile-time constant).
__device__ __forceinline__ void diag_block(double *__restrict__ Ms, int o, double *__restrict__ W, int lane) {
    double a[16], l[16], w[16];
    const int row = lane & 15;
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = Ms[(o + row) * kLd + o + j];
    double rinv_own = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        double x = a[j];
#pragma unroll
        for (int k = 0; k < j; k++) x -= l[k] * readlane_f64(l[k], j);
        const double d = readlane_f64(x, j);
        const double rinv = 1.0 / sqrt(d);      // not positive definite -> NaN -> the message stays undefined
        l[j] = x * rinv;                        // row == j: d / sqrt(d) = sqrt(d)
        if (row == j) rinv_own = rinv;
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) Ms[(o + row) * kLd + o + j] = (j <= row) ? l[j] : 0.0;
    }
    // inverse: lane c computes column c of W = L^-1 by forward substitution; L[i][k] is broadcast from lane i
#pragma unroll
    for (int i = 0; i < 16; i++) {
        double s = (row == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s -= readlane_f64(l[k], i) * w[k];
        w[i] = s * readlane_f64(rinv_own, i);
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) W[i * kLdw + row] = w[i];
    }
}

constexpr int kFlagFixed = 1;     // the sender's message is stored (user-set / observed), not a product of others

// MODE 0: factor→variable message of work item (sender slot, sender variable) into out[partner]
// MODE 1: marginal (mean | covariance) of variable work_vars[w] into out[w]  (P = 0, B = I, C = 0, sign flipped)
template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void k_rule64(int nwork, const int32_t *__restrict__ work_slots, const int32_t *__restrict__ work_vars,
                                                      const int32_t *__restrict__ work_flags, const int32_t *__restrict__ vbase,
                                                      const uint8_t *__restrict__ vinfo, const int32_t *__restrict__ partner,
                                                      const int32_t *__restrict__ spdir, const double *__restrict__ ptab,
                                                      const double *__restrict__ f2v_in, const double *__restrict__ v2f,
                                                      double *__restrict__ out) {
    __shared__ double Ms[kD * kLd];
    __shared__ double Yt[kD * kLd];
    __shared__ double Ws[4 * 16 * kLdw];
    __shared__ double eta[kD], zz[kD], tmp[16];
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = work_vars[w];
    const int slot = MODE == 0 ? work_slots[w] : -1;
    const int flags = MODE == 0 ? work_flags[w] : 0;
    const int deg = vinfo[v] & kDegMask;
    const int base = vbase[v];
    const double *tab = MODE == 0 ? ptab + (int64_t)spdir[slot] * 3 * kD * kD : nullptr;

    // ---- phase 0: M = P + sum of the other incoming Lambdas; Yt = B; eta = sum of the other etas -------------------
    for (int e = tid; e < kD * kD; e += kBlock) {
        const int r = e >> 6, c = e & 63;
        double acc = MODE == 0 ? tab[e] : 0.0;
        if (flags & kFlagFixed) {
            acc += v2f[(int64_t)slot * kMsg + kD + e];
        } else {
            for (int j = 0; j < deg; j++) {
                const int sj = base + j * kBlock;
                if (sj != slot) acc += f2v_in[(int64_t)sj * kMsg + kD + e];
            }
        }
        Ms[r * kLd + c] = acc;
        Yt[r * kLd + c] = MODE == 0 ? tab[kD * kD + e] : (r == c ? 1.0 : 0.0);
    }
    if (tid < kD) {
        double acc = 0.0;
        if (flags & kFlagFixed) {
            acc = v2f[(int64_t)slot * kMsg + tid];
        } else {
            for (int j = 0; j < deg; j++) {
                const int sj = base + j * kBlock;
                if (sj != slot) acc += f2v_in[(int64_t)sj * kMsg + tid];
            }
        }
        eta[tid] = acc;
    }
    __syncthreads();
    if (__builtin_isnan(Ms[0])) return;   // a dependency is undefined (whole messages are NaN together): not pending

    // ---- blocked Cholesky, NB = 16: Ms lower triangle <- L, Ws[kb] <- L_kk^-1 ---------------------------------------
    for (int kb = 0; kb < 4; kb++) {
        const int o = kb * 16;
        if (wave == 0) diag_block(Ms, o, Ws + kb * 16 * kLdw, lane);
        __syncthreads();
        // panel: L21 = A21 * W', one 16x16 tile per wave
        const int tr = kb + 1 + wave;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (tr < 4) acc = mfma_xzt(Ms, kLd, tr * 16, o, Ws + kb * 16 * kLdw, kLdw, 0, 0, 4, acc, lane);
        __syncthreads();                       // every lane has read its A21 tile before the tile is overwritten
        if (tr < 4) tile_store(Ms, kLd, tr * 16, o, acc, lane);
        __syncthreads();
        // trailing update: A22[ti][tj] -= L21[ti] L21[tj]'  for kb < tj <= ti <= 3, tiles dealt round-robin to waves
        int t = 0;
        for (int ti = kb + 1; ti < 4; ti++)
            for (int tj = kb + 1; tj <= ti; tj++, t++) {
                if ((t & 3) != wave) continue;
                d4 c = tile_load(Ms, kLd, ti * 16, tj * 16, lane);
                d4 p = {0.0, 0.0, 0.0, 0.0};
                p = mfma_xzt(Ms, kLd, ti * 16, o, Ms, kLd, tj * 16, o, 4, p, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) c[r] -= p[r];
                tile_store(Ms, kLd, ti * 16, tj * 16, c, lane);
            }
        __syncthreads();
    }

    // ---- z = L^-1 eta (wave 0, blocked) --------------------------------------------------------------------------------
    if (wave == 0) {
        for (int ib = 0; ib < 4; ib++) {
            const int r = lane & 15;
            double s = eta[ib * 16 + r];
            for (int k = 0; k < ib * 16; k++) s -= Ms[(ib * 16 + r) * kLd + k] * zz[k];
            if (lane < 16) tmp[r] = s;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            double zi = 0.0;
            for (int c = 0; c < 16; c++) zi += Ws[ib * 16 * kLdw + r * kLdw + c] * tmp[c];
            if (lane < 16) zz[ib * 16 + r] = zi;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- Yt = B L^-T, block column by block column; wave = tile row of Yt -------------------------------------------------
    for (int ib = 0; ib < 4; ib++) {
        d4 tq = tile_load(Yt, kLd, wave * 16, ib * 16, lane);
        if (ib > 0) {
            d4 p = {0.0, 0.0, 0.0, 0.0};
            p = mfma_xzt(Yt, kLd, wave * 16, 0, Ms, kLd, ib * 16, 0, 4 * ib, p, lane);   // sum_{k < 16 ib} Yt[c][k] L[r][k]
#pragma unroll
            for (int r = 0; r < 4; r++) tq[r] -= p[r];
        }
        tile_store(Yt, kLd, wave * 16, ib * 16, tq, lane);
        __syncthreads();
        d4 y = {0.0, 0.0, 0.0, 0.0};
        y = mfma_xzt(Yt, kLd, wave * 16, ib * 16, Ws + ib * 16 * kLdw, kLdw, 0, 0, 4, y, lane);   // (rhs tile) * W_ii'
        __syncthreads();
        tile_store(Yt, kLd, wave * 16, ib * 16, y, lane);
        __syncthreads();
    }

    // ---- G = Yt Yt' into Ms (free now); eta_out = Yt z --------------------------------------------------------------------
    for (int tj = 0; tj < 4; tj++) {
        d4 g = {0.0, 0.0, 0.0, 0.0};
        g = mfma_xzt(Yt, kLd, wave * 16, 0, Yt, kLd, tj * 16, 0, 16, g, lane);
        tile_store(Ms, kLd, wave * 16, tj * 16, g, lane);
    }
    if (tid < kD) {
        double s = 0.0;
        for (int k = 0; k < kD; k++) s += Yt[tid * kLd + k] * zz[k];
        eta[tid] = s;
    }
    __syncthreads();
    if (__builtin_isnan(Ms[0]) || __builtin_isnan(eta[0])) return;   // not positive definite: leave the old value

    // ---- store --------------------------------------------------------------------------------------------------------------
    const int64_t dst = MODE == 0 ? (int64_t)partner[slot] * kMsg : (int64_t)w * kMsg;
    for (int e = tid; e < kD * kD; e += kBlock) {
        const int r = e >> 6, c = e & 63;
        const double g = 0.5 * (Ms[r * kLd + c] + Ms[c * kLd + r]);
        out[dst + kD + e] = MODE == 0 ? tab[2 * kD * kD + e] - g : g;
    }
    if (tid < kD) out[dst + tid] = eta[tid];
}

// observed senders: Lambda_out = C, eta_out = B y  (one workgroup per message; pure streaming)
__global__ __launch_bounds__(kBlock) void k_point64(int nwork, const int32_t *__restrict__ work_slots, const int32_t *__restrict__ partner,
                                                    const int32_t *__restrict__ spdir, const double *__restrict__ ptab,
                                                    const double *__restrict__ v2f, double *__restrict__ out_a, double *__restrict__ out_b) {
    __shared__ double y[kD];
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int slot = work_slots[w], tid = threadIdx.x;
    const double *tab = ptab + (int64_t)spdir[slot] * 3 * kD * kD;
    if (tid < kD) y[tid] = v2f[(int64_t)slot * kMsg + tid];
    __syncthreads();
    if (__builtin_isnan(y[0])) return;
    const int64_t dst = (int64_t)partner[slot] * kMsg;
    for (int e = tid; e < kD * kD; e += kBlock) {
        const double c = tab[2 * kD * kD + e];
        out_a[dst + kD + e] = c;
        out_b[dst + kD + e] = c;
    }
    if (tid < kD) {
        double s = 0.0;
        for (int k = 0; k < kD; k++) s += tab[kD * kD + tid * kD + k] * y[k];
        out_a[dst + tid] = s;
        out_b[dst + tid] = s;
    }
}

// variable→factor message of a listed slot, on demand: the sum of the variable's other incoming messages
__global__ __launch_bounds__(kBlock) void k_v2f64(int n, const int32_t *__restrict__ slots, const int32_t *__restrict__ vars,
                                                  const int32_t *__restrict__ vbase, const uint8_t *__restrict__ vinfo,
                                                  const double *__restrict__ f2v, double *__restrict__ v2f) {
    const int w = blockIdx.x;
    if (w >= n) return;
    const int slot = slots[w], v = vars[w];
    const int info = vinfo[v], deg = info & kDegMask;
    if (deg < 2 || (info & (kClamped | kGhost))) return;
    const int base = vbase[v];
    for (int e = threadIdx.x; e < kMsg; e += kBlock) {
        double acc = 0.0;
        for (int j = 0; j < deg; j++) {
            const int sj = base + j * kBlock;
            if (sj != slot) acc += f2v[(int64_t)sj * kMsg + e];
        }
        v2f[(int64_t)slot * kMsg + e] = acc;
    }
}

__global__ void k_fill64(double *__restrict__ buf, int64_t nslots, double eta, double lam, const int32_t *__restrict__ partner) {
    const int64_t s = blockIdx.x;
    if (s >= nslots || partner[s] < 0) return;
    if (!__builtin_isnan(buf[s * kMsg + kD])) return;
    for (int e = threadIdx.x; e < kMsg; e += blockDim.x) {
        double v = eta;
        if (e >= kD) { const int q = e - kD; v = ((q >> 6) == (q & 63)) ? lam : 0.0; }
        buf[s * kMsg + e] = v;
    }
}

// row-wise staging helpers for the message-major layout
__global__ void k_rows_scatter(double *__restrict__ dst, const int32_t *__restrict__ idx, const double *__restrict__ val, int64_t n) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    for (int e = threadIdx.x; e < kMsg; e += blockDim.x) dst[(int64_t)idx[i] * kMsg + e] = val[i * kMsg + e];
}
__global__ void k_rows_gather(const double *__restrict__ src, const int32_t *__restrict__ idx, double *__restrict__ val, int64_t n) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    for (int e = threadIdx.x; e < kMsg; e += blockDim.x) val[i * kMsg + e] = src[(int64_t)idx[i] * kMsg + e];
}
// observed data: eta = y, Lambda[0][0] = +inf marks the point mass (the rest of the slot is never read)
__global__ void k_set_point64(double *__restrict__ dst, const int32_t *__restrict__ idx, const double *__restrict__ y, int64_t n) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    if (threadIdx.x < kD) dst[(int64_t)idx[i] * kMsg + threadIdx.x] = y[i * kD + threadIdx.x];
    if (threadIdx.x == kD) dst[(int64_t)idx[i] * kMsg + kD] = __builtin_inf();
}

// ------------------------------------------------------------------------------------------------ launchers
void mv64_rows_scatter(cx_handle *h, double *dst, const int32_t *d_idx, const double *d_val, int64_t n) {
    if (n) hipLaunchKernelGGL(k_rows_scatter, dim3((unsigned)n), dim3(256), 0, h->stream, dst, d_idx, d_val, n);
}
void mv64_rows_gather(cx_handle *h, const double *src, const int32_t *d_idx, double *d_val, int64_t n) {
    if (n) hipLaunchKernelGGL(k_rows_gather, dim3((unsigned)n), dim3(256), 0, h->stream, src, d_idx, d_val, n);
}
void mv64_set_point(cx_handle *h, double *dst, const int32_t *d_idx, const double *d_y, int64_t n) {
    if (n) hipLaunchKernelGGL(k_set_point64, dim3((unsigned)n), dim3(128), 0, h->stream, dst, d_idx, d_y, n);
}

void mv64_launch_rule(cx_handle *h, int nwork, const int32_t *d_slots, const int32_t *d_vars, const int32_t *d_flags,
                      const double *f2v_in, double *f2v_out, int kernel_id) {
    if (nwork == 0) return;
    h->prof_armed = false;
    if (h->profiling && (h->prof_count[kernel_id]++ % h->prof_stride) == 0) {
        h->prof_armed = true;
        ProfileRec r; r.kernel = kernel_id;
        (void)hipEventCreate(&r.start); (void)hipEventCreate(&r.stop); (void)hipEventRecord(r.start, h->stream);
        h->recs.push_back(r);
    }
    hipLaunchKernelGGL((k_rule64<0>), dim3(nwork), dim3(kBlock), 0, h->stream, nwork, d_slots, d_vars, d_flags, h->d_vbase, h->d_vinfo,
                       h->d_partner, h->d_spdir, h->d_ptab, f2v_in, h->d_mv_v2f, f2v_out);
    if (h->profiling && h->prof_armed) (void)hipEventRecord(h->recs.back().stop, h->stream);
}

void mv64_launch_marginals(cx_handle *h, int n, const int32_t *d_vars, const double *f2v, double *out) {
    if (n == 0) return;
    hipLaunchKernelGGL((k_rule64<1>), dim3(n), dim3(kBlock), 0, h->stream, n, (const int32_t *)nullptr, d_vars, (const int32_t *)nullptr,
                       h->d_vbase, h->d_vinfo, h->d_partner, h->d_spdir, h->d_ptab, f2v, h->d_mv_v2f, out);
}

void mv64_launch_point(cx_handle *h, int nwork, const int32_t *d_slots, double *out_a, double *out_b) {
    if (nwork == 0) return;
    hipLaunchKernelGGL(k_point64, dim3(nwork), dim3(kBlock), 0, h->stream, nwork, d_slots, h->d_partner, h->d_spdir, h->d_ptab,
                       h->d_mv_v2f, out_a, out_b);
}

void mv64_launch_v2f(cx_handle *h, int n, const int32_t *d_slots, const int32_t *d_vars, const double *f2v) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_v2f64, dim3(n), dim3(kBlock), 0, h->stream, n, d_slots, d_vars, h->d_vbase, h->d_vinfo, f2v, h->d_mv_v2f);
}

void mv64_launch_seed(cx_handle *h, double *buf, double eta, double lam) {
    hipLaunchKernelGGL(k_fill64, dim3((unsigned)h->nslots), dim3(256), 0, h->stream, buf, h->nslots, eta, lam, h->d_partner);
}

}  // namespace cx
