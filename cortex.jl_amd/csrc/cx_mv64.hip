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

#include <cstdlib>

#include <cstdio>
#include <vector>
#include "cx_internal.h"

namespace cx {

constexpr int kD = 64;
constexpr int kLd = 65;                 // leading dimension of the 64x64 LDS matrices (doubles): odd, so that the 16 rows an MFMA
                                        // operand read touches per quarter-wave fall into 16 different bank pairs (with 66 rows r and
                                        // r + 8 collided: PMC counted 30 % of the LDS cycles as bank conflicts)
constexpr int kLdw = 17;                // leading dimension of the 16x16 inverse blocks (odd, as above)
constexpr int kMsg = kD + kD * kD;      // doubles per message slot

using d4 = __attribute__((ext_vector_type(4))) double;

// 1/sqrt(x) to double precision: v_rsq_f64 seed (≈ 2^-27) + two Newton steps.  The library's sqrt() and 1.0/x expand to
// ≈ 40 dependent instructions; inside the 64 pivot steps of a message that chain was the kernel's critical path.
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y = y * __builtin_fma(-hx * y, y, 1.5);
    return y;       // x <= 0 or NaN -> NaN/inf: the message stays undefined
}

// acc += X[xr0 .. xr0+15][kx0 .. kx0+4*ksteps) * Z[zr0 .. zr0+15][kz0 .. kz0+4*ksteps)'   (16x16 tile of X Z')
__device__ __forceinline__ d4 mfma_xzt(const double *__restrict__ X, int ldx, int xr0, int kx0, const double *__restrict__ Z, int ldz,
                                       int zr0, int kz0, int ksteps, d4 acc, int lane) {
    const int r = lane & 15, kk = lane >> 4;
    const double *xp = X + (xr0 + r) * ldx + kx0 + kk;
    const double *zp = Z + (zr0 + r) * ldz + kz0 + kk;
#pragma unroll
    for (int s = 0; s < ksteps; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[4 * s], zp[4 * s], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ void tile_store(double *__restrict__ T, int ld, int r0, int c0, d4 acc, int lane) {
    const int c = lane & 15, rb = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) T[(r0 + rb + 4 * r) * ld + c0 + c] = acc[r];
}

// store only the first `nrows` rows of a tile (the eta row of Yt is a 1-row "tile")
__device__ __forceinline__ void tile_store_rows(double *__restrict__ T, int ld, int r0, int c0, d4 acc, int lane, int nrows) {
    const int c = lane & 15, rb = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (rb + 4 * r < nrows) T[(r0 + rb + 4 * r) * ld + c0 + c] = acc[r];
}

__device__ __forceinline__ d4 tile_load(const double *__restrict__ T, int ld, int r0, int c0, int lane) {
    const int c = lane & 15, rb = lane >> 4;
    d4 a;
#pragma unroll
    for (int r = 0; r < 4; r++) a[r] = T[(r0 + rb + 4 * r) * ld + c0 + c];
    return a;
}

// forward substitution with a 4x4 lower factor given by its strict lower part and reciprocal diagonal: x <- L44^-1 x
__device__ __forceinline__ void sub4(double (&x)[4], double l10, double l20, double l21, double l30, double l31, double l32,
                                     const double (&ri)[4]) {
    x[0] = x[0] * ri[0];
    x[1] = (x[1] - l10 * x[0]) * ri[1];
    x[2] = (x[2] - l20 * x[0] - l21 * x[1]) * ri[2];
    x[3] = (x[3] - l30 * x[0] - l31 * x[1] - l32 * x[2]) * ri[3];
}

// Cholesky of the 16x16 diagonal block at (o, o) of Ms and its inverse into W (row-major, ld kLdw), by the whole
// workgroup: thread (i, k) = (tid >> 4, tid & 15) keeps A[i][k] and W[i][k] in registers.  Four rounds of four columns,
// ONE barrier per round: the owners publish the four raw columns and the four rows of W (double-buffered staging), every
// thread factors the 4x4 pivot block redundantly in registers (the only sqrt chain) and applies the rank-4 update
//     l_i = raw_i L44^-T;   A[i][k] -= l_i . l_k;   W_J = L44^-1 W_J;   W[i][:] -= l_i W_J.
// History: a 16-lane v_readlane version took 12.7k cycles per block, a one-column-per-barrier version 11k (each pivot
// step pays barrier + LDS round trip + rsqrt chain); this one pays that latency 4 times instead of 16.
__device__ __forceinline__ void diag_block(double *__restrict__ Ms, int o, double *__restrict__ W, double *__restrict__ stage /* 2 x 128 */,
                                           int tid) {
    const int i = tid >> 4, k = tid & 15;
    double a = Ms[(o + i) * kLd + o + k];
    double w = (i == k) ? 1.0 : 0.0;
#pragma unroll
    for (int R = 0; R < 4; R++) {
        const int j0 = 4 * R;
        double *col = stage + (R & 1) * 128;     // col[row * 4 + m]: raw A[row][j0 + m]
        double *wrw = col + 64;                  // wrw[m * 16 + c]:  current W[j0 + m][c]
        if (k >= j0 && k < j0 + 4) col[i * 4 + (k - j0)] = a;
        if (i >= j0 && i < j0 + 4) wrw[(i - j0) * 16 + k] = w;
        __syncthreads();
        // 4x4 pivot block (rows j0..j0+3 of the staged columns), factored by every thread
        const double p00 = col[(j0 + 0) * 4 + 0];
        const double p10 = col[(j0 + 1) * 4 + 0], p11 = col[(j0 + 1) * 4 + 1];
        const double p20 = col[(j0 + 2) * 4 + 0], p21 = col[(j0 + 2) * 4 + 1], p22 = col[(j0 + 2) * 4 + 2];
        const double p30 = col[(j0 + 3) * 4 + 0], p31 = col[(j0 + 3) * 4 + 1], p32 = col[(j0 + 3) * 4 + 2], p33 = col[(j0 + 3) * 4 + 3];
        double ri[4];
        ri[0] = rsqrt_f64(p00);
        const double l10 = p10 * ri[0], l20 = p20 * ri[0], l30 = p30 * ri[0];
        ri[1] = rsqrt_f64(p11 - l10 * l10);
        const double l21 = (p21 - l20 * l10) * ri[1], l31 = (p31 - l30 * l10) * ri[1];
        ri[2] = rsqrt_f64(p22 - l20 * l20 - l21 * l21);
        const double l32 = (p32 - l30 * l20 - l31 * l21) * ri[2];
        ri[3] = rsqrt_f64(p33 - l30 * l30 - l31 * l31 - l32 * l32);   // not positive definite -> NaN -> message stays undefined
        double li[4], lk[4], wj[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { li[m] = col[i * 4 + m]; lk[m] = col[k * 4 + m]; wj[m] = wrw[m * 16 + k]; }
        sub4(li, l10, l20, l21, l30, l31, l32, ri);
        sub4(lk, l10, l20, l21, l30, l31, l32, ri);
        sub4(wj, l10, l20, l21, l30, l31, l32, ri);
        const int n = k - j0, mi = i - j0;
        if (i >= j0 && k >= j0) {
            if (n < 4) {
                const double lin = n == 0 ? li[0] : n == 1 ? li[1] : n == 2 ? li[2] : li[3];
                a = (i >= k) ? lin : 0.0;
            } else {
                a -= li[0] * lk[0] + li[1] * lk[1] + li[2] * lk[2] + li[3] * lk[3];
            }
        }
        if (mi >= 0 && mi < 4) w = mi == 0 ? wj[0] : mi == 1 ? wj[1] : mi == 2 ? wj[2] : wj[3];
        else if (mi >= 4) w -= li[0] * wj[0] + li[1] * wj[1] + li[2] * wj[2] + li[3] * wj[3];
    }
    Ms[(o + i) * kLd + o + k] = (k <= i) ? a : 0.0;
    W[i * kLdw + k] = w;
}

constexpr int kFlagFixed = 1;     // the sender's message is stored (user-set / observed), not a product of others

// LDS carve (doubles): Ms 64x66 | Yt 65x66 (row 64 carries eta, so z = L^-1 eta and eta_out = Yt z fall out of the same
// MFMA tiles as Yt and Yt Yt') | Ws 4 x 16x18 | eta_out 64.  The fifth tile row of Yt reads 15 rows past Yt's end: they
// land in Ws/eta_out (finite or not, they only feed output rows that are never used).
constexpr int kOffYt = kD * kLd;
constexpr int kOffWs = kOffYt + (kD + 1) * kLd;
constexpr int kOffEo = kOffWs + 4 * 16 * kLdw;
constexpr int kOffSt = kOffEo + kD;                 // 2 x 128 staging doubles of diag_block
constexpr int kLdsDoubles = kOffSt + 256;
static_assert((kD + 16) * kLd <= (kD + 1) * kLd + 4 * 16 * kLdw + kD, "fifth tile row must stay inside the LDS block");

// MODE 0: factor→variable message of work item (sender slot, sender variable) into out[partner]
// MODE 1: marginal (mean | covariance) of variable work_vars[w] into out[w]  (P = 0, B = I, C = 0, sign flipped)
// MODE 0 reads its work item from an 8-word record built on the host (cx_api.hip, build_work64):
//   {sender slot, up to three source slots (-1: none), rule-table index, destination slot, flags, 0}
// — one scalar round trip instead of the chain work list → vinfo / vbase / spdir / partner → loads.
template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void k_rule64(int nwork, const int32_t *__restrict__ work_rec, const int32_t *__restrict__ work_vars,
                                                      const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                                      const double *__restrict__ ptab,
                                                      const double *__restrict__ f2v_in, const double *__restrict__ v2f,
                                                      double *__restrict__ out) {
    __shared__ double lds[kLdsDoubles];
    double *Ms = lds, *Yt = lds + kOffYt, *Ws = lds + kOffWs, *eo = lds + kOffEo, *stage = lds + kOffSt;
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int slot = -1, flags = 0, dst_slot = 0, s0 = -1, s1 = -1, s2 = -1;
    const double *tab = nullptr;
    if (MODE == 0) {
        const int32_t *rec = work_rec + 8 * (int64_t)w;
        slot = rec[0]; s0 = rec[1]; s1 = rec[2]; s2 = rec[3];
        tab = ptab + (int64_t)rec[4] * 3 * kD * kD;
        dst_slot = rec[5]; flags = rec[6];
    }

    // ---- phase 0: M = P + sum of the other incoming Lambdas; Yt = [B; eta'] ----------------------------------------------
    // In-kernel stamps put HALF of a workgroup's lifetime here when the groups of loads were issued one after the other
    // (P, then each message, then B: four dependent round trips at two workgroups per CU).  Now every load of the phase —
    // 16 doubles of P, of B and of each source message per thread — is in flight before the first use.
    if (MODE == 0) {
        double pa[16], pb[16], m0[16], m1[16], m2[16];
        const double *src0 = (flags & kFlagFixed) ? v2f + (int64_t)slot * kMsg : (s0 >= 0 ? f2v_in + (int64_t)s0 * kMsg : nullptr);
        const double *src1 = (!(flags & kFlagFixed) && s1 >= 0) ? f2v_in + (int64_t)s1 * kMsg : nullptr;
        const double *src2 = (!(flags & kFlagFixed) && s2 >= 0) ? f2v_in + (int64_t)s2 * kMsg : nullptr;
#pragma unroll
        for (int i = 0; i < 16; i++) pa[i] = tab[tid + kBlock * i];
#pragma unroll
        for (int i = 0; i < 16; i++) m0[i] = src0 ? src0[kD + tid + kBlock * i] : 0.0;      // workgroup-uniform branches
#pragma unroll
        for (int i = 0; i < 16; i++) m1[i] = src1 ? src1[kD + tid + kBlock * i] : 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) m2[i] = src2 ? src2[kD + tid + kBlock * i] : 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) pb[i] = tab[kD * kD + tid + kBlock * i];
        double e0 = 0.0, e1 = 0.0, e2 = 0.0;
        if (tid < kD) { e0 = src0 ? src0[tid] : 0.0; e1 = src1 ? src1[tid] : 0.0; e2 = src2 ? src2[tid] : 0.0; }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + kBlock * i;
            double a = pa[i];
            if (src0) a += m0[i];
            if (src1) a += m1[i];
            if (src2) a += m2[i];
            Ms[(e >> 6) * kLd + (e & 63)] = a;
            Yt[(e >> 6) * kLd + (e & 63)] = pb[i];
        }
        if (tid < kD) {
            double ea = 0.0;
            if (src0) ea += e0;
            if (src1) ea += e1;
            if (src2) ea += e2;
            Yt[kD * kLd + tid] = ea;
        }
    } else {
        const int v = work_vars[w];
        const int info = vinfo[v];
        const bool big = (info & kDegMask) == kBigDeg;      // degree > 8: consecutive slots in the CSR tail, the degree in vdeg
        const int deg = big ? vdeg[v] : (info & kDegMask), stride = big ? 1 : kBlock;
        const int base = vbase[v];
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0;
        double ea = 0.0;
        for (int j = 0; j < deg; j++) {
            const double *src = f2v_in + (int64_t)(base + j * stride) * kMsg;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] += src[kD + tid + kBlock * i];
            if (tid < kD) ea += src[tid];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + kBlock * i;
            Ms[(e >> 6) * kLd + (e & 63)] = acc[i];
            Yt[(e >> 6) * kLd + (e & 63)] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
        }
        if (tid < kD) Yt[kD * kLd + tid] = ea;
    }
    __syncthreads();
    if (__builtin_isnan(Ms[0])) return;   // a dependency is undefined (whole messages are NaN together): not pending

    // ---- blocked Cholesky, NB = 16: Ms lower triangle <- L, Ws[kb] <- L_kk^-1 ---------------------------------------
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        const int o = kb * 16;
        diag_block(Ms, o, Ws + kb * 16 * kLdw, stage, tid);
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

    // ---- [Yt; z'] = [B; eta'] L^-T, block column by block column; wave w owns tile row w, wave 0 also the eta row ----------
    const int nrows = wave == 0 ? 2 : 1;          // tile rows of this wave: wave, and 4 for wave 0
#pragma unroll
    for (int ib = 0; ib < 4; ib++) {
        for (int q = 0; q < nrows; q++) {
            const int tr = q == 0 ? wave : 4;
            d4 tq = tile_load(Yt, kLd, tr * 16, ib * 16, lane);
            if (ib > 0) {
                d4 p = {0.0, 0.0, 0.0, 0.0};
                p = mfma_xzt(Yt, kLd, tr * 16, 0, Ms, kLd, ib * 16, 0, 4 * ib, p, lane);   // sum_{k < 16 ib} Yt[c][k] L[r][k]
#pragma unroll
                for (int r = 0; r < 4; r++) tq[r] -= p[r];
                tile_store_rows(Yt, kLd, tr * 16, ib * 16, tq, lane, tr == 4 ? 1 : 16);
            }
        }
        __syncthreads();
        d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
        y0 = mfma_xzt(Yt, kLd, wave * 16, ib * 16, Ws + ib * 16 * kLdw, kLdw, 0, 0, 4, y0, lane);   // (rhs tile) * W_ii'
        if (wave == 0) y1 = mfma_xzt(Yt, kLd, 4 * 16, ib * 16, Ws + ib * 16 * kLdw, kLdw, 0, 0, 4, y1, lane);
        __syncthreads();
        tile_store_rows(Yt, kLd, wave * 16, ib * 16, y0, lane, 16);
        if (wave == 0) tile_store_rows(Yt, kLd, 4 * 16, ib * 16, y1, lane, 1);
        __syncthreads();
    }

    // ---- G = Yt Yt' into Ms (free now); eta_out = Yt z is the 65th column of the same product --------------------------------
    // C is fetched now, before the MFMA phase: on gfx950 vmcnt retires loads and stores in issue order, so a load issued
    // after this kernel's own output stores would wait for them (that mistake cost 40 % of the kernel's time).
    double creg[16];
#pragma unroll
    for (int i = 0; i < 16; i++) creg[i] = MODE == 0 ? tab[2 * kD * kD + tid + kBlock * i] : 0.0;
    {
        // five accumulators share the A operand (this wave's 16 rows of Yt): 6 LDS reads per 5 MFMA, independent chains
        d4 g[5];
#pragma unroll
        for (int tj = 0; tj < 5; tj++) g[tj] = d4{0.0, 0.0, 0.0, 0.0};
        const int r = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int s4 = 0; s4 < 16; s4++) {
            const double av = Yt[(wave * 16 + r) * kLd + 4 * s4 + kk];
#pragma unroll
            for (int tj = 0; tj < 5; tj++)
                g[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Yt[(tj * 16 + r) * kLd + 4 * s4 + kk], g[tj], 0, 0, 0);
        }
#pragma unroll
        for (int tj = 0; tj < 4; tj++) tile_store(Ms, kLd, wave * 16, tj * 16, g[tj], lane);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) eo[wave * 16 + (lane >> 4) + 4 * rr] = g[4][rr];
        }
    }
    __syncthreads();
    if (__builtin_isnan(Ms[0]) || __builtin_isnan(eo[0])) return;   // not positive definite: leave the old value

    // ---- store --------------------------------------------------------------------------------------------------------------
    const int64_t dst = MODE == 0 ? (int64_t)dst_slot * kMsg : (int64_t)w * kMsg;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int e = tid + kBlock * i;
        const int r = e >> 6, c = e & 63;
        const double g = Ms[r * kLd + c];   // Yt Yt' is bitwise symmetric: (i,j) and (j,i) sum the same products in the same order
        out[dst + kD + e] = MODE == 0 ? creg[i] - g : g;
    }
    if (tid < kD) out[dst + tid] = eo[tid];
}

// ---- the MODE 0 rule with THREE workgroups per CU ------------------------------------------------------------------------
// Stamps and PMC of k_rule64<0>: with 80 KB of LDS two workgroups fit on a CU and each spends 40 % of its
// life waiting for its own source messages at the miss parallelism of one loader per CU.  This form needs 53 KB:
//   * Yt lives where M / L lived.  After the factorisation wave i keeps the row panel L[16 i .. 16 i + 15][0 .. 16 i) as
//     accumulator tiles in registers (<= 3 tiles = 12 doubles per lane) and publishes it through a 16-row LDS panel when
//     solve step i needs it; only then is the region overwritten with [B; eta'].
//   * the 16-row panel shares its LDS with the staging area of the diagonal blocks (used in disjoint phases).
// A third workgroup per CU overlaps one workgroup's load phase with two others' compute.
constexpr int kOffLp_s = (kD + 1) * kLd;                 // after the [M | Yt] region (65 x 66)
constexpr int kOffWs_s = kOffLp_s + 16 * kLd;            // 16 x 66 panel / diag staging
constexpr int kOffEo_s = kOffWs_s + 4 * 16 * kLdw;
constexpr int kOffEt_s = kOffEo_s + kD;
constexpr int kLdsDoubles_s = kOffEt_s + kD;
static_assert(kLdsDoubles_s * 8 * 3 <= 160 * 1024, "three workgroups per CU");
static_assert(16 * kLd >= 256, "diag_block staging fits in the panel");

__global__ __launch_bounds__(kBlock, 3) void k_rule64s(int nwork, const int32_t *__restrict__ work_rec, const double *__restrict__ ptab,
                                                       const double *__restrict__ f2v_in, const double *__restrict__ v2f,
                                                       double *__restrict__ out) {
    __shared__ double lds[kLdsDoubles_s];
    double *Ms = lds, *Yt = lds, *Lp = lds + kOffLp_s, *Ws = lds + kOffWs_s, *eo = lds + kOffEo_s, *et = lds + kOffEt_s;
    double *stage = Lp;
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *rec = work_rec + 8 * (int64_t)w;
    const int slot = rec[0], s0 = rec[1], s1 = rec[2], s2 = rec[3], dst_slot = rec[5], flags = rec[6];
    const double *tab = ptab + (int64_t)rec[4] * 3 * kD * kD;

    // ---- phase 0: M = P + sum of the other incoming Lambdas (ascending neighbour order); eta aside --------------------------
    {
        const double *src0 = (flags & kFlagFixed) ? v2f + (int64_t)slot * kMsg : (s0 >= 0 ? f2v_in + (int64_t)s0 * kMsg : nullptr);
        const double *src1 = (!(flags & kFlagFixed) && s1 >= 0) ? f2v_in + (int64_t)s1 * kMsg : nullptr;
        const double *src2 = (!(flags & kFlagFixed) && s2 >= 0) ? f2v_in + (int64_t)s2 * kMsg : nullptr;
        double pa[16], m0[16], m1[16];
#pragma unroll
        for (int i = 0; i < 16; i++) pa[i] = tab[tid + kBlock * i];
#pragma unroll
        for (int i = 0; i < 16; i++) m0[i] = src0 ? src0[kD + tid + kBlock * i] : 0.0;      // workgroup-uniform branches
#pragma unroll
        for (int i = 0; i < 16; i++) m1[i] = src1 ? src1[kD + tid + kBlock * i] : 0.0;
        double e0 = 0.0, e1 = 0.0, e2 = 0.0;
        if (tid < kD) { e0 = src0 ? src0[tid] : 0.0; e1 = src1 ? src1[tid] : 0.0; e2 = src2 ? src2[tid] : 0.0; }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + kBlock * i;
            double a = pa[i];
            if (src0) a += m0[i];
            if (src1) a += m1[i];
            if (src2) a += src2[kD + e];           // a third source (degree 4) is rare: loaded in place
            Ms[(e >> 6) * kLd + (e & 63)] = a;
        }
        if (tid < kD) {
            double ea = 0.0;
            if (src0) ea += e0;
            if (src1) ea += e1;
            if (src2) ea += e2;
            et[tid] = ea;
        }
    }
    __syncthreads();
    if (__builtin_isnan(Ms[0])) return;   // a dependency is undefined (whole messages are NaN together): not pending

    // ---- blocked Cholesky, NB = 16: Ms lower triangle <- L, Ws[kb] <- L_kk^-1 ---------------------------------------
    double pb[16];                         // B, fetched during the last block step, stored once L has left the region
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        const int o = kb * 16;
        if (kb == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) pb[i] = tab[kD * kD + tid + kBlock * i];
        }
        diag_block(Ms, o, Ws + kb * 16 * kLdw, stage, tid);
        __syncthreads();
        const int tr = kb + 1 + wave;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        // each wave rewrites the tile it alone read (LDS operations of one wave execute in order): no barrier in between
        if (tr < 4) { acc = mfma_xzt(Ms, kLd, tr * 16, o, Ws + kb * 16 * kLdw, kLdw, 0, 0, 4, acc, lane); tile_store(Ms, kLd, tr * 16, o, acc, lane); }
        __syncthreads();
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

    // ---- L leaves the region: wave i keeps row panel i (tiles (i, 0 .. i-1)); then the region becomes [B; eta'] ---------------
    d4 lt[3];
#pragma unroll
    for (int j = 0; j < 3; j++) lt[j] = (j < wave) ? tile_load(Ms, kLd, wave * 16, j * 16, lane) : d4{0.0, 0.0, 0.0, 0.0};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int e = tid + kBlock * i;
        Yt[(e >> 6) * kLd + (e & 63)] = pb[i];
    }
    if (tid < kD) Yt[kD * kLd + tid] = et[tid];
    __syncthreads();

    // ---- [Yt; z'] = [B; eta'] L^-T, block column by block column; wave w owns tile row w, wave 0 also the eta row ----------
    // The rows of Yt are independent: a wave reads and writes only its own tile row(s); the only data shared between waves
    // is the L panel, so a step needs two barriers (panel published / panel no longer read), the first step none.
    const int nrows = wave == 0 ? 2 : 1;
#pragma unroll
    for (int ib = 0; ib < 4; ib++) {
        if (ib > 0) {
            if (wave == ib) {
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (j < ib) tile_store(Lp, kLd, 0, j * 16, lt[j], lane);      // L[16 ib .. +15][0 .. 16 ib)
            }
            __syncthreads();
            for (int q = 0; q < nrows; q++) {
                const int tr = q == 0 ? wave : 4;
                d4 tq = tile_load(Yt, kLd, tr * 16, ib * 16, lane);
                d4 p = {0.0, 0.0, 0.0, 0.0};
                p = mfma_xzt(Yt, kLd, tr * 16, 0, Lp, kLd, 0, 0, 4 * ib, p, lane);   // sum_{k < 16 ib} Yt[c][k] L[r][k]
#pragma unroll
                for (int r = 0; r < 4; r++) tq[r] -= p[r];
                tile_store_rows(Yt, kLd, tr * 16, ib * 16, tq, lane, tr == 4 ? 1 : 16);
            }
            if (ib < 3) __syncthreads();          // the panel may be overwritten by the next step's owner
        }
        d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
        y0 = mfma_xzt(Yt, kLd, wave * 16, ib * 16, Ws + ib * 16 * kLdw, kLdw, 0, 0, 4, y0, lane);   // (rhs tile) * W_ii'
        if (wave == 0) y1 = mfma_xzt(Yt, kLd, 4 * 16, ib * 16, Ws + ib * 16 * kLdw, kLdw, 0, 0, 4, y1, lane);
        tile_store_rows(Yt, kLd, wave * 16, ib * 16, y0, lane, 16);
        if (wave == 0) tile_store_rows(Yt, kLd, 4 * 16, ib * 16, y1, lane, 1);
    }
    __syncthreads();                              // G reads every row of Yt

    // ---- G = Yt Yt'; eta_out = Yt z is the 65th column of the same product ---------------------------------------------------
    double creg[16];
#pragma unroll
    for (int i = 0; i < 16; i++) creg[i] = tab[2 * kD * kD + tid + kBlock * i];
    {
        d4 g[5];
#pragma unroll
        for (int tj = 0; tj < 5; tj++) g[tj] = d4{0.0, 0.0, 0.0, 0.0};
        const int r = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int s4 = 0; s4 < 16; s4++) {
            const double av = Yt[(wave * 16 + r) * kLd + 4 * s4 + kk];
#pragma unroll
            for (int tj = 0; tj < 5; tj++)
                g[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Yt[(tj * 16 + r) * kLd + 4 * s4 + kk], g[tj], 0, 0, 0);
        }
        __syncthreads();                    // every wave has read Yt before G is written over it
#pragma unroll
        for (int tj = 0; tj < 4; tj++) tile_store(Ms, kLd, wave * 16, tj * 16, g[tj], lane);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) eo[wave * 16 + (lane >> 4) + 4 * rr] = g[4][rr];
        }
    }
    __syncthreads();
    if (__builtin_isnan(Ms[0]) || __builtin_isnan(eo[0])) return;   // not positive definite: leave the old value

    const int64_t dst = (int64_t)dst_slot * kMsg;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int e = tid + kBlock * i;
        const double g = Ms[(e >> 6) * kLd + (e & 63)];
        out[dst + kD + e] = creg[i] - g;
    }
    if (tid < kD) out[dst + tid] = eo[tid];
}

// observed senders: Lambda_out = C, eta_out = B y  (one workgroup per message; pure streaming)
__global__ __launch_bounds__(kBlock) void k_point64(const int kd, int nwork, const int32_t *__restrict__ work_slots, const int32_t *__restrict__ partner,
                                                    const int32_t *__restrict__ spdir, const double *__restrict__ ptab,
                                                    const double *__restrict__ v2f, double *__restrict__ out_a, double *__restrict__ out_b) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    __shared__ double y[64];
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int slot = work_slots[w], tid = threadIdx.x;
    const double *tab = ptab + (int64_t)spdir[slot] * 3 * kd * kd;
    if (tid < kd) y[tid] = v2f[(int64_t)slot * km + tid];
    __syncthreads();
    if (__builtin_isnan(y[0])) return;
    const int64_t dst = (int64_t)partner[slot] * km;
    for (int e = tid; e < kd * kd; e += kBlock) {
        const double c = tab[2 * kd * kd + e];
        out_a[dst + kd + e] = c;
        out_b[dst + kd + e] = c;
    }
    if (tid < kd) {
        double s = 0.0;
        for (int k = 0; k < kd; k++) s += tab[kd * kd + tid * kd + k] * y[k];
        out_a[dst + tid] = s;
        out_b[dst + tid] = s;
    }
}

// variable→factor message of a listed slot, on demand: the sum of the variable's other incoming messages
__global__ __launch_bounds__(kBlock) void k_v2f64(const int kd, int n, const int32_t *__restrict__ slots, const int32_t *__restrict__ vars,
                                                  const int32_t *__restrict__ vbase, const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                                  const double *__restrict__ f2v, double *__restrict__ v2f) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    const int w = blockIdx.x;
    if (w >= n) return;
    const int slot = slots[w], v = vars[w];
    const int info = vinfo[v];
    const bool big = (info & kDegMask) == kBigDeg;          // degree > 8: consecutive slots in the CSR tail
    const int deg = big ? vdeg[v] : (info & kDegMask), stride = big ? 1 : kBlock;
    if (deg < 2 || (info & (kClamped | kGhost))) return;
    const int base = vbase[v];
    for (int e = threadIdx.x; e < km; e += kBlock) {
        double acc = 0.0;
        for (int j = 0; j < deg; j++) {
            const int sj = base + j * stride;
            if (sj != slot) acc += f2v[(int64_t)sj * km + e];
        }
        v2f[(int64_t)slot * km + e] = acc;
    }
}

// ProductOfMessages of dim 64: the sum of a range of a variable's messages into a row of the product table; nothing is stored when one
// of them is undefined (the signal is not pending)
__global__ __launch_bounds__(kBlock) void k_range_sum64(const int kd, int n, const int32_t *__restrict__ rec, const double *__restrict__ f2v, double *__restrict__ out) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    const int w = blockIdx.x;
    if (w >= n) return;
    const int32_t *r = rec + 4 * (int64_t)w;      // {row of the product table, messages, first slot, slot stride (1 in the CSR tail, 256 in a slice)}
    const int dst = r[0], ns = r[1], first = r[2], stride = r[3];
    for (int j = 0; j < ns; j++) if (__builtin_isnan(f2v[(int64_t)(first + j * stride) * km + kd])) return;      // (whole messages are NaN together)
    for (int e = threadIdx.x; e < km; e += kBlock) {
        double acc = 0.0;
        for (int j = 0; j < ns; j++) acc += f2v[(int64_t)(first + j * stride) * km + e];
        out[(int64_t)dst * km + e] = acc;
    }
}

// damping (cx_set_damping) of a flooding sweep of dim 64: every message a rule record has just written into `out` becomes
// (1 - lambda) out + lambda old in natural form, `old` the message it replaces in the sweep's input buffer; an old message that is
// undefined does not damp (cx_scalar_core.h: damped).  One workgroup per rule record, after the rule launch.
__global__ __launch_bounds__(kBlock) void k_damp64(const int kd, int n, const int32_t *__restrict__ rec, const double *__restrict__ old, double *__restrict__ out, double lam) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    const int w = blockIdx.x;
    if (w >= n) return;
    const int64_t dst = (int64_t)rec[8 * (int64_t)w + 5] * km;
    if (__builtin_isnan(old[dst + kd]) || __builtin_isnan(out[dst + kd])) return;      // (whole messages are NaN together)
    for (int e = threadIdx.x; e < km; e += kBlock) out[dst + e] = (1.0 - lam) * out[dst + e] + lam * old[dst + e];
}

__global__ void k_fill64(const int kd, double *__restrict__ buf, int64_t nslots, double eta, double lam, const int32_t *__restrict__ partner) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    const int64_t s = blockIdx.x;
    if (s >= nslots || partner[s] < 0) return;
    if (!__builtin_isnan(buf[s * km + kd])) return;
    for (int e = threadIdx.x; e < km; e += blockDim.x) {
        double v = eta;
        if (e >= kd) { const int q = e - kd; v = (q / kd == q % kd) ? lam : 0.0; }
        buf[s * km + e] = v;
    }
}

// row-wise staging helpers for the message-major layout
__global__ void k_rows_scatter(const int kd, double *__restrict__ dst, const int32_t *__restrict__ idx, const double *__restrict__ val, int64_t n) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    for (int e = threadIdx.x; e < km; e += blockDim.x) dst[(int64_t)idx[i] * km + e] = val[i * km + e];
}
__global__ void k_rows_gather(const int kd, const double *__restrict__ src, const int32_t *__restrict__ idx, double *__restrict__ val, int64_t n) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    for (int e = threadIdx.x; e < km; e += blockDim.x) val[i * km + e] = src[(int64_t)idx[i] * km + e];
}
// observed data: eta = y, Lambda[0][0] = +inf marks the point mass (the rest of the slot is never read)
__global__ void k_set_point64(const int kd, double *__restrict__ dst, const int32_t *__restrict__ idx, const double *__restrict__ y, int64_t n) {
    const int km = kd + kd * kd;      // doubles per message record: eta[kd] | Lambda[kd][kd]
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    if (threadIdx.x < kd) dst[(int64_t)idx[i] * km + threadIdx.x] = y[i * kd + threadIdx.x];
    if (threadIdx.x == kd) dst[(int64_t)idx[i] * km + kd] = __builtin_inf();
}

// marginals of dim 16 / 32 (the wave kernel forms them: cx_mv64w.hip): sums[w] = the sum of EVERY incoming message of variable vars[w]
// (NaN if one is undefined); the rule records that read them, {-, w, -1, -1, table 0, destination row w, 0, 0}, are made on the host
// (written here by thread 0 as eight int stores, hipcc 7.2 merged them into two 16-byte stores and LOST the two -1: the first store went
// out with whatever v2, v3 held — the loop counter — and the rule kernel read a third source at slot 512)
__global__ __launch_bounds__(kBlock) void k_sum_all64(const int kd, int n, const int32_t *__restrict__ vars, const int32_t *__restrict__ vbase,
                                                      const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo, const double *__restrict__ f2v,
                                                      double *__restrict__ sums) {
    const int km = kd + kd * kd;
    const int w = blockIdx.x;
    if (w >= n) return;
    const int v = vars[w], info = vinfo[v];
    const bool big = (info & kDegMask) == kBigDeg;
    const int deg = big ? vdeg[v] : (info & kDegMask), stride = big ? 1 : kBlock, base = vbase[v];
    for (int e = threadIdx.x; e < km; e += kBlock) {
        double acc = deg > 0 ? 0.0 : __builtin_nan("");
        for (int j = 0; j < deg; j++) acc += f2v[(int64_t)(base + j * stride) * km + e];
        sums[(int64_t)w * km + e] = acc;
    }
}
// (mean | minus covariance) -> (mean | covariance), in place
__global__ void k_negate_cov64(const int kd, double *__restrict__ rows, int64_t n) {
    const int km = kd + kd * kd;
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    for (int e = kd + threadIdx.x; e < km; e += blockDim.x) rows[i * km + e] = -rows[i * km + e];
}

// ------------------------------------------------------------------------------------------------ launchers
void mv64_rows_scatter(cx_handle *h, double *dst, const int32_t *d_idx, const double *d_val, int64_t n) {
    if (n) hipLaunchKernelGGL(k_rows_scatter, dim3((unsigned)n), dim3(256), 0, h->stream, h->cfg.dim, dst, d_idx, d_val, n);
}
void mv64_rows_gather(cx_handle *h, const double *src, const int32_t *d_idx, double *d_val, int64_t n) {
    if (n) hipLaunchKernelGGL(k_rows_gather, dim3((unsigned)n), dim3(256), 0, h->stream, h->cfg.dim, src, d_idx, d_val, n);
}
void mv64_set_point(cx_handle *h, double *dst, const int32_t *d_idx, const double *d_y, int64_t n) {
    if (n) hipLaunchKernelGGL(k_set_point64, dim3((unsigned)n), dim3(128), 0, h->stream, h->cfg.dim, dst, d_idx, d_y, n);
}

void mv64_launch_rule(cx_handle *h, int nwork, const int32_t *d_rec, const double *f2v_in, double *f2v_out, int kernel_id) {
    if (nwork == 0) return;
    h->prof_armed = false;
    if (h->profiling && (h->prof_count[kernel_id]++ % h->prof_stride) == 0) {
        h->prof_armed = true;
        ProfileRec r; r.kernel = kernel_id;
        (void)hipEventCreate(&r.start); (void)hipEventCreate(&r.stop); (void)hipEventRecord(r.start, h->stream);
        h->recs.push_back(r);
    }
    static const bool classic = getenv("CX_RULE64_CLASSIC") != nullptr;   // A/B switch: the 80 KB form, two workgroups per CU
    // default: the wave-per-message form (cx_mv64w.hip, 5.1 ms per C5 sweep); CX_RULE64=g selects the workgroup-per-message
    // form below (8.8 ms).  Read per launch so that a test can switch forms inside one process.
    const char *form_env = getenv("CX_RULE64");
    const int wave_form = (form_env && form_env[0] == 'g' && h->cfg.dim == kD) ? 0 : 1;      // (the workgroup forms are written for 64)
    if (wave_form && (!classic || h->cfg.dim != kD))
        mv64w_launch_rule(h, nwork, d_rec, f2v_in, f2v_out);
    else if (classic)
        hipLaunchKernelGGL((k_rule64<0>), dim3(nwork), dim3(kBlock), 0, h->stream, nwork, d_rec, (const int32_t *)nullptr, h->d_vbase, h->d_var_deg, h->d_vinfo,
                           h->d_ptab, f2v_in, h->d_mv_v2f, f2v_out);
    else
        hipLaunchKernelGGL(k_rule64s, dim3(nwork), dim3(kBlock), 0, h->stream, nwork, d_rec, h->d_ptab, f2v_in, h->d_mv_v2f, f2v_out);
    if (h->profiling && h->prof_armed) (void)hipEventRecord(h->recs.back().stop, h->stream);
}

void mv64_launch_marginals(cx_handle *h, int n, const int32_t *d_vars, const double *f2v, double *out) {
    if (n == 0) return;
    if (h->cfg.dim != kD) {
        // dim 16 / 32: the sum of a variable's messages through the wave kernel's rule with P = 0, B = I, C = 0: eta_out = M^-1 eta (the
        // mean), Lambda_out = -M^-1; scratch (sums + records) and the identity table live with the handle
        const int kd = h->cfg.dim, km = kd + kd * kd;
        if (h->marg64_cap < n) {
            if (h->d_marg64_sums) (void)hipFree(h->d_marg64_sums);
            if (h->d_marg64_rec) (void)hipFree(h->d_marg64_rec);
            h->d_marg64_sums = nullptr; h->d_marg64_rec = nullptr; h->marg64_cap = 0;
            if (hipMalloc((void **)&h->d_marg64_sums, (size_t)n * km * 8) != hipSuccess || hipMalloc((void **)&h->d_marg64_rec, (size_t)n * 32) != hipSuccess) { (void)hipGetLastError(); return; }      // (the rows stay UndefValue())
            h->marg64_cap = n;
            std::vector<int32_t> rec((size_t)n * 8, 0);
            for (int w = 0; w < n; w++) { rec[8 * (size_t)w + 1] = w; rec[8 * (size_t)w + 2] = -1; rec[8 * (size_t)w + 3] = -1; rec[8 * (size_t)w + 5] = w; }
            if (hipMemcpy(h->d_marg64_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); h->marg64_cap = 0; return; }
        }
        if (!h->d_marg64_tab) {
            std::vector<double> tab((size_t)4 * kd * kd, 0.0);      // P = 0 | B = I | C = 0 | B' = I
            for (int r = 0; r < kd; r++) { tab[(size_t)kd * kd + (size_t)r * kd + r] = 1.0; tab[(size_t)3 * kd * kd + (size_t)r * kd + r] = 1.0; }
            if (hipMalloc((void **)&h->d_marg64_tab, tab.size() * 8) != hipSuccess) { (void)hipGetLastError(); h->d_marg64_tab = nullptr; return; }
            (void)hipMemcpy(h->d_marg64_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
        }
        hipLaunchKernelGGL(k_sum_all64, dim3(n), dim3(kBlock), 0, h->stream, kd, n, d_vars, h->d_vbase, h->d_var_deg, h->d_vinfo, f2v, h->d_marg64_sums);
        mv64w_launch_marginal_rule(h, n, h->d_marg64_rec, h->d_marg64_tab, h->d_marg64_tab + (size_t)3 * kd * kd, h->d_marg64_sums, out);
        hipLaunchKernelGGL(k_negate_cov64, dim3(n), dim3(256), 0, h->stream, kd, out, (int64_t)n);
        return;
    }
    hipLaunchKernelGGL((k_rule64<1>), dim3(n), dim3(kBlock), 0, h->stream, n, (const int32_t *)nullptr, d_vars,
                       h->d_vbase, h->d_var_deg, h->d_vinfo, h->d_ptab, f2v, h->d_mv_v2f, out);
}

void mv64_launch_point(cx_handle *h, int nwork, const int32_t *d_slots, double *out_a, double *out_b) {
    if (nwork == 0) return;
    hipLaunchKernelGGL(k_point64, dim3(nwork), dim3(kBlock), 0, h->stream, h->cfg.dim, nwork, d_slots, h->d_partner, h->d_spdir, h->d_ptab,
                       h->d_mv_v2f, out_a, out_b);
}

void mv64_launch_v2f(cx_handle *h, int n, const int32_t *d_slots, const int32_t *d_vars, const double *f2v) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_v2f64, dim3(n), dim3(kBlock), 0, h->stream, h->cfg.dim, n, d_slots, d_vars, h->d_vbase, h->d_var_deg, h->d_vinfo, f2v, h->d_mv_v2f);
}

void mv64_launch_range_sums(cx_handle *h, int n, const int32_t *d_rec4, const double *f2v, double *out) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_range_sum64, dim3(n), dim3(kBlock), 0, h->stream, h->cfg.dim, n, d_rec4, f2v, out);
}

void mv64_launch_damp(cx_handle *h, int n, const int32_t *d_rec, const double *old, double *out, double lam) {
    if (n == 0 || lam == 0.0) return;
    hipLaunchKernelGGL(k_damp64, dim3(n), dim3(kBlock), 0, h->stream, h->cfg.dim, n, d_rec, old, out, lam);
}

void mv64_launch_seed(cx_handle *h, double *buf, double eta, double lam) {
    hipLaunchKernelGGL(k_fill64, dim3((unsigned)h->nslots), dim3(256), 0, h->stream, h->cfg.dim, buf, h->nslots, eta, lam, h->d_partner);
}

}  // namespace cx
