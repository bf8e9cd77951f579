// cx_mv64w.hip — the d = 64 factor→variable rule with ONE WAVE PER MESSAGE and the matrices resident in registers.
//
// Why: in the workgroup-per-message form (cx_mv64.hip, k_rule64s) the f64 vector pipe carries as many SIMD
// cycles as its 584 matrix instructions — 77 % of them the 4 x 4 pivot factorisations that all four waves repeat — and the
// ≈ 40 barrier-separated phases of a message leave issue slots empty.  Here a message belongs to one wave: no workgroup
// barrier, every pivot computed once, and no operand ever goes through LDS for a matrix product.
//
// The rule (same as cx_mv64.hip; P, B, C the receiving edge's tables, M = Lambda_in + P):
//     Lambda_out = C - B M^-1 B',      eta_out = B M^-1 eta_in.
// With the UPPER factor M = U'U, Yt = U^-T B' and z = U^-T eta_in this is  Lambda_out = C - Yt' Yt,  eta_out = Yt' z.
//
// Accumulators as operands.  A 16 x 16 tile T lives in the layout v_mfma_f64_16x16x4_f64 returns: lane l = (g, c) =
// (l >> 4, l & 15) holds T[g + 4 r][c] in register r = 0..3.  For that instruction lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; so register s of a tile T IS the A operand of T' and register s of a tile S the B operand of S, for
// the k-step that covers rows 4s..4s+3:        T' S  =  sum_{s = 0..3} mfma(T.reg[s], S.reg[s]).
// Every product of the upper-factor formulation contracts over tile ROWS — panel  U[k][j] = V_k' M[k][j]  (V_k = U_kk^-1),
// trailing update  M[i][j] -= U[k][i]' U[k][j],  solve  Yt[j] = V_j' R[j],  R[j'] -= U[j][j']' Yt[j],  Gram  G[a][b] += Yt[j][a]' Yt[j][b]
// — so all 384 matrix instructions (Cholesky 64, solve 160, Gram 160) read their operands straight from the registers the
// previous ones wrote.  eta rides on the vector pipe (z = U^-T eta, eta_out = Yt' z).
//
// The only work outside the matrix pipe is the 16 x 16 diagonal tile: it goes through a 2 KB LDS transpose into "lane c holds
// column c", is factored and inverted there with DPP row broadcasts (pivot i: one rsqrt chain, 15 - i independent
// v_fmac_f64_dpp per array), and comes back as V_k in tile layout.  255 VGPRs: two waves per SIMD, eight messages per CU in flight.
//
// The reference has no such rule (parity unpinned for d > 1); the kernel is checked against the numpy / C
// restatements every sweep, the exact block-tridiagonal smoother, and the workgroup-per-message kernel (same results to
// rounding: the factorisation order differs, upper instead of lower).

#include <cstdlib>

#include "cx_internal.h"
#include "cx_mv64w_core.h"

namespace cx {

using namespace w64;

// lab only (tools/lab/w64_phases.hip defines CX_W64_STAMPS and the counters): shader-clock cycles per phase and wave
#ifdef CX_W64_STAMPS
#define W64_STAMP(i)                                                                                  \
    do {                                                                                              \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();                                             \
        if (lane == 0) cx_w64_stamps[8 * (size_t)blockIdx.x + i] += (unsigned long long)(t_ - t_prev); \
        t_prev = t_;                                                                                  \
    } while (0)
#define W64_STAMP_INIT uint64_t t_prev = __builtin_amdgcn_s_memtime()
#else
#define W64_STAMP(i)
#define W64_STAMP_INIT
#endif

// work record (8 int32, built by build_work64 in cx_api.hip): {sender slot, three source slots (-1: none), rule-table index,
// destination slot, flags, 0}.  ptab: per table index (P, B, C); btab: per table index B' (the transpose of B).
//
// Register plan for TWO waves per SIMD (<= 256 registers each), so that one wave's diagonal-tile chains and memory round trips
// hide behind the other's matrix instructions:
//   factorisation   M upper tiles (80) + the column-layout temporaries of diag_factor (~100); V_k go to LDS (2 KB each)
//   solve           U off-diagonal (48) + Yt column blocks as they are finished (<= 160) + one V tile / negated U tile (16)
//   Gram + store    Yt (160) + one G tile, one C tile; every G tile leaves for HBM as soon as its 16 instructions are done
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES_PER_SIMD, WAVES_PER_SIMD)))
void k_rule64w(int nwork, const int32_t *__restrict__ work_rec, const double *__restrict__ ptab, const double *__restrict__ btab,
               const double *__restrict__ zero_msg, const double *__restrict__ f2v_in, const double *__restrict__ v2f,
               double *__restrict__ out) {
    __shared__ double S[16 * kLdT];
    __shared__ double Vs[4][16 * kLdT];
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    W64_STAMP_INIT;
    const int32_t *rec = work_rec + 8 * (int64_t)w;
    const int slot = rec[0], s0 = rec[1], s1 = rec[2], s2 = rec[3], dst_slot = rec[5], flags = rec[6];
    const double *tab = ptab + (int64_t)rec[4] * 3 * kD * kD;
    const double *bt = btab + (int64_t)rec[4] * kD * kD;
    const bool fixed = (flags & kFlagFixed) != 0;
    // an absent source reads a message of zeros: a "load or skip" choice per element — even a wave-uniform one — makes hipcc
    // branch around every load and wait for it alone (336 dependent round trips per message in the first version of this kernel)
    const double *src0 = fixed ? v2f + (int64_t)slot * kMsg : (s0 >= 0 ? f2v_in + (int64_t)s0 * kMsg : zero_msg);
    const double *src1 = (!fixed && s1 >= 0) ? f2v_in + (int64_t)s1 * kMsg : zero_msg;
    const bool has2 = !fixed && s2 >= 0;
    const double *src2 = has2 ? f2v_in + (int64_t)s2 * kMsg : zero_msg;

    // ---- M = P + sum of the other incoming Lambdas (ascending neighbour order), upper tiles only ------------------------------
    d4 M[10];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = a; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = tile_off(a, b, r, g, c);
                M[ut(a, b)][r] = (tab[o] + src0[kD + o]) + src1[kD + o];
            }
    if (has2) {      // a third source (a sender of degree 4) is rare: ONE branch around the whole block of loads
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = a; b < 4; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) M[ut(a, b)][r] += src2[kD + tile_off(a, b, r, g, c)];
    }
    // a dependency is undefined (whole messages are NaN together): the signal is not pending
    if (__builtin_isnan(bcast(M[0][0], 0))) return;
    W64_STAMP(0);

    // ---- blocked upper Cholesky, NB = 16: off-diagonal tiles of M become U, V_k = U_kk^-1 goes to LDS -------------------------
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const d4 Vk = diag_factor(M[ut(k, k)], S, g, c);
        W64_STAMP(1);
#pragma unroll
        for (int r = 0; r < 4; r++) Vs[k][(g + 4 * r) * kLdT + c] = Vk[r];
#pragma unroll
        for (int j = k + 1; j < 4; j++) M[ut(k, j)] = tts(Vk, M[ut(k, j)], d4{0.0, 0.0, 0.0, 0.0});        // U[k][j] = V_k' M[k][j]
#pragma unroll
        for (int i = k + 1; i < 4; i++) {
            const d4 nu = neg(M[ut(k, i)]);
#pragma unroll
            for (int j = i; j < 4; j++) M[ut(i, j)] = tts(nu, M[ut(k, j)], M[ut(i, j)]);                       // M[i][j] -= U[k][i]' U[k][j]
        }
        W64_STAMP(2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- z = U^-T eta_in on the vector pipe (as a fifth block column of the matrix solve it cost 104 of 488 matrix
    //      instructions, on tiles that are 15/16 zeros).  Two vector layouts: RV — lane (g, c) holds x[g + 4 r] in register r
    //      (indexed like tile rows); CV — lane (g, c) holds y[c] (indexed like tile columns).  T' x for a tile T: four FMAs per lane
    //      and a sum over the four lane groups gives CV; CV -> RV is four lane reads. -----------------------------------------------
    double zrv[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int e = 16 * j + c;
        double wcv = src0[e] + src1[e];
        if (has2) wcv += src2[e];
#pragma unroll
        for (int k = 0; k < j; k++) {
            double p = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) p += M[ut(k, j)][r] * zrv[k][r];
            wcv -= sum_groups(p);                                                      // eta_j - sum_k U[k][j]' z_k
        }
        double p = 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++) p += Vs[j][(g + 4 * r) * kLdT + c] * cv_to_rv(wcv, g, r);
        const double zcv = sum_groups(p);                                              // z_j = V_j' w_j
#pragma unroll
        for (int r = 0; r < 4; r++) zrv[j][r] = cv_to_rv(zcv, g, r);
    }

    W64_STAMP(3);
    // ---- Yt = U^-T B', one block COLUMN at a time (forward substitution over its four row blocks) ------------------------------
    d4 Y[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) Y[j][b][r] = bt[tile_off(j, b, r, g, c)];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d4 Vj;
#pragma unroll
            for (int r = 0; r < 4; r++) Vj[r] = Vs[j][(g + 4 * r) * kLdT + c];
            Y[j][b] = tts(Vj, Y[j][b], d4{0.0, 0.0, 0.0, 0.0});                                                  // Yt[j] = V_j' R[j]
#pragma unroll
            for (int jj = j + 1; jj < 4; jj++) Y[jj][b] = tts(neg(M[ut(j, jj)]), Y[j][b], Y[jj][b]);           // R[jj] -= U[j][jj]' Yt[j]
        }
    }
    // not positive definite somewhere: NaN everywhere downstream — leave the old message
    if (__builtin_isnan(bcast(Y[3][0][0], 0)) || __builtin_isnan(bcast(zrv[3][0], 0))) return;
    W64_STAMP(4);

    // ---- Gram tile by tile: G[a][b] = sum_j Yt[j][a]' Yt[j][b];  Lambda_out = C - G (C symmetric: the lower tiles are the
    //      transposes of the same differences, turned through LDS);  eta_out = Yt' z on the vector pipe ---------------------------
    double *dst = out + (int64_t)dst_slot * kMsg;
    const double *C = tab + 2 * kD * kD;
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
        for (int b = a; b < 4; b++) {
            d4 Ct;
#pragma unroll
            for (int r = 0; r < 4; r++) Ct[r] = C[tile_off(a, b, r, g, c)];
            d4 G = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 4; j++) G = tts(Y[j][a], Y[j][b], G);
            d4 D;
#pragma unroll
            for (int r = 0; r < 4; r++) D[r] = Ct[r] - G[r];
#pragma unroll
            for (int r = 0; r < 4; r++) dst[kD + tile_off(a, b, r, g, c)] = D[r];
            if (b > a) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the previous tile's reads have returned
#pragma unroll
                for (int r = 0; r < 4; r++) S[(g + 4 * r) * kLdT + c] = D[r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; r++) dst[kD + tile_off(b, a, r, g, c)] = S[c * kLdT + g + 4 * r];
            }
        }
        double p = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) p += Y[j][a][r] * zrv[j][r];
        const double ecv = sum_groups(p);                                              // (Yt' z)[16 a + c]
        if (g == 0) dst[16 * a + c] = ecv;
    }
    W64_STAMP(5);
}

void mv64w_launch_rule(cx_handle *h, int nwork, const int32_t *d_rec, const double *f2v_in, double *f2v_out) {
    static const int one = [] { const char *e = getenv("CX_RULE64_WAVES"); return (e && e[0] == '1') ? 1 : 0; }();
    // (holding back the odd wave slot of every SIMD's first pair by half a message, so that the two waves would not run their
    // vector and matrix phases in lockstep, was measured in tools/ab_c5.py: no difference at 2, 3, 4 or 6 x 8k cycles)
    if (one)
        hipLaunchKernelGGL(k_rule64w<1>, dim3(nwork), dim3(64), 0, h->stream, nwork, d_rec, h->d_ptab, h->d_ptab_bt, h->d_zero_msg, f2v_in,
                           h->d_mv_v2f, f2v_out);
    else
        hipLaunchKernelGGL(k_rule64w<2>, dim3(nwork), dim3(64), 0, h->stream, nwork, d_rec, h->d_ptab, h->d_ptab_bt, h->d_zero_msg, f2v_in,
                           h->d_mv_v2f, f2v_out);
}

}  // namespace cx
