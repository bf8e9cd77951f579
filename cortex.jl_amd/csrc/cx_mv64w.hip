// cx_mv64w.hip — the d = 64 (round 6: also 16 and 32) factor→variable rule with ONE WAVE PER MESSAGE and the matrices resident in registers.
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

// work record (8 int32, built by build_work64 in cx_api.hip): {sender slot, three source slots (-1: none), rule-table index,
// destination slot, flags, 0}.  ptab: per table index (P, B, C); btab: per table index B' (the transpose of B).
//
// Register plan for TWO waves per SIMD (<= 256 registers each), so that one wave's diagonal-tile chains and memory round trips
// hide behind the other's matrix instructions:
//   factorisation   M upper tiles (80) + the column-layout temporaries of diag_factor (~100); V_k go to LDS (2 KB each)
//   solve           U off-diagonal (48) + Yt column blocks as they are finished (<= 160) + one V tile / negated U tile (16)
//   Gram + store    Yt (160) + one G tile, one C tile; every G tile leaves for HBM as soon as its 16 instructions are done
template <int WAVES_PER_SIMD, int NT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES_PER_SIMD, WAVES_PER_SIMD)))
void k_rule64w(int nwork, const int32_t *__restrict__ work_rec, const double *__restrict__ ptab, const double *__restrict__ btab,
               const double *__restrict__ zero_msg, const double *__restrict__ f2v_in, const double *__restrict__ v2f,
               double *__restrict__ out) {
    constexpr int KD = 16 * NT, KM = KD + KD * KD;      // (round 6) NT x NT tiles of 16: d = 16, 32, 64; a record is eta[KD] | Lambda[KD][KD]
    __shared__ double S[16 * kLdT];
    __shared__ double Vs[NT][16 * kLdT];
    const int w = blockIdx.x;
    if (w >= nwork) return;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int32_t *rec = work_rec + 8 * (int64_t)w;
    const int slot = rec[0], s0 = rec[1], s1 = rec[2], s2 = rec[3], dst_slot = rec[5], flags = rec[6];
    const double *tab = ptab + (int64_t)rec[4] * 3 * KD * KD;
    const double *bt = btab + (int64_t)rec[4] * KD * KD;
    const bool fixed = (flags & kFlagFixed) != 0;
    // an absent source reads a message of zeros: a "load or skip" choice per element — even a wave-uniform one — makes hipcc
    // branch around every load and wait for it alone (336 dependent round trips per message in the first version of this kernel)
    const double *src0 = fixed ? v2f + (int64_t)slot * KM : (s0 >= 0 ? f2v_in + (int64_t)s0 * KM : zero_msg);
    const double *src1 = (!fixed && s1 >= 0) ? f2v_in + (int64_t)s1 * KM : zero_msg;
    const bool has2 = !fixed && s2 >= 0;
    const double *src2 = has2 ? f2v_in + (int64_t)s2 * KM : zero_msg;

    (void)rule64w_apply<false, false, NT>((gcdp)tab, (gcdp)bt, (gcdp)(tab + 2 * KD * KD), nullptr, nullptr, (gcdp)src0, (gcdp)src1, (gcdp)src2, has2,
                                          (gdp)(out + (int64_t)dst_slot * KM), S, Vs, lane, g, c);
}

// the launch for the handle's tile count; the tables and the stored variable→factor messages may be given (the marginal read-out of
// dim 16 / 32 runs the rule on a table of its own: P = 0, B = I, C = 0 gives (M^-1 eta, -M^-1))
static void launch_rule_tiles(cx_handle *h, int nwork, const int32_t *d_rec, const double *ptab, const double *btab, const double *f2v_in, const double *v2f, double *out) {
    static const int one = [] { const char *e = getenv("CX_RULE64_WAVES"); return (e && e[0] == '1') ? 1 : 0; }();
    // (holding back the odd wave slot of every SIMD's first pair by half a message, so that the two waves would not run their
    // vector and matrix phases in lockstep, was measured in tools/ab_c5.py: no difference at 2, 3, 4 or 6 x 8k cycles)
    // 1 x 1 and 2 x 2 tiles hold their matrices in a few registers: four waves per SIMD (128 registers: the diagonal tile's column arrays)
#define CX_R(W, N) hipLaunchKernelGGL((k_rule64w<W, N>), dim3(nwork), dim3(64), 0, h->stream, nwork, d_rec, ptab, btab, h->d_zero_msg, f2v_in, v2f, out)
    if (h->cfg.dim == 16) CX_R(4, 1);
    else if (h->cfg.dim == 32) CX_R(4, 2);
    else if (one) CX_R(1, 4);
    else CX_R(2, 4);
#undef CX_R
}

void mv64w_launch_rule(cx_handle *h, int nwork, const int32_t *d_rec, const double *f2v_in, double *f2v_out) {
    launch_rule_tiles(h, nwork, d_rec, h->d_ptab, h->d_ptab_bt, f2v_in, h->d_mv_v2f, f2v_out);
}

// marginals of dim 16 / 32 (dim 64: k_rule64<1>, cx_mv64.hip): `sums` holds the sum of every incoming message of the w-th listed
// variable (k_sum_all64), `d_rec` one record per variable {-, w, -1, -1, 0, w, 0, 0}; out[w] = (mean | MINUS the covariance)
void mv64w_launch_marginal_rule(cx_handle *h, int n, const int32_t *d_rec, const double *ident_tab, const double *ident_bt, const double *sums, double *out) {
    if (n) launch_rule_tiles(h, n, d_rec, ident_tab, ident_bt, sums, sums, out);
}

}  // namespace cx
