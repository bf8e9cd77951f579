// cx_kary_mv_core.h — one factor→variable message of a linear-Gaussian factor of MORE than two d-dimensional variables (d = 2, 3, 4),
//     x_out = A_1 x_1 + ... + A_k x_k + N(0, Q),   k = 2 .. 6 inputs  (CX_FACTOR_GAUSS_LINEAR_N, dim > 1; round 5)
// by ONE thread (entry = 8 * row + edge position in the k-ary table: the OUT edge first, then the IN edges by ascending variable id).
// The reference wires every message out of a factor to ALL its other variables' messages into it (src/dependencies.jl:17-31); the rule
// itself is the user's in the reference and exists nowhere in it for d > 1 ("parity unpinned": pinned by dense joint solves and by the
// moment-form formulas evaluated in numpy, tests/test_gpu_kary_mv.py).  With the other variables' messages in moment form (m_i, V_i):
//     to x_out:  N( sum_i A_i m_i,  Q + sum_i A_i V_i A_i' )
//     to x_j:    y = A_j x_j + e with y ~ N(mu, S),  mu = m_out - sum_{i != j} A_i m_i,  S = V_out + Q + sum_{i != j} A_i V_i A_i'
//                => natural parameters  Lambda = A_j' S^-1 A_j,  eta = A_j' S^-1 mu
// A point mass (an observed variable) is (y, V = 0).  An undefined input leaves the message what it was (the signal is not pending).
// Shared by the sweep's kernel (cx_kary_mv.hip) and the batch / tree-stage items (cx_mvbatch.hip), like cx_kary_core.h for scalars.
#pragma once

#include "cx_mv_core.h"

namespace cx {

struct KaryMvTab {
    const int32_t *slot;      // [8 rows]: slot of the entry's edge, -1 = no such edge
    const int32_t *pset;      // [8 rows]: parameter set of the entry — an IN entry's A, the OUT entry's Q
    const double *aq;         // [sets][2][D * D]: A | Q, row-major
};

// moment form of a stored natural-form message: (mean, packed covariance); a point mass (y, +inf) is (y, 0)
template <int D>
__device__ __forceinline__ Msg<D> kary_mv_moment(const Msg<D> &nat) {
    if (nat.lam[0] == __builtin_inf()) {
        Msg<D> m;
#pragma unroll
        for (int i = 0; i < D; i++) m.eta[i] = nat.eta[i];
#pragma unroll
        for (int i = 0; i < Msg<D>::NT; i++) m.lam[i] = 0.0;
        return m;
    }
    return mv_to_moment<D>(nat);
}

template <int D>
__device__ __forceinline__ double sym_at(const double (&s)[Msg<D>::NT], int i, int j) { return s[i <= j ? tri<D>(i, j) : tri<D>(j, i)]; }

template <int D>
__device__ __forceinline__ void kary_item_mv(int en, const KaryMvTab kt, const double *__restrict__ v2f, double *__restrict__ f2v_out,
                                             const double *__restrict__ prev, double lam) {
    const int row = en >> 3, e = en & 7, dst = kt.slot[en];
    if (dst < 0) return;
    const double *Q = kt.aq + (size_t)kt.pset[8 * row] * 2 * D * D + D * D;
    // mu / S accumulate: the OUT entry's own message enters with +, the IN entries' transformed messages with + (to x_out) or - (to x_j)
    double mu[D], S[Msg<D>::NT];
#pragma unroll
    for (int i = 0; i < D; i++) mu[i] = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = i; j < D; j++) S[tri<D>(i, j)] = Q[i * D + j];
    for (int r = 0; r < 8; r++) {
        const int o = 8 * row + r, s = kt.slot[o];
        if (s < 0 || r == e) continue;
        const Msg<D> mo = kary_mv_moment<D>(slot_load<D>(v2f, s));
        if (r == 0) {                       // the OUT variable's message: (m_out, V_out)
#pragma unroll
            for (int i = 0; i < D; i++) mu[i] += mo.eta[i];
#pragma unroll
            for (int i = 0; i < Msg<D>::NT; i++) S[i] += mo.lam[i];
        } else {                            // an IN variable's: (A m, A V A')
            const double *A = kt.aq + (size_t)kt.pset[o] * 2 * D * D;
            double AV[D][D];
#pragma unroll
            for (int i = 0; i < D; i++) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < D; k++) t += A[i * D + k] * mo.eta[k];
                mu[i] += e == 0 ? t : -t;
#pragma unroll
                for (int j = 0; j < D; j++) {
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < D; k++) a += A[i * D + k] * sym_at<D>(mo.lam, k, j);
                    AV[i][j] = a;
                }
            }
#pragma unroll
            for (int i = 0; i < D; i++)
#pragma unroll
                for (int j = i; j < D; j++) {
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < D; k++) a += AV[i][k] * A[j * D + k];
                    S[tri<D>(i, j)] += a;
                }
        }
    }
    Msg<D> out;
    double Lm[D][D], ri[D];
    chol<D>(S, nullptr, Lm, ri);            // an undefined or indefinite input: NaN, nothing is stored
    double z[D];
#pragma unroll
    for (int i = 0; i < D; i++) z[i] = mu[i];
    fwd_solve<D>(Lm, ri, z);
    double Y[D][D];                          // Y = L^-1 G, G = I (to x_out) or A_j (to x_j): Lambda = Y'Y, eta = Y'z
    const double *Aj = e == 0 ? nullptr : kt.aq + (size_t)kt.pset[en] * 2 * D * D;
#pragma unroll
    for (int c = 0; c < D; c++) {
        double col[D];
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = Aj ? Aj[k * D + c] : (k == c ? 1.0 : 0.0);
        fwd_solve<D>(Lm, ri, col);
#pragma unroll
        for (int k = 0; k < D; k++) Y[k][c] = col[k];
    }
#pragma unroll
    for (int i = 0; i < D; i++) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; k++) t += Y[k][i] * z[k];
        out.eta[i] = t;
#pragma unroll
        for (int j = i; j < D; j++) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) a += Y[k][i] * Y[k][j];
            out.lam[tri<D>(i, j)] = a;
        }
    }
    if (__builtin_isnan(out.lam[0]) || __builtin_isnan(out.eta[0])) return;
    if (lam != 0.0 && prev) {                // cx_set_damping: against the message this one replaces
        const Msg<D> old = slot_load<D>(prev, dst);
        if (!__builtin_isnan(old.lam[0])) {
#pragma unroll
            for (int c = 0; c < D; c++) out.eta[c] = (1.0 - lam) * out.eta[c] + lam * old.eta[c];
#pragma unroll
            for (int c = 0; c < Msg<D>::NT; c++) out.lam[c] = (1.0 - lam) * out.lam[c] + lam * old.lam[c];
        }
    }
    slot_store<D>(f2v_out, dst, out);
}

}  // namespace cx
