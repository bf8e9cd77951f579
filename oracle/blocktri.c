/* oracle/blocktri.c — exact posterior of a linear-Gaussian state-space chain by a block-tridiagonal solve, in C so that the
 * full-size configs (T = 1e6, d = 4) can be checked at EVERY time step.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference pins nothing numeric for Gaussian belief propagation (its SSM test asserts signs and monotonicity only,
 * test/inference_engine_tests.jl:485-487); on a tree the sum-product marginals equal the marginals of the joint Gaussian
 * whatever the schedule, so the mathematics is the oracle.  Same statement as oracle/exact.py:block_tridiag_posterior
 * (which pins this file in tests/test_blocktri_checker.py): Schur complements from both ends,
 *     L_t = J_tt - J_{t,t-1} L_{t-1}^-1 J_{t-1,t},   R_t = J_tt - J_{t,t+1} R_{t+1}^-1 J_{t+1,t},
 *     precision_t = L_t + R_t - J_tt,   mean_t = precision_t^-1 (hl_t + hr_t - h_t).
 * Dense d x d solves by Gaussian elimination with partial pivoting — no Cholesky, no message-passing form: nothing here shares
 * a formulation with the device kernels it checks.
 *
 * Model: x_{t+1} = A x_t + w, w ~ N(0, Q);  y_t = x_t + v, v ~ N(0, R);  no prior on x_1 (oracle/exact.py:lgssm_posterior with H = I).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* solve M X = B in place (M d x d, B d x m, row-major); returns 0 if singular */
static int lu_solve(int d, int m, double *M, double *B) {
    for (int c = 0; c < d; c++) {
        int piv = c;
        for (int r = c + 1; r < d; r++) if (fabs(M[r * d + c]) > fabs(M[piv * d + c])) piv = r;
        if (M[piv * d + c] == 0.0) return 0;
        if (piv != c) {
            for (int k = 0; k < d; k++) { double t = M[c * d + k]; M[c * d + k] = M[piv * d + k]; M[piv * d + k] = t; }
            for (int k = 0; k < m; k++) { double t = B[c * m + k]; B[c * m + k] = B[piv * m + k]; B[piv * m + k] = t; }
        }
        for (int r = c + 1; r < d; r++) {
            const double f = M[r * d + c] / M[c * d + c];
            if (f == 0.0) continue;
            for (int k = c; k < d; k++) M[r * d + k] -= f * M[c * d + k];
            for (int k = 0; k < m; k++) B[r * m + k] -= f * B[c * m + k];
        }
    }
    for (int r = d - 1; r >= 0; r--)
        for (int k = 0; k < m; k++) {
            double s = B[r * m + k];
            for (int c = r + 1; c < d; c++) s -= M[r * d + c] * B[c * m + k];
            B[r * m + k] = s / M[r * d + r];
        }
    return 1;
}

static void matmul(int d, const double *X, const double *Y, double *Z) {   /* Z = X Y */
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int k = 0; k < d; k++) s += X[i * d + k] * Y[k * d + j];
            Z[i * d + j] = s;
        }
}

/* y[T][d]; A, Q, R d x d row-major.  Out: mean[T][d], cov[T][d][d].  Returns 0 on success, -1 on a singular block, -2 on allocation failure. */
int32_t cxo_lgssm_posterior(int32_t d, int64_t T, const double *y, const double *A, const double *Q, const double *R, double *mean, double *cov) {
    const int dd = d * d;
    double *Qi = malloc(sizeof(double) * dd), *Ri = malloc(sizeof(double) * dd), *AtQi = malloc(sizeof(double) * dd), *AtQiA = malloc(sizeof(double) * dd);
    double *tmpM = malloc(sizeof(double) * dd), *G = malloc(sizeof(double) * dd), *W = malloc(sizeof(double) * (dd + d));
    double *Ld = malloc(sizeof(double) * (size_t)T * dd), *hl = malloc(sizeof(double) * (size_t)T * d), *h = malloc(sizeof(double) * (size_t)T * d);
    int rc = 0;
    if (!Qi || !Ri || !AtQi || !AtQiA || !tmpM || !G || !W || !Ld || !hl || !h) { rc = -2; goto done; }
    /* Qi = Q^-1, Ri = R^-1 */
    for (int i = 0; i < dd; i++) { Qi[i] = (i / d == i % d) ? 1.0 : 0.0; Ri[i] = Qi[i]; }
    memcpy(tmpM, Q, sizeof(double) * dd); if (!lu_solve(d, d, tmpM, Qi)) { rc = -1; goto done; }
    memcpy(tmpM, R, sizeof(double) * dd); if (!lu_solve(d, d, tmpM, Ri)) { rc = -1; goto done; }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) { double s = 0.0; for (int k = 0; k < d; k++) s += A[k * d + i] * Qi[k * d + j]; AtQi[i * d + j] = s; }
    matmul(d, AtQi, A, AtQiA);
    /* J_tt = Ri + [t < T-1] A'QiA + [t > 0] Qi;  J_{t,t+1} = -A'Qi;  h_t = Ri y_t */
#define JDIAG(t, out) do { for (int i_ = 0; i_ < dd; i_++) (out)[i_] = Ri[i_] + ((t) < T - 1 ? AtQiA[i_] : 0.0) + ((t) > 0 ? Qi[i_] : 0.0); } while (0)
    for (int64_t t = 0; t < T; t++)
        for (int i = 0; i < d; i++) { double s = 0.0; for (int k = 0; k < d; k++) s += Ri[i * d + k] * y[t * d + k]; h[t * d + i] = s; }
    /* forward: L_t, hl_t.  L_t = J_tt - Jo' L_{t-1}^-1 Jo with Jo = J_{t-1,t} = -A'Qi;  hl_t = h_t - Jo' L_{t-1}^-1 hl_{t-1} */
    JDIAG(0, Ld);
    memcpy(hl, h, sizeof(double) * d);
    for (int64_t t = 1; t < T; t++) {
        /* W = L_{t-1}^-1 [Jo | hl_{t-1}]  (d x (d+1)) */
        memcpy(tmpM, Ld + (t - 1) * dd, sizeof(double) * dd);
        for (int i = 0; i < d; i++) { for (int j = 0; j < d; j++) W[i * (d + 1) + j] = -AtQi[i * d + j]; W[i * (d + 1) + d] = hl[(t - 1) * d + i]; }
        if (!lu_solve(d, d + 1, tmpM, W)) { rc = -1; goto done; }
        JDIAG(t, Ld + t * dd);
        for (int i = 0; i < d; i++) {
            double sh = h[t * d + i];
            for (int k = 0; k < d; k++) sh -= (-AtQi[k * d + i]) * W[k * (d + 1) + d];
            hl[t * d + i] = sh;
            for (int j = 0; j < d; j++) {
                double s = 0.0;
                for (int k = 0; k < d; k++) s += (-AtQi[k * d + i]) * W[k * (d + 1) + j];
                Ld[t * dd + i * d + j] -= s;
            }
        }
    }
    /* backward: R_t, hr_t, combined on the fly.  R_t = J_tt - Jo R_{t+1}^-1 Jo',  hr_t = h_t - Jo R_{t+1}^-1 hr_{t+1} */
    {
        double *Rd = malloc(sizeof(double) * dd), *hr = malloc(sizeof(double) * d), *Rn = malloc(sizeof(double) * dd), *hn = malloc(sizeof(double) * d);
        double *Pm = malloc(sizeof(double) * dd), *rhs = malloc(sizeof(double) * (dd + d)), *Jd = malloc(sizeof(double) * dd);
        if (!Rd || !hr || !Rn || !hn || !Pm || !rhs || !Jd) { rc = -2; free(Rd); free(hr); free(Rn); free(hn); free(Pm); free(rhs); free(Jd); goto done; }
        for (int64_t t = T - 1; t >= 0; t--) {
            JDIAG(t, Jd);
            if (t == T - 1) { memcpy(Rd, Jd, sizeof(double) * dd); memcpy(hr, h + t * d, sizeof(double) * d); }
            else {
                /* W = R_{t+1}^-1 [Jo' | hr_{t+1}],  Jo' = -(A'Qi)' */
                memcpy(tmpM, Rd, sizeof(double) * dd);
                for (int i = 0; i < d; i++) { for (int j = 0; j < d; j++) W[i * (d + 1) + j] = -AtQi[j * d + i]; W[i * (d + 1) + d] = hr[i]; }
                if (!lu_solve(d, d + 1, tmpM, W)) { rc = -1; break; }
                for (int i = 0; i < d; i++) {
                    double sh = h[t * d + i];
                    for (int k = 0; k < d; k++) sh -= (-AtQi[i * d + k]) * W[k * (d + 1) + d];
                    hn[i] = sh;
                    for (int j = 0; j < d; j++) {
                        double s = 0.0;
                        for (int k = 0; k < d; k++) s += (-AtQi[i * d + k]) * W[k * (d + 1) + j];
                        Rn[i * d + j] = Jd[i * d + j] - s;
                    }
                }
                memcpy(Rd, Rn, sizeof(double) * dd); memcpy(hr, hn, sizeof(double) * d);
            }
            /* precision = L + R - J;  [cov | mean] = precision^-1 [I | hl + hr - h] */
            for (int i = 0; i < dd; i++) Pm[i] = Ld[t * dd + i] + Rd[i] - Jd[i];
            for (int i = 0; i < d; i++) {
                for (int j = 0; j < d; j++) rhs[i * (d + 1) + j] = (i == j) ? 1.0 : 0.0;
                rhs[i * (d + 1) + d] = hl[t * d + i] + hr[i] - h[t * d + i];
            }
            if (!lu_solve(d, d + 1, Pm, rhs)) { rc = -1; break; }
            for (int i = 0; i < d; i++) {
                mean[t * d + i] = rhs[i * (d + 1) + d];
                for (int j = 0; j < d; j++) cov[t * dd + i * d + j] = rhs[i * (d + 1) + j];
            }
        }
        free(Rd); free(hr); free(Rn); free(hn); free(Pm); free(rhs); free(Jd);
    }
#undef JDIAG
done:
    free(Qi); free(Ri); free(AtQi); free(AtQiA); free(tmpM); free(G); free(W); free(Ld); free(hl); free(h);
    return rc;
}
