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

/* Z (d x m) -= X' Y with X d x d, Y d x m (row-major): the update  L_t = J_tt - Jo' W  in axpy form (k outer) so that gcc vectorises it
 * without reassociating any sum */
static void sub_xty(int d, int m, const double *X, const double *Y, int ldy, double *Z, int ldz) {
    for (int k = 0; k < d; k++)
        for (int i = 0; i < d; i++) {
            const double x = X[k * d + i];
            for (int j = 0; j < m; j++) Z[i * ldz + j] -= x * Y[k * ldy + j];
        }
}

/* y[T][d]; A, Q, R d x d row-major.  Out: mean[T][d], cov[T][d][d].  Returns 0 on success, -1 on a singular block, -2 on allocation failure.
 * The two Schur recursions are independent of each other (one OpenMP section each, when built with -fopenmp); the T combinations
 * precision_t^-1 are independent of one another (a parallel loop).  d = 64, T = 1e5 (config C5): about a minute on two cores. */
int32_t cxo_lgssm_posterior(int32_t d, int64_t T, const double *y, const double *A, const double *Q, const double *R, double *mean, double *cov) {
    const int dd = d * d, d1 = d + 1;
    double *Qi = malloc(sizeof(double) * dd), *Ri = malloc(sizeof(double) * dd), *Jo = malloc(sizeof(double) * dd), *JoT = malloc(sizeof(double) * dd), *AtQiA = malloc(sizeof(double) * dd);
    double *tmpM = malloc(sizeof(double) * dd);
    double *Ld = malloc(sizeof(double) * (size_t)T * dd), *hl = malloc(sizeof(double) * (size_t)T * d), *h = malloc(sizeof(double) * (size_t)T * d);
    double *Rd = malloc(sizeof(double) * (size_t)T * dd), *hr = malloc(sizeof(double) * (size_t)T * d);
    int rc = 0, rc_f = 0, rc_b = 0;
    if (!Qi || !Ri || !Jo || !JoT || !AtQiA || !tmpM || !Ld || !hl || !h || !Rd || !hr) { rc = -2; goto done; }
    /* Qi = Q^-1, Ri = R^-1 */
    for (int i = 0; i < dd; i++) { Qi[i] = (i / d == i % d) ? 1.0 : 0.0; Ri[i] = Qi[i]; }
    memcpy(tmpM, Q, sizeof(double) * dd); if (!lu_solve(d, d, tmpM, Qi)) { rc = -1; goto done; }
    memcpy(tmpM, R, sizeof(double) * dd); if (!lu_solve(d, d, tmpM, Ri)) { rc = -1; goto done; }
    /* Jo = J_{t,t+1} = -A'Qi (d x d), JoT its transpose */
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) { double s = 0.0; for (int k = 0; k < d; k++) s += A[k * d + i] * Qi[k * d + j]; Jo[i * d + j] = -s; JoT[j * d + i] = -s; }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) { double s = 0.0; for (int k = 0; k < d; k++) s -= Jo[i * d + k] * A[k * d + j]; AtQiA[i * d + j] = s; }
    /* J_tt = Ri + [t < T-1] A'QiA + [t > 0] Qi;  h_t = Ri y_t */
#define JDIAG(t, out) do { for (int i_ = 0; i_ < dd; i_++) (out)[i_] = Ri[i_] + ((t) < T - 1 ? AtQiA[i_] : 0.0) + ((t) > 0 ? Qi[i_] : 0.0); } while (0)
    for (int64_t t = 0; t < T; t++)
        for (int i = 0; i < d; i++) { double s = 0.0; for (int k = 0; k < d; k++) s += Ri[i * d + k] * y[t * d + k]; h[t * d + i] = s; }
#pragma omp parallel sections num_threads(2)
    {
#pragma omp section
        {   /* forward: L_t = J_tt - Jo' L_{t-1}^-1 Jo,  hl_t = h_t - Jo' L_{t-1}^-1 hl_{t-1} */
            double *M = malloc(sizeof(double) * dd), *W = malloc(sizeof(double) * dd * 2);
            if (!M || !W) rc_f = -2;
            else {
                JDIAG(0, Ld);
                memcpy(hl, h, sizeof(double) * d);
                for (int64_t t = 1; t < T && !rc_f; t++) {
                    memcpy(M, Ld + (t - 1) * dd, sizeof(double) * dd);
                    for (int i = 0; i < d; i++) { memcpy(W + i * d1, Jo + i * d, sizeof(double) * d); W[i * d1 + d] = hl[(t - 1) * d + i]; }      /* W = L_{t-1}^-1 [Jo | hl_{t-1}] */
                    if (!lu_solve(d, d1, M, W)) { rc_f = -1; break; }
                    JDIAG(t, Ld + t * dd);
                    sub_xty(d, d, Jo, W, d1, Ld + t * dd, d);
                    memcpy(hl + t * d, h + t * d, sizeof(double) * d);
                    sub_xty(d, 1, Jo, W + d, d1, hl + t * d, 1);
                }
            }
            free(M); free(W);
        }
#pragma omp section
        {   /* backward: R_t = J_tt - Jo R_{t+1}^-1 Jo',  hr_t = h_t - Jo R_{t+1}^-1 hr_{t+1} */
            double *M = malloc(sizeof(double) * dd), *W = malloc(sizeof(double) * dd * 2);
            if (!M || !W) rc_b = -2;
            else {
                JDIAG(T - 1, Rd + (T - 1) * dd);
                memcpy(hr + (T - 1) * d, h + (T - 1) * d, sizeof(double) * d);
                for (int64_t t = T - 2; t >= 0 && !rc_b; t--) {
                    memcpy(M, Rd + (t + 1) * dd, sizeof(double) * dd);
                    for (int i = 0; i < d; i++) { memcpy(W + i * d1, JoT + i * d, sizeof(double) * d); W[i * d1 + d] = hr[(t + 1) * d + i]; }   /* W = R_{t+1}^-1 [Jo' | hr_{t+1}] */
                    if (!lu_solve(d, d1, M, W)) { rc_b = -1; break; }
                    JDIAG(t, Rd + t * dd);
                    sub_xty(d, d, JoT, W, d1, Rd + t * dd, d);
                    memcpy(hr + t * d, h + t * d, sizeof(double) * d);
                    sub_xty(d, 1, JoT, W + d, d1, hr + t * d, 1);
                }
            }
            free(M); free(W);
        }
    }
    if (rc_f || rc_b) { rc = rc_f ? rc_f : rc_b; goto done; }
    /* precision_t = L_t + R_t - J_tt;  [cov | mean] = precision_t^-1 [I | hl_t + hr_t - h_t] */
#pragma omp parallel
    {
        double *Pm = malloc(sizeof(double) * dd), *rhs = malloc(sizeof(double) * (dd + d)), *Jd = malloc(sizeof(double) * dd);
#pragma omp for schedule(static)
        for (int64_t t = 0; t < T; t++) {
            if (!Pm || !rhs || !Jd) { rc = -2; continue; }
            JDIAG(t, Jd);
            for (int i = 0; i < dd; i++) Pm[i] = Ld[t * dd + i] + Rd[t * dd + i] - Jd[i];
            for (int i = 0; i < d; i++) {
                for (int j = 0; j < d; j++) rhs[i * d1 + j] = (i == j) ? 1.0 : 0.0;
                rhs[i * d1 + d] = hl[t * d + i] + hr[t * d + i] - h[t * d + i];
            }
            if (!lu_solve(d, d1, Pm, rhs)) { rc = -1; continue; }
            for (int i = 0; i < d; i++) {
                mean[t * d + i] = rhs[i * d1 + d];
                for (int j = 0; j < d; j++) cov[t * dd + i * d + j] = rhs[i * d1 + j];
            }
        }
        free(Pm); free(rhs); free(Jd);
    }
#undef JDIAG
done:
    free(Qi); free(Ri); free(Jo); free(JoT); free(AtQiA); free(tmpM); free(Ld); free(hl); free(h); free(Rd); free(hr);
    return rc;
}
