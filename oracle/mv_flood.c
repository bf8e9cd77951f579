/*
 * oracle/mv_flood.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU checker for d-dimensional linear-Gaussian sum-product in the device's flooding order: the same role
 * bp_flood.c plays for scalar messages, so that per-sweep parity of the dim > 1 kernels (cx_mv.hip, cx_mv64.hip)
 * can run at chain lengths the numpy restatement (oracle/mv.py, Python loops) cannot reach.
 *
 * PARITY UNPINNED: the reference holds no d-dimensional rule.  What is restated here is the
 * d-dimensional analogue of its scalar test rules, in the form those tests use:
 *   product                  test/runtests.jl:79-85     precision-weighted product (MvNormalMeanPrecision relatives)
 *   message to factor        test/inference_engine_tests.jl:405-413   reduce(product, others), left fold, neighbour order
 *   message to variable      test/inference_engine_tests.jl:415-432   forward  N(A m, A S A' + Q);  data y -> N(A y, Q)
 *                                                                      backward in information form (A' (S+Q)^-1 A)
 *   individual marginal      test/inference_engine_tests.jl:385-393   reduce(product, all incoming)
 * Messages are (mean[d], covariance[d*d]) in moment form, exactly as oracle/mv.py states them; this file is pinned
 * against mv.py on small chains (tests/test_mv_flood_checker.py) and both against the exact block-tridiagonal
 * smoother (oracle/exact.py).  Dependency sets: dependencies.jl:17-31 (factor side), :60-88 (variable side).
 *
 * Conventions: edges sorted by (variable, factor); def[e] == 0 is UndefValue(); is_point[e] marks a clamped datum
 * on the variable→factor side (the `Real` branch), whose value sits in point_y[e*d ..].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out = inverse of the d x d matrix a (Gauss-Jordan, partial pivoting); returns 0 if singular */
static int mat_inv(int d, const double *a, double *out, double *work /* d*2d */) {
    const int w = 2 * d;
    for (int i = 0; i < d; i++) {
        for (int j = 0; j < d; j++) { work[i * w + j] = a[i * d + j]; work[i * w + d + j] = (i == j) ? 1.0 : 0.0; }
    }
    for (int c = 0; c < d; c++) {
        int piv = c;
        double best = fabs(work[c * w + c]);
        for (int r = c + 1; r < d; r++) if (fabs(work[r * w + c]) > best) { best = fabs(work[r * w + c]); piv = r; }
        if (!(best > 0.0) || !isfinite(best)) return 0;
        if (piv != c) for (int j = 0; j < w; j++) { double t = work[c * w + j]; work[c * w + j] = work[piv * w + j]; work[piv * w + j] = t; }
        const double ip = 1.0 / work[c * w + c];
        for (int j = 0; j < w; j++) work[c * w + j] *= ip;
        for (int r = 0; r < d; r++) {
            if (r == c) continue;
            const double f = work[r * w + c];
            if (f == 0.0) continue;
            for (int j = 0; j < w; j++) work[r * w + j] -= f * work[c * w + j];
        }
    }
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) out[i * d + j] = work[i * w + d + j];
    return 1;
}

static void mat_mul(int d, const double *a, const double *b, double *c) {           /* c = a b */
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
        double s = 0; for (int k = 0; k < d; k++) s += a[i * d + k] * b[k * d + j];
        c[i * d + j] = s;
    }
}
static void mat_mul_bt(int d, const double *a, const double *b, double *c) {        /* c = a b' */
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
        double s = 0; for (int k = 0; k < d; k++) s += a[i * d + k] * b[j * d + k];
        c[i * d + j] = s;
    }
}
static void mat_mul_at(int d, const double *a, const double *b, double *c) {        /* c = a' b */
    for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
        double s = 0; for (int k = 0; k < d; k++) s += a[k * d + i] * b[k * d + j];
        c[i * d + j] = s;
    }
}
static void mat_vec(int d, const double *a, const double *x, double *y) {           /* y = a x */
    for (int i = 0; i < d; i++) { double s = 0; for (int k = 0; k < d; k++) s += a[i * d + k] * x[k]; y[i] = s; }
}
static void mat_vec_t(int d, const double *a, const double *x, double *y) {         /* y = a' x */
    for (int i = 0; i < d; i++) { double s = 0; for (int k = 0; k < d; k++) s += a[k * d + i] * x[k]; y[i] = s; }
}

typedef struct { double *W1, *W2, *S, *t1, *t2, *work, *v1, *v2; } scratch;

static int scratch_init(scratch *s, int d) {
    const size_t n = (size_t)d * d;
    double *b = (double *)malloc(sizeof(double) * (5 * n + 2 * n + 2 * (size_t)d));
    if (!b) return 0;
    s->W1 = b; s->W2 = b + n; s->S = b + 2 * n; s->t1 = b + 3 * n; s->t2 = b + 4 * n; s->work = b + 5 * n; s->v1 = b + 7 * n; s->v2 = b + 7 * n + d;
    return 1;
}

/* (m, S) <- (m, S) x (m2, S2); returns 0 when a covariance is singular */
static int product_into(int d, double *m, double *S, const double *m2, const double *S2, scratch *w) {
    if (!mat_inv(d, S, w->W1, w->work) || !mat_inv(d, S2, w->W2, w->work)) return 0;
    for (int i = 0; i < d * d; i++) w->t1[i] = w->W1[i] + w->W2[i];
    if (!mat_inv(d, w->t1, w->S, w->work)) return 0;
    mat_vec(d, w->W1, m, w->v1);
    mat_vec(d, w->W2, m2, w->v2);
    for (int i = 0; i < d; i++) w->v1[i] += w->v2[i];
    mat_vec(d, w->S, w->v1, m);
    memcpy(S, w->S, sizeof(double) * (size_t)d * d);
    return 1;
}

/* one flooding sweep: phase A (variable -> factor, from the current factor -> variable messages), then phase B
 * (factor -> variable, from the fresh variable -> factor messages).  Returns the number of messages (re)computed,
 * or -1 on allocation failure. */
int64_t cxo_mv_flood_sweep(int32_t d, int64_t nv, const int64_t *var_off, int64_t ne, const int64_t *partner,
                           const int32_t *pset, const int32_t *role, const double *A, const double *Q,
                           const uint8_t *is_point, const double *point_y,
                           double *f2v_m, double *f2v_S, uint8_t *f2v_def,
                           double *v2f_m, double *v2f_S, uint8_t *v2f_def, int32_t use_omp) {
    const size_t dd = (size_t)d * d;
    int64_t updates = 0;
    int failed = 0;
#pragma omp parallel if (use_omp) reduction(+ : updates)
    {
        scratch w;
        double *am = NULL, *aS = NULL;
        int ok_alloc = scratch_init(&w, d);
        if (ok_alloc) { am = (double *)malloc(sizeof(double) * d); aS = (double *)malloc(sizeof(double) * dd); }
        if (!ok_alloc || !am || !aS) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(static)
            for (int64_t v = 0; v < nv; v++) {
                const int64_t s = var_off[v], t = var_off[v + 1];
                for (int64_t e = s; e < t; e++) {
                    if (is_point[e] || partner[e] < 0 || t - s < 2) continue;   /* data / no listener / no dependencies */
                    int first = 1, ok = 1;
                    for (int64_t o = s; o < t; o++) {
                        if (o == e) continue;
                        if (!f2v_def[o]) { ok = 0; break; }
                        if (first) { memcpy(am, f2v_m + o * d, sizeof(double) * d); memcpy(aS, f2v_S + o * dd, sizeof(double) * dd); first = 0; }
                        else if (!product_into(d, am, aS, f2v_m + o * d, f2v_S + o * dd, &w)) { ok = 0; break; }
                    }
                    if (!ok) continue;
                    memcpy(v2f_m + e * d, am, sizeof(double) * d);
                    memcpy(v2f_S + e * dd, aS, sizeof(double) * dd);
                    v2f_def[e] = 1;
                    updates++;
                }
            }
#pragma omp for schedule(static)
            for (int64_t e = 0; e < ne; e++) {
                const int64_t p = partner[e];
                if (p < 0) continue;
                const double *Ap = A + (size_t)pset[e] * dd, *Qp = Q + (size_t)pset[e] * dd;
                const int forward = role[e] == 0;   /* the receiver is the OUT edge of x_out = A x_in + N(0, Q) */
                double *om = f2v_m + e * d, *oS = f2v_S + e * dd;
                if (is_point[p]) {
                    const double *y = point_y + p * d;
                    if (forward) { mat_vec(d, Ap, y, om); memcpy(oS, Qp, sizeof(double) * dd); }
                    else {
                        if (!mat_inv(d, Qp, w.W1, w.work)) continue;            /* Qi */
                        mat_mul_at(d, Ap, w.W1, w.t1);                          /* A' Qi */
                        mat_mul(d, w.t1, Ap, w.t2);                             /* A' Qi A */
                        if (!mat_inv(d, w.t2, oS, w.work)) continue;
                        mat_vec(d, w.t1, y, w.v1);                              /* A' Qi y */
                        mat_vec(d, oS, w.v1, om);
                    }
                    f2v_def[e] = 1; updates++;
                    continue;
                }
                if (!v2f_def[p]) continue;
                const double *m = v2f_m + p * d, *S = v2f_S + p * dd;
                if (forward) {
                    mat_vec(d, Ap, m, om);
                    mat_mul(d, Ap, S, w.t1);
                    mat_mul_bt(d, w.t1, Ap, oS);
                    for (size_t i = 0; i < dd; i++) oS[i] += Qp[i];
                } else {
                    for (size_t i = 0; i < dd; i++) w.t1[i] = S[i] + Qp[i];
                    if (!mat_inv(d, w.t1, w.W1, w.work)) continue;              /* Wq = (S + Q)^-1 */
                    mat_mul_at(d, Ap, w.W1, w.t1);                              /* A' Wq */
                    mat_mul(d, w.t1, Ap, w.t2);                                 /* W = A' Wq A */
                    if (!mat_inv(d, w.t2, oS, w.work)) continue;                /* Sx */
                    mat_vec(d, w.t1, m, w.v1);                                  /* A' Wq m */
                    mat_vec(d, oS, w.v1, om);
                }
                f2v_def[e] = 1; updates++;
            }
        }
        free(am); free(aS);
        if (ok_alloc) free(w.W1);
    }
    (void)mat_vec_t;
    return failed ? -1 : updates;
}

/* marginal of each variable = product of all incoming factor -> variable messages (left fold); def_out[v] = 0 when a
 * message is still undefined */
int32_t cxo_mv_flood_marginals(int32_t d, int64_t nv, const int64_t *var_off, const double *f2v_m, const double *f2v_S,
                               const uint8_t *f2v_def, double *marg_m, double *marg_S, uint8_t *def_out) {
    const size_t dd = (size_t)d * d;
    scratch w;
    if (!scratch_init(&w, d)) return -1;
    for (int64_t v = 0; v < nv; v++) {
        const int64_t s = var_off[v], t = var_off[v + 1];
        int ok = t > s;
        for (int64_t o = s; o < t && ok; o++) {
            if (!f2v_def[o]) { ok = 0; break; }
            if (o == s) { memcpy(marg_m + v * d, f2v_m + o * d, sizeof(double) * d); memcpy(marg_S + v * dd, f2v_S + o * dd, sizeof(double) * dd); }
            else if (!product_into(d, marg_m + v * d, marg_S + v * dd, f2v_m + o * d, f2v_S + o * dd, &w)) ok = 0;
        }
        def_out[v] = (uint8_t)ok;
    }
    free(w.W1);
    return 0;
}
