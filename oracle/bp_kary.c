/*
 * oracle/bp_kary.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The factor→variable phase of a flooding sweep for linear-Gaussian factors with more than two edges,
 *     x_out = a_1 x_1 + ... + a_k x_k + b + N(0, q),
 * in moment form.  The reference wires every message out of a factor to ALL the other messages into it
 * (src/dependencies.jl:17-31) and leaves the arithmetic to the user's compute_message_to_variable!
 * (src/inference_engine.jl:351-361); its only Gaussian instance is the pairwise rule of
 * test/inference_engine_tests.jl:415-432 (N(m, v + q); data y -> N(y, q)), of which this is the k-input member:
 *     to x_out:  N( b + sum_i a_i m_i,                       q + sum_i a_i^2 v_i )
 *     to x_j:    N( (m_out - b - sum_{i != j} a_i m_i) / a_j,  (v_out + q + sum_{i != j} a_i^2 v_i) / a_j^2 )
 * written with the two cases spelled out (the device uses one signed-coefficient form for both) and with direct
 * sums over the other edges in edge order.  The reference has no such factor: PARITY UNPINNED by anything of the
 * reference's; pinned by mathematics — on a tree the fixed point equals the marginals of the joint Gaussian
 * (tests/test_kary_checker.py: dense solve).
 *
 * Conventions as oracle/bp_flood.c: edges sorted by (variable, factor); variance NaN = UndefValue(); variance 0 =
 * point-mass data.  foff/fedge: CSR of the k-ary factors' edges, the OUT edge first; a[]: coefficient per CSR
 * entry (ignored for the OUT entry); q[], b[] per factor.
 */
#include <math.h>
#include <stdint.h>

int64_t cxo_kary_factor_phase(int64_t nfac, const int64_t *foff, const int64_t *fedge, const double *a, const double *q, const double *b,
                              const double *v2f_m, const double *v2f_v, double *f2v_m, double *f2v_v) {
    int64_t updates = 0;
    for (int64_t f = 0; f < nfac; f++) {
        const int64_t s = foff[f], t = foff[f + 1], eo = fedge[s];
        /* to x_out */
        {
            double m = b[f], v = q[f];
            int ok = 1;
            for (int64_t i = s + 1; i < t; i++) {
                const int64_t e = fedge[i];
                if (isnan(v2f_v[e])) { ok = 0; break; }
                m += a[i] * v2f_m[e];
                v += a[i] * a[i] * v2f_v[e];
            }
            if (ok) { f2v_m[eo] = m; f2v_v[eo] = v; updates++; }
        }
        /* to every x_j */
        for (int64_t j = s + 1; j < t; j++) {
            if (isnan(v2f_v[eo])) continue;
            double m = v2f_m[eo] - b[f], v = v2f_v[eo] + q[f];
            int ok = 1;
            for (int64_t i = s + 1; i < t; i++) {
                if (i == j) continue;
                const int64_t e = fedge[i];
                if (isnan(v2f_v[e])) { ok = 0; break; }
                m -= a[i] * v2f_m[e];
                v += a[i] * a[i] * v2f_v[e];
            }
            if (!ok) continue;
            f2v_m[fedge[j]] = m / a[j];
            f2v_v[fedge[j]] = v / (a[j] * a[j]);
            updates++;
        }
    }
    return updates;
}
