/*
 * oracle/bp_flood.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the scalar-Gaussian sum-product rules of the reference's test
 * processor, applied in the *flooding* order the device uses (all variable→factor
 * messages, then all factor→variable messages, then marginals), on a flattened
 * edge list.  Arithmetic is the reference's, in moment form and in its operation
 * order:
 *   product                  test/runtests.jl:40-46
 *   message to factor        test/inference_engine_tests.jl:405-413   reduce(product, others), left fold
 *   message to variable      test/inference_engine_tests.jl:415-432   N(m, v + q)  /  data y -> N(y, q)
 *   individual marginal      test/inference_engine_tests.jl:385-393   reduce(product, all incoming)
 * The dependency sets are those of dependencies.jl:17-31 (factor side) and :60-88
 * (variable side, "product of the others", ascending neighbour order).
 *
 * The reference itself never runs this order (its scheduler is sequential,
 * inference_engine.jl:559-632); on trees both orders reach the same fixed point,
 * on loopy graphs only this file is the per-sweep oracle of the device kernels.
 *
 * Conventions shared with the tests: edges are sorted by (variable, factor);
 * variance NaN = UndefValue(); variance 0 on a variable→factor message = point-mass
 * data (the `Real` branch of :424); fixed_v2f[e] != 0 marks user-set messages that
 * have no dependencies (clamped data / halo inputs) and are never recomputed.
 */
#include <stdint.h>
#include <math.h>

typedef struct { double mean, variance; } nmv;

static inline nmv product(nmv l, nmv r) {
    double xi = l.mean / l.variance + r.mean / r.variance;
    double w = 1 / l.variance + 1 / r.variance;
    double variance = 1 / w;
    double mean = variance * xi;
    nmv o = { mean, variance };
    return o;
}

/* one flooding sweep; returns number of directed messages (re)computed */
int64_t cxo_flood_sweep(int64_t nv, const int64_t *var_off, int64_t ne, const int64_t *partner, const double *q,
                        const uint8_t *fixed_v2f, double *f2v_m, double *f2v_v, double *v2f_m, double *v2f_v,
                        int32_t use_omp, int32_t phases) {
    /* phases: bit 0 = phase A (variable -> factor), bit 1 = phase B (factor -> variable); 3 = one full sweep.
     * A partitioned run does A, exchanges the cut messages, then B. */
    int64_t updates = 0;
    (void)ne;
    if (phases & 1) {
    /* phase A: variable -> factor */
#pragma omp parallel for schedule(static) reduction(+ : updates) if (use_omp)
    for (int64_t v = 0; v < nv; v++) {
        int64_t s = var_off[v], t = var_off[v + 1];
        for (int64_t e = s; e < t; e++) {
            if (fixed_v2f[e] || partner[e] < 0 || t - s < 2) continue; /* no listeners / no dependencies */
            int first = 1, ok = 1;
            nmv acc = { 0, 0 };
            for (int64_t o = s; o < t; o++) {
                if (o == e) continue;
                if (isnan(f2v_v[o])) { ok = 0; break; } /* a dependency is not computed -> not pending */
                nmv in = { f2v_m[o], f2v_v[o] };
                acc = first ? in : product(acc, in);
                first = 0;
            }
            if (!ok) continue;
            v2f_m[e] = acc.mean; v2f_v[e] = acc.variance;
            updates++;
        }
    }
    }
    if (phases & 2) {
    /* phase B: factor -> variable (pairwise additive-Gaussian factors) */
#pragma omp parallel for schedule(static) reduction(+ : updates) if (use_omp)
    for (int64_t e = 0; e < ne; e++) {
        int64_t p = partner[e];
        if (p < 0) continue;
        if (isnan(v2f_v[p])) continue;
        f2v_m[e] = v2f_m[p];
        f2v_v[e] = v2f_v[p] + q[e];
        updates++;
    }
    }
    return updates;
}

/* marginals = product of all incoming factor->variable messages */
void cxo_flood_marginals(int64_t nv, const int64_t *var_off, const double *f2v_m, const double *f2v_v,
                         double *marg_m, double *marg_v, int32_t use_omp) {
#pragma omp parallel for schedule(static) if (use_omp)
    for (int64_t v = 0; v < nv; v++) {
        int64_t s = var_off[v], t = var_off[v + 1];
        int ok = t > s;
        nmv acc = { 0, 0 };
        for (int64_t o = s; o < t && ok; o++) {
            if (isnan(f2v_v[o])) { ok = 0; break; }
            nmv in = { f2v_m[o], f2v_v[o] };
            acc = (o == s) ? in : product(acc, in);
        }
        if (ok) { marg_m[v] = acc.mean; marg_v[v] = acc.variance; }
        else { marg_m[v] = NAN; marg_v[v] = NAN; }
    }
}
