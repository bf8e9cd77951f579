/* tests/c/abi_smoke.c — the C ABI used from plain C (what a Julia `ccall` does): build the reference's
 * SSM test graph (test/inference_engine_tests.jl:436-453) for T = 5, inject data, run the chain-scan schedule once and
 * print the marginals.  Compiled with gcc against include/cortex_hip.h only; no C++, no torch.
 *   gcc -std=c11 -Iinclude tests/c/abi_smoke.c -o abi_smoke -L cortex.jl_amd -lcortex_hip -Wl,-rpath,$PWD/cortex.jl_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cortex_hip.h"

#define T 5
#define CHECK(call)                                                                       \
    do {                                                                                  \
        int32_t rc_ = (call);                                                             \
        if (rc_ != CX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cx_last_error(h)); return 1; } \
    } while (0)

/* dim = 4 through the batched entry point: the reference's schedule on a 3-step state-space chain (SURVEY.md §3.3 hand trace:
 * lik_t->x_t, x1->tr1, tr1->x2, x2->tr2, tr2->x3 | x3->tr2, tr2->x2, x2->tr1, tr1->x1 | marginals), one cx_update_batch per
 * wavefront of independent signals, as a processor whose process! enqueues and flushes would issue them. */
#define D 4
#define T3 3
static int mv_batches(void) {
    cx_handle *h = NULL;
    cx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.device = 0; cfg.dim = D; cfg.schedule = CX_SCHED_FUSED; cfg.compute_marginals_in_sweep = 1;
    int32_t rc = cx_create(&cfg, &h);
    if (rc != CX_OK) { fprintf(stderr, "cx_create(dim 4) -> %d: %s\n", rc, cx_last_error(NULL)); return 1; }
    double A[D * D], Q[D * D], I4[D * D], R[D * D];
    for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++) {
            A[i * D + j] = i == j ? 0.9 : (j == i + 1 ? 0.1 : 0.0);
            Q[i * D + j] = i == j ? 0.1 : 0.0; I4[i * D + j] = i == j ? 1.0 : 0.0; R[i * D + j] = i == j ? 1.0 : 0.0;
        }
    CHECK(cx_set_factor_matrices(h, 0, A, Q));      /* transitions */
    CHECK(cx_set_factor_matrices(h, 1, I4, R));     /* likelihoods */
    /* x 1..3, y 4..6, likelihood 7..9, transition 10..11 */
    int64_t ev[4 * T3 - 2], ef[4 * T3 - 2], fid[2 * T3 - 1];
    int32_t role[4 * T3 - 2], fkind[2 * T3 - 1];
    double fpar[(2 * T3 - 1) * CX_NPARAM];
    int n = 0;
    memset(fpar, 0, sizeof fpar);
    for (int i = 0; i < T3; i++) { ev[n] = T3 + 1 + i; ef[n] = 2 * T3 + 1 + i; role[n++] = CX_ROLE_OUT; ev[n] = 1 + i; ef[n] = 2 * T3 + 1 + i; role[n++] = CX_ROLE_IN; }
    for (int i = 0; i < T3 - 1; i++) { ev[n] = 1 + i; ef[n] = 3 * T3 + 1 + i; role[n++] = CX_ROLE_IN; ev[n] = 2 + i; ef[n] = 3 * T3 + 1 + i; role[n++] = CX_ROLE_OUT; }
    for (int f = 0; f < 2 * T3 - 1; f++) { fid[f] = 2 * T3 + 1 + f; fkind[f] = CX_FACTOR_GAUSS_LINEAR; fpar[f * CX_NPARAM] = f < T3 ? 1.0 : 0.0; }
    CHECK(cx_graph_create(h, n, ev, ef, role, 2 * T3 - 1, fid, fkind, fpar));
    int64_t yv[T3] = {4, 5, 6}, yf[T3] = {7, 8, 9}, xv[T3] = {1, 2, 3};
    double y[T3 * D] = {0.5, -1.0, 2.0, 0.25, 1.5, -0.5, 1.0, 0.75, 2.5, 0.5, 0.0, 1.25};
    CHECK(cx_set_messages(h, T3, yv, yf, CX_TO_FACTOR, CX_FORM_POINT, y));
#define ITEM(k, v, f) {k, 0, v, f}
    const cx_item w0[] = {ITEM(CX_ITEM_MESSAGE_TO_VARIABLE, 1, 7), ITEM(CX_ITEM_MESSAGE_TO_VARIABLE, 2, 8), ITEM(CX_ITEM_MESSAGE_TO_VARIABLE, 3, 9)};
    const cx_item seq[] = {ITEM(CX_ITEM_MESSAGE_TO_FACTOR, 1, 10), ITEM(CX_ITEM_MESSAGE_TO_VARIABLE, 2, 10), ITEM(CX_ITEM_MESSAGE_TO_FACTOR, 2, 11),
                           ITEM(CX_ITEM_MESSAGE_TO_VARIABLE, 3, 11), ITEM(CX_ITEM_MESSAGE_TO_FACTOR, 3, 11), ITEM(CX_ITEM_MESSAGE_TO_VARIABLE, 2, 11),
                           ITEM(CX_ITEM_MESSAGE_TO_FACTOR, 2, 10), ITEM(CX_ITEM_MESSAGE_TO_VARIABLE, 1, 10)};
    const cx_item wm[] = {ITEM(CX_ITEM_INDIVIDUAL_MARGINAL, 1, 0), ITEM(CX_ITEM_INDIVIDUAL_MARGINAL, 2, 0), ITEM(CX_ITEM_INDIVIDUAL_MARGINAL, 3, 0)};
    CHECK(cx_update_batch(h, w0, 3));
    /* the sequential part of the schedule, one signal per wavefront, without waiting between the launches (what a scheduler does
     * between two of them does not read the device); cx_get_marginals below waits for the stream */
    for (int i = 0; i < 8; i++) CHECK(cx_update_batch_async(h, &seq[i], 1));
    CHECK(cx_update_batch(h, wm, 3));
    double marg[T3 * (D + D * D)];
    CHECK(cx_get_marginals(h, T3, xv, marg));
    for (int t = 0; t < T3; t++) {
        printf("m4 %d", t + 1);
        for (int k = 0; k < D + D * D; k++) printf(" %.15g", marg[t * (D + D * D) + k]);
        printf("\n");
    }
    CHECK(cx_destroy(h));
    return 0;
}

int main(void) {
    cx_handle *h = NULL;
    cx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.device = 0; cfg.dim = 1; cfg.schedule = CX_SCHED_CHAIN_SCAN; cfg.compute_marginals_in_sweep = 1;
    int32_t rc = cx_create(&cfg, &h);
    if (rc != CX_OK) { fprintf(stderr, "cx_create -> %d: %s\n", rc, cx_last_error(NULL)); return rc == CX_ERR_NO_DEVICE ? 77 : 1; }

    /* ids as BipartiteFactorGraphs hands them out: x 1..T, y T+1..2T, likelihood 2T+1..3T, transition 3T+1..4T-1 */
    int64_t ev[4 * T - 2], ef[4 * T - 2], fid[2 * T - 1];
    int32_t fkind[2 * T - 1];
    double fpar[(2 * T - 1) * CX_NPARAM];
    int n = 0;
    memset(fpar, 0, sizeof fpar);
    for (int i = 0; i < T; i++) { ev[n] = T + 1 + i; ef[n++] = 2 * T + 1 + i; ev[n] = 1 + i; ef[n++] = 2 * T + 1 + i; }
    for (int i = 0; i < T - 1; i++) { ev[n] = 1 + i; ef[n++] = 3 * T + 1 + i; ev[n] = 2 + i; ef[n++] = 3 * T + 1 + i; }
    for (int f = 0; f < 2 * T - 1; f++) { fid[f] = 2 * T + 1 + f; fkind[f] = CX_FACTOR_GAUSS_ADDITIVE; fpar[f * CX_NPARAM] = 1.0; }
    CHECK(cx_graph_create(h, n, ev, ef, NULL, 2 * T - 1, fid, fkind, fpar));

    int64_t yv[T], yf[T], xv[T];
    double y[T] = {2.1, 3.9, 6.2, 8.0, 9.7}, marg[2 * T];
    for (int i = 0; i < T; i++) { yv[i] = T + 1 + i; yf[i] = 2 * T + 1 + i; xv[i] = 1 + i; }
    CHECK(cx_set_messages(h, T, yv, yf, CX_TO_FACTOR, CX_FORM_POINT, y));   /* set_value!(message_to_factor(y_i, lik_i), data) */
    CHECK(cx_sweep(h, 1));                                                  /* update_marginals!(engine, x) */
    CHECK(cx_get_marginals(h, T, xv, marg));
    for (int i = 0; i < T; i++) printf("x%d %.15g %.15g\n", i + 1, marg[2 * i], marg[2 * i + 1]);
    int64_t health[4];
    CHECK(cx_message_health(h, health));                                    /* the numerical guards as counters: nothing undefined after the sweep */
    printf("health %lld %lld %lld %lld\n", (long long)health[0], (long long)health[1], (long long)health[2], (long long)health[3]);

    int64_t bad_v = 1, bad_f = 12345;
    double tmp[2];
    rc = cx_get_messages(h, 1, &bad_v, &bad_f, CX_TO_VARIABLE, CX_FORM_MOMENT, tmp);
    printf("unknown-edge status %d: %s\n", rc, cx_last_error(h));
    CHECK(cx_destroy(h));
    return mv_batches();
}
