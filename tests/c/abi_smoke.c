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

    int64_t bad_v = 1, bad_f = 12345;
    double tmp[2];
    rc = cx_get_messages(h, 1, &bad_v, &bad_f, CX_TO_VARIABLE, CX_FORM_MOMENT, tmp);
    printf("unknown-edge status %d: %s\n", rc, cx_last_error(h));
    CHECK(cx_destroy(h));
    return 0;
}
