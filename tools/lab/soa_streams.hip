// lab: does the number of concurrent far-apart streams cost HBM bandwidth?  Every thread moves NC doubles of "its" slot:
//   planes  — component-major planes n slots apart (the d <= 4 message layout: c * nslots + slot)
//   blocked — the same data with the NC planes of each 256-slot block next to each other (((slot >> 8) * NC + c) * 256 + (slot & 255))
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NC, bool BLOCKED>
__global__ __launch_bounds__(256) void k(const double *__restrict__ in, double *__restrict__ out, long n) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    double v[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) v[c] = __builtin_nontemporal_load(&in[BLOCKED ? ((s >> 8) * NC + c) * 256 + (s & 255) : (long)c * n + s]);
    double t = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) t += v[c];
#pragma unroll
    for (int c = 0; c < NC; c++) out[BLOCKED ? ((s >> 8) * NC + c) * 256 + (s & 255) : (long)c * n + s] = v[c] + t * 1e-30;
}
template <int NC, bool BLOCKED>
void run(const double *in, double *out, long n, const char *name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k<NC, BLOCKED>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, in, out, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s NC = %2d: %.1f us per pass, %.2f TB/s (read + write)\n", name, NC, ms * 100, 2.0 * NC * n * 8 / (ms / 10 * 1e-3) / 1e12);
}
int main() {
    const long n = 4 << 20;          // 4 Mi slots, like C3
    double *in, *out;
    (void)hipMalloc(&in, n * 42 * 8); (void)hipMalloc(&out, n * 42 * 8);
    (void)hipMemset(in, 0, n * 42 * 8);
    run<14, false>(in, out, n, "planes"); run<14, true>(in, out, n, "blocked");
    run<42, false>(in, out, n, "planes"); run<42, true>(in, out, n, "blocked");
    return 0;
}
