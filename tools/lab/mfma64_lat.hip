// lab: issue interval vs dependent latency of v_mfma_f64_16x16x4_f64 on gfx950 (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ __launch_bounds__(64) void k(double *out, int iters, double a, double b) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = s; out[2 * blockIdx.x + 1] = (double)(t1 - t0); }
}
// chain where each result feeds the next one's B operand (accumulator as operand)
__global__ __launch_bounds__(64) void kop(double *out, int iters, double a) {
    d4 x = d4{1e-3, 2e-3, 3e-3, 4e-3};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        d4 y = d4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; s++) y = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x[s], y, 0, 0, 0);
        x = y;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x[0]; out[2 * blockIdx.x + 1] = (double)(t1 - t0); }
}
int main() {
    double *d; hipMalloc(&d, 1 << 16); double h[4];
    const int iters = 2000;
#define RUN(N) { hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, d, iters, 1e-3, 1e-3); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%d accumulator(s): %.1f cycles per MFMA\n", N, h[1] / (iters * N)); }
    RUN(1) RUN(2) RUN(3) RUN(4) RUN(8)
    hipLaunchKernelGGL(kop, dim3(1), dim3(64), 0, 0, d, iters, 1e-3); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("result-as-operand chain (4 accumulating MFMAs, then the tile becomes the B operand): %.1f cycles per MFMA\n", h[1] / (iters * 4));
    return 0;
}
