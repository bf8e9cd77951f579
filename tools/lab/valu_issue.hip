// lab: how fast can ONE wave issue f64 vector instructions, and how many waves saturate the pipe?  v_fmac_f64 with NACC independent
// accumulators, W waves per SIMD; wall clock (hipEvent) and s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC, int KIND>
__global__ __launch_bounds__(64) void k(double *out, const double *in, int iters, unsigned long long *cyc) {
    double x[NACC], y[NACC];
    const int lane = threadIdx.x;
    for (int i = 0; i < NACC; i++) { x[i] = in[lane + 64 * i]; y[i] = in[lane + 64 * i + 1024]; }
    const double m = in[7];
    float xf[NACC], yf[NACC];
    for (int i = 0; i < NACC; i++) { xf[i] = (float)x[i]; yf[i] = (float)y[i]; }
    const float mf = (float)m;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (KIND == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(m));
            if (KIND == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y[i]), "v"(m));
            if (KIND == 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(xf[i]) : "v"(yf[i]), "v"(mf));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += x[i] + xf[i];
    if (s == 123.456) out[blockIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int KIND>
int run(double *out, double *in, unsigned long long *cyc) {
    const int total = 64000;
    const int iters = total / NACC;
    for (int waves : {1, 2, 3, 4, 8}) {
        const int grid = 256 * 4 * waves;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<NACC, KIND>), dim3(grid), dim3(64), 0, 0, out, in, iters, cyc);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NACC, KIND>), dim3(grid), dim3(64), 0, 0, out, in, iters, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid);
        CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (auto v : h) s += (double)v;
        const double n = (double)iters * NACC;
        printf("kind %d  %2d accumulators, %d wave(s)/SIMD: wall %.3f ms -> %6.2f ns per instruction and wave, %6.2f ns per instruction and SIMD (= %5.1f cycles @2.4 GHz);  s_memtime %.2f ticks per instruction and wave\n",
               KIND, NACC, waves, ms, ms * 1e6 / n, ms * 1e6 / n / waves, ms * 1e6 / n / waves * 2.4, s / grid / n);
    }
    return 0;
}

int main() {
    double *out, *in; unsigned long long *cyc;
    CK(hipMalloc(&out, 16384 * 8)); CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&cyc, 16384 * 8));
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = 1e-3 * (i % 97);
    CK(hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    run<1, 0>(out, in, cyc); run<2, 0>(out, in, cyc); run<4, 0>(out, in, cyc); run<8, 0>(out, in, cyc); run<16, 0>(out, in, cyc);
    run<8, 1>(out, in, cyc); run<16, 1>(out, in, cyc);
    run<8, 2>(out, in, cyc);
    return 0;
}
