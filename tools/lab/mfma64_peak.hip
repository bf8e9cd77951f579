// lab: what v_mfma_f64_16x16x4_f64 sustains on the whole chip (wall clock), by accumulators per wave and waves per SIMD.
// One workgroup of 4 x W waves per CU (100 KB of LDS each, so a second one does not fit): every SIMD holds exactly W waves.
// (With 1024 x W one-wave workgroups the dispatcher packs some SIMDs and leaves others empty: the chip then reads 0.64 busy
// and "50 TFLOP/s" whatever the loop does — that figure was an artefact of placement, not the instruction's ceiling.)
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ __launch_bounds__(1024) void k(double *out, int iters, const double *in) {
    __shared__ double pad[12500];
    if (in[0] == 77.0) pad[threadIdx.x] = 1.0;
    d4 acc[NACC];
    double a[NACC], b[NACC];
    for (int i = 0; i < NACC; i++) { acc[i] = d4{0, 0, 0, 0}; a[i] = in[(threadIdx.x & 63) + 64 * i]; b[i] = in[(threadIdx.x & 63) + 64 * i + 512]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[blockIdx.x] = s;
}
template <int NACC>
void run(double *d, double *in, int waves_per_simd, int iters = 4000) {
    const int grid = 256, waves = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256 * waves_per_simd), 0, 0, d, iters, in);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256 * waves_per_simd), 0, 0, d, iters, in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)waves * iters * NACC * 2048.0;
    printf("%d accumulators, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", NACC, waves_per_simd, ms, flop / ms / 1e9,
           ms * 1e6 / ((double)iters * NACC * waves_per_simd));
}
int main() {
    double *d, *in; (void)hipMalloc(&d, 1 << 20); (void)hipMalloc(&in, 1 << 16);
    double h[1024]; for (int i = 0; i < 1024; i++) h[i] = 0.001 * (i % 97) - 0.03;
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<1>(d, in, 1); run<2>(d, in, 1); run<4>(d, in, 1); run<8>(d, in, 1);
    run<1>(d, in, 2); run<4>(d, in, 2); run<1>(d, in, 4); run<4>(d, in, 4);
    run<4>(d, in, 2, 40000); run<4>(d, in, 4, 40000);      // long enough (tens of ms) for the clock the chip settles at
    return 0;
}
