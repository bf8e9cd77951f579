// tools/lab/f64_pipes.hip — do v_mfma_f64_16x16x4_f64 and the f64 vector instructions of a SIMD run side by side on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/lab/f64_pipes.hip -o /tmp/f64_pipes && /tmp/f64_pipes
// Four timings, 8 waves per CU (two per SIMD) on every CU: matrix instructions only, vector FMAs only, both in every wave,
// and one wave of each kind per SIMD.  If the last two cost the SUM of the first two, the two kinds share the f64 datapath.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, typename VT>      // 0 matrix only, 1 vector only, 2 both in each wave, 3 by wave: waves 0-3 matrix, waves 4-7 vector
__global__ __launch_bounds__(512) void k(int reps, double *out) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    VT v[16];
    for (int i = 0; i < 16; i++) v[i] = (VT)(threadIdx.x + i) * (VT)1e-3;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
    const VT x = (VT)0.999, y = (VT)1e-6;
    for (int r = 0; r < reps; r++) {
        if (do_m) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                if (do_v) {
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = v[i] * x + y;
                }
            }
        } else if (do_v) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = v[i] * x + y;
        }
    }
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; i++) s += (double)v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, typename VT>
float run(int reps, double *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, VT><<<256, 512>>>(reps, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, VT><<<256, 512>>>(reps, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double *d;
    hipMalloc(&d, 256 * 512 * 8);
    const int reps = 20000;
    // per wave and rep: 16 matrix instructions (2048 flop each per wave) and 64 vector FMAs (128 flop each per wave)
    const double mflop = 256.0 * 8 * reps * 16 * 2048, vflop = 256.0 * 8 * reps * 64 * 128;
    float m = run<0, double>(reps, d), v = run<1, double>(reps, d), both = run<2, double>(reps, d), bywave = run<3, double>(reps, d);
    printf("f64 matrix only      %8.3f ms  %6.1f TFLOP/s (8 waves per CU)\n", m, mflop / m / 1e9);
    printf("f64 vector only      %8.3f ms  %6.1f TFLOP/s\n", v, vflop / v / 1e9);
    printf("both, every wave     %8.3f ms  (sum %.3f, max %.3f)\n", both, m + v, m > v ? m : v);
    printf("matrix | vector waves %7.3f ms  (half the waves each: sum %.3f, max %.3f)\n", bywave, (m + v) / 2, (m > v ? m : v) / 2);
    float v32 = run<1, float>(reps, d), both32 = run<2, float>(reps, d), bywave32 = run<3, float>(reps, d);
    printf("f32 vector only      %8.3f ms\n", v32);
    printf("f64 matrix + f32 vector, every wave %8.3f ms (sum %.3f, max %.3f)\n", both32, m + v32, m > v32 ? m : v32);
    printf("f64 matrix | f32 vector waves       %8.3f ms (sum %.3f, max %.3f)\n", bywave32, (m + v32) / 2, (m > v32 ? m : v32) / 2);
    return 0;
}
