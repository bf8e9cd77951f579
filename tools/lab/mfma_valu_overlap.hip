// lab: does vector work of one wave overlap the f64 matrix instructions of another wave on the same SIMD?
// grid = 256 workgroups of 8 waves (100 KB of LDS each: one per CU); on every SIMD the first wave to arrive runs `nm` MFMAs
// (4 accumulators), the second `nv` vector instructions of one kind (roles handed out through an LDS counter per SIMD_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
template <int KIND>
__global__ __launch_bounds__(512) void k(double *out, int nm, int nv, const double *in) {
    __shared__ int role[4];
    __shared__ double pad[12500];
    if (threadIdx.x < 4) role[threadIdx.x] = 0;
    if (in[0] == 77.0) pad[threadIdx.x] = 1.0;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all bits
    const int simd = (hw >> 4) & 3;
    int r = 0;
    if ((threadIdx.x & 63) == 0) r = atomicAdd(&role[simd], 1);
    r = __builtin_amdgcn_readfirstlane(r);
    const bool mat = r == 0 && nm > 0;          // nm == 0: both waves of the SIMD run the vector loop
    if (r > 1) out[0] = -1.0;                            // more than two waves of the workgroup on one SIMD: the layout assumption fails
    double s = 0;
    if (mat) {
        d4 acc[4]; double a[4], b[4];
        for (int i = 0; i < 4; i++) { acc[i] = d4{0, 0, 0, 0}; a[i] = in[(threadIdx.x & 63) + 64 * i]; b[i] = in[(threadIdx.x & 63) + 64 * i + 512]; }
        for (int it = 0; it < nm / 4; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
        for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double x[8]; float f[8]; int q[8];
        for (int i = 0; i < 8; i++) { x[i] = in[(threadIdx.x & 63) + 64 * i]; f[i] = (float)x[i]; q[i] = threadIdx.x + i; }
        const double m = in[7];
        for (int it = 0; it < nv / 8; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) x[i] = __builtin_fma(x[i], m, 0.25);                          // f64 FMA
                if (KIND == 1) f[i] = __builtin_fmaf(f[i], (float)m, 0.25f);                  // f32 FMA
                if (KIND == 2) q[i] = __builtin_amdgcn_readlane(q[i], (i * 7) & 63) + q[(i + 1) & 7];   // readlane + int add
            }
        }
        for (int i = 0; i < 8; i++) s += x[i] + f[i] + q[i];
    }
    if (s == 123.456) out[blockIdx.x] = s;
}
template <int KIND>
float run(double *d, double *in, int nm, int nv) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, nm, nv, in);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, nm, nv, in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int KIND>
void trio(double *d_, double *in, const char *name, int nm, int nv) {
    const float a = run<KIND>(d_, in, nm, 8), b = run<KIND>(d_, in, 4, nv), c = run<KIND>(d_, in, nm, nv);
    const float d = run<KIND>(d_, in, 0, nv);
    printf("%-22s matrix alone %.3f ms, vector alone %.3f ms, together %.3f ms  (sum %.3f, max %.3f);  two vector waves %.3f ms (%.1f cycles per instruction and SIMD at 2.39 GHz; one wave: %.1f)\n",
           name, a, b, c, a + b, a > b ? a : b, d, d * 1e-3 * 2.39e9 / (2.0 * nv), b * 1e-3 * 2.39e9 / nv);
}
int main() {
    double *d, *in; (void)hipMalloc(&d, 1 << 20); (void)hipMalloc(&in, 1 << 16);
    double h[1024]; for (int i = 0; i < 1024; i++) h[i] = 0.001 * (i % 97) - 0.03;
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMemset(d, 0, 8);
    trio<0>(d, in, "f64 FMA", 16000, 160000);
    trio<1>(d, in, "f32 FMA", 16000, 320000);
    trio<2>(d, in, "readlane + int add", 16000, 160000);
    double flag; (void)hipMemcpy(&flag, d, 8, hipMemcpyDeviceToHost);
    printf("one matrix + one vector wave per SIMD: %s\n", flag == 0.0 ? "yes" : "NO (role counter went past 1)");
    return 0;
}
