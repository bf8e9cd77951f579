// lab: issue cost (cycles per instruction and SIMD) of the vector instructions the diagonal-tile chain of k_rule64w is made of,
// measured with s_memtime around an unrolled run of INDEPENDENT instructions (8 accumulators), one and two waves per SIMD.
// usage: dpp_rate            (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(64) void k(double *out, const double *in, int iters, unsigned long long *cyc) {
    double x[8], y[8];
    const int lane = threadIdx.x;
    for (int i = 0; i < 8; i++) { x[i] = in[lane + 64 * i]; y[i] = in[lane + 64 * i + 512]; }
    const double m = in[7];
    int q[8];
    for (int i = 0; i < 8; i++) q[i] = lane * 4 + i;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(m));
            if (KIND == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y[i]), "v"(m));
            if (KIND == 2) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(m));
            if (KIND == 3) asm volatile("v_rsq_f64_e32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
            if (KIND == 4) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(y[i]));
            if (KIND == 5) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(q[i]) : "v"(q[(i + 1) & 7]));
            if (KIND == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(q[i]), "+v"(q[(i + 4) & 7]));
            if (KIND == 7) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(q[i]), "+v"(q[(i + 4) & 7]));
            if (KIND == 8) asm volatile("v_add_f64 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(m));
            if (KIND == 9) asm volatile("v_mov_b64 %0, %1" : "=v"(x[i]) : "v"(y[i]));
            if (KIND == 10) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(q[i]) : "v"(q[(i + 1) & 7]), "v"(q[(i + 2) & 7]));
            if (KIND == 11) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(q[i]) : "s20");
            if (KIND == 12) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(q[i]) : "v"(q[(i + 1) & 7]), "v"(q[(i + 2) & 7]));
            if (KIND == 13) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x[i]) : "v"(y[i]), "v"(m), "v"(y[(i + 1) & 7]));
        }
    }
    if (KIND == 12) asm volatile("s_waitcnt lgkmcnt(0)");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + q[i];
    if (s == 123.456) out[blockIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
int run(const char *name, double *out, double *in, unsigned long long *cyc) {
    const int iters = 2000;
    for (int waves = 1; waves <= 2; waves++) {
        const int grid = 256 * 4 * waves;                 // one launch wave: `waves` 64-thread workgroups per SIMD
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, in, iters, cyc);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(grid);
        CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (auto v : h) s += (double)v;
        // s_memtime counts at 100 MHz; shader clock ~2.4 GHz under load: report both raw ticks and estimated shader cycles
        const double ticks = s / grid / (iters * 8.0);
        printf("%-34s %d wave(s)/SIMD: %7.3f ticks (100 MHz) per instruction and wave = %6.1f shader cycles @2.4 GHz; per SIMD %6.1f\n", name, waves, ticks,
               ticks * 24.0, ticks * 24.0 / waves);
    }
    return 0;
}

int main() {
    double *out, *in; unsigned long long *cyc;
    CK(hipMalloc(&out, 8192 * 8)); CK(hipMalloc(&in, 2048 * 8)); CK(hipMalloc(&cyc, 8192 * 8));
    std::vector<double> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = 1.0 + 1e-3 * (i % 97);
    CK(hipMemcpy(in, h.data(), 2048 * 8, hipMemcpyHostToDevice));
    run<0>("v_fmac_f64", out, in, cyc);
    run<1>("v_fmac_f64_dpp row_newbcast", out, in, cyc);
    run<13>("v_fma_f64", out, in, cyc);
    run<2>("v_mul_f64", out, in, cyc);
    run<8>("v_add_f64", out, in, cyc);
    run<3>("v_rsq_f64", out, in, cyc);
    run<9>("v_mov_b64", out, in, cyc);
    run<4>("v_mov_b64_dpp row_newbcast", out, in, cyc);
    run<5>("v_mov_b32_dpp row_newbcast", out, in, cyc);
    run<6>("v_permlane32_swap_b32", out, in, cyc);
    run<7>("v_permlane16_swap_b32", out, in, cyc);
    run<10>("v_cndmask_b32", out, in, cyc);
    run<11>("v_readlane_b32", out, in, cyc);
    run<12>("ds_bpermute_b32", out, in, cyc);
    return 0;
}
