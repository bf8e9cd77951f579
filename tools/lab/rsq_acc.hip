// lab: relative error of v_rsq_f64 and of one / two Newton steps on it (host long double as the reference)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *y0, double *y1, double *y2, double *y1c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    y0[i] = y;
    const double hx = 0.5 * v;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y1[i] = y;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y2[i] = y;
        double z = y0[i];
    const double e = __builtin_fma(-v * z, z, 1.0);
    // one third-order step: h = 1 - x y^2, y (1 + h/2 + 3 h^2 / 8)
    const double pp = __builtin_fma(e, 0.375, 0.5);
    y1c[i] = __builtin_fma(z * e, pp, z);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), r[4];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *d, *o[4]; (void)hipMalloc(&d, n * 8); (void)hipMemcpy(d, x.data(), n * 8, hipMemcpyHostToDevice);
    for (auto &p : o) (void)hipMalloc(&p, n * 8);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o[0], o[1], o[2], o[3], n);
    const char *nm[4] = {"v_rsq_f64", "+ 1 Newton step", "+ 2 Newton steps", "+ 1 third-order step"};
    for (int j = 0; j < 4; j++) {
        r[j].resize(n); (void)hipMemcpy(r[j].data(), o[j], n * 8, hipMemcpyDeviceToHost);
        long double worst = 0, sum = 0;
        for (int i = 0; i < n; i++) { const long double t = 1.0L / sqrtl((long double)x[i]); const long double e = fabsl(((long double)r[j][i] - t) / t); worst = e > worst ? e : worst; sum += e; }
        printf("%-26s max rel err %.3Le (%.2Lf ulp of f64), mean %.3Le\n", nm[j], worst, worst / 2.220446049250313e-16L, sum / n);
    }
    return 0;
}
