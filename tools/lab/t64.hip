// standalone lab for cx_mv64w.hip pieces: diag_factor and tts on one wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdint>
#define CX_LAB 1
namespace cx { constexpr int kBlock = 256; }
struct cx_handle;
#include "../../cortex.jl_amd/csrc/cx_mv64w_core.h"
using namespace cx::w64;

__global__ __launch_bounds__(64) void k_test(const double *Tin, const double *Sin, double *Vout, double *Pout, double *Uout) {
    __shared__ double S[16 * kLdT];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    d4 T, X;
    for (int r = 0; r < 4; r++) { T[r] = Tin[(g + 4 * r) * 16 + c]; X[r] = Sin[(g + 4 * r) * 16 + c]; }
    d4 V = diag_factor(T, S, g, c, Uout);
    d4 P = tts(T, X, d4{0.0, 0.0, 0.0, 0.0});
    for (int r = 0; r < 4; r++) { Vout[(g + 4 * r) * 16 + c] = V[r]; Pout[(g + 4 * r) * 16 + c] = P[r]; }
}

int main() {
    std::vector<double> T(256), X(256), V(256), P(256), A(256);
    srand(1);
    for (auto &a : A) a = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = (i == j) ? 16.0 : 0.0; for (int k = 0; k < 16; k++) s += A[i * 16 + k] * A[j * 16 + k]; T[i * 16 + j] = s; }
    for (auto &a : X) a = (rand() / (double)RAND_MAX) - 0.5;
    double *dT, *dX, *dV, *dP, *dU; hipMalloc(&dU, 2048); std::vector<double> Ug(256);
    hipMalloc(&dT, 2048); hipMalloc(&dX, 2048); hipMalloc(&dV, 2048); hipMalloc(&dP, 2048);
    hipMemcpy(dT, T.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dT, dX, dV, dP, dU); hipMemcpy(Ug.data(), dU, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(V.data(), dV, 2048, hipMemcpyDeviceToHost); hipMemcpy(P.data(), dP, 2048, hipMemcpyDeviceToHost);
    // host: upper Cholesky U (T = U'U), check V U = I; P = T' X
    std::vector<double> U(256, 0.0);
    for (int i = 0; i < 16; i++) for (int j = i; j < 16; j++) { double s = T[i * 16 + j]; for (int k = 0; k < i; k++) s -= U[k * 16 + i] * U[k * 16 + j]; U[i * 16 + j] = (i == j) ? sqrt(s) : s / U[i * 16 + i]; }
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0, p = 0; for (int k = 0; k < 16; k++) { s += V[i * 16 + k] * U[k * 16 + j]; p += T[k * 16 + i] * X[k * 16 + j]; } e1 = fmax(e1, fabs(s - (i == j))); e2 = fmax(e2, fabs(p - P[i * 16 + j])); }
    double e3 = 0; int wi = -1, wj = -1;
    for (int i = 0; i < 16; i++) for (int j = i; j < 16; j++) { double e = fabs(Ug[i * 16 + j] - U[i * 16 + j]); if (e > e3) { e3 = e; wi = i; wj = j; } }
    printf("max |U_gpu - U| (upper) = %.3e at (%d,%d): gpu %g host %g\n", e3, wi, wj, wi >= 0 ? Ug[wi * 16 + wj] : 0.0, wi >= 0 ? U[wi * 16 + wj] : 0.0);
    for (int i = 0; i < 4; i++) { for (int j = 0; j < 6; j++) printf("%9.4f/%9.4f ", Ug[i * 16 + j], U[i * 16 + j]); printf("\n"); }
    printf("max |V U - I| = %.3e   max |T'X - tts| = %.3e   V[0][0] = %g (1/U00 = %g)\n", e1, e2, V[0], 1.0 / U[0]);
    return 0;
}
