// tools/hbm_calib.hip — known-byte-count streaming kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 for the access widths the sweep kernel uses (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads exactly 1/2 of
// the bytes of a 16 B/lane coalesced stream; other widths are uncalibrated, "calibrate on a known byte count in
// your own access pattern").  Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib
// Each kernel touches BYTES = 1 GiB (4x the 256 MiB Infinity Cache) exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class T> __device__ double as_sum(T v);
template <> __device__ double as_sum<int>(int v) { return (double)v; }
template <> __device__ double as_sum<double>(double v) { return v; }
template <> __device__ double as_sum<double2>(double2 v) { return v.x + v.y; }

template <class T>
__global__ void calib_read(const T *__restrict__ in, size_t n, double *out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += as_sum<T>(in[i]);
    if (acc == 12345.678) out[0] = acc;  // keep the loads alive
}
template <class T>
__global__ void calib_write(T *__restrict__ out, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
// scattered 16-byte stores at a fixed stride of `stride` elements inside 256-element groups: the partner scatter's shape
__global__ void calib_write16_perm(double2 *__restrict__ out, size_t n, int shift) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t j = (i + (size_t)shift) % n;
        out[j] = make_double2(1.0, 2.0);
    }
}

int main() {
    const size_t BYTES = 1ull << 30;
    void *buf; double *d_out;
    CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&d_out, 64));
    CK(hipMemset(buf, 0, BYTES));
    const int grid = 256 * 8, block = 256;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read<int>, dim3(grid), dim3(block), 0, 0, (const int *)buf, BYTES / 4, d_out);
        hipLaunchKernelGGL(calib_read<double>, dim3(grid), dim3(block), 0, 0, (const double *)buf, BYTES / 8, d_out);
        hipLaunchKernelGGL(calib_read<double2>, dim3(grid), dim3(block), 0, 0, (const double2 *)buf, BYTES / 16, d_out);
        hipLaunchKernelGGL(calib_write<double2>, dim3(grid), dim3(block), 0, 0, (double2 *)buf, BYTES / 16, make_double2(1.0, 2.0));
        hipLaunchKernelGGL(calib_write16_perm, dim3(grid), dim3(block), 0, 0, (double2 *)buf, BYTES / 16, 7075);
    }
    CK(hipDeviceSynchronize());
    printf("calib done: each kernel moved %zu bytes\n", BYTES);
    return 0;
}
