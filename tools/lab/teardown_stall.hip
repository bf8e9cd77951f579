// lab: what makes the first ~20 ms of work after a handle's teardown stall for ~70 ms (tools/lab/tiles_after_c5.py)?  A stream of
// 0.4 ms kernels is timed in batches of 20 after (a) nothing, (b) a stream created, used and destroyed, (c) 1 GB allocated, touched and
// freed, (d) events created and destroyed, (e) a host-mapped allocation freed.
// hipcc --offload-arch=gfx950 -O3 -o teardown_stall teardown_stall.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void spin(double *p, int iters) {
    double x = p[threadIdx.x];
    for (int i = 0; i < iters; i++) x = x * 1.0000001 + 1e-9;
    p[threadIdx.x + blockIdx.x * blockDim.x] = x;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void batches(const char *tag, int iters) {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double *p;
    hipMalloc(&p, 1024 * 256 * 8);
    hipMemset(p, 0, 1024 * 256 * 8);
    printf("%-44s", tag);
    for (int b = 0; b < 10; b++) {
        const double t0 = now();
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, p, iters);
        hipStreamSynchronize(s);
        printf(" %6.2f", (now() - t0) / 20 * 1e3);
    }
    printf("  ms per kernel\n");
    fflush(stdout);
    hipFree(p);
    hipStreamDestroy(s);
}

int main() {
    const int iters = 60000;
    batches("first", iters);
    batches("after the first's teardown", iters);
    {
        void *q; hipMalloc(&q, 1ull << 30); hipMemset(q, 1, 1ull << 30); hipDeviceSynchronize(); hipFree(q);
    }
    batches("after 1 GB allocated, set, freed", iters);
    {
        std::vector<hipEvent_t> ev(2000);
        for (auto &e : ev) hipEventCreate(&e);
        for (auto &e : ev) hipEventDestroy(e);
    }
    batches("after 2000 events created and destroyed", iters);
    {
        void *hw; hipHostMalloc(&hw, 64, hipHostMallocMapped); hipHostFree(hw);
    }
    batches("after a mapped host word freed", iters);
    batches("again", iters);
    std::this_thread::sleep_for(std::chrono::milliseconds(2000));
    batches("after two idle seconds", 20000);
    std::this_thread::sleep_for(std::chrono::milliseconds(2000));
    batches("after two idle seconds", 20000);
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    batches("after 0.2 idle seconds", 20000);
    return 0;
}
