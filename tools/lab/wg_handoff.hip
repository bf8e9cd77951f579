// tools/lab/wg_handoff.hip — what ONE stage of a one-workgroup run (cx_batch.hip: k_batch_run) costs at the least on gfx950: a value handed
// from one thread of a workgroup to another through memory, round after round.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/wg_handoff.hip -o /tmp/wg_handoff && /tmp/wg_handoff
// R rounds of [lane t stores f(x) to slot t of the round's buffer, workgroup-scope release, barrier, acquire, lane t loads slot (t + 1) mod n]:
// a dependent chain of R store -> barrier -> load steps, the arithmetic one multiply-add.  Variants: the load plain (through the compute
// unit's vector cache) or with the scope bit sc1 (past it, from the XCD's L2); the value through LDS instead (the floor); and with an index
// looked up in memory before the load (one more dependent trip, what a record that names its source costs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>      // 0 plain loads, 1 sc1 loads, 2 LDS, 3 plain + index from memory, 4 plain, and the workgroup waits for the store's acknowledgement,
                         // 5 plain, and four dependent f64 divisions on the value (a variational rule's arithmetic)
__global__ __launch_bounds__(1024) void k_chain(double *buf, const int *idx, int n, int R, double *out) {
    __shared__ double lds[2][1024];
    const int t = threadIdx.x;
    double x = 1.0 + t;
    for (int r = 0; r < R; r++) {
        double *cur = buf + (size_t)(r & 1) * 1024;
        if (t < n) {
            if (MODE == 2) lds[r & 1][t] = x * 1.0000001 + 0.5;
            else if (MODE == 5) { const double a = 1.0 / x, b = 1.0 / (a + 0.25), c = b / (x + 1.0), d = 1.0 / (c + 1.0); cur[t] = d + x * 1e-9; }
            else cur[t] = x * 1.0000001 + 0.5;
        }
        if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (t < n) {
            int s = (t + 1) % n;
            if (MODE == 3) s = idx[(size_t)r * 1024 + t];
            if (MODE == 2) x = lds[r & 1][s];
            else if (MODE == 1) x = __hip_atomic_load(cur + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else x = cur[s];
        }
    }
    if (t < n) out[t] = x;
}

int main() {
    const int R = 20000;
    double *buf, *out; int *idx;
    CK(hipMalloc(&buf, 2 * 1024 * 8)); CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&idx, (size_t)R * 1024 * 4));
    int *h = (int *)malloc((size_t)R * 1024 * 4);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[] = {"plain load", "sc1 load", "through LDS", "plain load, index from memory first", "plain load, store acknowledged before the barrier", "plain load, four dependent f64 divisions"};
    for (int n : {8, 64, 1024}) {
        for (size_t i = 0; i < (size_t)R * 1024; i++) h[i] = (int)((i % 1024 + 1) % n);
        CK(hipMemcpy(idx, h, (size_t)R * 1024 * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 6; mode++) {
            float best = 1e30f;
            double first = 0;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(buf, 0, 2 * 1024 * 8));
                CK(hipEventRecord(a));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(1024), 0, 0, buf, idx, n, R, out); break;
                    case 1: hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(1024), 0, 0, buf, idx, n, R, out); break;
                    case 2: hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(1024), 0, 0, buf, idx, n, R, out); break;
                    case 3: hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(1024), 0, 0, buf, idx, n, R, out); break;
                    case 4: hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(1024), 0, 0, buf, idx, n, R, out); break;
                    default: hipLaunchKernelGGL(k_chain<5>, dim3(1), dim3(1024), 0, 0, buf, idx, n, R, out); break;
                }
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
                CK(hipMemcpy(&first, out, 8, hipMemcpyDeviceToHost));
            }
            printf("n = %4d  %-52s %7.3f us per round   (x0 = %.6f)\n", n, names[mode], 1e3 * best / R, first);
        }
    }
    return 0;
}
