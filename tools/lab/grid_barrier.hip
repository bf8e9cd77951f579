// tools/lab/grid_barrier.hip — what a device-wide barrier costs on gfx950, against a kernel boundary.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Question: the 1/8 strip of config C4 runs 16 dependent sweeps of ≈ 9.4 us between two halo exchanges, each a
// launch of 1,156 workgroups that are all resident at once.  Would ONE persistent launch with a grid barrier between the sweeps be
// cheaper than 16 launches?  Measured here, per round of G workgroups x 256 threads (all resident: G <= 2048):
//   launches      R dependent launches of the round's work (what the library does)
//   flat          one launch, R rounds, a barrier on ONE counter (agent-scope release before, acquire after)
//   by XCD        the same with one counter per XCD (hardware XCC_ID) and a second level over the eight
// each with no work (the bare cost) and with the strip's traffic (every thread reads two 16-byte values another workgroup wrote in
// the round before and writes two: 47 MB per round at G = 1156 when `bytes` says so).  Every wait is bounded: a barrier that does not
// complete raises an abort flag that every waiter sees, and the program reports it instead of hanging.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Bar {
    unsigned *flat;      // one counter
    unsigned *xcd;       // 8 counters, 64 B apart
    unsigned *top;       // second level
    unsigned *abort_;    // raised by a waiter that gave up
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(v));      // HW_REG_XCC_ID
    return v & 7u;
}

__device__ __forceinline__ bool wait_for(unsigned *p, unsigned target, unsigned *abort_) {
    for (unsigned spins = 0;; spins++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (spins > (1u << 18)) { __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        if ((spins & 63u) == 63u && __hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// returns false when the barrier was abandoned
template <int KIND>
__device__ __forceinline__ bool grid_barrier(const Bar &b, unsigned round, unsigned G, const unsigned *per_xcd) {
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        bool ok = true;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (KIND == 0) {
            __hip_atomic_fetch_add(b.flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = wait_for(b.flat, (round + 1) * G, b.abort_);
        } else {
            const unsigned x = xcc_id();
            const unsigned mine = per_xcd[x];
            const unsigned prev = __hip_atomic_fetch_add(b.xcd + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1 == (round + 1) * mine) __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the XCD's last arrival reports upwards
            ok = wait_for(b.top, (round + 1) * per_xcd[8], b.abort_);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    return ok_s != 0;
}

// the round's work: thread t of workgroup g reads what the "partner" workgroup wrote in the round before (two 16-byte values) and
// writes two; buffers alternate.  words == 0: no memory work at all.
constexpr int kVals = 5;      // 16-byte values read and written per thread and round: 160 B per thread, 47 MB at G = 1156 (the strip's sweep)
__device__ __forceinline__ void round_work(const double2 *__restrict__ in, double2 *__restrict__ out, size_t words, unsigned G) {
    if (words == 0) return;
    const size_t n = words / kVals;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t j = ((size_t)((blockIdx.x + 7) % G) * blockDim.x + threadIdx.x) % n;      // another workgroup's slots (another XCD's)
    double2 v[kVals];
#pragma unroll
    for (int k = 0; k < kVals; k++) v[k] = in[j + k * n];
#pragma unroll
    for (int k = 0; k < kVals; k++) out[i + k * n] = make_double2(v[k].x + v[(k + 1) % kVals].y, v[k].y + 1.0);
}

__global__ __launch_bounds__(256) void k_round(const double2 *in, double2 *out, size_t words, unsigned G) { round_work(in, out, words, G); }

template <int KIND>
__global__ __launch_bounds__(256) void k_persist(Bar b, unsigned R, unsigned G, const unsigned *per_xcd, double2 *x, double2 *y, size_t words) {
    for (unsigned r = 0; r < R; r++) {
        round_work((r & 1) ? y : x, (r & 1) ? x : y, words, G);
        if (!grid_barrier<KIND>(b, r, G, per_xcd)) return;
    }
}

// which XCD each workgroup of a G-wide launch lands on (the by-XCD barrier needs the counts)
__global__ void k_census(unsigned *count) {
    if (threadIdx.x == 0) atomicAdd(count + xcc_id(), 1u);
}

int main(int argc, char **argv) {
    const unsigned R = argc > 1 ? atoi(argv[1]) : 64;
    CK(hipSetDevice(0));
    unsigned *ctr;
    CK(hipMalloc(&ctr, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# device-wide barrier against a kernel boundary, %u rounds, 256 threads per workgroup; us per round\n", R);
    printf("# G      MB/round  launches   flat-barrier  by-XCD-barrier   workgroups per XCD\n");
    for (unsigned G : {256u, 1156u, 2048u}) {
        for (int with_bytes = 0; with_bytes < 2; with_bytes++) {
            const size_t words = with_bytes ? (size_t)G * 256 * kVals : 0;      // double2 values per buffer
            double2 *x = nullptr, *y = nullptr;
            CK(hipMalloc(&x, (words ? words : 1) * 16)); CK(hipMalloc(&y, (words ? words : 1) * 16));
            CK(hipMemset(x, 0, (words ? words : 1) * 16)); CK(hipMemset(y, 0, (words ? words : 1) * 16));
            // census
            CK(hipMemset(ctr, 0, 4096));
            k_census<<<G, 256>>>(ctr + 512);
            std::vector<unsigned> cnt(9, 0);
            CK(hipMemcpy(cnt.data(), ctr + 512, 32, hipMemcpyDeviceToHost));
            cnt[8] = 0;
            for (int i = 0; i < 8; i++) cnt[8] += cnt[i] ? 1 : 0;
            CK(hipMemcpy(ctr + 640, cnt.data(), 36, hipMemcpyHostToDevice));
            float ms[3] = {0, 0, 0};
            unsigned aborted[3] = {0, 0, 0};
            for (int kind = 0; kind < 3; kind++) {
                for (int rep = 0; rep < 2; rep++) {          // first repetition warms up
                    CK(hipMemset(ctr, 0, 2048));
                    Bar b{ctr, ctr + 64, ctr + 320, ctr + 400};
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    if (kind == 0) for (unsigned r = 0; r < R; r++) k_round<<<G, 256>>>((r & 1) ? y : x, (r & 1) ? x : y, words, G);
                    else if (kind == 1) k_persist<0><<<G, 256>>>(b, R, G, ctr + 640, x, y, words);
                    else k_persist<1><<<G, 256>>>(b, R, G, ctr + 640, x, y, words);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms[kind], e0, e1));
                    CK(hipMemcpy(&aborted[kind], ctr + 400, 4, hipMemcpyDeviceToHost));
                }
            }
            printf("%5u  %8.1f  %8.2f   %8.2f%s     %8.2f%s      ", G, words * 16 * 2 / 1e6, ms[0] * 1e3 / R, ms[1] * 1e3 / R, aborted[1] ? " ABORTED" : "", ms[2] * 1e3 / R,
                   aborted[2] ? " ABORTED" : "");
            for (int i = 0; i < 8; i++) printf("%u ", cnt[i]);
            printf("\n");
            CK(hipFree(x)); CK(hipFree(y));
        }
    }
    return 0;
}
