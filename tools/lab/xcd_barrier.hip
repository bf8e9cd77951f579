// tools/lab/xcd_barrier.hip — what a barrier among the workgroups of ONE XCD costs on gfx950, and whether values pass between them
// through the XCD's L2 without the agent-scope release / acquire (an L2 write-back and invalidate on this part) a device-wide barrier needs.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
// Question: a reference-order plan is thousands of dependent stages of a few thousand items; as launches a stage
// costs ≈ 9 us, a device-wide barrier 19 us and more (profiles/r04_grid_barrier.txt: eight L2s to write back and invalidate).  The
// workgroups of one XCD share ONE L2: stores are written through the vector cache to it, loads that bypass the vector cache (scope bits:
// a relaxed atomic load at agent scope) read from it, and nothing has to be written back or invalidated between them.  Measured here:
// a launch of 256 x T threads; the workgroups that find themselves on XCD 0 (hardware XCC_ID) take part, the others leave at once;
// R rounds of [every thread stores a value, barrier on one counter in that L2, every thread loads what a thread of ANOTHER workgroup
// stored and checks it].  Every wait is bounded; a wrong or missing value is counted, not assumed away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(v));      // HW_REG_XCC_ID
    return v & 7u;
}

struct Ctl { unsigned registered, members, arrive, abort_, wrong, rank_next; };

__device__ __forceinline__ bool wait_for(unsigned *p, unsigned target, unsigned *abort_) {
    for (unsigned spins = 0;; spins++) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (spins > (1u << 20)) { __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        if ((spins & 63u) == 63u && __hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
}

// WORK: 0 = the bare barrier, 1 = one 8-byte value per thread through the L2, 2 = five 16-byte values per thread (a sweep's traffic)
template <int WORK>
__global__ __launch_bounds__(1024) void k_cluster(Ctl *c, unsigned G, unsigned R, unsigned long long *buf, unsigned want_xcd) {
    __shared__ unsigned rank_s, members_s, ok_s;
    const bool mine = xcc_id() == want_xcd;
    if (threadIdx.x == 0) {
        rank_s = mine ? __hip_atomic_fetch_add(&c->rank_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (mine) __hip_atomic_fetch_add(&c->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&c->registered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok_s = 1;
        if (mine) {      // everybody has said where it is: the membership is final
            ok_s = wait_for(&c->registered, G, &c->abort_) ? 1u : 0u;
            members_s = __hip_atomic_load(&c->members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!mine || !ok_s) return;
    const unsigned P = members_s, rank = rank_s, T = blockDim.x;
    const unsigned partner = (rank + 1) % P;
    unsigned wrong = 0;
    for (unsigned r = 0; r < R; r++) {
        unsigned long long *cur = buf + (size_t)(r & 1) * (size_t)G * 1024 * 10;
        if (WORK == 1) __hip_atomic_store(cur + (size_t)rank * T + threadIdx.x, ((unsigned long long)(r + 1) << 32) | (rank * T + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (WORK == 2)
            for (int k = 0; k < 10; k++) cur[((size_t)k * P + rank) * T + threadIdx.x] = ((unsigned long long)(r + 1) << 32) | (unsigned)(k * 1000003u + rank * T + threadIdx.x);
        // the barrier: this thread's stores have left for the L2, the workgroup has arrived, one thread reports and waits
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok_s = wait_for(&c->arrive, (r + 1) * P, &c->abort_) ? 1u : 0u;
        }
        __syncthreads();
        if (!ok_s) return;
        if (WORK == 1) {
            const unsigned long long v = __hip_atomic_load(cur + (size_t)partner * T + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (((unsigned long long)(r + 1) << 32) | (partner * T + threadIdx.x))) wrong++;
        }
        if (WORK == 2)
            for (int k = 0; k < 10; k++) {
                const unsigned long long v = __hip_atomic_load(cur + ((size_t)k * P + partner) * T + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (((unsigned long long)(r + 1) << 32) | (unsigned)(k * 1000003u + partner * T + threadIdx.x))) wrong++;
            }
    }
    if (wrong) atomicAdd(&c->wrong, wrong);
}

int main(int argc, char **argv) {
    const unsigned R = argc > 1 ? atoi(argv[1]) : 2000;
    CK(hipSetDevice(0));
    Ctl *c;
    CK(hipMalloc(&c, sizeof(Ctl)));
    unsigned long long *buf;
    const unsigned G = argc > 2 ? atoi(argv[2]) : 256;      // (round 6: 1360 workgroups of 256 threads = 170 members per XCD, a 1/8 strip of C4 as XCD slabs)
    CK(hipMalloc(&buf, (size_t)2 * G * 1024 * 10 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# a barrier among the workgroups of ONE XCD (launch of %u workgroups, members = those on XCD 0), %u rounds; us per round\n", G, R);
    printf("# threads/wg  work                          members   us/round   wrong values   aborted\n");
    for (unsigned T : {256u, 1024u}) {
        if (G > 256 && T > 256) continue;      // (all members must be resident at once)
        for (int work = 0; work < 3; work++) {
            float ms = 0;
            Ctl h{};
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemset(c, 0, sizeof(Ctl)));
                CK(hipMemset(buf, 0, (size_t)2 * G * 1024 * 10 * 8));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (work == 0) k_cluster<0><<<G, T>>>(c, G, R, buf, 0);
                else if (work == 1) k_cluster<1><<<G, T>>>(c, G, R, buf, 0);
                else k_cluster<2><<<G, T>>>(c, G, R, buf, 0);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
            }
            printf("%8u    %-28s %6u   %8.2f   %10u   %s\n", T, work == 0 ? "none" : work == 1 ? "8 B per thread through L2" : "80 B per thread through L2", h.members,
                   ms * 1e3 / R, h.wrong, h.abort_ ? "ABORTED" : "no");
        }
    }
    return 0;
}
