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

// What a stage of k_flat_run does, piece by piece (FEAT bits): 1 the value through five 16-byte sc1 buffer loads + five that point past the
// buffer, 2 a byte buffer load (the observed flag) with them, 4 the record from LDS (four ds_read_b128), 8 two stores (a sum, then a rule's
// result with four dependent divisions), 16 the stage's bounds from memory (a scalar load), 32 item t on wavefront t mod 15 (one lane each)
// instead of consecutive lanes
typedef double __attribute__((ext_vector_type(2))) d2v;
template <int FEAT>
__global__ __launch_bounds__(1024) void k_stage(double *buf, double *buf2, const unsigned char *flags, const long *bounds, int n, int R, double *out) {
    __shared__ int4 recs[1024][2];
    const int tid = threadIdx.x;
    const int t = (FEAT & 32) ? ((tid >> 6) < 15 ? (tid >> 6) + 15 * (tid & 63) : 1 << 20) : tid;
    if (tid < 1024) { recs[tid][0] = make_int4(tid, (tid + 1) % n, (tid + 2) % n, (tid + 3) % n); recs[tid][1] = make_int4((tid + 4) % n, (tid + 5) % n, 0, 0); }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)flags, 0, 0x7fffffff, 0x00020000);
    double x = 1.0 + tid;
    for (int r = 0; r < R; r++) {
        long hi = n;
        if (FEAT & 16) hi = bounds[r];
        if (t < hi) {
            int s[5] = {(t + 1) % n, (t + 2) % n, (t + 3) % n, (t + 4) % n, (t + 5) % n};
            if (FEAT & 4) { const int4 a = recs[t][0], b = recs[t][1]; s[0] = a.y; s[1] = a.z; s[2] = a.w; s[3] = b.x; s[4] = b.y; }
            double acc = 0.0;
            const int base = (r & 1) * 2048 * 16;
            if (FEAT & 1) {
                d2v v[5];
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const d2v a = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rf, base + s[j] * 16, 0, 16));
                    const d2v b = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rf, -1, 0, 16));
                    v[j] = a + b;
                }
                for (int j = 0; j < 5; j++) acc += v[j][0];
            } else acc = buf[base / 8 + s[0] * 2];
            if (FEAT & 2) acc += (double)__builtin_amdgcn_raw_buffer_load_b8(rg, t, 0, 0);
            const int nb = ((r + 1) & 1) * 2048 * 2;
            if (FEAT & 8) {
                buf2[nb + 2 * t] = acc;
                const double a = 1.0 / (acc + 2.0), b = 1.0 / (a + 0.25), c = b / (acc + 3.0), d = 1.0 / (c + 1.0);
                buf[nb + 2 * t] = d + 1e-9 * acc; buf[nb + 2 * t + 1] = a;
            } else buf[nb + 2 * t] = acc * 1e-3 + 0.5;
            x = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (tid < n) out[tid] = x;
}

template <int FEAT>
static float run_stage(double *buf, double *buf2, unsigned char *flags, long *bounds, int n, int R, double *out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(buf, 0, 2 * 2048 * 16));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_stage<FEAT>, dim3(1), dim3(1024), 0, 0, buf, buf2, flags, bounds, n, R, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return 1e3f * best / R;
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
    {
        const int R2 = 20000, n = 7;
        double *b1, *b2, *o; unsigned char *fl; long *bd;
        CK(hipMalloc(&b1, 2 * 2048 * 16)); CK(hipMalloc(&b2, 2 * 2048 * 16)); CK(hipMalloc(&o, 1024 * 8)); CK(hipMalloc(&fl, 4096)); CK(hipMalloc(&bd, (size_t)R2 * 8));
        CK(hipMemset(fl, 0, 4096));
        long *hb = (long *)malloc((size_t)R2 * 8);
        for (int i = 0; i < R2; i++) hb[i] = n;
        CK(hipMemcpy(bd, hb, (size_t)R2 * 8, hipMemcpyHostToDevice));
        printf("\na stage of %d items, features added one by one (us per round):\n", n);
        printf("  plain load, one store                                   %7.3f\n", run_stage<0>(b1, b2, fl, bd, n, R2, o));
        printf("  + five sc1 buffer loads (and five past the end)         %7.3f\n", run_stage<1>(b1, b2, fl, bd, n, R2, o));
        printf("  + the flag byte                                         %7.3f\n", run_stage<3>(b1, b2, fl, bd, n, R2, o));
        printf("  + the record from LDS                                   %7.3f\n", run_stage<7>(b1, b2, fl, bd, n, R2, o));
        printf("  + two stores, four divisions between them               %7.3f\n", run_stage<15>(b1, b2, fl, bd, n, R2, o));
        printf("  + the stage's bounds from memory                        %7.3f\n", run_stage<31>(b1, b2, fl, bd, n, R2, o));
        printf("  + one item per wavefront                                %7.3f\n", run_stage<63>(b1, b2, fl, bd, n, R2, o));
    }
    return 0;
}
