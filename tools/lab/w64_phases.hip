// lab: where a wave of k_rule64w spends its cycles.  The kernel is compiled here with phase stamps (s_memtime at the phase
// boundaries, one row of counters per wave, summed on the host) and driven with synthetic messages of C5's shape: 2e5 work records, two sources each,
// chain-like slot numbering.  Timing does not depend on the values (Lambda = 2 I, P = 2 I, B small random, C = 4 I).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ unsigned long long *cx_w64_stamps;
#define CX_W64_STAMPS 1
#include "../../cortex.jl_amd/csrc/cx_mv64w.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const int nwork = 199998, nslots = 400000, kMsg = 64 + 64 * 64;
    const int waves = argc > 1 ? atoi(argv[1]) : 2;
    std::vector<double> msg((size_t)kMsg, 0.0), tab(3 * 4096, 0.0), bt(4096);
    for (int i = 0; i < 64; i++) { msg[64 + i * 65] = 2.0; msg[i] = 0.01 * i; tab[i * 65] = 2.0; tab[2 * 4096 + i * 65] = 4.0; }
    for (int i = 0; i < 4096; i++) { bt[i] = 0.001 * ((i * 37) % 101 - 50); tab[4096 + (i % 64) * 64 + i / 64] = bt[i]; }
    double *f2v, *out, *d_tab, *d_bt, *zero;
    int32_t *rec;
    CK(hipMalloc(&f2v, (size_t)nslots * kMsg * 8)); CK(hipMalloc(&out, (size_t)nslots * kMsg * 8));
    CK(hipMalloc(&d_tab, tab.size() * 8)); CK(hipMalloc(&d_bt, bt.size() * 8)); CK(hipMalloc(&zero, (size_t)kMsg * 8));
    CK(hipMemset(zero, 0, (size_t)kMsg * 8));
    CK(hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_bt, bt.data(), bt.size() * 8, hipMemcpyHostToDevice));
    for (int s = 0; s < nslots; s++) CK(hipMemcpyAsync(f2v + (size_t)s * kMsg, msg.data(), (size_t)kMsg * 8, hipMemcpyHostToDevice, 0));
    std::vector<int32_t> r((size_t)nwork * 8);
    for (int w = 0; w < nwork; w++) { int32_t *q = &r[(size_t)w * 8]; q[0] = 2 * w; q[1] = (2 * w + 1) % nslots; q[2] = (2 * w + 3) % nslots; q[3] = -1; q[4] = 0; q[5] = (2 * w + 5) % nslots; q[6] = 0; q[7] = 0; }
    CK(hipMalloc(&rec, r.size() * 4)); CK(hipMemcpy(rec, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    unsigned long long *stamps; CK(hipMalloc(&stamps, (size_t)nwork * 64));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(cx_w64_stamps), &stamps, sizeof stamps));
    std::vector<unsigned long long> hs((size_t)nwork * 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(stamps, 0, (size_t)nwork * 64));
        CK(hipEventRecord(e0));
        if (waves == 1) hipLaunchKernelGGL((cx::k_rule64w<1, 4>), dim3(nwork), dim3(64), 0, 0, nwork, rec, d_tab, d_bt, zero, f2v, f2v, out);
        else hipLaunchKernelGGL((cx::k_rule64w<2, 4>), dim3(nwork), dim3(64), 0, 0, nwork, rec, d_tab, d_bt, zero, f2v, f2v, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hs.data(), stamps, (size_t)nwork * 64, hipMemcpyDeviceToHost));
        double z[8] = {0};
        for (int w = 0; w < nwork; w++) for (int i = 0; i < 6; i++) z[i] += (double)hs[(size_t)w * 8 + i];
        double tot = 0; for (int i = 0; i < 6; i++) tot += (double)z[i];
        static const char *nm[6] = {"record + source loads + sums", "diagonal tiles (4)", "panel + trailing update", "z = U^-T eta (vector pipe)", "solve Yt = U^-T B'", "Gram + stores"};
        printf("launch %d: %.3f ms with stamps, %d wave(s) per SIMD; cycles per message (s_memtime) %.0f\n", rep, ms, waves, tot / nwork);
        for (int i = 0; i < 6; i++) printf("   %-30s %8.0f cycles  %5.1f %%\n", nm[i], (double)z[i] / nwork, 100.0 * z[i] / tot);
    }
    double chk[4]; CK(hipMemcpy(chk, out + (size_t)5 * kMsg + 64, sizeof chk, hipMemcpyDeviceToHost));
    printf("Lambda_out[0][0..3] of slot 5: %.6f %.6f %.6f %.6f\n", chk[0], chk[1], chk[2], chk[3]);
    return 0;
}
