// cx_mv64w_core.h — device helpers of the wave-per-message d = 64 rule (cx_mv64w.hip), shared with tools/lab/t64.hip
#pragma once
#include <hip/hip_runtime.h>

namespace cx {
namespace w64 {

constexpr int kD = 64;
constexpr int kMsg = kD + kD * kD;      // doubles per message slot: eta[64] | Lambda[64][64]
constexpr int kLdT = 17;                // leading dimension of the 16 x 16 LDS transpose tile
constexpr int kFlagFixed = 1;

using d4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y = y * __builtin_fma(-hx * y, y, 1.5);
    return y;       // x <= 0 or NaN -> NaN/inf: the message stays undefined
}

// value of lane `src` (a compile-time constant after unrolling) broadcast to the wave through SGPRs
__device__ __forceinline__ double bcast(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// T' S accumulated into acc (contraction over the 16 tile rows)
__device__ __forceinline__ d4 tts(const d4 &T, const d4 &S, d4 acc) {
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[s], S[s], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ d4 neg(const d4 &T) { return d4{-T[0], -T[1], -T[2], -T[3]}; }

// index of the upper tile (a, b), a <= b, in a 10-entry array
__device__ __forceinline__ constexpr int ut(int a, int b) { return a * 4 - a * (a - 1) / 2 + (b - a); }

// Upper Cholesky of a symmetric 16 x 16 tile and the inverse of its factor: T = U'U, returns V = U^-1 (tile layout).
// S: 16 x 17 doubles of LDS private to this wave.
__device__ __forceinline__ d4 diag_factor(const d4 &T, double *__restrict__ S, int g, int c, double *dbg_u = nullptr) {
#pragma unroll
    for (int r = 0; r < 4; r++) S[(g + 4 * r) * kLdT + c] = T[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = S[i * kLdT + c];      // column c (all four lane groups hold a copy)
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        d[i] = rsqrt_f64(bcast(m[i], i));                      // lane i holds column i: its m[i] is the pivot
        const double ui = m[i] * d[i];                         // U[i][c] (meaningful for c >= i)
        m[i] = ui;
#pragma unroll
        for (int k = i + 1; k < 16; k++) m[k] -= bcast(ui, k) * ui;   // U[i][k] lives in lane k
    }
    // hipcc (ROCm 7.2) was seen to move LDS accesses of this wave-private tile across each other without these compiler
    // barriers (tools/lab/t64.hip: |V U - I| = 78 without, 2e-16 with any one of them); they cost no instruction
    asm volatile("" ::: "memory");
    if (dbg_u && g == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) dbg_u[i * 16 + c] = m[i];
    }
    // column c of V = U^-1 by back substitution: V[c][c] = 1 / U[c][c], V[i][c] = -(1 / U[i][i]) sum_{k > i} U[i][k] V[k][c], 0 below
    double v[16];
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        double acc = 0.0;
#pragma unroll
        for (int k = i + 1; k < 16; k++) acc += bcast(m[i], k) * v[k];
        v[i] = (i == c) ? d[i] : ((i < c) ? -d[i] * acc : 0.0);
    }
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) S[i * kLdT + c] = v[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d4 V;
#pragma unroll
    for (int r = 0; r < 4; r++) V[r] = S[(g + 4 * r) * kLdT + c];
    return V;
}

// element (row, col) of a row-major 64 x 64 matrix for tile (tr, tc), register r, this lane
__device__ __forceinline__ int tile_off(int tr, int tc, int r, int g, int c) { return (16 * tr + g + 4 * r) * kD + 16 * tc + c; }


}  // namespace w64
}  // namespace cx
