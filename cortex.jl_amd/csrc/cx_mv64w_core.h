// cx_mv64w_core.h — device helpers of the wave-per-message d = 64 rule (cx_mv64w.hip), shared with tools/lab/t64.hip
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

namespace cx {
namespace w64 {

constexpr int kD = 64;
constexpr int kMsg = kD + kD * kD;      // doubles per message slot: eta[64] | Lambda[64][64]
constexpr int kLdT = 17;                // leading dimension of the 16 x 16 LDS transpose tile
constexpr int kFlagFixed = 1;

using d4 = __attribute__((ext_vector_type(4))) double;
// pointers into device memory, typed as such: a pointer that a kernel READS from a record (cx_mv64chain.hip) is generic to hipcc
// unless it says so, and every access through a generic pointer is a flat_load / flat_store
using gdp = __attribute__((address_space(1))) double *;
using gcdp = const __attribute__((address_space(1))) double *;

// 1 / sqrt(x): v_rsq_f64 (2^-24 relative, measured) and ONE third-order step y (1 + h/2 + 3 h^2/8), h = 1 - x y^2:
// 0.62 ulp at worst over 2^20 arguments (tools/lab/rsq_acc.hip; two Newton steps: 1.06 ulp and two more dependent operations)
__device__ __forceinline__ double rsqrt_f64(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double h = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * h, __builtin_fma(h, 0.375, 0.5), y);       // x <= 0 or NaN -> NaN/inf: the message stays undefined
}

// value of lane `src` (a compile-time constant after unrolling) broadcast to the wave through SGPRs
__device__ __forceinline__ double bcast(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// sum of a per-lane value over the four lane groups (lanes c, c + 16, c + 32, c + 48): every lane ends with the total
__device__ __forceinline__ double sum_groups(double p) {
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    return p;
}
// CV -> RV: lane (g, c) reads the value that the lanes of column g + 4 r hold
__device__ __forceinline__ double cv_to_rv(double cv, int g, int r) { return __shfl(cv, g + 4 * r, 64); }

// T' S accumulated into acc (contraction over the 16 tile rows)
__device__ __forceinline__ d4 tts(const d4 &T, const d4 &S, d4 acc) {
#pragma unroll
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[s], S[s], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ d4 neg(const d4 &T) { return d4{-T[0], -T[1], -T[2], -T[3]}; }

// index of the upper tile (a, b), a <= b, among the NT (NT + 1) / 2 upper tiles of a matrix of NT x NT tiles (4 x 4: a 10-entry array)
template <int NT>
__device__ __forceinline__ constexpr int utn(int a, int b) { return a * NT - a * (a - 1) / 2 + (b - a); }
__device__ __forceinline__ constexpr int ut(int a, int b) { return utn<4>(a, b); }

// acc += (lane I of this lane's 16-lane row).row_val * mul — v_fmac_f64 with the DPP control row_newbcast (gfx90a and later: the only
// DPP control 64-bit vector instructions take).  A VGPR written by a vector instruction needs two wait states before a DPP read
// and the compiler's hazard pass does not look inside inline assembly: dpp_ready() puts them right after the value is made.
template <int I>
__device__ __forceinline__ void fmac_rowbcast(double &acc, double row_val, double mul) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(row_val), "v"(mul), "n"(I));
}
__device__ __forceinline__ void dpp_ready(double &x) { asm("s_nop 1" : "+v"(x)); }
template <int I>
__device__ __forceinline__ double rowbcast(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + I, 0xf, 0xf, false); }

template <int K, int... Is>
__device__ __forceinline__ void diag_eliminate(double (&m)[16], double (&aw)[16], double uk, double nuk, double wk, std::integer_sequence<int, Is...>) {
    ((fmac_rowbcast<K + 1 + Is>(m[K + 1 + Is], uk, nuk), fmac_rowbcast<K + 1 + Is>(aw[K + 1 + Is], uk, wk)), ...);
}
template <int K>
__device__ __forceinline__ void diag_step(double (&m)[16], double (&aw)[16], int c) {
    // The pivot chain is what a tile's time hangs on (16 steps, each waiting for the previous one's update of ITS pivot): the row
    // U[K][.] = m[K] / sqrt(pivot) comes out of the same third-order step as 1 / sqrt(pivot), not as one more multiplication after it
    const double pk = rowbcast<K>(m[K]);                        // lane K of the row holds column K: its m[K] is the pivot
    const double y = __builtin_amdgcn_rsq(pk);
    const double h = __builtin_fma(-pk * y, y, 1.0);
    const double q = __builtin_fma(h, 0.375, 0.5);
    const double my = m[K] * y;
    const double uk = __builtin_fma(my * h, q, my);             // U[K][c] (meaningful for c >= K)
    const double dk = __builtin_fma(y * h, q, y);
    const double wk = dk * (((K == c) ? 1.0 : 0.0) - aw[K]);    // W[K][c] (zero for c > K by construction)
    double ur = uk;
    dpp_ready(ur);
    m[K] = uk; aw[K] = wk;
    diag_eliminate<K>(m, aw, ur, -uk, wk, std::make_integer_sequence<int, 15 - K>{});      // m[i] -= U[K][i] uk;  aw[i] += U[K][i] wk
}
template <int... Ks>
__device__ __forceinline__ void diag_steps(double (&m)[16], double (&aw)[16], int c, std::integer_sequence<int, Ks...>) {
    (diag_step<Ks>(m, aw, c), ...);
}

// Upper Cholesky of a symmetric 16 x 16 tile and the inverse of its factor: T = U'U, returns V = U^-1 (tile layout).
// S: 16 x 17 doubles of LDS private to this wave.  Lane c holds column c of the tile (all four lane groups hold a copy).
// Pivot k: d = 1 / sqrt(pivot) from lane k; u = U[k][.] = m[k] d; every later row i gets  m[i] -= U[k][i] u  with U[k][i]
// broadcast from lane i.  The SAME broadcast builds W = (U')^-1 by forward substitution, W[i][c] = d_i ([i == c] - sum_{k < i}
// U[k][i] W[k][c]), so each of the 120 off-diagonal entries is broadcast once; lane c ends with column c of W = row c of V.
__device__ __forceinline__ d4 diag_factor(const d4 &T, double *__restrict__ S, int g, int c, double *dbg_u = nullptr) {
#pragma unroll
    for (int r = 0; r < 4; r++) S[(g + 4 * r) * kLdT + c] = T[r];
    // hipcc (ROCm 7.2) was seen to move LDS accesses of this wave-private tile across each other without these compiler
    // barriers (tools/lab/t64.hip: |V U - I| = 78 without, 2e-16 with); the waits are what the hardware needs anyway
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // Every lane group holds column c of both U (m) and W (aw).  The multiplier U[k][i] of row i sits in lane i of each 16-lane
    // row: v_fmac_f64 with the DPP control row_newbcast:i reads it from there, so an elimination step costs one instruction
    // per (k, i) and array and nothing goes through SGPRs.  (v_readlane broadcasts: three instructions per (k, i), and hipcc
    // kept each for a second use and spilled it to VGPR lanes — 612 v_writelane + as many reloads + ~1000 s_nop per message.)
    double m[16], aw[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { m[i] = S[i * kLdT + c]; aw[i] = 0.0; }
#ifdef CX_W64_PRIO      // lab (tools/lab/w64_phases.hip): the pivot chain at a raised wave priority — see there for what it bought
    __builtin_amdgcn_s_setprio(CX_W64_PRIO);
#endif
    diag_steps(m, aw, c, std::make_integer_sequence<int, 16>{});
#ifdef CX_W64_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (dbg_u && g == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) dbg_u[i * 16 + c] = m[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) S[c * kLdT + i] = aw[i];   // V[c][i] = W[i][c]
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d4 V;
#pragma unroll
    for (int r = 0; r < 4; r++) V[r] = S[(g + 4 * r) * kLdT + c];
    return V;
}

// element (row, col) of a row-major 64 x 64 matrix for tile (tr, tc), register r, this lane
__device__ __forceinline__ int tile_off(int tr, int tc, int r, int g, int c) { return (16 * tr + g + 4 * r) * kD + 16 * tc + c; }

// ---- how the rule body addresses memory: element = base + a PER-LANE part (one register for every access of a kind) + a constant ----
//   tile (a, b), register r of a row-major 64 x 64 matrix:  lane part g * 64 + c,  constant tile_const(a, b, r)
//   element 16 j + c of a vector:                           lane part c,           constant 16 j
// PtrAcc: plain global pointers (k_rule64w, k_step64: straight-line kernels, hipcc folds the constants into its own addressing).
// BufAcc: buffer descriptors — base in four scalar registers, the constant in a scalar register or the instruction: NOTHING per
// lane but the one offset, which is what a kernel that LOOPS over rules needs to stay under 256 registers (with pointers hipcc
// keeps a 64-bit address or a 32-bit offset per tile row that is out of reach of the 13-bit immediate, hoists them out of the
// loop and spills: cx_mv64chain.hip).  Reads past the end of a buffer return zero; nothing relies on that.
template <int NT>
__device__ __forceinline__ constexpr int tile_const_n(int a, int b, int r) { return (16 * a + 4 * r) * (16 * NT) + 16 * b; }
__device__ __forceinline__ constexpr int tile_const(int a, int b, int r) { return tile_const_n<4>(a, b, r); }
struct PtrAcc {
    gdp p;
    __device__ __forceinline__ double ld(int lane_part, int cst) const { return p[lane_part + cst]; }
    __device__ __forceinline__ void st(int lane_part, int cst, double v) const { p[lane_part + cst] = v; }
};
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
struct BufAcc {
    rsrc_t r;
    __device__ __forceinline__ double ld(int lane_part, int cst) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane_part * 8, cst * 8, 0));
    }
    __device__ __forceinline__ void st(int lane_part, int cst, double v) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), r, lane_part * 8, cst * 8, 0);
    }
};
__device__ __forceinline__ PtrAcc acc_of(gcdp p) { return PtrAcc{(gdp)p}; }
__device__ __forceinline__ BufAcc buf_of(gcdp p) { return BufAcc{__builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)p, 0, 0x7fffffff, 0x00020000)}; }

// lab only (tools/lab/w64_phases.hip defines CX_W64_STAMPS and the counters): shader-clock cycles per phase and wave
#ifdef CX_W64_STAMPS
#define W64_STAMP(i)                                                                                  \
    do {                                                                                              \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();                                             \
        if (lane == 0) cx_w64_stamps[8 * (size_t)blockIdx.x + i] += (unsigned long long)(t_ - t_prev); \
        t_prev = t_;                                                                                  \
    } while (0)
#define W64_STAMP_INIT uint64_t t_prev = __builtin_amdgcn_s_memtime()
#else
#define W64_STAMP(i)
#define W64_STAMP_INIT
#endif

// ONE application of the d = 64 rule by one wave (the body of k_rule64w, cx_mv64w.hip; the walks of the chain-scan schedule,
// cx_mv64chain.hip, call it once per step):
//     M = tabP + sum of the sources' Lambdas,   Lambda_out = tabC - B M^-1 B',   eta_out = [cv +] B M^-1 (sum of the sources' etas [+ hv]),
// bt = B' row-major.  Sources are whole message records (eta[64] | Lambda[64][64]); an absent source is a record of zeros.
// AFFINE: the rule of a composed potential carries the offsets hv, cv (a single factor's rule has none).
// S: 16 x 17 doubles, Vs: 4 x 16 x 17 doubles of LDS private to this wave.  Returns false — and stores nothing — when an input is
// undefined (NaN) or M is not positive definite.
// ZS (optional): 64 doubles of LDS private to this wave.  With it z = U^-T eta_in waits there between its solve and its use (eta_out =
// Yt' z) instead of in 32 registers across the matrix solve — what the loops of cx_mv64chain.hip need to stay under 256 registers at two
// waves per SIMD without spilling (over the limit hipcc also un-clusters the loads: one memory round trip per load).
// NT (round 6): the matrices are NT x NT tiles of 16 — d = 16, 32 or 64; a message record is eta[16 NT] | Lambda[16 NT][16 NT].
template <bool AFFINE, bool Z_IN_LDS, bool CHUNKED, class A, int NT = 4>
__device__ __forceinline__ bool rule64_body(const A tabP, const A bt, const A tabC, const A hv, const A cv, const A src0, const A src1, const A src2, const bool has2,
                                            const A dst, double *S, double (*Vs)[16 * kLdT], const int lane, const int g, const int c, double *ZS = nullptr) {
    W64_STAMP_INIT;
    (void)lane;
    static_assert(!CHUNKED || NT == 4, "the chunked operand loads are written for 4 x 4 tiles");
    constexpr int KD = 16 * NT, NU = NT * (NT + 1) / 2;      // the dimension; upper tiles
    const int mo = g * KD + c;          // the per-lane part of every matrix access
    // ---- M = P + sum of the other incoming Lambdas (ascending neighbour order), upper tiles only ------------------------------
    d4 M[NU];
    if constexpr (CHUNKED) {
        // a kernel that loops over rules: four tiles' worth of operands in flight at a time, each sum pinned where it is made.
        // Left to itself hipcc, at 256 registers, falls back to its minimum-pressure schedule for this block: one memory round trip
        // per element (load, load, wait, add, load, wait, add), 120 of them per rule.
#pragma unroll
        for (int t0 = 0; t0 < 10; t0 += 4) {
            d4 X0[4], X1[4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++) {
                    const int t = ut(a, b);
                    if (t < t0 || t >= t0 + 4) continue;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = tile_const(a, b, r);
                        M[t][r] = tabP.ld(mo, o); X0[t - t0][r] = src0.ld(mo, KD + o); X1[t - t0][r] = src1.ld(mo, KD + o);
                    }
                }
            // (the loads above are written first and the additions after them: under register pressure hipcc keeps source order)
#pragma unroll
            for (int t = t0; t < t0 + 4 && t < 10; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    M[t][r] = (M[t][r] + X0[t - t0][r]) + X1[t - t0][r];
                    asm volatile("" : "+v"(M[t][r]));
                }
            asm volatile("" ::: "memory");
        }
        if (has2) {
#pragma unroll
            for (int t0 = 0; t0 < 10; t0 += 5) {
                d4 X2[5];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = a; b < 4; b++) {
                        const int t = ut(a, b);
                        if (t < t0 || t >= t0 + 5) continue;
#pragma unroll
                        for (int r = 0; r < 4; r++) X2[t - t0][r] = src2.ld(mo, KD + tile_const(a, b, r));
                    }
#pragma unroll
                for (int t = t0; t < t0 + 5; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        M[t][r] += X2[t - t0][r];
                        asm volatile("" : "+v"(M[t][r]));
                    }
                asm volatile("" ::: "memory");
            }
        }
    } else {
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
        for (int b = a; b < NT; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = tile_const_n<NT>(a, b, r);
                M[utn<NT>(a, b)][r] = (tabP.ld(mo, o) + src0.ld(mo, KD + o)) + src1.ld(mo, KD + o);
            }
    if (has2) {      // a third source (a sender of degree 4) is rare: ONE branch around the whole block of loads
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = a; b < NT; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) M[utn<NT>(a, b)][r] += src2.ld(mo, KD + tile_const_n<NT>(a, b, r));
    }
    }
    // a dependency is undefined (whole messages are NaN together): the signal is not pending
    if (__builtin_isnan(bcast(M[0][0], 0))) return false;
    W64_STAMP(0);

    // ---- blocked upper Cholesky, NB = 16: off-diagonal tiles of M become U, V_k = U_kk^-1 goes to LDS -------------------------
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const d4 Vk = diag_factor(M[utn<NT>(k, k)], S, g, c);
        W64_STAMP(1);
#pragma unroll
        for (int r = 0; r < 4; r++) Vs[k][(g + 4 * r) * kLdT + c] = Vk[r];
#pragma unroll
        for (int j = k + 1; j < NT; j++) M[utn<NT>(k, j)] = tts(Vk, M[utn<NT>(k, j)], d4{0.0, 0.0, 0.0, 0.0});        // U[k][j] = V_k' M[k][j]
#pragma unroll
        for (int i = k + 1; i < NT; i++) {
            const d4 nu = neg(M[utn<NT>(k, i)]);
#pragma unroll
            for (int j = i; j < NT; j++) M[utn<NT>(i, j)] = tts(nu, M[utn<NT>(k, j)], M[utn<NT>(i, j)]);                       // M[i][j] -= U[k][i]' U[k][j]
        }
        W64_STAMP(2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- z = U^-T eta_in on the vector pipe (as a fifth block column of the matrix solve it cost 104 of 488 matrix
    //      instructions, on tiles that are 15/16 zeros).  Two vector layouts: RV — lane (g, c) holds x[g + 4 r] in register r
    //      (indexed like tile rows); CV — lane (g, c) holds y[c] (indexed like tile columns).  T' x for a tile T: four FMAs per lane
    //      and a sum over the four lane groups gives CV; CV -> RV is four lane reads. -----------------------------------------------
    double zrv[NT][4];
#pragma unroll
    for (int j = 0; j < NT; j++) {
        double wcv = src0.ld(c, 16 * j) + src1.ld(c, 16 * j);
        if (has2) wcv += src2.ld(c, 16 * j);
        if (AFFINE) wcv += hv.ld(c, 16 * j);
#pragma unroll
        for (int k = 0; k < j; k++) {
            double p = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) p += M[utn<NT>(k, j)][r] * zrv[k][r];
            wcv -= sum_groups(p);                                                      // eta_j - sum_k U[k][j]' z_k
        }
        double p = 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++) p += Vs[j][(g + 4 * r) * kLdT + c] * cv_to_rv(wcv, g, r);
        const double zcv = sum_groups(p);                                              // z_j = V_j' w_j
#pragma unroll
        for (int r = 0; r < 4; r++) zrv[j][r] = cv_to_rv(zcv, g, r);
        if (Z_IN_LDS && g == 0) ZS[16 * j + c] = zcv;
    }
    const bool z_undefined = __builtin_isnan(bcast(zrv[NT - 1][0], 0));

    W64_STAMP(3);
    // ---- Yt = U^-T B', one block COLUMN at a time (forward substitution over its four row blocks) ------------------------------
    d4 Y[NT][NT];
#pragma unroll
    for (int b = 0; b < NT; b++) {
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) Y[j][b][r] = bt.ld(mo, tile_const_n<NT>(j, b, r));
#pragma unroll
        for (int j = 0; j < NT; j++) {
            d4 Vj;
#pragma unroll
            for (int r = 0; r < 4; r++) Vj[r] = Vs[j][(g + 4 * r) * kLdT + c];
            Y[j][b] = tts(Vj, Y[j][b], d4{0.0, 0.0, 0.0, 0.0});                                                  // Yt[j] = V_j' R[j]
#pragma unroll
            for (int jj = j + 1; jj < NT; jj++) Y[jj][b] = tts(neg(M[utn<NT>(j, jj)]), Y[j][b], Y[jj][b]);           // R[jj] -= U[j][jj]' Yt[j]
        }
    }
    // not positive definite somewhere: NaN everywhere downstream — leave the old message
    if (__builtin_isnan(bcast(Y[NT - 1][0][0], 0)) || z_undefined) return false;
    W64_STAMP(4);

    // ---- Gram tile by tile: G[a][b] = sum_j Yt[j][a]' Yt[j][b];  Lambda_out = C - G (C symmetric: the lower tiles are the
    //      transposes of the same differences, turned through LDS);  eta_out = Yt' z on the vector pipe ---------------------------
#pragma unroll
    for (int a = 0; a < NT; a++) {
#pragma unroll
        for (int b = a; b < NT; b++) {
            d4 Ct;
#pragma unroll
            for (int r = 0; r < 4; r++) Ct[r] = tabC.ld(mo, tile_const_n<NT>(a, b, r));
            d4 G = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < NT; j++) G = tts(Y[j][a], Y[j][b], G);
            d4 D;
#pragma unroll
            for (int r = 0; r < 4; r++) D[r] = Ct[r] - G[r];
#pragma unroll
            for (int r = 0; r < 4; r++) dst.st(mo, KD + tile_const_n<NT>(a, b, r), D[r]);
            if (b > a) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the previous tile's reads have returned
#pragma unroll
                for (int r = 0; r < 4; r++) S[(g + 4 * r) * kLdT + c] = D[r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; r++) dst.st(mo, KD + tile_const_n<NT>(b, a, r), S[c * kLdT + g + 4 * r]);
            }
        }
        double p = 0.0;
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) p += Y[j][a][r] * (Z_IN_LDS ? ZS[16 * j + g + 4 * r] : zrv[j][r]);
        double ecv = sum_groups(p);                                                    // (Yt' z)[16 a + c]
        if (AFFINE) ecv += cv.ld(c, 16 * a);
        if (g == 0) dst.st(c, 16 * a, ecv);
    }
    W64_STAMP(5);
    return true;
}



// the rule on plain pointers (k_rule64w, k_step64)
template <bool AFFINE, bool Z_IN_LDS = false, int NT = 4>
__device__ __forceinline__ bool rule64w_apply(gcdp tabP, gcdp bt, gcdp tabC, gcdp hv, gcdp cv, gcdp src0, gcdp src1, gcdp src2, const bool has2, gdp dst,
                                              double *S, double (*Vs)[16 * kLdT], const int lane, const int g, const int c, double *ZS = nullptr) {
    return rule64_body<AFFINE, Z_IN_LDS, false, PtrAcc, NT>(acc_of(tabP), acc_of(bt), acc_of(tabC), acc_of(hv), acc_of(cv), acc_of(src0), acc_of(src1), acc_of(src2), has2,
                                                 acc_of((gcdp)dst), S, Vs, lane, g, c, ZS);
}
// the rule on buffer descriptors (the walks that loop inside a kernel: cx_mv64chain.hip)
template <bool AFFINE, bool Z_IN_LDS = false>
__device__ __forceinline__ bool rule64b_apply(gcdp tabP, gcdp bt, gcdp tabC, gcdp hv, gcdp cv, gcdp src0, gcdp src1, gcdp src2, const bool has2, gdp dst,
                                              double *S, double (*Vs)[16 * kLdT], const int lane, const int g, const int c, double *ZS = nullptr) {
    return rule64_body<AFFINE, Z_IN_LDS, true, BufAcc>(buf_of(tabP), buf_of(bt), buf_of(tabC), buf_of(hv), buf_of(cv), buf_of(src0), buf_of(src1), buf_of(src2), has2,
                                                 buf_of((gcdp)dst), S, Vs, lane, g, c, ZS);
}

}  // namespace w64
}  // namespace cx
