// cx_mv64chain.hip — CX_SCHED_CHAIN_SCAN for dim 64: ONE cx_sweep on a state-space chain is what ONE update_marginals! of
// the reference computes there — the exact forward/backward pass (/root/reference/src/inference_engine.jl:575-608; the SSM
// of test/inference_engine_tests.jl:436-487) — with no seeding and at every time step.
//
// The plan (cx_chain64_plan.h, host, GPU-free) cuts every path into blocks and names two kinds of work:
//   compose   a PAIR of waves (k_compose64p) folds its children — links of a block, or the potentials of the level below — into ONE
//             pairwise potential (P, B, C, h, c) of the segment's two end variables.  Per pair of children (M = C1 + side + P2 = U'U):
//                 Y1 = U^-T B1,  Y2 = U^-T B2',  P = P1 - Y1'Y1,  C = C2 - Y2'Y2,  B = Y2'Y1,  h = h1 + Y1'z,  c = c2 + Y2'z,  z = U^-T g
//             = 64 (Cholesky) + 2 x 160 (solves) + 2 x 160 (Grams) + 256 (product) = 960 v_mfma_f64_16x16x4_f64, on register-resident
//             tiles in the accumulator layout of cx_mv64w_core.h.  The SAME potential serves the forward and the backward pass (read
//             from its other end it is (C, B', P, c, h)): one tree for both directions.
//   walk      a wave applies rules in sequence (k_walk64b: the body of k_rule64w on buffer descriptors, looping inside one launch): the
//             potentials of a group, to hand every child the message that enters it, and finally the links of a level-0 block, which
//             writes the exact factor→variable messages into their slots.
// Every kernel here runs two waves per SIMD under 256 registers (the walks of the tree's thin upper levels: one).
//
// Records name operands by handle (space << 56 | offset), resolved against the six base pointers of the moment (kernel arguments).
// The reference has no d-dimensional rule (parity unpinned for d > 1); pinned by the exact block-tridiagonal
// solve at every time step (tests/test_gpu_mv64_chain.py) and by the numpy execution of the same plan (tests/test_chain64_plan.py).

#include <cstdlib>

#include "cx_host.h"
#include "cx_chain64_plan.h"
// lab build (CX_BUILD_STAMPS=1 python -m cortex.jl_amd.build): shader-clock stamps at the phase boundaries of both kernels, summed per
// launch and printed to stderr after every launch — where a wave's cycles go (never defined in the shipped library)
#ifdef CX_C64_STAMPS
__device__ unsigned long long *cx_w64_stamps;      // the rule body's phases: 8 counters per workgroup (cx_mv64w_core.h)
__device__ unsigned long long *cx_c64_stamps;      // the composition's phases: 16 counters per workgroup
#define CX_W64_STAMPS 1
#define C64_STAMP(i)                                                                                   \
    do {                                                                                               \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();                                              \
        if (lane == 0) cx_c64_stamps[16 * (size_t)blockIdx.x + i] += (unsigned long long)(t_ - t_prev); \
        t_prev = t_;                                                                                   \
    } while (0)
#define C64_STAMP_INIT uint64_t t_prev = __builtin_amdgcn_s_memtime()
#else
#define C64_STAMP(i)
#define C64_STAMP_INIT
#endif
#include "cx_mv64w_core.h"

namespace cx {

using namespace w64;
namespace p64 = plan64;

// Device records: the plan's handles resolved to pointers (chain64_resolve, host) — ten 64-bit words each, like the plan's.
struct DStep { gcdp src[3], P, Bt, C, h, c; gdp dst; int64_t has2; };
struct DChild { gcdp P, B, Bt, C, h, c, side[3]; int64_t nside; };      // nside: sides that are not the zero page (they come first)
struct DJob { gdp out; int32_t first, n; };
static_assert(sizeof(DStep) == sizeof(p64::Step) && sizeof(DChild) == sizeof(p64::Child) && sizeof(DJob) == sizeof(p64::Job), "records mirror the plan's");

// A record is read ONCE per step by the whole wave.  Behind the stores of the loop body hipcc no longer treats such a load as
// uniform-and-unclobbered: it becomes a vector load, every pointer derived from it lives in two VGPRs per lane and every access gets
// its own 64-bit address arithmetic.  Reading the records through the constant address space keeps them scalar loads and the
// pointers in SGPRs (base SGPR + one lane offset per access, as in k_rule64w).  The first version resolved handles against six base
// pointers on the device: selects of pointers ended up on the vector pipe, the loop body's address arithmetic was hoisted and
// spilled, and at two waves per SIMD the M loads of a step were serialised into one memory round trip each (52 k cycles per step).
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T *as_const(const T *p) {
    return (const __attribute__((address_space(4))) T *)(uintptr_t)p;
}

// The walks LOOP inside one kernel (k_walk64b; the first version launched every step — k_step64, 98 launches per level — because
// the loop compiled badly: DESIGN.md §4): what made the loop compile badly were per-lane values kept across
// iterations.  With buffer addressing (BufAcc: one per-lane offset for every access) and the lane index made opaque at the top of
// every step there is nothing per-lane left to hoist, and the body keeps the allocation of the straight-line kernel.  What the loop
// buys: the waves of a launch drift apart, so one wave's loads (its side information comes from HBM, once per sweep) run beside its
// SIMD neighbour's arithmetic — per-step launches start every wave of a step together, and all of them wait for memory together —
// and ≈ 100 launch boundaries per sweep go.
template <int WAVES_PER_SIMD, bool AFFINE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES_PER_SIMD, WAVES_PER_SIMD)))
void k_walk64b(int njobs, const DJob *__restrict__ jobs, const DStep *__restrict__ steps) {
    __shared__ double S[16 * kLdT];
    __shared__ double Vs[4][16 * kLdT];
    const int w = blockIdx.x;
    if (w >= njobs) return;
    int lane = threadIdx.x;
    const int first = as_const(jobs)[w].first, n = as_const(jobs)[w].n;
    for (int s = 0; s < n; s++) {
        const auto *st = as_const(steps) + (first + s);
        asm volatile("" : "+v"(lane));
        lane &= 63;
        const int g = lane >> 4, c = lane & 15;
        if (!rule64b_apply<AFFINE>(st->P, st->Bt, st->C, st->h, st->c, st->src[0], st->src[1], st->src[2], st->has2 != 0, st->dst, S, Vs, lane, g, c)) break;
        // the next step reads what this one stored (its entering message): the stores have to have left the wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// Yt = U^-T R for the four tiles of block column b of R (in place): forward substitution over the row blocks
__device__ __forceinline__ void solve_col(d4 (&R)[4][4], const int b, const d4 (&M)[10], const double (*Vs)[16 * kLdT], const int g, const int c) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        d4 Vj;
#pragma unroll
        for (int r = 0; r < 4; r++) Vj[r] = Vs[j][(g + 4 * r) * kLdT + c];
        R[j][b] = tts(Vj, R[j][b], d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
        for (int jj = j + 1; jj < 4; jj++) R[jj][b] = tts(neg(M[ut(j, jj)]), R[j][b], R[jj][b]);
    }
}

// ---- the composition as a PAIR of waves ------------------------------------------------------------------------------------------
// The first version (one wave per composition, removed) kept B1, U and Y2 — 336 registers of matrices — in ONE wave: they need the
// accumulator half of the register file, hipcc shuffles them back and forth (≈ 1,000 v_accvgpr_read and 330 scratch accesses per
// step), and with one wave per SIMD every memory round trip of a step is exposed: 0.26 of the matrix pipe, 8.6 ms for the 99,995
// compositions of C5.
// Here a composition is split between the two waves of a 128-thread workgroup so that each stays under 256 registers (two waves per
// SIMD) with nothing spilled:
//   wave A (the RULE part: what k_rule64w does)   M = C1 + side + P2 = U'U,  z = U^-T g;  hands -U, V = U_kk^-1 and z over (LDS);
//                                                 Y2 = U^-T B2' -> hands it over (32 KB per workgroup in global memory: the "ring");
//                                                 C1 <- C2 - Y2'Y2,  c1 <- c2 + Y2'z                       384 MFMA + the diagonal tiles
//   wave B (the EXTENSION)                        Y1 = U^-T B1 in place,  P1 -= Y1'Y1 (in the output record),  h1 += Y1'z;
//                                                 B1 <- Y2'Y1, one block column at a time, Y2 streamed from the ring     576 MFMA
// Both waves read U tile by tile from LDS while they solve (A's 80 registers of M are free once U is published; B never holds U).
// Two workgroup barriers per step: (1) U, V, z are in LDS;  (2) Y2 is in the ring — and both waves have finished with U, V, z.  A is
// already factoring the next joint while B forms the cross product; A's next write of the ring comes after barrier (1) of the next
// step, which B only reaches with the cross product done.  Tiles travel in the accumulator layout (element r of lane l at
// tile * 256 + r * 64 + l): both waves use the same lane <-> element map, nothing is transposed.  The waves of a workgroup share
// their CU's vector cache, so the ring needs a workgroup-scope release / acquire around barrier (2) and nothing wider.
constexpr int kTileD = 256;      // doubles per 16 x 16 tile

__device__ __forceinline__ void tile_put(double *dst, int t, const d4 &T, int lane) {
#pragma unroll
    for (int r = 0; r < 4; r++) dst[t * kTileD + r * 64 + lane] = T[r];
}
__device__ __forceinline__ d4 tile_get(const double *src, int t, int lane) {
    d4 T;
#pragma unroll
    for (int r = 0; r < 4; r++) T[r] = src[t * kTileD + r * 64 + lane];
    return T;
}
__device__ __forceinline__ d4 tile_get_g(gcdp src, int t, int lane) {
    d4 T;
#pragma unroll
    for (int r = 0; r < 4; r++) T[r] = src[t * kTileD + r * 64 + lane];
    return T;
}
__device__ __forceinline__ constexpr int uo(int a, int b) { return ut(a, b) - a - 1; }      // the off-diagonal upper tile (a < b) among the six

// solve_col with -U read from LDS (nU: six tiles, accumulator layout)
__device__ __forceinline__ void solve_col_lds(d4 (&R)[4][4], const int b, const double *nU, const double (*Vs)[16 * kLdT], const int lane, const int g, const int c) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        d4 Vj;
#pragma unroll
        for (int r = 0; r < 4; r++) Vj[r] = Vs[j][(g + 4 * r) * kLdT + c];
        R[j][b] = tts(Vj, R[j][b], d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
        for (int jj = j + 1; jj < 4; jj++) R[jj][b] = tts(tile_get(nU, uo(j, jj), lane), R[j][b], R[jj][b]);
    }
}

__device__ __forceinline__ void wg_barrier_acquire() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void wg_barrier_release_acquire() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Buffer addressing for this kernel: address = base (four scalar registers) + ONE per-lane byte offset + a compile-time constant
// (scalar).  With plain pointers hipcc keeps a 64-bit per-lane address or a separate 32-bit offset for every tile row that is out
// of reach of the 13-bit immediate — dozens of registers of addresses, which at 256 registers per wave were spilled (1.2 KB of
// scratch per lane).  Reads past the end of a buffer return zero; nothing here relies on that.
__device__ __forceinline__ rsrc_t buf(const double __attribute__((address_space(1))) *p) { return buf_of(p).r; }
__device__ __forceinline__ double bld(rsrc_t r, int lane_bytes, int const_bytes) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, const_bytes, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, int lane_bytes, int const_bytes, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), r, lane_bytes, const_bytes, 0);
}
// element r of tile (a, b) of a row-major 64 x 64 matrix: lane part (g * 64 + c) * 8, the rest is a constant
__device__ __forceinline__ constexpr int mconst(int a, int b, int r) { return ((16 * a + 4 * r) * kD + 16 * b) * 8; }

// M += P2 + (the first NS sides), tile by tile in chunks: the loads of a chunk are all in flight together (one memory round trip per
// chunk), the sums are pinned where they are made — hipcc otherwise sinks the additions below the factorisation of the first diagonal
// tile (the first use of the other tiles) and keeps, that is spills, the loaded operands of every element instead of their sum.
// Chunks are sized for ≈ 100 registers of operands in flight beside the 80 of M.
template <int NS>
__device__ __forceinline__ void add_joint(d4 (&M)[10], double (&gv)[4], const rsrc_t P2, const rsrc_t h2, const rsrc_t s0, const rsrc_t s1, const rsrc_t s2,
                                          const int mo, const int vo) {
    constexpr int kChunk = NS == 0 ? 10 : NS == 1 ? 5 : NS == 2 ? 4 : 3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double v = bld(h2, vo, 128 * j);
        if (NS > 0) v += bld(s0, vo, 128 * j);
        if (NS > 1) v += bld(s1, vo, 128 * j);
        if (NS > 2) v += bld(s2, vo, 128 * j);
        gv[j] += v;
    }
#pragma unroll
    for (int t0 = 0; t0 < 10; t0 += kChunk) {
        d4 X[kChunk];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = a; b < 4; b++) {
                const int t = ut(a, b);
                if (t < t0 || t >= t0 + kChunk) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int o = mconst(a, b, r);
                    double v = bld(P2, mo, o);
                    if (NS > 0) v += bld(s0, mo, kD * 8 + o);
                    if (NS > 1) v += bld(s1, mo, kD * 8 + o);
                    if (NS > 2) v += bld(s2, mo, kD * 8 + o);
                    X[t - t0][r] = v;
                }
            }
#pragma unroll
        for (int t = t0; t < t0 + kChunk && t < 10; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                M[t][r] += X[t - t0][r];
                asm volatile("" : "+v"(M[t][r]));
            }
        asm volatile("" ::: "memory");
    }
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_compose64p(int njobs, const DJob *__restrict__ jobs, const DChild *__restrict__ children, double *__restrict__ ring_base) {
    __shared__ double Us[6 * kTileD];           // -U off the diagonal, accumulator layout
    __shared__ double Vs[4][16 * kLdT];         // V_k = U_kk^-1, row-major with pitch (what the solves read)
    __shared__ double SA[16 * kLdT], SB[16 * kLdT];
    __shared__ double zs[kD];
    const int w = blockIdx.x;
    if (w >= njobs) return;
    // uniform: each wave takes ONE side of the branch below.  (Alternating the parts with the workgroup's parity, so that the
    // factorisations do not all land on the SIMDs of first waves: measured 10.50 against 10.10 ms per C5 sweep — not kept.  Neither is
    // the variant with wave A ONE STEP AHEAD of wave B (one barrier fewer to wait at, the ring doubled): both waves then compete for
    // the SIMD's one f64 pipe all the time and the joint's loads stretch from 12 k to 49 k cycles per step: 10.67 ms.)
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int first = as_const(jobs)[w].first, n = as_const(jobs)[w].n;
    gdp out = as_const(jobs)[w].out;
    const rsrc_t ring = buf((gdp)(ring_base + (size_t)w * 16 * kTileD));
    const rsrc_t oP = buf(out), oB = buf(out + kD * kD), oBt = buf(out + 2 * kD * kD), oC = buf(out + 3 * kD * kD), ohc = buf(out + 4 * kD * kD);

    if (role == 0) {
        // ================================================================ wave A: the rule part =========================================
        d4 C1[10];
        double c1[4];
        C64_STAMP_INIT;
        {
            const auto *ch = as_const(children) + first;
            const rsrc_t C = buf(ch->C), cc = buf(ch->c);
            const int mo = (g * kD + c) * 8;
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = a; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) C1[ut(a, b)][r] = bld(C, mo, mconst(a, b, r));
                c1[a] = bld(cc, c * 8, 128 * a);
            }
        }
        for (int k = 1; k < n; k++) {
            const auto *ch = as_const(children) + (first + k);
            asm volatile("" : "+v"(lane));      // (opaque per step, range restated: nothing lane-derived is hoisted out of the loop)
            lane &= 63;
            g = lane >> 4; c = lane & 15;
            const int mo = (g * kD + c) * 8, vo = c * 8, to = lane * 8;
            const rsrc_t P2 = buf(ch->P), Bt2 = buf(ch->Bt), C2 = buf(ch->C), h2 = buf(ch->h), c2 = buf(ch->c), s0 = buf(ch->side[0]), s1 = buf(ch->side[1]), s2 = buf(ch->side[2]);
            double gv[4] = {c1[0], c1[1], c1[2], c1[3]};
            switch ((int)ch->nside) {       // uniform
            case 0: add_joint<0>(C1, gv, P2, h2, s0, s1, s2, mo, vo); break;
            case 1: add_joint<1>(C1, gv, P2, h2, s0, s1, s2, mo, vo); break;
            case 2: add_joint<2>(C1, gv, P2, h2, s0, s1, s2, mo, vo); break;
            default: add_joint<3>(C1, gv, P2, h2, s0, s1, s2, mo, vo); break;
            }
            C64_STAMP(0);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const d4 Vk = diag_factor(C1[ut(kk, kk)], SA, g, c);
#pragma unroll
                for (int r = 0; r < 4; r++) Vs[kk][(g + 4 * r) * kLdT + c] = Vk[r];
#pragma unroll
                for (int j = kk + 1; j < 4; j++) C1[ut(kk, j)] = tts(Vk, C1[ut(kk, j)], d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
                for (int i = kk + 1; i < 4; i++) {
                    const d4 nu = neg(C1[ut(kk, i)]);
#pragma unroll
                    for (int j = i; j < 4; j++) C1[ut(i, j)] = tts(nu, C1[ut(kk, j)], C1[ut(i, j)]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // z = U^-T g (vector pipe), left in LDS in column order for both waves
            {
                double zrv[4][4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    double wcv = gv[j];
#pragma unroll
                    for (int q = 0; q < j; q++) {
                        double p = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; r++) p += C1[ut(q, j)][r] * zrv[q][r];
                        wcv -= sum_groups(p);
                    }
                    double p = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; r++) p += Vs[j][(g + 4 * r) * kLdT + c] * cv_to_rv(wcv, g, r);
                    const double zcv = sum_groups(p);
#pragma unroll
                    for (int r = 0; r < 4; r++) zrv[j][r] = cv_to_rv(zcv, g, r);
                    if (g == 0) zs[16 * j + c] = zcv;
                }
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a + 1; b < 4; b++) tile_put(Us, uo(a, b), neg(C1[ut(a, b)]), lane);
            C64_STAMP(1);
            wg_barrier_release_acquire();                      // (1) -U, V, z are in LDS (this wave's M is dead from here)
            C64_STAMP(2);
            // Y2 = U^-T B2' -> the ring
            d4 Y2[4][4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Y2[j][b][r] = bld(Bt2, mo, mconst(j, b, r));
                solve_col_lds(Y2, b, Us, Vs, lane, g, c);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) bst(ring, to, ((4 * j + b) * kTileD + r * 64) * 8, Y2[j][b][r]);
            }
            // c1 <- c2 + Y2'z (z is rewritten after barrier (2))
#pragma unroll
            for (int a = 0; a < 4; a++) {
                double p = 0.0;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) p += Y2[j][a][r] * zs[16 * j + g + 4 * r];
                c1[a] = bld(c2, vo, 128 * a) + sum_groups(p);
            }
            C64_STAMP(3);
            wg_barrier_release_acquire();                      // (2) Y2 is in the ring; both waves are done with -U, V, z
            C64_STAMP(4);
            // C1 <- C2 - Y2'Y2, last block column first: a column of Y2 is dead once its Gram tiles exist
#pragma unroll
            for (int b = 3; b >= 0; b--)
#pragma unroll
                for (int a = 0; a <= b; a++) {
                    d4 G = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int j = 0; j < 4; j++) G = tts(Y2[j][a], Y2[j][b], G);
#pragma unroll
                    for (int r = 0; r < 4; r++) C1[ut(a, b)][r] = bld(C2, mo, mconst(a, b, r)) - G[r];
                }
            C64_STAMP(5);
        }
        {
            const int mo = (g * kD + c) * 8;
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = a; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) bst(oC, mo, mconst(a, b, r), C1[ut(a, b)][r]);
                if (g == 0) bst(ohc, c * 8, (kD + 16 * a) * 8, c1[a]);
            }
        }
    } else {
        // ================================================================ wave B: the extension =========================================
        d4 B1[4][4];
        double h1[4];
        C64_STAMP_INIT;
        {
            const auto *ch = as_const(children) + first;
            const rsrc_t P = buf(ch->P), B = buf(ch->B), hh = buf(ch->h);
            const int mo = (g * kD + c) * 8;
            // (all fetches first, then the copy of P into the output record: written as load, store, load, store hipcc waits for every
            // element — 40 round trips before the first step, most of what a launch of the tree's upper levels took)
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) B1[a][b][r] = bld(B, mo, mconst(a, b, r));
                h1[a] = bld(hh, c * 8, 128 * a);
            }
            d4 Pc[10];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Pc[ut(a, b)][r] = bld(P, mo, mconst(a, b, r));
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) bst(oP, mo, mconst(a, b, r), Pc[ut(a, b)][r]);
        }
        for (int k = 1; k < n; k++) {
            asm volatile("" : "+v"(lane));
            lane &= 63;
            g = lane >> 4; c = lane & 15;
            const int mo = (g * kD + c) * 8, to = lane * 8;
            // P1's tiles, for the update below: fetched while this wave waits for the factorisation anyway
            d4 Pt[10];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = a; b < 4; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Pt[ut(a, b)][r] = bld(oP, mo, mconst(a, b, r));
            wg_barrier_acquire();                              // (1) (this wave has published nothing: no release, the fetches stay in flight)
            C64_STAMP(8);
#pragma unroll
            for (int b = 0; b < 4; b++) solve_col_lds(B1, b, Us, Vs, lane, g, c);        // Y1 = U^-T B1 in place
            C64_STAMP(9);
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = a; b < 4; b++) {
                    d4 G = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int j = 0; j < 4; j++) G = tts(B1[j][a], B1[j][b], G);
#pragma unroll
                    for (int r = 0; r < 4; r++) bst(oP, mo, mconst(a, b, r), Pt[ut(a, b)][r] - G[r]);
                }
                double p = 0.0;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) p += B1[j][a][r] * zs[16 * j + g + 4 * r];
                h1[a] += sum_groups(p);
            }
            C64_STAMP(10);
            wg_barrier_release_acquire();                      // (2)
            C64_STAMP(11);
            // B1 <- Y2'Y1, one block column of Y1 at a time; block column a of Y2 (four tiles) is fetched from the ring while the
            // products with the previous one run (the memory clobber keeps hipcc from fetching further ahead and spilling B1 for it)
            {
                d4 Yc[4], Yn[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Yc[j][r] = bld(ring, to, ((4 * j + 0) * kTileD + r * 64) * 8);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    d4 T[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const int an = (a + 1) & 3;
                        if (!(b == 3 && a == 3)) {
#pragma unroll
                            for (int j = 0; j < 4; j++)
#pragma unroll
                                for (int r = 0; r < 4; r++) Yn[j][r] = bld(ring, to, ((4 * j + an) * kTileD + r * 64) * 8);
                        }
                        T[a] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int j = 0; j < 4; j++) T[a] = tts(Yc[j], B1[j][b], T[a]);
#pragma unroll
                        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(T[a][r]));
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int j = 0; j < 4; j++) Yc[j] = Yn[j];
                    }
#pragma unroll
                    for (int a = 0; a < 4; a++) B1[a][b] = T[a];
                }
            }
            C64_STAMP(12);
        }
        {
            const int mo = (g * kD + c) * 8;
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) bst(oB, mo, mconst(a, b, r), B1[a][b][r]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 4; r++) SB[(g + 4 * r) * kLdT + c] = B1[a][b][r];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 4; r++) bst(oBt, mo, mconst(b, a, r), SB[c * kLdT + g + 4 * r]);
                }
                if (g == 0) bst(ohc, c * 8, 16 * a * 8, h1[a]);
            }
        }
    }
}

// ---- host: the plan on the device -----------------------------------------------------------------------------------------------
struct Chain64 {
    DJob *d_jobs = nullptr;
    DChild *d_children = nullptr;
    DStep *d_steps = nullptr;
    std::vector<p64::Job> jobs;            // the plan's records (handles), kept to be resolved again when a base pointer moves
    std::vector<p64::Child> children;
    std::vector<p64::Step> steps;
    double *bases[p64::kSpaces] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // what the device records were resolved against
    double *d_pot = nullptr, *d_ent = nullptr, *d_ring = nullptr;      // d_ring: the Y2 hand-off of k_compose64p, 32 KB per job of the widest launch
    // positions with three or more side slots (a path variable of degree 5 .. 8): their side information is summed into one message of
    // this arena at the head of every sweep (k_side64), and the plan reads that one
    double *d_aux = nullptr;
    int32_t *d_aux_src = nullptr;          // 8 slots per arena message, -1 = none
    int64_t n_aux = 0;
    struct Launch { int kind; int64_t first; int n; int steps; };      // kind 0: compose, 1: walk over potentials, 2: walk along links; steps: longest job
    std::vector<Launch> launches;
    int64_t n_pot = 0, n_ent = 0, n_compositions = 0, n_rules = 0;
    int K0 = 0, fan = 0, levels = 0;
    int64_t bytes = 0;
    // a time block of a partitioned chain (cx_chain_block_maps): the plan also composes ONE potential per path
    int n_roots = 0;
    int64_t root_off = -1;                 // path 0's potential, in doubles from d_pot
    int32_t side_ends[6] = {-1, -1, -1, -1, -1, -1};      // side slots of path 0's first / last position
};

// ---- 1 x 1 and 2 x 2 tiles (round 6: d = 5 .. 32 in their native size under the chain-scan and tree schedules too) ---------------------
// The same plan, the same records, the same algebra — on matrices of NT x NT tiles of 16 with NT = 1, 2, where a whole composition fits
// ONE wave's registers (18 tiles at NT = 2) and nothing has to be split, staged or streamed: P1, M (in C1's place), Y1 (in B1's place),
// Y2 and the new B stay in registers over a job's children.  A composition of 1 x 1 tiles is 24 matrix instructions behind one
// diagonal tile's pivot chain: these kernels are bound by that chain's latency, not by the matrix pipe (as k_rule64w<4, 1> is).
template <int NT, bool AFFINE>
__global__ __launch_bounds__(64) void k_walk_nt(int njobs, const DJob *__restrict__ jobs, const DStep *__restrict__ steps) {
    __shared__ double S[16 * kLdT];
    __shared__ double Vs[NT][16 * kLdT];
    const int w = blockIdx.x;
    if (w >= njobs) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int first = as_const(jobs)[w].first, n = as_const(jobs)[w].n;
    for (int s = 0; s < n; s++) {
        const auto *st = as_const(steps) + (first + s);
        if (!rule64w_apply<AFFINE, false, NT>(st->P, st->Bt, st->C, st->h, st->c, st->src[0], st->src[1], st->src[2], st->has2 != 0, st->dst, S, Vs, lane, g, c)) break;
        // the next step reads what this one stored (its entering message): the stores have to have left the wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int NT>
__global__ __launch_bounds__(64) void k_compose_nt(int njobs, const DJob *__restrict__ jobs, const DChild *__restrict__ children) {
    constexpr int KD = 16 * NT, NU = NT * (NT + 1) / 2;
    __shared__ double S[16 * kLdT];
    __shared__ double Vs[NT][16 * kLdT];
    const int w = blockIdx.x;
    if (w >= njobs) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, mo = g * KD + c;
    const int first = as_const(jobs)[w].first, n = as_const(jobs)[w].n;
    gdp out = as_const(jobs)[w].out;
    // the running potential: P1 and C1 as upper tiles, B1[i][j] = tile (row block i: the joint's side, column block j: the far end's)
    d4 P1[NU], C1[NU], B1[NT][NT];
    double h1[NT], c1[NT];      // element 16 j + c in every lane group
    {
        const auto *ch = as_const(children) + first;
        const PtrAcc P = acc_of(ch->P), B = acc_of(ch->B), C = acc_of(ch->C), hh = acc_of(ch->h), cc = acc_of(ch->c);
#pragma unroll
        for (int a = 0; a < NT; a++) {
#pragma unroll
            for (int b = 0; b < NT; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int o = tile_const_n<NT>(a, b, r);
                    B1[a][b][r] = B.ld(mo, o);
                    if (b >= a) { P1[utn<NT>(a, b)][r] = P.ld(mo, o); C1[utn<NT>(a, b)][r] = C.ld(mo, o); }
                }
            h1[a] = hh.ld(c, 16 * a); c1[a] = cc.ld(c, 16 * a);
        }
    }
    for (int k = 1; k < n; k++) {
        const auto *ch = as_const(children) + (first + k);
        const PtrAcc P2 = acc_of(ch->P), Bt2 = acc_of(ch->Bt), C2 = acc_of(ch->C), h2 = acc_of(ch->h), c2 = acc_of(ch->c);
        const int nside = (int)ch->nside;
        // the joint: M = C1 + P2 + the sides' Lambdas (in C1's registers), gv = c1 + h2 + the sides' etas
        double gv[NT];
#pragma unroll
        for (int a = 0; a < NT; a++) {
#pragma unroll
            for (int b = a; b < NT; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) C1[utn<NT>(a, b)][r] += P2.ld(mo, tile_const_n<NT>(a, b, r));
            gv[a] = c1[a] + h2.ld(c, 16 * a);
        }
        for (int sd = 0; sd < nside; sd++) {      // (uniform; side messages are whole records eta[KD] | Lambda[KD][KD])
            const PtrAcc sm = acc_of(ch->side[sd]);
#pragma unroll
            for (int a = 0; a < NT; a++) {
#pragma unroll
                for (int b = a; b < NT; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) C1[utn<NT>(a, b)][r] += sm.ld(mo, KD + tile_const_n<NT>(a, b, r));
                gv[a] += sm.ld(c, 16 * a);
            }
        }
        // M = U'U: off-diagonal tiles of C1 become U, V_k = U_kk^-1 goes to LDS
#pragma unroll
        for (int kk = 0; kk < NT; kk++) {
            const d4 Vk = diag_factor(C1[utn<NT>(kk, kk)], S, g, c);
#pragma unroll
            for (int r = 0; r < 4; r++) Vs[kk][(g + 4 * r) * kLdT + c] = Vk[r];
#pragma unroll
            for (int j = kk + 1; j < NT; j++) C1[utn<NT>(kk, j)] = tts(Vk, C1[utn<NT>(kk, j)], d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
            for (int i = kk + 1; i < NT; i++) {
                const d4 nu = neg(C1[utn<NT>(kk, i)]);
#pragma unroll
                for (int j = i; j < NT; j++) C1[utn<NT>(i, j)] = tts(nu, C1[utn<NT>(kk, j)], C1[utn<NT>(i, j)]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // z = U^-T gv on the vector pipe (as in rule64_body)
        double zrv[NT][4];
#pragma unroll
        for (int j = 0; j < NT; j++) {
            double wcv = gv[j];
#pragma unroll
            for (int kk = 0; kk < j; kk++) {
                double p = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) p += C1[utn<NT>(kk, j)][r] * zrv[kk][r];
                wcv -= sum_groups(p);
            }
            double p = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) p += Vs[j][(g + 4 * r) * kLdT + c] * cv_to_rv(wcv, g, r);
            const double zcv = sum_groups(p);
#pragma unroll
            for (int r = 0; r < 4; r++) zrv[j][r] = cv_to_rv(zcv, g, r);
        }
        // Y1 = U^-T B1 (in B1's registers), Y2 = U^-T B2'
        d4 Y2[NT][NT];
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = 0; b < NT; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) Y2[a][b][r] = Bt2.ld(mo, tile_const_n<NT>(a, b, r));
#pragma unroll
        for (int b = 0; b < NT; b++)
#pragma unroll
            for (int j = 0; j < NT; j++) {
                d4 Vj;
#pragma unroll
                for (int r = 0; r < 4; r++) Vj[r] = Vs[j][(g + 4 * r) * kLdT + c];
                B1[j][b] = tts(Vj, B1[j][b], d4{0.0, 0.0, 0.0, 0.0});
                Y2[j][b] = tts(Vj, Y2[j][b], d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
                for (int jj = j + 1; jj < NT; jj++) {
                    const d4 nu = neg(C1[utn<NT>(j, jj)]);
                    B1[jj][b] = tts(nu, B1[j][b], B1[jj][b]);
                    Y2[jj][b] = tts(nu, Y2[j][b], Y2[jj][b]);
                }
            }
        // P1 -= Y1'Y1, h1 += Y1'z;  C1 = C2 - Y2'Y2, c1 = c2 + Y2'z;  B1 = Y2'Y1
        d4 Bn[NT][NT];
#pragma unroll
        for (int a = 0; a < NT; a++) {
#pragma unroll
            for (int b = 0; b < NT; b++) {
                d4 G = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < NT; j++) G = tts(Y2[j][a], B1[j][b], G);
                Bn[a][b] = G;
            }
#pragma unroll
            for (int b = a; b < NT; b++) {
                d4 G1 = d4{0.0, 0.0, 0.0, 0.0}, G2 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < NT; j++) { G1 = tts(B1[j][a], B1[j][b], G1); G2 = tts(Y2[j][a], Y2[j][b], G2); }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    P1[utn<NT>(a, b)][r] -= G1[r];
                    C1[utn<NT>(a, b)][r] = C2.ld(mo, tile_const_n<NT>(a, b, r)) - G2[r];
                }
            }
            double p1 = 0.0, p2 = 0.0;
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) { p1 += B1[j][a][r] * zrv[j][r]; p2 += Y2[j][a][r] * zrv[j][r]; }
            h1[a] += sum_groups(p1);
            c1[a] = c2.ld(c, 16 * a) + sum_groups(p2);
        }
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = 0; b < NT; b++) B1[a][b] = Bn[a][b];
    }
    // the record: P | B | B' | C (row-major; P and C: their upper tiles) | h | c
    const PtrAcc oP = acc_of((gcdp)out), oB = acc_of((gcdp)(out + KD * KD)), oBt = acc_of((gcdp)(out + 2 * KD * KD)), oC = acc_of((gcdp)(out + 3 * KD * KD)),
                 ohc = acc_of((gcdp)(out + 4 * KD * KD));
#pragma unroll
    for (int a = 0; a < NT; a++) {
#pragma unroll
        for (int b = 0; b < NT; b++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = tile_const_n<NT>(a, b, r);
                oB.st(mo, o, B1[a][b][r]);
                if (b >= a) { oP.st(mo, o, P1[utn<NT>(a, b)][r]); oC.st(mo, o, C1[utn<NT>(a, b)][r]); }
            }
            // B' tile (b, a) is this tile transposed: through LDS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; r++) S[(g + 4 * r) * kLdT + c] = B1[a][b][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; r++) oBt.st(mo, tile_const_n<NT>(b, a, r), S[c * kLdT + g + 4 * r]);
        }
        if (g == 0) { ohc.st(c, 16 * a, h1[a]); ohc.st(c, KD + 16 * a, c1[a]); }
    }
}

static void chain64_drop(cx_handle *h, Chain64 *c) {
    if (!c) return;
    for (void *p : {(void *)c->d_jobs, (void *)c->d_children, (void *)c->d_steps, (void *)c->d_pot, (void *)c->d_ent, (void *)c->d_ring, (void *)c->d_aux, (void *)c->d_aux_src}) if (p) (void)hipFree(p);
    h->device_bytes -= c->bytes;
    delete c;
}

// the side information of a position with three or more side slots, summed: one workgroup per arena message
__global__ __launch_bounds__(256) void k_side64(int n, int msg, const int32_t *__restrict__ src, const double *__restrict__ f2v, double *__restrict__ aux) {
    const int w = blockIdx.x;
    if (w >= n) return;
    int s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = src[8 * w + j];
    for (int e = threadIdx.x; e < msg; e += 256) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++) if (s[j] >= 0) acc += f2v[(int64_t)s[j] * msg + e];
        aux[(int64_t)w * msg + e] = acc;
    }
}

void chain64_free(cx_handle *h) {
    chain64_drop(h, (Chain64 *)h->chain64);
    h->chain64 = nullptr;
}

// CX_SCHED_TREE over heavy paths (cx_tree_plan.h): one plan per light depth and direction of travel
void chain64_tree_free(cx_handle *h) {
    for (void *p : h->tree_c64) chain64_drop(h, (Chain64 *)p);
    h->tree_c64.clear();
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}

// Build the plan from the chain decomposition (host arrays of build_chains) and upload it.
static int32_t chain64_make(cx_handle *h, Chain64 **out, bool root, const std::vector<int32_t> &pos_var, const std::vector<int32_t> &skip0,
                            const std::vector<int32_t> &skip1, const std::vector<int32_t> &link_pos, const std::vector<int32_t> &from,
                            const std::vector<int32_t> &to, const std::vector<uint8_t> &head_fwd, const std::vector<uint8_t> &head_bwd,
                            const std::vector<int32_t> &tab_fwd, const std::vector<int32_t> &tab_bwd);

int32_t chain64_build(cx_handle *h, const std::vector<int32_t> &pos_var, const std::vector<int32_t> &skip0, const std::vector<int32_t> &skip1,
                      const std::vector<int32_t> &link_pos, const std::vector<int32_t> &from, const std::vector<int32_t> &to,
                      const std::vector<uint8_t> &head_fwd, const std::vector<uint8_t> &head_bwd, const std::vector<int32_t> &tab_fwd,
                      const std::vector<int32_t> &tab_bwd) {
    chain64_free(h);
    h->pot64_fresh = false;
    Chain64 *c = nullptr;
    const int32_t rc = chain64_make(h, &c, h->chain_partition, pos_var, skip0, skip1, link_pos, from, to, head_fwd, head_bwd, tab_fwd, tab_bwd);
    h->chain64 = c;      // (a plan that failed half-way is freed with the handle's)
    return rc;
}

// the plan of ONE light depth of the tree schedule's heavy paths, for one direction of travel (the way up skips a head's slot towards
// its parent: other side information); appended to h->tree_c64, *index = its place there
int32_t chain64_tree_build(cx_handle *h, int *index, const std::vector<int32_t> &pos_var, const std::vector<int32_t> &skip0, const std::vector<int32_t> &skip1,
                           const std::vector<int32_t> &link_pos, const std::vector<int32_t> &from, const std::vector<int32_t> &to,
                           const std::vector<uint8_t> &head_fwd, const std::vector<uint8_t> &head_bwd, const std::vector<int32_t> &tab_fwd,
                           const std::vector<int32_t> &tab_bwd) {
    Chain64 *c = nullptr;
    const int32_t rc = chain64_make(h, &c, false, pos_var, skip0, skip1, link_pos, from, to, head_fwd, head_bwd, tab_fwd, tab_bwd);
    if (rc != CX_OK) { chain64_drop(h, c); return rc; }
    *index = (int)h->tree_c64.size();
    h->tree_c64.push_back(c);
    return CX_OK;
}

static int32_t chain64_make(cx_handle *h, Chain64 **out, bool root, const std::vector<int32_t> &pos_var, const std::vector<int32_t> &skip0,
                            const std::vector<int32_t> &skip1, const std::vector<int32_t> &link_pos, const std::vector<int32_t> &from,
                            const std::vector<int32_t> &to, const std::vector<uint8_t> &head_fwd, const std::vector<uint8_t> &head_bwd,
                            const std::vector<int32_t> &tab_fwd, const std::vector<int32_t> &tab_bwd) {
    using cxh::fail;
    const int64_t npos = (int64_t)pos_var.size(), nlinks = (int64_t)link_pos.size();
    std::vector<int32_t> side((size_t)3 * npos, -1), side_aux((size_t)npos, -1), aux_src;
    for (int64_t p = 0; p < npos; p++) {
        const int32_t v = pos_var[p], deg = h->var_off[v + 1] - h->var_off[v];
        int32_t all[8];
        int n = 0;
        for (int32_t j = 0; j < deg && j < 8; j++) {
            const int32_t sj = h->vbase[v] + j * kBlock;
            if (sj == skip0[p] || sj == skip1[p]) continue;
            all[n++] = sj;
        }
        if (n <= 2 || (root && n == 3)) { for (int k = 0; k < n; k++) side[3 * p + k] = all[k]; continue; }      // (a time block's end variables hand their side slots to the caller as they are)
        // three or more side slots (with the entering message: more than the three sources a rule or a joint sums): summed first
        if (root) return fail(h, CX_ERR_UNSUPPORTED, "cx_chain_block_maps, dim 64: variable " + std::to_string(h->var_ids[v]) + " has more than three inputs besides its chain links");
        side_aux[p] = (int32_t)(aux_src.size() / 8);
        for (int k = 0; k < 8; k++) aux_src.push_back(k < n ? all[k] : -1);
    }
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    p64::Input in;
    in.d = h->cfg.dim; in.npos = npos; in.nlinks = nlinks;      // (16, 32 or 64: cx_const.h is_mfma_dim)
    in.link_pos = link_pos.data(); in.from = from.data(); in.to = to.data(); in.tab_fwd = tab_fwd.data(); in.tab_bwd = tab_bwd.data();
    in.head_fwd = head_fwd.data(); in.head_bwd = head_bwd.data(); in.side = side.data(); in.side_aux = side_aux.data();
    in.K0 = env_int("CX_MVC64_K", 0);            // links per level-0 block (default: one block per SIMD)
    in.fan = std::max(2, env_int("CX_MVC64_FAN", 2));      // binary tree: the shortest dependent chain above level 0 (fan 2 / 4 / 8: 14.19 / 14.37 / 15.08 ms on C5)
    // a composition of 4 x 4 tiles is one wave per SIMD (a pair of waves per two SIMDs); the small-tile kernels run three (1 x 1: four in
    // the walks) or two (2 x 2) waves per SIMD, and their time is a wave's chain of dependent pivots: more, shorter blocks
    in.lanes = 4 * (int64_t)ncu * (h->cfg.dim == 16 ? env_int("CX_MVC_NT_WAVES", 3) : h->cfg.dim == 32 ? env_int("CX_MVC_NT_WAVES", 2) : 1);
    in.root = root;                              // a time block hands its whole potential to the other blocks
    p64::Plan plan;
    try { plan = p64::build(in); }
    catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "chain-scan schedule, dim 64: host allocation failed"); }
    catch (const std::exception &e) { return fail(h, CX_ERR_UNSUPPORTED, std::string("chain-scan schedule, dim 64: ") + e.what()); }
    Chain64 *c = new (std::nothrow) Chain64();
    if (!c) return fail(h, CX_ERR_OUT_OF_MEMORY, "chain-scan schedule, dim 64: host allocation failed");
    *out = c;
    c->n_pot = plan.n_pot; c->n_ent = plan.n_ent; c->K0 = plan.K0; c->fan = plan.fan; c->levels = plan.levels;
    c->n_compositions = plan.n_compositions; c->n_rules = plan.n_rules;
    c->n_roots = (int)plan.root_pot.size();
    if (c->n_roots > 0) {
        c->root_off = plan.root_pot[0] & p64::kOffMask;
        for (int k = 0; k < 3; k++) { c->side_ends[k] = side[3 * plan.end_pos[0] + k]; c->side_ends[3 + k] = side[3 * plan.end_pos[1] + k]; }
    }
    auto &jobs = c->jobs;
    auto longest = [](const std::vector<p64::Job> &L) { int m = 0; for (const auto &j : L) m = std::max(m, (int)j.n); return m; };
    int widest_compose = 1;
    for (const auto &L : plan.compose_launches) widest_compose = std::max(widest_compose, (int)L.size());
    for (const auto &L : plan.compose_launches) if (!L.empty()) { c->launches.push_back({0, (int64_t)jobs.size(), (int)L.size(), longest(L)}); jobs.insert(jobs.end(), L.begin(), L.end()); }
    for (size_t i = 0; i < plan.walk_launches.size(); i++) {      // the last walk launch is the one along the links (plain rules, h = c = 0)
        const auto &L = plan.walk_launches[i];
        if (L.empty()) continue;
        c->launches.push_back({i + 1 == plan.walk_launches.size() ? 2 : 1, (int64_t)jobs.size(), (int)L.size(), longest(L)});
        jobs.insert(jobs.end(), L.begin(), L.end());
    }
    c->children = std::move(plan.children);
    c->steps = std::move(plan.steps);
    const int64_t before = h->device_bytes;
    int32_t rc;
    if ((rc = cxh::dev_alloc(h, &c->d_jobs, (int64_t)jobs.size())) != CX_OK) return rc;
    if ((rc = cxh::dev_alloc(h, &c->d_children, (int64_t)c->children.size())) != CX_OK) return rc;
    if ((rc = cxh::dev_alloc(h, &c->d_steps, (int64_t)c->steps.size())) != CX_OK) return rc;
    if ((rc = cxh::dev_alloc(h, &c->d_pot, plan.n_pot * plan.pot)) != CX_OK) return rc;
    if ((rc = cxh::dev_alloc(h, &c->d_ent, plan.n_ent * plan.msg)) != CX_OK) return rc;
    if (plan.n_pot > 0 && h->cfg.dim == kD && (rc = cxh::dev_alloc(h, &c->d_ring, (int64_t)widest_compose * 16 * kTileD)) != CX_OK) return rc;      // (the pair of waves' hand-over)
    c->n_aux = (int64_t)aux_src.size() / 8;
    if (c->n_aux > 0) {
        if ((rc = cxh::dev_alloc(h, &c->d_aux, c->n_aux * plan.msg)) != CX_OK) return rc;
        if ((rc = cxh::dev_upload(h, &c->d_aux_src, aux_src)) != CX_OK) return rc;
        CX_HIP(h, hipStreamSynchronize(h->stream));      // (aux_src is a local)
    }
    // a potential or entry message that was never computed reads as UndefValue()
    CX_HIP(h, hipMemsetAsync(c->d_pot, 0xff, (size_t)std::max<int64_t>(1, plan.n_pot * plan.pot) * 8, h->stream));
    CX_HIP(h, hipMemsetAsync(c->d_ent, 0xff, (size_t)std::max<int64_t>(1, plan.n_ent * plan.msg) * 8, h->stream));
    c->bytes = h->device_bytes - before;
    return CX_OK;
}

// The device records hold POINTERS: the plan's handles resolved against the six base pointers of the moment.  Redone (host loop +
// one upload per array) only when a base has moved: new rule tables (cx_set_factor_matrices) or a new plan.
static int32_t chain64_resolve(cx_handle *h, Chain64 *c) {
    double *bases[p64::kSpaces] = {h->d_zero_msg, h->d_mv_f2v, h->d_ptab, h->d_ptab_bt, c->d_pot, c->d_ent, c->d_aux};
    if (std::memcmp(bases, c->bases, sizeof bases) == 0) return CX_OK;
    try {
        auto ptr = [&](int64_t hd) { return (uint64_t)(uintptr_t)(bases[hd >> 56] + (hd & p64::kOffMask)); };
        std::vector<uint64_t> w;
        w.resize(c->jobs.size() * 2);
        for (size_t i = 0; i < c->jobs.size(); i++) {
            w[2 * i] = ptr(c->jobs[i].out);
            w[2 * i + 1] = (uint64_t)(uint32_t)c->jobs[i].first | ((uint64_t)(uint32_t)c->jobs[i].n << 32);
        }
        CX_HIP(h, hipMemcpy(c->d_jobs, w.data(), w.size() * 8, hipMemcpyHostToDevice));
        w.resize(c->children.size() * 10);
        for (size_t i = 0; i < c->children.size(); i++) {
            const p64::Child &r = c->children[i];
            const int64_t hs[9] = {r.P, r.B, r.Bt, r.C, r.h, r.c, r.side[0], r.side[1], r.side[2]};
            for (int k = 0; k < 9; k++) w[10 * i + k] = ptr(hs[k]);
            uint64_t ns = 0;
            for (int k = 0; k < 3; k++) ns += (r.side[k] >> 56) != p64::kZero;
            w[10 * i + 9] = ns;
        }
        if (!w.empty()) CX_HIP(h, hipMemcpy(c->d_children, w.data(), w.size() * 8, hipMemcpyHostToDevice));
        w.resize(c->steps.size() * 10);
        for (size_t i = 0; i < c->steps.size(); i++) {
            const p64::Step &r = c->steps[i];
            const int64_t hs[9] = {r.src[0], r.src[1], r.src[2], r.P, r.Bt, r.C, r.h, r.c, r.dst};
            for (int k = 0; k < 9; k++) w[10 * i + k] = ptr(hs[k]);
            w[10 * i + 9] = (r.src[2] >> 56) != p64::kZero ? 1 : 0;      // a third source: one extra block of loads in the rule
        }
        if (!w.empty()) CX_HIP(h, hipMemcpy(c->d_steps, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    } catch (const std::bad_alloc &) { return cxh::fail(h, CX_ERR_OUT_OF_MEMORY, "chain-scan schedule, dim 64: host allocation failed"); }
    std::memcpy(c->bases, bases, sizeof bases);
    return CX_OK;
}

// one exact sweep: every launch of the plan, in order, on the handle's stream
static int32_t chain64_run(cx_handle *h, Chain64 *c, bool skip_compose);

int32_t chain64_sweep(cx_handle *h) {
    Chain64 *c = (Chain64 *)h->chain64;
    if (!c) return cxh::fail(h, CX_ERR_STATE, "chain-scan schedule, dim 64: no plan");
    // a time block right after cx_chain_block_maps: the potentials are on the device already (what changed since — the messages that
    // enter the block at its two ends — is side information of END positions, which no composition reads)
    const bool skip_compose = h->pot64_fresh;
    h->pot64_fresh = false;
    return chain64_run(h, c, skip_compose);
}

// the tree schedule's plans: their device records resolved against the current base pointers (host copies: NOT inside a stream capture),
// their launches, and one of them run
int32_t chain64_tree_resolve(cx_handle *h) {
    for (void *p : h->tree_c64) { const int32_t rc = chain64_resolve(h, (Chain64 *)p); if (rc != CX_OK) return rc; }
    return CX_OK;
}
int64_t chain64_tree_launches(const cx_handle *h, int index) { return (int64_t)((const Chain64 *)h->tree_c64[index])->launches.size(); }
int32_t chain64_tree_sweep(cx_handle *h, int index) { return chain64_run(h, (Chain64 *)h->tree_c64[index], false); }

static int32_t chain64_run(cx_handle *h, Chain64 *c, bool skip_compose) {
    { int32_t rc = chain64_resolve(h, c); if (rc != CX_OK) return rc; }
    static const int walk_waves = env_int("CX_MVC64_WALK_WAVES", 2);
#ifdef CX_C64_STAMPS
    static unsigned long long *d_st = nullptr;
    const size_t st_n = (size_t)16 * 8192;
    if (!d_st) {
        (void)hipMalloc(&d_st, st_n * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(cx_w64_stamps), &d_st, sizeof d_st);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(cx_c64_stamps), &d_st, sizeof d_st);
    }
    std::vector<unsigned long long> hs(st_n);
#endif
    const int dim = h->cfg.dim;
    if (c->n_aux > 0) hipLaunchKernelGGL(k_side64, dim3((unsigned)c->n_aux), dim3(256), 0, h->stream, (int)c->n_aux, dim + dim * dim, c->d_aux_src, h->d_mv_f2v, c->d_aux);
    for (const auto &L : c->launches) {
#ifdef CX_C64_STAMPS
        (void)hipMemsetAsync(d_st, 0, st_n * 8, h->stream);
#endif
        if (L.kind == 0 && skip_compose) continue;
        if (dim != kD) {      // 1 x 1 / 2 x 2 tiles: one wave per job, whatever the launch
            const bool two = dim == 32;
            if (L.kind == 0) { if (two) hipLaunchKernelGGL(k_compose_nt<2>, dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_children); else hipLaunchKernelGGL(k_compose_nt<1>, dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_children); }
            else if (L.kind == 1) { if (two) hipLaunchKernelGGL((k_walk_nt<2, true>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps); else hipLaunchKernelGGL((k_walk_nt<1, true>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps); }
            else { if (two) hipLaunchKernelGGL((k_walk_nt<2, false>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps); else hipLaunchKernelGGL((k_walk_nt<1, false>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps); }
        } else if (L.kind == 0) {
            hipLaunchKernelGGL(k_compose64p, dim3(L.n), dim3(128), 0, h->stream, L.n, c->d_jobs + L.first, c->d_children, c->d_ring);
        } else if (L.kind == 1) {          // few jobs: a wave alone on its SIMD
            hipLaunchKernelGGL((k_walk64b<1, true>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps);
        } else if (walk_waves == 1) {
            hipLaunchKernelGGL((k_walk64b<1, false>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps);
        } else {
            hipLaunchKernelGGL((k_walk64b<2, false>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps);
        }
#ifdef CX_C64_STAMPS
        (void)hipStreamSynchronize(h->stream);
        (void)hipMemcpy(hs.data(), d_st, st_n * 8, hipMemcpyDeviceToHost);
        const int per = L.kind == 0 ? 16 : 8, nph = L.kind == 0 ? 13 : 6;
        double tot[16] = {0};
        for (int wg = 0; wg < std::min(L.n, (int)(st_n / per)); wg++) for (int i = 0; i < nph; i++) tot[i] += (double)hs[(size_t)per * wg + i];
        fprintf(stderr, "[c64 stamps] kind %d, %d jobs: cycles per job and phase:", L.kind, L.n);
        for (int i = 0; i < nph; i++) fprintf(stderr, " %.0f", tot[i] / L.n);
        fprintf(stderr, "\n");
#endif
    }
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

// cx_chain_block_maps, dim 64: the compose launches of the plan (no walks), then the ONE potential of the handle's path
// (P | B | B' | C row-major 64 x 64 — P and C hold their upper 16 x 16 tile blocks only — | h | c) and the side slots of its two ends.
// *no_root: the plan was built without a root (the caller rebuilds it as a partition's and asks again).
int32_t chain64_block_potential(cx_handle *h, double *pot, int32_t *side_first3, int32_t *side_last3, bool *no_root) {
    Chain64 *c = (Chain64 *)h->chain64;
    if (!c) return cxh::fail(h, CX_ERR_STATE, "chain-scan schedule, dim 64: no plan");
    if (h->cfg.dim != kD) return cxh::fail(h, CX_ERR_UNSUPPORTED, "cx_chain_block_maps: time blocks of a chain on the matrix-core path are exchanged as 64 x 64 potentials (dim 64)");
    *no_root = c->n_roots < 1;
    if (*no_root) return CX_OK;
    if (c->n_roots != 1) return cxh::fail(h, CX_ERR_UNSUPPORTED, "cx_chain_block_maps: the non-observed variables of this handle must form ONE path (a time block of a chain)");
    { int32_t rc = chain64_resolve(h, c); if (rc != CX_OK) return rc; }
    for (const auto &L : c->launches)
        if (L.kind == 0) hipLaunchKernelGGL(k_compose64p, dim3(L.n), dim3(128), 0, h->stream, L.n, c->d_jobs + L.first, c->d_children, c->d_ring);
    CX_HIP(h, hipGetLastError());
    CX_HIP(h, hipMemcpyAsync(pot, c->d_pot + c->root_off, (size_t)(4 * kD * kD + 2 * kD) * 8, hipMemcpyDeviceToHost, h->stream));
    CX_HIP(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 3; k++) { side_first3[k] = c->side_ends[k]; side_last3[k] = c->side_ends[3 + k]; }
    for (int k = 0; k < 6; k++) h->pot64_end_slots[k] = c->side_ends[k];
    h->pot64_fresh = true;
    return CX_OK;
}

void chain64_stats(const cx_handle *h, int64_t *out8) {
    const Chain64 *c = (const Chain64 *)h->chain64;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    if (!c) return;
    out8[0] = c->K0; out8[1] = c->fan; out8[2] = c->levels; out8[3] = c->n_pot; out8[4] = c->n_compositions; out8[5] = c->n_rules;
    out8[6] = (int64_t)c->launches.size() + (c->n_aux > 0 ? 1 : 0);      // kernel launches per sweep: a walk loops over its steps INSIDE one launch (chain64_run); + k_side64
    out8[7] = c->bytes;
}

}  // namespace cx
