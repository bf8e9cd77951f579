// cx_chain.hip — exact sum-product on chain-structured graphs by two associative scans (CX_SCHED_CHAIN_SCAN).
//
// What it replaces: on a state-space chain the reference's `update_marginals!` is strictly sequential — a forward
// pass and a reverse pass of process! calls, 5T−4 message computations (src/inference_engine.jl:575-608; hand trace
// in SURVEY.md §3.3).  On the device the same messages come from two parallel prefix scans over per-link maps.
//
// With messages in natural form m = (xi, w), adding the side information u_t of variable t (the sum of its
// non-chain incoming messages: likelihoods, priors) and passing through the factor rule
//     (xi, w) -> ((a (xi+u_xi) + b (w+u_w)) s, (w+u_w) s),   s = 1 / (a² + q (w+u_w))
// is a projective-linear map on (xi, w, 1):
//     [xi_num]   [e f g] [xi]
//     [w_num ] = [0 A B] [w ]      with  e = a, f = b, g = a u_xi + b u_w, A = 1, B = u_w, C = q, D = a² + q u_w
//     [den   ]   [0 C D] [1 ]
// Such matrices are closed under multiplication, so the forward messages α_{t+1} = (F_t ∘ … ∘ F_1)(0, 0) and the
// backward messages β_t = (G_t ∘ … ∘ G_{T-1})(0, 0) are inclusive scans; every product is rescaled to D = 1, which
// keeps all entries bounded (A, B, C, D ≥ 0 never cancel).  Applied to (0,0) a prefix yields (g, B).
// Path boundaries are handled by a segmented scan (head flags).  The arithmetic is re-associated relative to the
// sequential schedule: results agree to rounding (tests hold 1e-9), not bitwise.

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cx_internal.h"
#include "cx_lin.h"

namespace cx {

constexpr int kItems = 4;                    // links per thread
constexpr int kTile = kBlock * kItems;       // links per workgroup

// the factor variance of a receiving slot: the per-slot table, or — when an owner of this handle names precision variables
// (cx_vmp.hip: q = 1 / E[precision], the same for every factor of one precision variable) — read through the slot's index
__device__ __forceinline__ double slot_q(const double *__restrict__ q, const int32_t *__restrict__ qg, const double *__restrict__ gm, int slot) {
    if (qg) { const int g = qg[slot]; if (g >= 0) return 1.0 / gm[g]; }
    return q[slot];
}

// side information of chain position i: the sum of the variable's incoming messages except its (≤2) chain slots.
// FUSED_LEAVES: those messages are first recomputed from the variable→factor messages of their senders (observed leaves,
// priors keep their stored value) and stored — the factor phase of the flooding schedule restricted to the slots the
// scans do not produce, so that no separate pass over all slots is needed.
template <bool FUSED_LEAVES>
__global__ __launch_bounds__(kBlock) void k_chain_side(int npos, const int32_t *__restrict__ pos_var, const int32_t *__restrict__ pos_skip0,
                                                       const int32_t *__restrict__ pos_skip1, const int32_t *__restrict__ vbase,
                                                       const int32_t *__restrict__ vdeg, const uint8_t *__restrict__ vinfo,
                                                       const int32_t *__restrict__ partner, const double *__restrict__ q,
                                                       const int32_t *__restrict__ qg, const double *__restrict__ gm,
                                                       const double *__restrict__ pa, const double *__restrict__ pb,
                                                       const double2 *__restrict__ v2f, double2 *__restrict__ f2v,
                                                       double2 *__restrict__ side) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= npos) return;
    const int v = pos_var[i], s0 = pos_skip0[i], s1 = pos_skip1[i];
    const int stride = ((vinfo[v] & kDegMask) == kBigDeg) ? 1 : kBlock;
    const int b = vbase[v], deg = vdeg[v];
    double2 acc = make_double2(0.0, 0.0);
    for (int k = 0; k < deg; k++) {
        const int slot = b + k * stride;
        if (slot == s0 || slot == s1) continue;
        double2 m = f2v[slot];
        if (FUSED_LEAVES) {
            const int p = partner[slot];
            if (p >= 0) {
                const double2 in = v2f[p];
                if (!__builtin_isnan(in.y)) {
                    m = chain_factor_rule(in, slot_q(q, qg, gm, slot), pa ? pa[slot] : 1.0, pb ? pb[slot] : 0.0);
                    f2v[slot] = m;
                }
            }
        }
        acc.x += m.x; acc.y += m.y;
    }
    side[i] = acc;
}

// Direction dir = +1: link l maps α at position l to α at position l+1 (uses side[l], parameters of to_slot[l]).
// Direction dir = -1: links are visited in reverse order; link l maps β at position l+1 to β at position l
// (uses side[l+1], parameters of from_slot[l]).  `pos_of_link[l]` is the chain position of the link's left end.
// The receiving slot's rule parameters copied into link order, one array each per direction (round 6): q[to_slot[l]] is a load behind a
// load, and in a launch that is a chain of five trips to memory every one of them shows (k_chain_onepass).  Arrays, not records: a
// thread owns K CONSECUTIVE links, so its K values of one array are one or two 16-byte loads.
struct LinkPar {
    const double *q, *a, *b;      // (a, b null on a graph without linear factors)
    const int32_t *recv_seg;      // receiving slot | (first link of its path in this direction) << 31
    const int32_t *g;             // slot_q's precision index (-1: q as stored), null without one
};

struct ChainArgs {
    int nlinks;
    const int32_t *link_pos;     // position of the left variable of link l
    const int32_t *from_slot;    // slot (left variable, factor)
    const int32_t *to_slot;      // slot (right variable, factor)
    const uint8_t *head_fwd;     // link l is the first link of its path
    const uint8_t *head_bwd;     // link l is the last link of its path
    const double *q, *a, *b;     // rule parameters per RECEIVING slot (a, b may be null: additive)
    const int32_t *qg;           // slot_q: optional indirection of q through a precision variable's mean
    const double *gm;
    const double2 *side;
    const int32_t *pos_var;      // variable at a chain position (for the marginals k_chain_apply<.., true> writes)
    int pos0;                    // >= 0: link_pos[l] == pos0 + l for every link (one path): a side sum's address needs no load
    LinkPar par[2];              // the rule parameters in LINK order, forward / backward (k_chain_linkpar); par[0].q null: not made
};

// Tiles are blocks of K * T consecutive links in BOTH directions (K links per thread, T threads per direction): the tile at position p
// of the forward scan is block p, of the backward scan block ntiles - 1 - p, visited from its last link to its first (the ragged
// block comes first there).  One workgroup holds the same block for both directions, so the side sums and rule parameters of a link
// cross the fabric once (with separate forward / backward workgroups — on different XCDs, behind different L2s — the two scan kernels
// of a 1M-link chain fetched 184 + 88 MB: profiles/r02_vmp_rocprof.md).
//
// Round 4, second form.  The first form gave every thread four links, scanned the 1024 maps of a tile with a workgroup scan (twice: once
// for the tile totals, once in the apply kernel, which composed carry ∘ prefix for every link) and composed the tile carries with a
// third one: ≈ 14 instructions per link and direction.  Now a thread owns a RUN of K consecutive links: the totals kernel composes the
// run's maps in order, scans the run totals of a wave, and stores each thread's exclusive prefix within its tile; the apply kernel
// turns that prefix and the tile's entering message into the message at the head of the run and WALKS the run with the message rule
// itself (one division and six multiply-adds per link instead of a 30-flop composition): ≈ 7 instructions per link and direction.
// What that bought is less than the count suggests (C2 23.5 -> 21.5 us, the structured family unchanged): the counters show the vector
// pipe ≈ 30 % busy and the waves 55 - 73 % of their life in s_waitcnt — the kernels are bound by their dependent loads (link -> position
// -> side sum; link -> slot -> precision index -> mean) at four waves per SIMD, not by instruction issue.
struct LinkIn {       // what the rule of one link needs
    double2 u;        // side sum of the variable the link leaves
    double q, a, b;
    int recv, seg, link;
};

template <int K, int T>
__device__ __forceinline__ int run_link(const ChainArgs &A, int block, int j, int dir) {      // the link at place j of the tile's scan order, -1 past its end
    const int lo = block * (K * T), hi = min(lo + K * T, A.nlinks);
    if (j >= hi - lo) return -1;
    return dir > 0 ? lo + j : hi - 1 - j;
}

__device__ __forceinline__ int chain_pos(const ChainArgs &A, int l) { return A.pos0 >= 0 ? A.pos0 + l : A.link_pos[l]; }

__device__ __forceinline__ LinkIn load_link_in(const ChainArgs &A, int l, int dir) {
    LinkIn in;
    in.link = l;
    if (l < 0) { in.u = make_double2(0.0, 0.0); in.q = 0.0; in.a = 1.0; in.b = 0.0; in.recv = -1; in.seg = 0; return in; }
    in.u = A.side[chain_pos(A, l) + (dir > 0 ? 0 : 1)];
    if (A.par[0].q) {
        const LinkPar &P = A.par[dir > 0 ? 0 : 1];
        const int rs = P.recv_seg[l], g = P.g ? P.g[l] : -1;
        in.recv = rs & 0x7fffffff; in.seg = rs < 0 ? 1 : 0;
        in.q = g >= 0 ? 1.0 / A.gm[g] : P.q[l];
        in.a = P.a ? P.a[l] : 1.0; in.b = P.b ? P.b[l] : 0.0;
        return in;
    }
    in.recv = dir > 0 ? A.to_slot[l] : A.from_slot[l];
    in.seg = dir > 0 ? A.head_fwd[l] : A.head_bwd[l];
    in.q = slot_q(A.q, A.qg, A.gm, in.recv);
    in.a = A.a ? A.a[in.recv] : 1.0;
    in.b = A.b ? A.b[in.recv] : 0.0;
    return in;
}

// a thread's run of K links in scan order.  With link-ordered parameters and one path the run is K consecutive entries of every array
// (ascending in memory; the backward scan reads them in reverse): indexed from one base, so the compiler forms 16-byte loads.
template <int K, int T>
__device__ __forceinline__ void load_run(const ChainArgs &A, int block, int tid, int dir, LinkIn (&in)[K]) {
    const int lo = block * (K * T), hi = min(lo + K * T, A.nlinks), j0 = tid * K;
    if (A.par[0].q && A.pos0 >= 0 && hi - lo == K * T) {      // a whole tile: the run's base is a multiple of K in both directions
        const LinkPar &P = A.par[dir > 0 ? 0 : 1];
        const int base = dir > 0 ? lo + j0 : hi - j0 - K;
        const double2 *side = A.side + A.pos0 + base + (dir > 0 ? 0 : 1);
        const double *pq = (const double *)__builtin_assume_aligned(P.q + base, 8 * K), *pa = P.a ? (const double *)__builtin_assume_aligned(P.a + base, 8 * K) : nullptr,
                     *pb = P.b ? (const double *)__builtin_assume_aligned(P.b + base, 8 * K) : nullptr;
        const int32_t *prs = (const int32_t *)__builtin_assume_aligned(P.recv_seg + base, 4 * K), *pg = P.g ? (const int32_t *)__builtin_assume_aligned(P.g + base, 4 * K) : nullptr;
        double2 u[K]; double q[K], a[K], b[K]; int rs[K], g[K];
#pragma unroll
        for (int k = 0; k < K; k++) { u[k] = side[k]; q[k] = pq[k]; rs[k] = prs[k]; }
        if (pa) {
#pragma unroll
            for (int k = 0; k < K; k++) { a[k] = pa[k]; b[k] = pb[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) { a[k] = 1.0; b[k] = 0.0; }
        }
        if (pg) {
#pragma unroll
            for (int k = 0; k < K; k++) g[k] = pg[k];
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) g[k] = -1;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int kk = dir > 0 ? k : K - 1 - k;
            in[k].link = base + kk; in[k].u = u[kk]; in[k].a = a[kk]; in[k].b = b[kk];
            in[k].recv = rs[kk] & 0x7fffffff; in[k].seg = rs[kk] < 0 ? 1 : 0;
            in[k].q = g[kk] >= 0 ? 1.0 / A.gm[g[kk]] : q[kk];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < K; k++) in[k] = load_link_in(A, run_link<K, T>(A, block, j0 + k, dir), dir);
}

__global__ __launch_bounds__(kBlock) void k_chain_linkpar(int nlinks, const int32_t *__restrict__ from_slot, const int32_t *__restrict__ to_slot,
                                                          const uint8_t *__restrict__ head_fwd, const uint8_t *__restrict__ head_bwd, const double *__restrict__ q,
                                                          const double *__restrict__ a, const double *__restrict__ b, const int32_t *__restrict__ qg,
                                                          int stride, double *__restrict__ oq, double *__restrict__ oa, double *__restrict__ ob,
                                                          int32_t *__restrict__ ors, int32_t *__restrict__ og) {
    // (output arrays: forward half at [0, nlinks), backward half at [stride, stride + nlinks))
    const int l = blockIdx.x * kBlock + threadIdx.x;
    if (l >= nlinks) return;
    const int t = to_slot[l], f = from_slot[l];
    oq[l] = q[t]; oq[stride + l] = q[f];
    if (a) { oa[l] = a[t]; oa[stride + l] = a[f]; ob[l] = b[t]; ob[stride + l] = b[f]; }
    ors[l] = t | (head_fwd[l] ? (int)0x80000000 : 0); ors[stride + l] = f | (head_bwd[l] ? (int)0x80000000 : 0);
    if (qg) { og[l] = qg[t]; og[stride + l] = qg[f]; }
}

// workgroup-wide inclusive scan of kTile links; returns this thread's kItems inclusive prefixes and the tile total
// (tid: thread within its group of kBlock; a workgroup of 2 * kBlock threads runs two such scans side by side, each on its own
// LDS array — every thread of the workgroup has to make the call: __syncthreads)
__device__ __forceinline__ void tile_scan(Lin (&x)[kItems], Lin &tile_total, Lin *wave_tot /* LDS [kBlock/64] */, int tid) {
    const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int k = 1; k < kItems; k++) x[k] = lin_compose(x[k - 1], x[k]);
    Lin t = x[kItems - 1];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Lin o = lin_shfl_up(t, d);
        if (lane >= d) t = lin_compose(o, t);
    }
    if (lane == 63) wave_tot[wid] = t;
    __syncthreads();
    Lin excl = lin_shfl_up(t, 1);            // exclusive prefix within the wave
    if (lane == 0) excl = lin_identity();
    Lin carry = lin_identity();
    for (int w = 0; w < wid; w++) carry = lin_compose(carry, wave_tot[w]);
    excl = lin_compose(carry, excl);
#pragma unroll
    for (int k = 0; k < kItems; k++) x[k] = lin_compose(excl, x[k]);
    tile_total = wave_tot[0];
    for (int w = 1; w < kBlock / 64; w++) tile_total = lin_compose(tile_total, wave_tot[w]);
    __syncthreads();
}

// The forward and the backward scan are independent: every stage runs both in one launch, threads 0..T-1 of a workgroup forward, the
// other T backward over the same block of links, with the backward direction's tile totals stored after the forward ones (stride
// ntiles + 1) in scan order.  `run_excl`: per direction, tile (scan order) and thread, the composition of the runs before the thread's
// own within its tile.
template <int K, int T>
__global__ __launch_bounds__(2 * T) void k_chain_run_totals(ChainArgs A, Lin *__restrict__ totals, Lin *__restrict__ run_excl) {
    const int half = threadIdx.x / T, tid = threadIdx.x % T, dir = half == 0 ? 1 : -1, ntiles = gridDim.x;
    const int pos = half == 0 ? blockIdx.x : ntiles - 1 - blockIdx.x;       // this block's place in the direction's scan order
    const int lane = tid & 63, wid = tid >> 6;
    __shared__ Lin wave_tot[2][T / 64];
    LinkIn in[K];
    load_run<K, T>(A, blockIdx.x, tid, dir, in);
    Lin t = lin_identity();
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (in[k].link < 0) continue;
        const Lin m = lin_of_link(in[k].u, in[k].q, in[k].a, in[k].b, in[k].seg);
        t = k == 0 ? m : lin_compose(t, m);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Lin o = lin_shfl_up(t, d);
        if (lane >= d) t = lin_compose(o, t);
    }
    if (lane == 63) wave_tot[half][wid] = t;
    __syncthreads();
    Lin excl = lin_shfl_up(t, 1);            // exclusive prefix within the wave
    if (lane == 0) excl = lin_identity();
    if (wid > 0) {
        Lin carry = wave_tot[half][0];
        for (int w = 1; w < wid; w++) carry = lin_compose(carry, wave_tot[half][w]);
        excl = lin_compose(carry, excl);
    }
    run_excl[((size_t)half * ntiles + pos) * T + tid] = excl;
    if (tid == 0) {
        Lin tot = wave_tot[half][0];
        for (int w = 1; w < T / 64; w++) tot = lin_compose(tot, wave_tot[half][w]);
        totals[(size_t)half * (ntiles + 1) + pos] = tot;
    }
}

// exclusive scan of the tile totals by one workgroup (sequential over chunks of kTile tiles: ≤ 1M links per chunk)
__global__ __launch_bounds__(kBlock) void k_chain_scan_totals(int ntiles, Lin *__restrict__ totals) {
    totals += (size_t)blockIdx.x * (ntiles + 1);     // one workgroup per direction
    __shared__ Lin wave_tot[kBlock / 64];
    __shared__ Lin carry_s;
    if (threadIdx.x == 0) carry_s = lin_identity();
    __syncthreads();
    for (int chunk = 0; chunk < ntiles; chunk += kTile) {
        Lin x[kItems], incl[kItems];
        const int base = chunk + threadIdx.x * kItems;
#pragma unroll
        for (int k = 0; k < kItems; k++) x[k] = (base + k < ntiles) ? totals[base + k] : lin_identity();
#pragma unroll
        for (int k = 0; k < kItems; k++) incl[k] = x[k];
        Lin tot;
        tile_scan(incl, tot, wave_tot, threadIdx.x);
        const Lin carry = carry_s;
        // exclusive value of element k = carry ∘ (inclusive prefix of the previous element)
        Lin prev = lin_identity();
        {
            // previous thread's last inclusive value
            Lin lastv = incl[kItems - 1];
            Lin up = lin_shfl_up(lastv, 1);
            __shared__ Lin wave_last[kBlock / 64];
            if ((threadIdx.x & 63) == 63) wave_last[threadIdx.x >> 6] = lastv;
            __syncthreads();
            if ((threadIdx.x & 63) == 0) prev = (threadIdx.x == 0) ? lin_identity() : wave_last[(threadIdx.x >> 6) - 1];
            else prev = up;
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < kItems; k++) {
            const Lin ex = lin_compose(carry, k == 0 ? prev : incl[k - 1]);
            if (base + k < ntiles) totals[base + k] = ex;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = lin_compose(carry, tot);
        __syncthreads();
    }
}

// OWN_CARRY: the tile totals are NOT pre-scanned; every workgroup composes the totals of the tiles before it itself (each thread a
// run of them in order, a wave scan, the wave totals by one thread) — up to kOwnCarryTiles tiles that is cheaper than the
// one-workgroup scan kernel and its launch.
constexpr int kOwnCarryTiles = 2048;
__device__ __forceinline__ double2 chain_to_moment(double2 nat) {      // as cx_kernels.hip's to_moment
    const double var = 1.0 / nat.y;
    return make_double2(nat.x * var, var);
}

// MARG: the workgroup also writes the marginals of its block's chain variables.  It holds alpha (the forward message into the
// right end of every link of the block) and beta (the backward message into the left end): the marginal of a link's LEFT variable
// is side + beta(this link) + alpha(previous link) — across a block seam the message that enters the tile, none at the head of a
// path — and the last link of a path also owns its RIGHT variable (side + alpha).  With every reader of factor→variable messages
// on a chain this replaces the variable phase of the sweep (one launch less; variable→factor messages are then recomputed from
// the stored messages when somebody asks for them, like in the fused schedule).
template <int K, int T, bool OWN_CARRY, bool MARG>
__global__ __launch_bounds__(2 * T) void k_chain_run_apply(ChainArgs A, const Lin *__restrict__ tile_excl, const Lin *__restrict__ run_excl,
                                                           double2 *__restrict__ f2v, double2 *__restrict__ marg, int marg_form,
                                                           double2 *__restrict__ chain_v2f, double *__restrict__ split_mean,
                                                           double *__restrict__ split_prec, const bool store_msgs) {
    constexpr int kRunTile = K * T;
    const int half = threadIdx.x / T, tid = threadIdx.x % T, dir = half == 0 ? 1 : -1, ntiles = gridDim.x;
    const int pos = half == 0 ? blockIdx.x : ntiles - 1 - blockIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    tile_excl += (size_t)half * (ntiles + 1);
    __shared__ Lin wave_tot[2][T / 64];
    __shared__ double2 msg_s[MARG ? 2 : 1][MARG ? kRunTile : 1];      // [0]: alpha by link of the block, [1]: beta
    __shared__ double2 seam[2];                                       // the message that enters the tile: [0] alpha from the left, [1] beta from the right
    // this thread's run and its prefix within the tile: issued before the carry is worked out
    LinkIn in[K];
    load_run<K, T>(A, blockIdx.x, tid, dir, in);
    const Lin ex = run_excl[((size_t)half * ntiles + pos) * T + tid];
    if (OWN_CARRY) {
        // totals[0 .. pos) precede this tile in its direction: thread t composes the run [t * per, (t + 1) * per) of them in order
        const int per = (pos + T - 1) / T;
        if (wid * 64 * per < pos) {            // (a wave whose runs are all empty has nothing to add)
            const int b = tid * per, e = min(b + per, pos);
            Lin c = lin_identity();
            for (int j = b; j < e; j++) c = j == b ? tile_excl[j] : lin_compose(c, tile_excl[j]);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                Lin o = lin_shfl_up(c, d);
                if (lane >= d) c = lin_compose(o, c);
            }
            if (lane == 63) wave_tot[half][wid] = c;
        } else if (lane == 63) wave_tot[half][wid] = lin_identity();
        __syncthreads();
        if (tid == 0) {
            Lin c = wave_tot[half][0];
            for (int w = 1; w < T / 64; w++) c = lin_compose(c, wave_tot[half][w]);
            seam[half] = make_double2(c.g, c.B);      // the carry applied to the empty message
        }
    } else if (tid == 0) {
        const Lin c = tile_excl[pos];
        seam[half] = make_double2(c.g, c.B);
    }
    __syncthreads();
    const int lo = blockIdx.x * kRunTile;
    double2 m = lin_apply(ex, seam[half]);          // the message into the variable this thread's run starts at
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int l = in[k].link;
        if (l < 0) continue;
        if (in[k].seg) m = make_double2(0.0, 0.0);
        const double2 v = make_double2(m.x + in[k].u.x, m.y + in[k].u.y);
        const double s = 1.0 / (in[k].a * in[k].a + in[k].q * v.y);
        m = make_double2((in[k].a * v.x + in[k].b * v.y) * s, v.y * s);
        if (MARG) msg_s[half][l - lo] = m;      // stored below, in link order across the lanes
        else if (!__builtin_isnan(m.y)) f2v[in[k].recv] = m;
    }
    if (MARG) {
        __syncthreads();
        const double2 seam_alpha = seam[0], seam_beta = seam[1];
        const int cnt = min(kRunTile, A.nlinks - lo);
        for (int j = threadIdx.x; j < cnt; j += 2 * T) {
            const int l = lo + j, p = chain_pos(A, l);
            const double2 sd = A.side[p], be = msg_s[1][j];
            if (store_msgs) {
                const double2 al = msg_s[0][j];
                if (!__builtin_isnan(al.y)) f2v[A.to_slot[l]] = al;
                if (!__builtin_isnan(be.y)) f2v[A.from_slot[l]] = be;
            }
            double2 lx = sd;                                                  // everything the left variable hears except this link
            if (!A.head_fwd[l]) { const double2 al = j > 0 ? msg_s[0][j - 1] : seam_alpha; lx.x += al.x; lx.y += al.y; }
            const double2 t = make_double2(lx.x + be.x, lx.y + be.y);
            // marg_form 3: (mean, precision) into two arrays of the handle's owner (cx_vmp.hip keeps its marginals that way: no copy pass)
            if (marg_form == 3) { const int v = A.pos_var[p]; split_mean[v] = t.x / t.y; split_prec[v] = t.y; }
            else marg[A.pos_var[p]] = marg_form == 2 ? t : chain_to_moment(t);
            const double2 s1 = A.side[p + 1], al1 = msg_s[0][j];
            if (A.head_bwd[l]) {
                const double2 u = make_double2(s1.x + al1.x, s1.y + al1.y);
                if (marg_form == 3) { const int v = A.pos_var[p + 1]; split_mean[v] = u.x / u.y; split_prec[v] = u.y; }
                else marg[A.pos_var[p + 1]] = marg_form == 2 ? u : chain_to_moment(u);
            }
            if (chain_v2f) {
                // the two variable→factor messages of the link (the variational families read them: cx_vmp.hip, k_rate): what each
                // end hears from everybody else.  Undefined ones keep their stored value, like the variable phase does.
                if (!__builtin_isnan(lx.y)) chain_v2f[A.from_slot[l]] = lx;
                double2 rx = s1;
                if (!A.head_bwd[l]) { const double2 bn = j + 1 < cnt ? msg_s[1][j + 1] : seam_beta; rx.x += bn.x; rx.y += bn.y; }
                if (!__builtin_isnan(rx.y)) chain_v2f[A.to_slot[l]] = rx;
            }
        }
    }
}


// ---- ONE launch (round 6): the two kernels above merged ("single-pass scan with decoupled look-back", the totals tagged, not flagged) ------
// k_chain_run_totals ends where k_chain_run_apply begins only because a workgroup's carry is made of OTHER workgroups' tile totals: a
// kernel boundary used as a device-wide barrier, ≈ 5.7 us of a 21 us sweep at C2, plus every thread's prefix written out and read back
// (run_excl: 14 MB at C2).  Here a workgroup publishes its two tile totals (128 bytes each, write-through stores), keeps its runs'
// prefixes in registers, and composes the totals of the tiles before it as they appear: nobody waits for anything but data.
//   * Publishing comes BEFORE any wait, so the launch makes progress whenever every workgroup is or becomes resident; the launcher takes
//     this path only when the whole grid fits the device at once (occupancy x compute units), and every wait is bounded in time: a
//     workgroup that gives up raises a word in host memory (chain_abort), nobody stores a result, and the next call that waits for the
//     stream fails loudly and switches the handle back to the two launches (a chain-scan sweep is exact whatever it starts from: repeat it).
//   * Totals cross XCDs: written through (sc0 sc1), read past the caches (sc0 sc1), each 16-byte piece carrying the launch's tag beside its
//     value (below) — no flag, no fence: an agent-scope release / acquire would write back and invalidate the XCD's whole L2 under the
//     other workgroups' loads, and a flag is a second round trip through the memory the XCDs share.
//   * A wave fetches the totals its 64 threads compose with eight lanes per total — one 128-byte request each (stage_totals); the maps are
//     kept up to scale (cx_lin.h: LinP), so a composition has no division, and the wave scans run on DPP row shifts and broadcasts.
//   * The epoch the tag is made of lives on the device and is moved on inside the launch (by workgroup 0 when it is past its look-back:
//     it has then seen every other workgroup's total, i.e. every workgroup has read the word), so a captured graph replays.
constexpr int kOnepassMaxTiles = 2048;
constexpr int kTotalPieces = 8;        // 16-byte pieces per published total: records are 128 bytes apart
struct OnepassCtl { unsigned epoch, pad[15]; };
typedef unsigned int cx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long cx_u64;
// A published total is eight pieces (value, value XOR tag), the tag a 64-bit pattern of the launch's epoch: a reader that finds the
// relation in all eight has the values of THIS launch — no flag, so no second round trip through the memory the XCDs share, and no
// assumption that a 16-byte store is seen whole (a piece torn at any granularity fails the relation; pieces of an earlier launch carry
// another tag).
__device__ __forceinline__ cx_u64 epoch_tag(unsigned epoch) { return ((cx_u64)(epoch + 1u) * 0x9E3779B97F4A7C15ull) | 1ull; }
__device__ __forceinline__ void store_piece(double2 *p, double v, cx_u64 tag) {
    const cx_u64 b = (cx_u64)__double_as_longlong(v), c = b ^ tag;
    cx_u32x4 r;
    r[0] = (unsigned)b; r[1] = (unsigned)(b >> 32); r[2] = (unsigned)c; r[3] = (unsigned)(c >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");      // written through (system scope)
}
__device__ __forceinline__ void publish_total(double2 *p, const LinP &t, cx_u64 tag) {
    store_piece(p, t.e, tag); store_piece(p + 1, t.f, tag); store_piece(p + 2, t.g, tag); store_piece(p + 3, t.A, tag);
    store_piece(p + 4, t.B, tag); store_piece(p + 5, t.C, tag); store_piece(p + 6, t.D, tag); store_piece(p + 7, (double)t.seg, tag);
}
// A wave fetches the totals its 64 threads compose: lane l asks for piece l & 7 of the total of thread (l >> 3) + 8 i, i = 0 .. 7 — the
// eight pieces of a total are ONE 128-byte request (a thread reading its own total piece by piece made eight, and the 245 tiles of C2
// 480 k of them per round on 490 lines) — past the caches, again until every piece carries this launch's tag; the values go through
// LDS to the thread that composes them.  `first`: index of the total of the wave's first thread; `stride`: between two threads' totals;
// `limit`: totals exist below it.  false: gave up (time) or somebody else did.
__device__ __forceinline__ bool stage_totals(const double2 *tot, int first, int stride, int limit, cx_u64 tag, double (*out)[8] /* LDS, the wave's 64 rows */,
                                             int lane, const unsigned *abort_word, unsigned long long wait_limit) {
    const int piece = lane & 7;
    unsigned pending = 0;      // bit i: the i-th request of this lane is still to be answered with the tag
    const double2 *src[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int j = first + (i * 8 + (lane >> 3)) * stride;
        src[i] = tot + (size_t)j * kTotalPieces + piece;
        if (j < limit) pending |= 1u << i;
    }
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; spins++) {
        cx_u32x4 r[8];
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (pending & (1u << i)) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r[i]) : "v"(src[i]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (!(pending & (1u << i))) continue;
            const cx_u64 b = (cx_u64)r[i][0] | ((cx_u64)r[i][1] << 32), c = (cx_u64)r[i][2] | ((cx_u64)r[i][3] << 32);
            if ((b ^ c) == tag) { out[i * 8 + (lane >> 3)][piece] = __longlong_as_double((long long)b); pending &= ~(1u << i); }
        }
        if (!__any(pending != 0)) return true;
        if ((spins & 15u) == 15u) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return false;
            const unsigned long long now = wall_clock64();      // 100 MHz
            if (!t0) t0 = now; else if (now - t0 > wait_limit) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// results nobody reads again inside the launch: stored nontemporal on request, so that they do not push the links' parameters out of the L2s
typedef double cx_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void out_store(double2 *p, double2 v, bool nt) {
    if (nt) { cx_d2v t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, (cx_d2v *)p); }
    else *p = v;
}

// WIDE: held to 128 registers (two workgroups of 512 per compute unit: grids of up to twice the compute units stay resident) by
// loading what the marginal phase reads where it reads it
template <int K, int T, bool MARG, bool WIDE>
__global__ __launch_bounds__(2 * T, WIDE ? 4 : 2) void k_chain_onepass(ChainArgs A, double2 *__restrict__ totals64, OnepassCtl *__restrict__ ctl, unsigned *__restrict__ abort_word,
                                                         unsigned long long wait_limit_and_fault, double2 *__restrict__ f2v, double2 *__restrict__ marg, int marg_form,
                                                         double2 *__restrict__ chain_v2f, double *__restrict__ split_mean, double *__restrict__ split_prec,
                                                         const bool store_msgs, const bool nt, unsigned long long *__restrict__ stamps) {
    constexpr int kRunTile = K * T;
    constexpr int kOut = MARG ? (kRunTile + 2 * T - 1) / (2 * T) : 1;      // links per thread of the marginal phase
    const int half = threadIdx.x / T, tid = threadIdx.x % T, dir = half == 0 ? 1 : -1, ntiles = gridDim.x;
    const int pos = half == 0 ? blockIdx.x : ntiles - 1 - blockIdx.x;       // this block's place in the direction's scan order
    const int lane = tid & 63, wid = tid >> 6;
    __shared__ LinP wave_tot[2][T / 64], wave_run[2][T / 64];      // the look-back's and the runs' wave totals
    __shared__ double tot_s[2][T][8];                              // the published totals a wave has fetched, a row per composing thread
    __shared__ double2 msg_s[MARG ? 2 : 1][MARG ? kRunTile : 1];
    __shared__ double2 seam[2];
    // (CX_CHAIN_ONEPASS_STAMPS=1, lab: where a workgroup's time goes — the 100 MHz clock at six points, thread 0)
#define CX_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    CX_STAMP(0);
    double2 *tot_dir = totals64 + (size_t)half * kOnepassMaxTiles * kTotalPieces;
    __shared__ unsigned epoch_s;
    if (threadIdx.x == 0) epoch_s = __hip_atomic_load(&ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (needed after the first barrier: behind the links' loads)
    const unsigned long long wait_limit = wait_limit_and_fault & ~(1ull << 63);
    // 1. this thread's run, composed in order; the prefix of the runs before it within the tile stays in registers
    LinkIn in[K];
    load_run<K, T>(A, blockIdx.x, tid, dir, in);
    // what the marginal phase reads besides the messages is asked for NOW, so that nothing but stores is left behind the look-back
    const int lo = blockIdx.x * kRunTile, cnt = min(kRunTile, A.nlinks - lo);
    double2 o_sd[kOut], o_s1[kOut];
    int o_to[kOut], o_from[kOut], o_v0[kOut], o_v1[kOut], o_heads[kOut];
    auto load_out = [&](int i) {
        const int j = threadIdx.x + i * 2 * T;
        o_heads[i] = -1;
        if (j < cnt) {
            const int l = lo + j, p = chain_pos(A, l);
            o_sd[i] = A.side[p]; o_s1[i] = A.side[p + 1];
            o_to[i] = A.to_slot[l]; o_from[i] = A.from_slot[l];
            o_v0[i] = A.pos_var[p]; o_v1[i] = A.pos_var[p + 1];
            o_heads[i] = (A.head_fwd[l] ? 1 : 0) | (A.head_bwd[l] ? 2 : 0);
        }
    };
    if (MARG && !WIDE) {
#pragma unroll
        for (int i = 0; i < kOut; i++) load_out(i);
    }
    LinP t = linp_identity();
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (in[k].link < 0) continue;
        const LinP m = linp_of_link(in[k].u, in[k].q, in[k].a, in[k].b, in[k].seg);
        t = k == 0 ? m : linp_compose(t, m);
    }
    t = linp_wave_scan(t, lane);
    if (lane == 63) wave_run[half][wid] = t;
    const LinP ex = linp_wave_prev(t, lane);      // the runs before this one within the wave
    __syncthreads();
    CX_STAMP(1);
    // 2. the tile's total goes out before anything is waited for
    if (tid == 0) {
        LinP tot = wave_run[half][0];
        for (int w = 1; w < T / 64; w++) tot = linp_compose(tot, wave_run[half][w]);
        if (!((wait_limit_and_fault >> 63) && half == 0 && pos == 0)) publish_total(tot_dir + (size_t)pos * kTotalPieces, tot, epoch_tag(epoch_s));
    }
    __syncthreads();      // (nobody polls before the workgroup's own totals are on their way: early polls only crowd the memory the XCDs share)
    CX_STAMP(2);
    // 3. the carry: the totals of the tiles before this one, thread t the run [t per, (t + 1) per) of them as they are published
    const int per = (pos + T - 1) / T;
    const cx_u64 tag = epoch_tag(epoch_s);
    bool ok = true;
    if (per > 0 && wid * 64 * per < pos) {
        LinP c = linp_identity();
        for (int r = 0; r < per; r++) {      // round r: thread t's total t per + r
            if (!stage_totals(tot_dir, wid * 64 * per + r, per, pos, tag, &tot_s[half][wid * 64], lane, abort_word, wait_limit)) { ok = false; break; }
            if (tid * per + r < pos) {
                const double *v = tot_s[half][tid];
                const LinP x{v[0], v[1], v[2], v[3], v[4], v[5], v[6], (int)v[7]};
                c = r == 0 ? x : linp_compose(c, x);
            }
        }
        CX_STAMP(7);
        c = linp_wave_scan(c, lane);
        if (lane == 63) wave_tot[half][wid] = c;
    } else if (lane == 63) wave_tot[half][wid] = linp_identity();
    if (__syncthreads_or(ok ? 0 : 1)) {      // somebody gave up: say so where the host looks, store nothing (the sweep is repeated on two launches)
        if (threadIdx.x == 0) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    CX_STAMP(6);
    // the message that enters this thread's run: the empty message through the tile's carry (the look-back's wave totals), the waves
    // before this one, the runs before this one — applied to a message kept as (x, w) / d: no composition, no division
    MsgP mp{0.0, 0.0, 1.0};
#pragma unroll
    for (int w = 0; w < T / 64; w++) mp = linp_apply_p(wave_tot[half][w], mp);
    const MsgP tile_in = mp;      // (what enters the tile: the marginal phase reads it as `seam`; its division waits until the walk is on its way)
    mp = msgp_rescale(mp);        // (every few maps the triple goes back to d in [0.5, 1): a map's entries are bounded by nothing but its own D)
    for (int w = 0; w < wid; w++) mp = linp_apply_p(wave_run[half][w], mp);
    mp = linp_apply_p(ex, mp);
    CX_STAMP(3);
    // 4. the walk, as k_chain_run_apply's, with the message kept as (x, w) / d: the divisions leave the chain of dependent operations
    {
        const int sh0 = -__builtin_amdgcn_frexp_exp(mp.d);
        double mx = __builtin_ldexp(mp.x, sh0), mw = __builtin_ldexp(mp.w, sh0), md = __builtin_ldexp(mp.d, sh0);
        double ox[K], ow[K], od[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (in[k].link < 0) continue;
            if (in[k].seg) { mx = 0.0; mw = 0.0; md = 1.0; }
            const double vx = mx + in[k].u.x * md, vw = mw + in[k].u.y * md;
            double nd = in[k].a * in[k].a * md + in[k].q * vw, nx = in[k].a * vx + in[k].b * vw, nw = vw;
            const int sh = -__builtin_amdgcn_frexp_exp(nd);
            mx = ox[k] = __builtin_ldexp(nx, sh); mw = ow[k] = __builtin_ldexp(nw, sh); md = od[k] = __builtin_ldexp(nd, sh);
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int l = in[k].link;
            if (l < 0) continue;
            const double inv = 1.0 / od[k];
            const double2 r = make_double2(ox[k] * inv, ow[k] * inv);
            if (MARG) msg_s[half][l - lo] = r;
            else if (!__builtin_isnan(r.y)) out_store(&f2v[in[k].recv], r, nt);
        }
    }
    if (tid == 0) { const double inv = 1.0 / tile_in.d; seam[half] = make_double2(tile_in.x * inv, tile_in.w * inv); }
    if (MARG) {
        __syncthreads();      // (msg_s, seam)
        const double2 seam_alpha = seam[0], seam_beta = seam[1];
#pragma unroll
        for (int i = 0; i < kOut; i++) {
            const int j = threadIdx.x + i * 2 * T;
            if (WIDE) load_out(i);
            if (o_heads[i] < 0) continue;
            const bool hf = o_heads[i] & 1, hb = o_heads[i] & 2;
            const double2 sd = o_sd[i], be = msg_s[MARG ? 1 : 0][j], al1 = msg_s[0][j];
            if (store_msgs) {
                if (!__builtin_isnan(al1.y)) out_store(&f2v[o_to[i]], al1, nt);
                if (!__builtin_isnan(be.y)) out_store(&f2v[o_from[i]], be, nt);
            }
            double2 lx = sd;
            if (!hf) { const double2 al = j > 0 ? msg_s[0][j - 1] : seam_alpha; lx.x += al.x; lx.y += al.y; }
            const double2 tt = make_double2(lx.x + be.x, lx.y + be.y);
            if (marg_form == 3) { split_mean[o_v0[i]] = tt.x / tt.y; split_prec[o_v0[i]] = tt.y; }
            else out_store(&marg[o_v0[i]], marg_form == 2 ? tt : chain_to_moment(tt), nt);
            const double2 s1 = o_s1[i];
            if (hb) {
                const double2 u = make_double2(s1.x + al1.x, s1.y + al1.y);
                if (marg_form == 3) { split_mean[o_v1[i]] = u.x / u.y; split_prec[o_v1[i]] = u.y; }
                else out_store(&marg[o_v1[i]], marg_form == 2 ? u : chain_to_moment(u), nt);
            }
            if (chain_v2f) {
                if (!__builtin_isnan(lx.y)) out_store(&chain_v2f[o_from[i]], lx, nt);
                double2 rx = s1;
                if (!hb) { const double2 bn = j + 1 < cnt ? msg_s[MARG ? 1 : 0][j + 1] : seam_beta; rx.x += bn.x; rx.y += bn.y; }
                if (!__builtin_isnan(rx.y)) out_store(&chain_v2f[o_to[i]], rx, nt);
            }
        }
    }
    CX_STAMP(4);
    // The epoch moves on for the next launch (or replay of a captured graph): by workgroup 0, which is past its look-back, so every other
    // workgroup has published — with the tag of the epoch it read at its start: nobody will read the word again in this launch.
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&ctl->epoch, epoch_s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CX_STAMP(5);
#undef CX_STAMP
}

// the one-launch form is taken when it is on (CX_CHAIN_ONEPASS != 0, no wait of it ever timed out on this handle), the grid is within the
// totals' table and ALL its workgroups are resident at once on this device
template <int K, int T>
static bool onepass_ready(cx_handle *h, int ntiles, bool *wide) {
    *wide = false;
    if (h->chain_onepass_state < 0 || ntiles > kOnepassMaxTiles || ntiles < 1) return false;
    if (h->chain_onepass_state == 0) {
        h->chain_onepass_state = -1;
        if (const char *e = getenv("CX_CHAIN_ONEPASS")) if (e[0] == '0') return false;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); return false; }
        // 2 x kOnepassMaxTiles totals of 128 bytes, the control block behind them; the abort word in host memory the device can write
        void *d = nullptr;
        const size_t bytes = (size_t)2 * kOnepassMaxTiles * kTotalPieces * 16 + sizeof(OnepassCtl);
        if (hipMalloc(&d, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipMemset(d, 0, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return false; }
        void *hw = nullptr;
        if (hipHostMalloc(&hw, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return false; }
        std::memset(hw, 0, 64);
        void *dw = nullptr;
        if (hipHostGetDevicePointer(&dw, hw, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); (void)hipHostFree(hw); return false; }
        h->d_chain_onepass = d; h->chain_abort_host = (volatile unsigned *)hw; h->d_chain_abort = dw;
        h->device_bytes += (int64_t)bytes;
        h->chain_onepass_cus = cus;
        h->chain_onepass_state = 1;
    }
    // resident workgroups of each instance (the marginal form holds 32 KB of LDS more; the wide instances hold fewer registers): the
    // narrow pair when the grid fits it, the wide pair otherwise
    static int per_cu[2] = {0, 0};
    if (!per_cu[0]) {
        int a = 0, b = 0, c = 0, d = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_chain_onepass<K, T, true, false>, 2 * T, 0) != hipSuccess) { (void)hipGetLastError(); a = 0; }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_chain_onepass<K, T, false, false>, 2 * T, 0) != hipSuccess) { (void)hipGetLastError(); b = 0; }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, k_chain_onepass<K, T, true, true>, 2 * T, 0) != hipSuccess) { (void)hipGetLastError(); c = 0; }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, k_chain_onepass<K, T, false, true>, 2 * T, 0) != hipSuccess) { (void)hipGetLastError(); d = 0; }
        per_cu[0] = std::max(std::min(a, b), 0) + 1; per_cu[1] = std::max(std::min(c, d), 0) + 1;      // (+ 1: 0 means "not asked yet")
    }
    *wide = ntiles > (per_cu[0] - 1) * h->chain_onepass_cus;
    return ntiles <= (std::max(per_cu[0], per_cu[1]) - 1) * h->chain_onepass_cus;
}

template <int K, int T, bool WIDE>
static void launch_onepass(cx_handle *h, const ChainArgs &A, int ntiles, double2 *f2v, int marg_form, bool chain_v2f) {
    double2 *totals64 = (double2 *)h->d_chain_onepass;
    OnepassCtl *ctl = (OnepassCtl *)((char *)h->d_chain_onepass + (size_t)2 * kOnepassMaxTiles * kTotalPieces * 16);
    // every wait's bound in ticks of the 100 MHz clock (0.5 s; read per launch: a test shortens it) and the fault injection of that test
    // (CX_CHAIN_ONEPASS_FAULT=1: the first tile of the forward scan never publishes its total)
    const char *e = getenv("CX_CHAIN_ONEPASS_TIMEOUT_MS"), *ef = getenv("CX_CHAIN_ONEPASS_FAULT");
    const unsigned long long limit = (unsigned long long)((e ? std::max(1.0, std::min(20000.0, atof(e))) : 500.0) * 1e5) | (ef && ef[0] == '1' ? 1ull << 63 : 0ull);
    const bool store = !(h->chain_msgs_unread && marg_form == 3 && chain_v2f);
    double2 *v2f = chain_v2f ? h->d_v2f : nullptr;
    const dim3 g(ntiles), b(2 * T);
    // results stored nontemporal (C2: 18.0 -> 16.9 us: the links' parameters stay in the L2s from sweep to sweep) when the sweep's own
    // caller reads them next, not another kernel of the same iteration (CX_CHAIN_NT=0: never, lab)
    static const bool nt_on = [] { const char *v = getenv("CX_CHAIN_NT"); return !(v && v[0] == '0'); }();
    const bool nt = nt_on && (marg_form == 1 || marg_form == 2);
    static const bool want_stamps = [] { const char *v = getenv("CX_CHAIN_ONEPASS_STAMPS"); return v && v[0] == '1'; }();
    unsigned long long *stamps = nullptr;
    if (want_stamps && hipMalloc((void **)&stamps, (size_t)ntiles * 64) != hipSuccess) { (void)hipGetLastError(); stamps = nullptr; }
    if (stamps) (void)hipMemsetAsync(stamps, 0, (size_t)ntiles * 64, h->stream);
    if (marg_form) hipLaunchKernelGGL((k_chain_onepass<K, T, true, WIDE>), g, b, 0, h->stream, A, totals64, ctl, (unsigned *)h->d_chain_abort, limit, f2v, h->d_marg, marg_form, v2f, h->d_split_mean, h->d_split_prec, store, nt, stamps);
    else hipLaunchKernelGGL((k_chain_onepass<K, T, false, WIDE>), g, b, 0, h->stream, A, totals64, ctl, (unsigned *)h->d_chain_abort, limit, f2v, h->d_marg, 0, (double2 *)nullptr, (double *)nullptr, (double *)nullptr, true, nt, stamps);
    h->chain_onepass_launches++;
    if (stamps) {      // lab: per phase, the earliest, median and latest workgroup, in us after the first workgroup started
        std::vector<unsigned long long> st((size_t)ntiles * 8);
        (void)hipStreamSynchronize(h->stream);
        (void)hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(stamps);
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < ntiles; i++) t0 = std::min(t0, st[(size_t)i * 8]);
        static const char *names[8] = {"start", "runs composed", "total published", "carry composed", "results stored", "end", "all totals scanned", "wave 0 has its totals"};
        for (int ph = 0; ph < 8; ph++) {
            std::vector<double> v;
            for (int i = 0; i < ntiles; i++) if (st[(size_t)i * 8 + ph]) v.push_back((double)(st[(size_t)i * 8 + ph] - t0) * 0.01);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[onepass %d tiles] %-16s min %7.2f  median %7.2f  max %7.2f us\n", ntiles, names[ph], v.front(), v[v.size() / 2], v.back());
        }
    }
}

void chain_onepass_free(cx_handle *h) {
    if (h->d_chain_onepass) (void)hipFree(h->d_chain_onepass);
    if (h->chain_abort_host) (void)hipHostFree((void *)h->chain_abort_host);
    h->d_chain_onepass = nullptr; h->chain_abort_host = nullptr; h->d_chain_abort = nullptr;
    if (h->chain_onepass_state > 0) h->chain_onepass_state = 0;
    if (h->d_chain_linkpar) (void)hipFree(h->d_chain_linkpar);
    h->d_chain_linkpar = nullptr; h->chain_linkpar_cap = 0; h->chain_linkpar_dirty = true;
}

// the shape of a tile: K links per thread, T threads per direction (CX_CHAIN_SHAPE picks among the instances: lab switch)
constexpr int kRunTileLinks = 512;        // every instance below has K * T = 512 or a multiple of it
static int chain_shape() {
    static const int v = [] { const char *e = getenv("CX_CHAIN_SHAPE"); return e ? atoi(e) : 1; }();
    return v;
}

template <int K, int T>
static void launch_run_totals(cx_handle *h, const ChainArgs &A, int *ntiles_out) {
    const int ntiles = (A.nlinks + K * T - 1) / (K * T);
    Lin *totals = (Lin *)h->d_chain_totals, *run_excl = totals + (size_t)2 * (ntiles + 1);
    hipLaunchKernelGGL((k_chain_run_totals<K, T>), dim3(ntiles), dim3(2 * T), 0, h->stream, A, totals, run_excl);
    *ntiles_out = ntiles;
}

template <int K, int T>
static void launch_run_scan(cx_handle *h, const ChainArgs &A, double2 *f2v, int marg_form, bool chain_v2f) {
    int ntiles = (A.nlinks + K * T - 1) / (K * T);
    bool wide = false;
    if (onepass_ready<K, T>(h, ntiles, &wide)) {
        if (wide) launch_onepass<K, T, true>(h, A, ntiles, f2v, marg_form, chain_v2f);
        else launch_onepass<K, T, false>(h, A, ntiles, f2v, marg_form, chain_v2f);
        return;
    }
    launch_run_totals<K, T>(h, A, &ntiles);
    Lin *totals = (Lin *)h->d_chain_totals, *run_excl = totals + (size_t)2 * (ntiles + 1);
    const dim3 g(ntiles), b(2 * T);
    // CX_CHAIN_OWN_CARRY_TILES: the tile count up to which the apply workgroups compose their own carry (tests set it low to run the
    // one-workgroup scan of the totals on small chains)
    static const int own_carry_tiles = [] { const char *e = getenv("CX_CHAIN_OWN_CARRY_TILES"); return e ? atoi(e) : kOwnCarryTiles; }();
    const bool store = !(h->chain_msgs_unread && marg_form == 3 && chain_v2f);
    double2 *v2f = chain_v2f ? h->d_v2f : nullptr;
    if (ntiles <= own_carry_tiles) {
        if (marg_form) hipLaunchKernelGGL((k_chain_run_apply<K, T, true, true>), g, b, 0, h->stream, A, totals, run_excl, f2v, h->d_marg, marg_form, v2f, h->d_split_mean, h->d_split_prec, store);
        else hipLaunchKernelGGL((k_chain_run_apply<K, T, true, false>), g, b, 0, h->stream, A, totals, run_excl, f2v, h->d_marg, 0, (double2 *)nullptr, (double *)nullptr, (double *)nullptr, true);
    } else {
        hipLaunchKernelGGL(k_chain_scan_totals, dim3(2), dim3(kBlock), 0, h->stream, ntiles, totals);
        if (marg_form) hipLaunchKernelGGL((k_chain_run_apply<K, T, false, true>), g, b, 0, h->stream, A, totals, run_excl, f2v, h->d_marg, marg_form, v2f, h->d_split_mean, h->d_split_prec, store);
        else hipLaunchKernelGGL((k_chain_run_apply<K, T, false, false>), g, b, 0, h->stream, A, totals, run_excl, f2v, h->d_marg, 0, (double2 *)nullptr, (double *)nullptr, (double *)nullptr, true);
    }
}

static void launch_chain_side(cx_handle *h, double2 *f2v, bool fused_leaves) {
    const int npos = (int)h->chain_npos;
    const double *pa = h->any_linear ? h->d_a : nullptr, *pb = h->any_linear ? h->d_b : nullptr;
    if (fused_leaves)
        hipLaunchKernelGGL(k_chain_side<true>, dim3((npos + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, npos, h->d_chain_pos_var,
                           h->d_chain_skip0, h->d_chain_skip1, h->d_vbase, h->d_var_deg, h->d_vinfo, h->d_partner, h->d_q, h->d_q_gamma, h->d_q_gmean, pa, pb,
                           h->d_v2f, f2v, h->d_chain_side);
    else
        hipLaunchKernelGGL(k_chain_side<false>, dim3((npos + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, npos, h->d_chain_pos_var,
                           h->d_chain_skip0, h->d_chain_skip1, h->d_vbase, h->d_var_deg, h->d_vinfo, h->d_partner, h->d_q, h->d_q_gamma, h->d_q_gmean, pa, pb,
                           h->d_v2f, f2v, h->d_chain_side);
}

static ChainArgs chain_args(cx_handle *h) {
    return ChainArgs{(int)h->chain_nlinks, h->d_chain_link_pos, h->d_chain_from, h->d_chain_to, h->d_chain_head_fwd, h->d_chain_head_bwd,
                     h->d_q, h->any_linear ? h->d_a : nullptr, h->any_linear ? h->d_b : nullptr, h->d_q_gamma, h->d_q_gmean, h->d_chain_side, h->d_chain_pos_var,
                     h->chain_pos0, {LinkPar{nullptr, nullptr, nullptr, nullptr, nullptr}, LinkPar{nullptr, nullptr, nullptr, nullptr, nullptr}}};
}

// the rule parameters in link order: made when the chains are (re)built or another precision index is installed — the tables they are
// copied from do not change between those (a scalar graph's q, a, b are set at creation; q through the index is read at its source)
static bool chain_linkpar(cx_handle *h, ChainArgs &A) {
    static const bool on = [] { const char *e = getenv("CX_CHAIN_LINKPAR"); return !(e && e[0] == '0'); }();
    if (!on) return false;
    const int64_t n = h->chain_nlinks, ns = (n + 7) / 8 * 8, n2 = 2 * ns;      // (the backward half starts at a multiple of eight entries: 16-byte loads)
    if (h->chain_linkpar_dirty || !h->d_chain_linkpar || h->chain_linkpar_qg != (const void *)h->d_q_gamma) {
        if (h->chain_linkpar_cap < n) {
            if (h->d_chain_linkpar) (void)hipFree(h->d_chain_linkpar);
            h->d_chain_linkpar = nullptr; h->chain_linkpar_cap = 0;
            if (hipMalloc(&h->d_chain_linkpar, (size_t)n2 * (3 * 8 + 2 * 4)) != hipSuccess) { (void)hipGetLastError(); h->d_chain_linkpar = nullptr; return false; }
            h->chain_linkpar_cap = n;
            h->device_bytes += n2 * (3 * 8 + 2 * 4);
        }
    }
    const int64_t cs = (h->chain_linkpar_cap + 7) / 8 * 8;
    double *oq = (double *)h->d_chain_linkpar, *oa = oq + 2 * cs, *ob = oa + 2 * cs;
    int32_t *ors = (int32_t *)(ob + 2 * cs), *og = ors + 2 * cs;
    const bool lin = h->any_linear, idx = h->d_q_gamma != nullptr;
    if (h->chain_linkpar_dirty || h->chain_linkpar_qg != (const void *)h->d_q_gamma) {
        hipLaunchKernelGGL(k_chain_linkpar, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, (int)n, h->d_chain_from, h->d_chain_to, h->d_chain_head_fwd,
                           h->d_chain_head_bwd, h->d_q, lin ? h->d_a : nullptr, lin ? h->d_b : nullptr, h->d_q_gamma, (int)ns, oq, oa, ob, ors, og);
        h->chain_linkpar_dirty = false; h->chain_linkpar_qg = (const void *)h->d_q_gamma;
    }
    A.par[0] = LinkPar{oq, lin ? oa : nullptr, lin ? ob : nullptr, ors, idx ? og : nullptr};
    A.par[1] = LinkPar{oq + ns, lin ? oa + ns : nullptr, lin ? ob + ns : nullptr, ors + ns, idx ? og + ns : nullptr};
    return true;
}

// CX_CHAIN_SHAPE, measured on one box (C2 / structured family, us per sweep / iteration): 1 = <4, 256>: 21.0 / 119.7 (the default);
// 0 = <8, 128>: 23.8 / 144.8; 2 = <8, 256>: 32.7 / 135.3; 3 = <2, 256>: 24.8 / 153.3; 4 = <4, 128>: 21.8 / 148.1 (smaller tiles: more
// totals for every workgroup's own carry)
#define CX_CHAIN_SHAPES(CALL)                      \
    switch (chain_shape()) {                       \
    case 0: CALL(8, 128); break;                   \
    case 2: CALL(8, 256); break;                   \
    case 3: CALL(2, 256); break;                   \
    case 4: CALL(4, 128); break;                   \
    default: CALL(4, 256); break;                  \
    }

void launch_chain_scan(cx_handle *h, double2 *f2v, bool fused_leaves, int marg_form, bool chain_v2f) {
    // marg_form: 0 — messages only (the caller runs the variable phase); 1 / 2 — also the chain variables' marginals, moment / natural;
    // 3 — as (mean, precision) into h->d_split_mean / d_split_prec;
    // chain_v2f (with marg_form != 0): also the variable→factor messages of the chain links
    if (h->chain_nlinks == 0) return;
    // With fused leaves the side sums depend only on stored messages of fixed senders (data, priors) and on the rule
    // parameters: they are recomputed when one of those changed (cx_set_messages, cx_seed_messages, cx_update_batch, a state
    // import, a new q table), not on every sweep.
    if (!fused_leaves || h->chain_side_dirty) launch_chain_side(h, f2v, fused_leaves);
    h->chain_side_dirty = false;
    ChainArgs A = chain_args(h);
    (void)chain_linkpar(h, A);
#define CX_CALL(K, T) launch_run_scan<K, T>(h, A, f2v, marg_form, chain_v2f)
    CX_CHAIN_SHAPES(CX_CALL)
#undef CX_CALL
}

// CX_SCHED_TREE over heavy paths (cx_tree_plan.h: build_hp): the scan of ONE light depth — positions [pos_lo, pos_lo + npos) and links
// [link_lo, link_lo + nlinks) of the plan's arrays, which live in the handle's chain fields.  `skip1`: the second skipped slot of every
// position (on the way up a head also skips its slot towards its parent).  final: both directions are exact — marginals of the
// positions and the variable→factor messages of the links are written as well; otherwise messages only.
void launch_chain_scan_range(cx_handle *h, double2 *f2v, int64_t pos_lo, int64_t npos, int64_t link_lo, int64_t nlinks, const int32_t *skip1, bool final) {
    if (nlinks <= 0 || npos <= 0) return;
    // (paths through factors with more than two edges: their links' (a, b) live in the tree's own arrays on a graph that has none)
    const double *pa = h->d_tree_a ? h->d_tree_a : (h->any_linear ? h->d_a : nullptr), *pb = h->d_tree_b ? h->d_tree_b : (h->any_linear ? h->d_b : nullptr);
    hipLaunchKernelGGL(k_chain_side<false>, dim3((unsigned)((npos + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, (int)npos, h->d_chain_pos_var + pos_lo,
                       h->d_chain_skip0 + pos_lo, skip1 + pos_lo, h->d_vbase, h->d_var_deg, h->d_vinfo, h->d_partner, h->d_q, h->d_q_gamma, h->d_q_gmean, pa, pb,
                       h->d_v2f, f2v, h->d_chain_side + pos_lo);
    // the link arrays start at this depth's first link; positions stay global (link_pos, side and pos_var are indexed by them)
    const ChainArgs A{(int)nlinks, h->d_chain_link_pos + link_lo, h->d_chain_from + link_lo, h->d_chain_to + link_lo, h->d_chain_head_fwd + link_lo,
                      h->d_chain_head_bwd + link_lo, h->d_q, pa, pb, h->d_q_gamma, h->d_q_gmean, h->d_chain_side, h->d_chain_pos_var, -1, {LinkPar{nullptr, nullptr, nullptr, nullptr, nullptr}, LinkPar{nullptr, nullptr, nullptr, nullptr, nullptr}}};
    const int form = final ? (h->cfg.family == CX_FAMILY_NATURAL2 ? 2 : 1) : 0;
#define CX_CALL(K, T) launch_run_scan<K, T>(h, A, f2v, form, final)
    CX_CHAIN_SHAPES(CX_CALL)
#undef CX_CALL
}

// side sums + tile totals only (the first two stages of a sweep), for cx_chain_block_maps; totals stay un-scanned
void launch_chain_totals(cx_handle *h, double2 *f2v, bool fused_leaves, int64_t *ntiles_out) {
    launch_chain_side(h, f2v, fused_leaves);
    h->chain_side_dirty = true;     // the boundary messages will change before the sweep proper
    int ntiles = 0;
    if (h->chain_nlinks) {
        const ChainArgs A = chain_args(h);
#define CX_CALL(K, T) launch_run_totals<K, T>(h, A, &ntiles)
        CX_CHAIN_SHAPES(CX_CALL)
#undef CX_CALL
    }
    *ntiles_out = ntiles;
}

size_t chain_total_bytes(int64_t nlinks) {
    // forward and backward tile totals, then every thread's exclusive prefix within its tile (sized for the smallest tile and the
    // widest workgroup of the instances above)
    const int64_t ntiles = (nlinks + kRunTileLinks - 1) / kRunTileLinks;
    return (size_t)(2 * (ntiles + 1) + 2 * ntiles * 256) * sizeof(Lin);
}

}  // namespace cx
