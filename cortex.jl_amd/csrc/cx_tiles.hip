// cx_tiles.hip — TWO sweeps per launch: temporal blocking of the fused scalar sweep in LDS (gfx950).
//
// Why.  One fused sweep (cx_kernels.hip, k_sweep) streams every factor→variable message, every index and every rule
// parameter through HBM once: 427 MB per sweep on the 10M-edge grid, 63 us at the memory system's limit.  The only way past
// that limit is to move fewer bytes per sweep.  Here a workgroup owns a TILE of <= 256 variables (a compact cluster of the
// graph), loads the messages of the tile and of its two-hop neighbourhood once, runs the sweep t -> t+1 for the tile and its
// one-hop ring and the sweep t+1 -> t+2 for the tile itself out of LDS, and stores only the tile's own messages of time t+2
// (pull form: unit-stride runs, no scatter).  Redundant arithmetic in the rings buys one HBM pass per TWO sweeps, and one
// launch boundary instead of two (what a 1/8 strip of the grid, 8 us of kernel per sweep, is bound by).
//
// Semantics.  Exactly two sweeps of k_sweep<.., PUSH>: the same leave-one-out sums in the same order, the same rule arithmetic
// — results are bit-identical to two single-sweep launches (tests/test_gpu_tiled.py), so the deep-halo partition argument
// and every parity statement carry over.  What one process! computes in the reference (src/inference_engine.jl:479-509 with
// the rules of test/inference_engine_tests.jl:385-432) is unchanged; only where the intermediate time step lives is.
//
// Layout.  The tiles are an overlay on the SELL-256 slot layout (no renumbering): a tile names its variables through a
// per-tile table of (first slot, local variable number, LDS offset, degree); messages of variable v sit at slot
// vbase[v] + 256 k as before, so a run of variables that are neighbours in id order (a grid row inside the tile) is still
// read as one contiguous segment per message index.  Tiles come from recursive bisection of the variable graph along
// breadth-first levels from a pseudo-peripheral vertex (host, once): on a 2-D grid that yields compact polygons whose two-hop
// neighbourhood is ~1.5x the tile.
//
// Buffers.  The launch reads the input buffer (time t) and must not write into it (other tiles read their rings from it), so
// it writes time t+2 into the OTHER Jacobi buffer; the buffer of time t+1 never exists in HBM.  The host tracks that the
// retained buffer is two steps behind (cx_handle::alt_two_back) and regenerates time t+1 with one plain sweep when somebody
// asks for variable→factor messages or a checkpoint.

#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "cx_internal.h"
#include "cx_tiling.h"

namespace cx {

constexpr int kTileLocals = kBlock;   // a tile and its two rings hold at most one local variable per thread

struct TileRec {        // one per local variable of a tile (8 bytes); tables have kTileLocals entries per tile
    int32_t base;       // slot of the variable's first message
    uint16_t loff;      // first packed LDS slot of the variable's messages
    uint8_t deg;        // degree 0..8
    uint8_t cls;        // 0: owned by the tile, 1: ring 1, 2: ring 2, 3: unused entry
};

typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_stream(const double2 *p) {
    const d2v v = __builtin_nontemporal_load((const d2v *)p);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ double2 t_add2(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }

// the rule of cx_kernels.hip (factor_rule), receiving-edge parameters
template <bool LINEAR>
__device__ __forceinline__ double2 t_rule(double2 m, double q, double a, double b) {
    double2 o;
    if (m.y == __builtin_inf()) {
        double mean = LINEAR ? (a * m.x + b) : m.x;
        o.y = 1.0 / q;
        o.x = mean * o.y;
    } else {
        double s = 1.0 / ((LINEAR ? a * a : 1.0) + q * m.y);
        o.y = m.y * s;
        o.x = (LINEAR ? (a * m.x + b * m.y) : m.x) * s;
    }
    return o;
}

// variable phase for one local variable: leave-one-out sums of its incoming messages, in k_sweep's order
// (out[k] = (in[0] + … + in[k-1]) + (in[deg-1] + … + in[k+1])), written to `dst` unless the variable is fixed; returns the total.
// MAXD bounds the unrolled register arrays (the graph's largest degree); terms beyond a variable's degree are exact zeros, so
// the sums do not depend on MAXD.
template <int MAXD>
__device__ __forceinline__ double2 tile_var_phase(const double2 *__restrict__ src, double2 *__restrict__ dst, int loff, int deg, bool fixed) {
    double2 in[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; k++) in[k] = (k < deg) ? src[loff + k] : make_double2(0.0, 0.0);
    double2 out[MAXD];
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < MAXD; k++) { out[k] = acc; acc = t_add2(acc, in[k]); }
    const double2 total = acc;
    acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = MAXD - 1; k >= 0; k--) { out[k] = t_add2(out[k], acc); acc = t_add2(acc, in[k]); }
    if (!fixed) {
#pragma unroll
        for (int k = 0; k < MAXD; k++)
            if (k < deg) dst[loff + k] = out[k];
    }
    return total;
}

// workgroup barrier that orders LDS traffic only: `__syncthreads()` would also wait for every outstanding global load, i.e.
// for the NEXT tile's prefetch (cdna_hip_programming.md, "Pipelining across barriers")
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One thread = one local variable of the tile (own, ring 1 or ring 2) through every phase: its table entry, its partner slots
// and its rule parameters live in registers, LDS holds only the two message arrays.  Workgroups are persistent: while a tile
// is being computed out of LDS, the messages / parameters of the workgroup's NEXT tile are already in flight into registers
// and the table entry of the one after that is being fetched, so the two dependent memory round trips of a tile (table entry,
// then the slots it names) hide behind arithmetic instead of adding up per tile.
template <bool LINEAR, int MAXD>
__global__ __launch_bounds__(kBlock) void k_sweep2(int ntiles, const TileRec *__restrict__ recs, const int32_t *__restrict__ tvar,
                                                   const uint8_t *__restrict__ tinfo, const uint16_t *__restrict__ pl, int pl_stride,
                                                   const double *__restrict__ q, const double *__restrict__ pa,
                                                   const double *__restrict__ pb, const double2 *__restrict__ f2v_in,
                                                   double2 *__restrict__ f2v_out, const double2 *__restrict__ v2f,
                                                   double2 *__restrict__ marg, int write_marg, int max_slots) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *A = (double2 *)smem;                       // messages into the variables: time t, then (own + ring 1) time t+1
    double2 *V = A + max_slots;                         // variable→factor messages of the current half step
    const int tid = threadIdx.x;
    const int stride = gridDim.x;
    // a workgroup walks a contiguous run of tiles (neighbouring tiles share ring lines: keep them on one XCD's L2 in time)
    const int per = (ntiles + stride - 1) / stride;
    int tile = blockIdx.x * per;
    const int tile_end = min(ntiles, tile + per);
    if (tile >= tile_end) return;

    auto load_rec = [&](int t, TileRec &r, int &info) {
        r.base = 0; r.loff = 0; r.deg = 0; r.cls = 3; info = 0;
        if (t < tile_end) { r = recs[(int64_t)t * kTileLocals + tid]; info = tinfo[(int64_t)t * kTileLocals + tid]; }
    };
    // ---- prologue: table entries of the first two tiles, data of the first -------------------------------------------------
    TileRec rec_n, rec_nn;
    int info_n, info_nn;
    load_rec(tile, rec_n, info_n);
    load_rec(tile + 1, rec_nn, info_nn);
    double2 in_n[MAXD], fx_n[MAXD];
    double q_n[MAXD];
    unsigned short p_n[MAXD];
    int var_n = 0;
    auto issue = [&](int t, const TileRec &r, int info) {
        const bool fixed = (r.deg < 2) || (info & (kClamped | kGhost));
#pragma unroll
        for (int k = 0; k < MAXD; k++) {
            in_n[k] = make_double2(0.0, 0.0); fx_n[k] = make_double2(0.0, 0.0); q_n[k] = 0.0; p_n[k] = 0xffff;
            if (k < r.deg) {
                const int slot = r.base + k * kBlock;
                in_n[k] = ld_stream(&f2v_in[slot]);
                // fewer than two factors, observed, or a stand-in of another rank's variable: the stored message feeds the factor
                if (fixed) fx_n[k] = v2f[slot];
                if (r.cls <= 1) { p_n[k] = pl[(int64_t)t * pl_stride + r.loff + k]; q_n[k] = q[slot]; }
            }
        }
        var_n = (r.cls == 0) ? tvar[(int64_t)t * kTileLocals + tid] : 0;
    };
    issue(tile, rec_n, info_n);

    for (; tile < tile_end; tile++) {
        // ---- the prefetched tile becomes the current one: registers -> LDS ---------------------------------------------------
        const int base = rec_n.base, loff = rec_n.loff, deg = rec_n.deg, cls = rec_n.cls, var = var_n;
        const bool fixed = (deg < 2) || (info_n & (kClamped | kGhost));
        double qc[MAXD];
        unsigned short pc[MAXD];
        lds_barrier();                                   // every thread is done with the previous tile's A and V
#pragma unroll
        for (int k = 0; k < MAXD; k++) {
            qc[k] = q_n[k]; pc[k] = p_n[k];
            if (k < deg) { A[loff + k] = in_n[k]; if (fixed) V[loff + k] = fx_n[k]; }
        }
        // ---- prefetch: data of the next tile (its table entry arrived during the previous iteration), entry of the one after
        rec_n = rec_nn; info_n = info_nn;
        issue(tile + 1, rec_n, info_n);
        load_rec(tile + 2, rec_nn, info_nn);
        lds_barrier();

        // ---- half step 1: variable phase over own + ring 1 + ring 2 (time t) ------------------------------------------------
        if (cls <= 2 && !fixed) (void)tile_var_phase<MAXD>(A, V, loff, deg, false);
        lds_barrier();
        // ---- factor phase: messages of time t+1 into own + ring 1 (in place over A) -----------------------------------------
        if (cls <= 1) {
#pragma unroll
            for (int k = 0; k < MAXD; k++) {
                if (k >= deg) continue;
                const unsigned p = pc[k];
                if (p == 0xffffu) continue;              // unary factor: the message is the caller's (a prior), it stays
                const double2 m = V[p];
                if (__builtin_isnan(m.y)) continue;      // a dependency is undefined: the signal is not pending
                const int slot = base + k * kBlock;
                A[loff + k] = t_rule<LINEAR>(m, qc[k], LINEAR ? pa[slot] : 1.0, LINEAR ? pb[slot] : 0.0);
            }
        }
        lds_barrier();
        // ---- half step 2: variable phase over own + ring 1 (time t+1); the marginals of the own variables ------------------
        if (cls <= 1 && (!fixed || (cls == 0 && write_marg))) {
            const double2 total = tile_var_phase<MAXD>(A, V, loff, deg, fixed);
            if (write_marg && cls == 0) {
                d2v t;
                if (write_marg == 2) { t.x = total.x; t.y = total.y; }
                else { const double vv = 1.0 / total.y; t.x = total.x * vv; t.y = vv; }
                __builtin_nontemporal_store(t, (d2v *)&marg[var]);
            }
        }
        lds_barrier();
        // ---- factor phase: messages of time t+2 into the own variables, straight to the other buffer ------------------------
        if (cls == 0) {
#pragma unroll
            for (int k = 0; k < MAXD; k++) {
                if (k >= deg) continue;
                const unsigned p = pc[k];
                if (p == 0xffffu) continue;
                const int slot = base + k * kBlock;
                const double2 m = V[p];
                // undefined dependency: the slot keeps the value it had (A[..] is then still the message of time t, which is
                // what the single-sweep schedule leaves in this buffer)
                f2v_out[slot] = __builtin_isnan(m.y) ? A[loff + k]
                                                     : t_rule<LINEAR>(m, qc[k], LINEAR ? pa[slot] : 1.0, LINEAR ? pb[slot] : 0.0);
            }
        }
    }
}

// observed / stand-in flags per table entry, refreshed whenever the variable flags changed (data injected, halo configured)
__global__ void k_tile_info(int64_t n, const int32_t *__restrict__ tvar, const uint8_t *__restrict__ vinfo, uint8_t *__restrict__ tinfo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = tvar[i]; tinfo[i] = v >= 0 ? vinfo[v] : 0; }
}

// ------------------------------------------------------------------------------------------------ host: tiles (clustering in cx_tiling.h)

// Build (once per graph) the tile tables of the two-sweep kernel.  Returns false (with h->tiles_state = -1) when the graph is
// outside what the kernel handles: variables of degree > 8.
bool tiles_build(cx_handle *h, std::string &why) {
    if (h->tiles_state != 0) return h->tiles_state > 0;
    h->tiles_state = -1;
    if (!h->big_vars.empty()) { why = "variables of degree > 8"; return false; }
    const int32_t nv = (int32_t)h->nv;
    if (nv == 0) { why = "empty graph"; return false; }
    // variable adjacency through the 2-edge factors (partner slots)
    std::vector<int32_t> slot_var(h->nslots, -1);
    for (int64_t e = 0; e < h->ne; e++) slot_var[slot_of_edge(h, e)] = h->edge_var[e];
    std::vector<int32_t> adj_off(nv + 1, 0), adj;
    adj.reserve(h->ne);
    int max_deg = 0;
    for (int32_t v = 0; v < nv; v++) {
        for (int32_t e = h->var_off[v]; e < h->var_off[v + 1]; e++) {
            const int32_t p = h->partner[slot_of_edge(h, e)];
            if (p >= 0) adj.push_back(slot_var[p]);
        }
        adj_off[v + 1] = (int32_t)adj.size();
        max_deg = std::max(max_deg, h->var_off[v + 1] - h->var_off[v]);
    }
    int cap = 160;   // owned variables per tile: with its two rings a tile must fit one local variable per thread
    if (const char *s = std::getenv("CX_TILE_VARS")) { const int c = std::atoi(s); if (c >= 8 && c <= kTileLocals) cap = c; }
    std::vector<int32_t> order, piece_end;
    bisect(adj_off, adj, nv, cap, order, piece_end);
    // work list of pieces; a piece whose two-hop neighbourhood exceeds the workgroup (ragged or thin clusters) is cut again
    std::vector<std::vector<int32_t>> work;
    {
        int32_t lo = 0;
        for (int32_t hi : piece_end) { work.emplace_back(order.begin() + lo, order.begin() + hi); lo = hi; }
        std::reverse(work.begin(), work.end());      // popped from the back: keeps the spatial order of the bisection
    }
    std::vector<TileRec> recs;
    std::vector<int32_t> tvar;
    std::vector<std::vector<uint16_t>> pls;
    std::vector<int32_t> local_of(nv, -1), mark(nv, -1), scratch(nv, -1), ring1, ring2;
    int max_slots = 0, max_pl = 0;
    int32_t stamp = 0;
    int64_t own_total = 0, all_total = 0;
    auto deg_of = [&](int32_t v) { return h->var_off[v + 1] - h->var_off[v]; };
    while (!work.empty()) {
        std::vector<int32_t> members = std::move(work.back());
        work.pop_back();
        const int32_t t = stamp++;
        ring1.clear(); ring2.clear();
        for (int32_t v : members) mark[v] = t;
        for (int32_t v : members)
            for (int32_t e = adj_off[v]; e < adj_off[v + 1]; e++) { const int32_t w = adj[e]; if (mark[w] != t) { mark[w] = t; ring1.push_back(w); } }
        for (int32_t v : ring1)
            for (int32_t e = adj_off[v]; e < adj_off[v + 1]; e++) { const int32_t w = adj[e]; if (mark[w] != t) { mark[w] = t; ring2.push_back(w); } }
        const size_t n_all = members.size() + ring1.size() + ring2.size();
        if (n_all > (size_t)kTileLocals) {
            if (members.size() == 1) { why = "a single variable's two-hop neighbourhood exceeds a workgroup"; return false; }
            const int32_t nleft = split_piece(adj_off, adj, members, scratch);
            work.emplace_back(members.begin() + nleft, members.end());
            work.emplace_back(members.begin(), members.begin() + nleft);
            continue;
        }
        std::sort(members.begin(), members.end());       // id order inside the tile: runs of neighbours in id order are contiguous slots
        std::sort(ring1.begin(), ring1.end());
        std::sort(ring2.begin(), ring2.end());
        const size_t rec0 = recs.size();
        int32_t off = 0, l = 0;
        auto add = [&](int32_t v, int cls) {
            TileRec r{};
            r.base = h->vbase[v]; r.loff = (uint16_t)off; r.deg = (uint8_t)deg_of(v); r.cls = (uint8_t)cls;
            recs.push_back(r); tvar.push_back(v);
            local_of[v] = l++;
            off += r.deg;
        };
        for (int32_t v : members) add(v, 0);
        for (int32_t v : ring1) add(v, 1);
        const int32_t slots_ring1 = off, n_ring1 = l;
        for (int32_t v : ring2) add(v, 2);
        // partner table of the own + ring-1 slots
        std::vector<uint16_t> pl;
        pl.reserve(slots_ring1);
        for (int32_t i = 0; i < n_ring1; i++) {
            const TileRec &r = recs[rec0 + i];
            for (int32_t k = 0; k < r.deg; k++) {
                const int32_t p = h->partner[r.base + k * kBlock];
                if (p < 0) { pl.push_back(0xffff); continue; }
                const int32_t w = slot_var[p];
                const TileRec &rw = recs[rec0 + local_of[w]];
                pl.push_back((uint16_t)(rw.loff + (p - rw.base) / kBlock));
            }
        }
        for (; l < kTileLocals; l++) { TileRec r{}; r.cls = 3; recs.push_back(r); tvar.push_back(-1); }
        max_slots = std::max(max_slots, off);
        max_pl = std::max(max_pl, slots_ring1);
        own_total += (int64_t)members.size(); all_total += (int64_t)n_all;
        pls.push_back(std::move(pl));
    }
    const int32_t ntiles = (int32_t)pls.size();
    const int pl_stride = (max_pl + 7) / 8 * 8;
    std::vector<uint16_t> plflat((size_t)ntiles * pl_stride, 0xffff);
    for (int32_t t = 0; t < ntiles; t++) std::copy(pls[t].begin(), pls[t].end(), plflat.begin() + (size_t)t * pl_stride);
    const size_t lds = (size_t)2 * max_slots * 16;
    void *d_recs = nullptr, *d_pl = nullptr, *d_var = nullptr, *d_info = nullptr;
    bool ok = hipMalloc(&d_recs, recs.size() * sizeof(TileRec)) == hipSuccess && hipMalloc(&d_pl, std::max<size_t>(plflat.size(), 1) * 2) == hipSuccess &&
              hipMalloc(&d_var, tvar.size() * 4) == hipSuccess && hipMalloc(&d_info, tvar.size()) == hipSuccess;
    if (!ok) {
        for (void *p : {d_recs, d_pl, d_var, d_info}) if (p) (void)hipFree(p);
        why = "device allocation of the tile tables failed";
        return false;
    }
    (void)hipMemcpy(d_recs, recs.data(), recs.size() * sizeof(TileRec), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_var, tvar.data(), tvar.size() * 4, hipMemcpyHostToDevice);
    if (!plflat.empty()) (void)hipMemcpy(d_pl, plflat.data(), plflat.size() * 2, hipMemcpyHostToDevice);
    h->d_tile_recs = d_recs; h->d_tile_pl = d_pl; h->d_tile_var = d_var; h->d_tile_info = d_info;
    h->n_tiles = ntiles; h->n_tile_recs = (int64_t)tvar.size(); h->tile_max_slots = max_slots; h->tile_lds = (int64_t)lds;
    h->tile_pl_stride = pl_stride; h->tile_max_deg = max_deg;
    h->device_bytes += (int64_t)(recs.size() * sizeof(TileRec) + plflat.size() * 2 + tvar.size() * 5);
    h->tile_redundancy = own_total ? (double)all_total / (double)own_total : 0.0;
    h->tile_info_dirty = true;
    h->tiles_state = 1;
    return true;
}

void tiles_free(cx_handle *h) {
    for (void *p : {h->d_tile_hdr, h->d_tile_recs, h->d_tile_pl, h->d_tile_var, h->d_tile_info}) if (p) (void)hipFree(p);
    h->d_tile_hdr = h->d_tile_recs = h->d_tile_pl = h->d_tile_var = h->d_tile_info = nullptr;
    h->n_tiles = 0; h->tiles_state = 0;
}

template <bool LINEAR, int MAXD>
static void launch_sweep2_t(cx_handle *h, const double2 *f2v_in, double2 *f2v_out, int wm, int grid) {
    hipLaunchKernelGGL((k_sweep2<LINEAR, MAXD>), dim3((unsigned)grid), dim3(kBlock), (size_t)h->tile_lds, h->stream, h->n_tiles,
                       (const TileRec *)h->d_tile_recs, (const int32_t *)h->d_tile_var, (const uint8_t *)h->d_tile_info,
                       (const uint16_t *)h->d_tile_pl, h->tile_pl_stride, h->d_q, LINEAR ? h->d_a : (const double *)nullptr,
                       LINEAR ? h->d_b : (const double *)nullptr, f2v_in, f2v_out, h->d_v2f, h->d_marg, wm, h->tile_max_slots);
}

// two sweeps: f2v_in holds time t, f2v_out receives time t+2 (its other slots are untouched)
void launch_tiled2(cx_handle *h, const double2 *f2v_in, double2 *f2v_out, bool write_marg) {
    if (h->n_tiles == 0) return;
    if (h->tile_info_dirty) {
        hipLaunchKernelGGL(k_tile_info, dim3((unsigned)((h->n_tile_recs + 255) / 256)), dim3(256), 0, h->stream, h->n_tile_recs,
                           (const int32_t *)h->d_tile_var, h->d_vinfo, (uint8_t *)h->d_tile_info);
        h->tile_info_dirty = false;
    }
    const int wm = write_marg ? (h->cfg.family == CX_FAMILY_NATURAL2 ? 2 : 1) : 0;
    if (h->profiling && (h->prof_count[CX_KERNEL_TILED]++ % h->prof_stride) == 0) {
        ProfileRec r; r.kernel = CX_KERNEL_TILED;
        (void)hipEventCreate(&r.start); (void)hipEventCreate(&r.stop);
        (void)hipEventRecord(r.start, h->stream);
        h->recs.push_back(r);
        h->prof_armed = true;
    } else h->prof_armed = false;
    // persistent workgroups: as many as stay resident (LDS and registers), each walking a contiguous run of tiles
    const int grid = std::min<int>(h->n_tiles, h->tile_grid);
    const bool six = h->tile_max_deg <= 6;
    if (h->any_linear) { if (six) launch_sweep2_t<true, 6>(h, f2v_in, f2v_out, wm, grid); else launch_sweep2_t<true, 8>(h, f2v_in, f2v_out, wm, grid); }
    else { if (six) launch_sweep2_t<false, 6>(h, f2v_in, f2v_out, wm, grid); else launch_sweep2_t<false, 8>(h, f2v_in, f2v_out, wm, grid); }
    if (h->profiling && h->prof_armed) (void)hipEventRecord(h->recs.back().stop, h->stream);
}

bool tiles_prepare_kernel(cx_handle *h) {
    const int bytes = (int)h->tile_lds;
    const void *fn[4] = {(const void *)k_sweep2<false, 6>, (const void *)k_sweep2<false, 8>, (const void *)k_sweep2<true, 6>, (const void *)k_sweep2<true, 8>};
    for (const void *f : fn)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;   // > 64 KB needs the opt-in
    const void *mine = h->any_linear ? (h->tile_max_deg <= 6 ? fn[2] : fn[3]) : (h->tile_max_deg <= 6 ? fn[0] : fn[1]);
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mine, kBlock, (size_t)bytes) != hipSuccess || per_cu < 1) per_cu = 1;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (const char *s = std::getenv("CX_TILE_WGS_PER_CU")) { const int c = std::atoi(s); if (c >= 1 && c <= 8) per_cu = c; }
    h->tile_grid = std::max(1, per_cu * cus);
    return true;
}

}  // namespace cx
