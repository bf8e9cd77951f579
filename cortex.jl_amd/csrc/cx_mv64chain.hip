// cx_mv64chain.hip — CX_SCHED_CHAIN_SCAN for dim 64: ONE cx_sweep on a state-space chain is what ONE update_marginals! of
// the reference computes there — the exact forward/backward pass (/root/reference/src/inference_engine.jl:575-608; the SSM
// of test/inference_engine_tests.jl:436-487) — with no seeding and at every time step.
//
// The plan (cx_chain64_plan.h, host, GPU-free) cuts every path into blocks and names two kinds of work:
//   compose   a wave folds its children — links of a block, or the potentials of the level below — into ONE pairwise potential
//             (P, B, C, h, c) of the segment's two end variables.  Per pair of children (M = C1 + side + P2 = U'U):
//                 Y1 = U^-T B1,  Y2 = U^-T B2',  P = P1 - Y1'Y1,  C = C2 - Y2'Y2,  B = Y2'Y1,  h = h1 + Y1'z,  c = c2 + Y2'z,  z = U^-T g
//             = 64 (Cholesky) + 2 x 160 (solves) + 2 x 160 (Grams) + 256 (product) = 960 v_mfma_f64_16x16x4_f64, all on
//             register-resident tiles in the accumulator layout of cx_mv64w_core.h.  The SAME potential serves the forward and the
//             backward pass (read from its other end it is (C, B', P, c, h)): one tree for both directions.
//   walk      a wave applies rules in sequence (rule64w_apply, the body of k_rule64w): the potentials of a group, to hand every
//             child the message that enters it, and finally the links of a level-0 block, which writes the exact
//             factor→variable messages into their slots.
// Registers: a composition keeps B1 -> Y1 -> B (128), C1 -> M -> U (80) and Y2 (128) resident and updates P1 in its output record:
// one wave per SIMD (512 registers).  The walks run two waves per SIMD like the flooding rule.
//
// Records name operands by handle (space << 56 | offset), resolved against the six base pointers of the moment (kernel arguments).
// The reference has no d-dimensional rule (parity unpinned for d > 1); pinned by the exact block-tridiagonal
// solve at every time step (tests/test_gpu_mv64_chain.py) and by the numpy execution of the same plan (tests/test_chain64_plan.py).

#include <cstdlib>

#include "cx_host.h"
#include "cx_chain64_plan.h"
#include "cx_mv64w_core.h"

namespace cx {

using namespace w64;
namespace p64 = plan64;

// The six base pointers travel as six KERNEL ARGUMENTS and a handle selects among them: pointers read from memory (a table of
// bases) are generic to hipcc and every access through them becomes a flat_load / flat_store; selected kernel arguments stay global.
#define CX_BASES_PARAMS double *__restrict__ b_zero, double *__restrict__ b_f2v, double *__restrict__ b_ptab, double *__restrict__ b_btab, double *__restrict__ b_pot, double *__restrict__ b_ent
#define CX_RESOLVE(h) resolve64(b_zero, b_f2v, b_ptab, b_btab, b_pot, b_ent, (h))
__device__ __forceinline__ double *resolve64(double *b_zero, double *b_f2v, double *b_ptab, double *b_btab, double *b_pot, double *b_ent, int64_t h) {
    const int sp = (int)(h >> 56);
    double *base = sp == p64::kF2V ? b_f2v : sp == p64::kPtab ? b_ptab : sp == p64::kBtab ? b_btab : sp == p64::kPot ? b_pot : sp == p64::kEnt ? b_ent : b_zero;
    return base + (h & p64::kOffMask);
}

// AFFINE: the steps apply composed potentials (offsets h, c); the walks along the links of a block apply plain factor rules
template <int WAVES_PER_SIMD, bool AFFINE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES_PER_SIMD, WAVES_PER_SIMD)))
void k_walk64(int njobs, const p64::Job *__restrict__ jobs, const p64::Step *__restrict__ steps, CX_BASES_PARAMS) {
    __shared__ double S[16 * kLdT];
    __shared__ double Vs[4][16 * kLdT];
    const int w = blockIdx.x;
    if (w >= njobs) return;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int first = jobs[w].first, n = jobs[w].n;
    // (the body inlined into this loop needs 184 bytes of scratch per lane at two waves per SIMD where k_rule64w needs none; a real
    // call per step — the callee with its own register allocation — needs 540, opaque lane offsets per step 372: measured, kept inline)
    for (int s = 0; s < n; s++) {
        const p64::Step *st = steps + (first + s);
        const int64_t h2 = st->src[2];
        const bool has2 = (h2 >> 56) != p64::kZero;
        if (!rule64w_apply<AFFINE>(CX_RESOLVE(st->P), CX_RESOLVE(st->Bt), CX_RESOLVE(st->C), AFFINE ? CX_RESOLVE(st->h) : nullptr,
                                   AFFINE ? CX_RESOLVE(st->c) : nullptr, CX_RESOLVE(st->src[0]), CX_RESOLVE(st->src[1]),
                                   CX_RESOLVE(h2), has2, CX_RESOLVE(st->dst), S, Vs, lane, g, c))
            return;      // undefined input or not positive definite: nothing stored, and nothing downstream of it is defined either
        // the next step reads what this one stored (same wave, same CU: its stores must have left the wave first)
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

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_compose64(int njobs, const p64::Job *__restrict__ jobs, const p64::Child *__restrict__ children, CX_BASES_PARAMS) {
    __shared__ double S[16 * kLdT];
    __shared__ double Vs[4][16 * kLdT];
    const int w = blockIdx.x;
    if (w >= njobs) return;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int first = jobs[w].first, n = jobs[w].n;
    double *out = CX_RESOLVE(jobs[w].out);

    // ---- the accumulated potential starts as the first child -------------------------------------------------------------------
    // (P1 does not stay in registers: it is only ever updated tile by tile, so it lives in the output record — 940 bytes of scratch
    // per lane with it resident)
    d4 C1[10], B1[4][4];
    double h1[4], c1[4];         // CV layout: lane (g, c) holds x[16 a + c]
    double *oP = out;
    {
        const p64::Child *ch = children + first;
        const double *P = CX_RESOLVE(ch->P), *B = CX_RESOLVE(ch->B), *C = CX_RESOLVE(ch->C);
        const double *hh = CX_RESOLVE(ch->h), *cc = CX_RESOLVE(ch->c);
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int o = tile_off(a, b, r, g, c);
                    B1[a][b][r] = B[o];
                    if (b >= a) { oP[o] = P[o]; C1[ut(a, b)][r] = C[o]; }
                }
            h1[a] = hh[16 * a + c];
            c1[a] = cc[16 * a + c];
        }
    }

    for (int k = 1; k < n; k++) {
        const p64::Child *ch = children + (first + k);
        const double *P2 = CX_RESOLVE(ch->P), *Bt2 = CX_RESOLVE(ch->Bt), *C2 = CX_RESOLVE(ch->C);
        const double *h2 = CX_RESOLVE(ch->h), *c2 = CX_RESOLVE(ch->c);
        const double *s0 = CX_RESOLVE(ch->side[0]), *s1 = CX_RESOLVE(ch->side[1]), *s2 = CX_RESOLVE(ch->side[2]);
        // ---- M = C1 + side information of the joint + P2 (upper tiles; C1's registers become M, then U) ------------------------------
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = a; b < 4; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int o = tile_off(a, b, r, g, c);
                    C1[ut(a, b)][r] += (P2[o] + s0[kD + o]) + (s1[kD + o] + s2[kD + o]);
                }
        double gv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int e = 16 * j + c; gv[j] = c1[j] + (s0[e] + s1[e]) + (s2[e] + h2[e]); }
        // ---- blocked upper Cholesky (as in rule64w_apply) ---------------------------------------------------------------------------
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const d4 Vk = diag_factor(C1[ut(kk, kk)], S, g, c);
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
        // ---- z = U^-T g on the vector pipe ----------------------------------------------------------------------------------------------
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
        }
        // ---- Y1 = U^-T B1 in place;  P1 -= Y1'Y1;  h1 += Y1'z --------------------------------------------------------------------------
#pragma unroll
        for (int b = 0; b < 4; b++) solve_col(B1, b, C1, Vs, g, c);
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int b = a; b < 4; b++) {
                d4 G = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 4; j++) G = tts(B1[j][a], B1[j][b], G);
                // same lane, same address as the store that wrote it: in order behind it
#pragma unroll
                for (int r = 0; r < 4; r++) { const int o = tile_off(a, b, r, g, c); oP[o] = oP[o] - G[r]; }
            }
            double p = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) p += B1[j][a][r] * zrv[j][r];
            h1[a] += sum_groups(p);
        }
        // ---- Y2 = U^-T B2';  C1 = C2 - Y2'Y2;  c1 = c2 + Y2'z -------------------------------------------------------------------------
        d4 Y2[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) Y2[j][b][r] = Bt2[tile_off(j, b, r, g, c)];
            solve_col(Y2, b, C1, Vs, g, c);
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int b = a; b < 4; b++) {
                d4 G = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 4; j++) G = tts(Y2[j][a], Y2[j][b], G);
#pragma unroll
                for (int r = 0; r < 4; r++) C1[ut(a, b)][r] = C2[tile_off(a, b, r, g, c)] - G[r];      // (U is dead: every solve is done)
            }
            double p = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) p += Y2[j][a][r] * zrv[j][r];
            c1[a] = c2[16 * a + c] + sum_groups(p);
        }
        // ---- B = Y2'Y1, one block column of Y1 at a time, in place ------------------------------------------------------------------------
#pragma unroll
        for (int b = 0; b < 4; b++) {
            d4 T[4];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                T[a] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 4; j++) T[a] = tts(Y2[j][a], B1[j][b], T[a]);
            }
#pragma unroll
            for (int a = 0; a < 4; a++) B1[a][b] = T[a];
        }
    }

    // ---- the potential record: P | B | B' | C | h | c (P, C: upper tiles only — every reader takes the upper tiles) ------------------------
    double *oB = out + kD * kD, *oBt = out + 2 * kD * kD, *oC = out + 3 * kD * kD, *oh = out + 4 * kD * kD, *oc = oh + kD;
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = tile_off(a, b, r, g, c);
                oB[o] = B1[a][b][r];
                if (b >= a) oC[o] = C1[ut(a, b)][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; r++) S[(g + 4 * r) * kLdT + c] = B1[a][b][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; r++) oBt[tile_off(b, a, r, g, c)] = S[c * kLdT + g + 4 * r];
        }
        if (g == 0) { oh[16 * a + c] = h1[a]; oc[16 * a + c] = c1[a]; }
    }
}

// ---- host: the plan on the device -----------------------------------------------------------------------------------------------
struct Chain64 {
    p64::Job *d_jobs = nullptr;
    p64::Child *d_children = nullptr;
    p64::Step *d_steps = nullptr;
    double *d_pot = nullptr, *d_ent = nullptr;
    struct Launch { int kind; int64_t first; int n; };      // kind 0: compose, 1: walk over potentials, 2: walk along links
    std::vector<Launch> launches;
    int64_t n_pot = 0, n_ent = 0, n_compositions = 0, n_rules = 0;
    int K0 = 0, fan = 0, levels = 0;
    int64_t bytes = 0;
};

void chain64_free(cx_handle *h) {
    Chain64 *c = (Chain64 *)h->chain64;
    if (!c) return;
    for (void *p : {(void *)c->d_jobs, (void *)c->d_children, (void *)c->d_steps, (void *)c->d_pot, (void *)c->d_ent}) if (p) (void)hipFree(p);
    h->device_bytes -= c->bytes;
    delete c;
    h->chain64 = nullptr;
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}

// Build the plan from the chain decomposition (host arrays of build_chains) and upload it.
int32_t chain64_build(cx_handle *h, const std::vector<int32_t> &pos_var, const std::vector<int32_t> &skip0, const std::vector<int32_t> &skip1,
                      const std::vector<int32_t> &link_pos, const std::vector<int32_t> &from, const std::vector<int32_t> &to,
                      const std::vector<uint8_t> &head_fwd, const std::vector<uint8_t> &head_bwd, const std::vector<int32_t> &tab_fwd,
                      const std::vector<int32_t> &tab_bwd) {
    using cxh::fail;
    chain64_free(h);
    const int64_t npos = (int64_t)pos_var.size(), nlinks = (int64_t)link_pos.size();
    std::vector<int32_t> side((size_t)3 * npos, -1);
    for (int64_t p = 0; p < npos; p++) {
        const int32_t v = pos_var[p], deg = h->var_off[v + 1] - h->var_off[v];
        int n = 0;
        for (int32_t j = 0; j < deg; j++) {
            const int32_t sj = h->vbase[v] + j * kBlock;
            if (sj == skip0[p] || sj == skip1[p]) continue;
            if (n == 3) return fail(h, CX_ERR_UNSUPPORTED, "chain-scan schedule, dim 64: variable " + std::to_string(h->var_ids[v]) + " has more than three inputs besides its chain links");
            side[3 * p + n++] = sj;
        }
    }
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, h->cfg.device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    p64::Input in;
    in.d = 64; in.npos = npos; in.nlinks = nlinks;
    in.link_pos = link_pos.data(); in.from = from.data(); in.to = to.data(); in.tab_fwd = tab_fwd.data(); in.tab_bwd = tab_bwd.data();
    in.head_fwd = head_fwd.data(); in.head_bwd = head_bwd.data(); in.side = side.data();
    in.K0 = env_int("CX_MVC64_K", 0);            // links per level-0 block (default: one block per SIMD)
    in.fan = std::max(2, env_int("CX_MVC64_FAN", 4));
    in.lanes = 4 * (int64_t)ncu;                 // a composition is one wave per SIMD
    p64::Plan plan;
    try { plan = p64::build(in); }
    catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "chain-scan schedule, dim 64: host allocation failed"); }
    catch (const std::exception &e) { return fail(h, CX_ERR_UNSUPPORTED, std::string("chain-scan schedule, dim 64: ") + e.what()); }
    Chain64 *c = new (std::nothrow) Chain64();
    if (!c) return fail(h, CX_ERR_OUT_OF_MEMORY, "chain-scan schedule, dim 64: host allocation failed");
    h->chain64 = c;
    c->n_pot = plan.n_pot; c->n_ent = plan.n_ent; c->K0 = plan.K0; c->fan = plan.fan; c->levels = plan.levels;
    c->n_compositions = plan.n_compositions; c->n_rules = plan.n_rules;
    std::vector<p64::Job> jobs;
    for (const auto &L : plan.compose_launches) if (!L.empty()) { c->launches.push_back({0, (int64_t)jobs.size(), (int)L.size()}); jobs.insert(jobs.end(), L.begin(), L.end()); }
    for (size_t i = 0; i < plan.walk_launches.size(); i++) {      // the last walk launch is the one along the links (plain rules, h = c = 0)
        const auto &L = plan.walk_launches[i];
        if (L.empty()) continue;
        c->launches.push_back({i + 1 == plan.walk_launches.size() ? 2 : 1, (int64_t)jobs.size(), (int)L.size()});
        jobs.insert(jobs.end(), L.begin(), L.end());
    }
    const int64_t before = h->device_bytes;
    int32_t rc;
    if ((rc = cxh::dev_upload(h, &c->d_jobs, jobs)) != CX_OK) return rc;
    if ((rc = cxh::dev_upload(h, &c->d_children, plan.children)) != CX_OK) return rc;
    if ((rc = cxh::dev_upload(h, &c->d_steps, plan.steps)) != CX_OK) return rc;
    if ((rc = cxh::dev_alloc(h, &c->d_pot, plan.n_pot * plan.pot)) != CX_OK) return rc;
    if ((rc = cxh::dev_alloc(h, &c->d_ent, plan.n_ent * plan.msg)) != CX_OK) return rc;
    // a potential or entry message that was never computed reads as UndefValue()
    CX_HIP(h, hipMemsetAsync(c->d_pot, 0xff, (size_t)std::max<int64_t>(1, plan.n_pot * plan.pot) * 8, h->stream));
    CX_HIP(h, hipMemsetAsync(c->d_ent, 0xff, (size_t)std::max<int64_t>(1, plan.n_ent * plan.msg) * 8, h->stream));
    c->bytes = h->device_bytes - before;
    return CX_OK;
}

// one exact sweep: every launch of the plan, in order, on the handle's stream
int32_t chain64_sweep(cx_handle *h) {
    Chain64 *c = (Chain64 *)h->chain64;
    if (!c) return cxh::fail(h, CX_ERR_STATE, "chain-scan schedule, dim 64: no plan");
    double *b0 = h->d_zero_msg, *b1 = h->d_mv_f2v, *b2 = h->d_ptab, *b3 = h->d_ptab_bt, *b4 = c->d_pot, *b5 = c->d_ent;
    static const int walk_waves = env_int("CX_MVC64_WALK_WAVES", 2);
    for (const auto &L : c->launches) {
        if (L.kind == 0)
            hipLaunchKernelGGL(k_compose64, dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_children, b0, b1, b2, b3, b4, b5);
        else if (L.kind == 1)      // few jobs, long dependent chains: a wave alone on its SIMD
            hipLaunchKernelGGL((k_walk64<1, true>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps, b0, b1, b2, b3, b4, b5);
        else if (walk_waves == 1)
            hipLaunchKernelGGL((k_walk64<1, false>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps, b0, b1, b2, b3, b4, b5);
        else
            hipLaunchKernelGGL((k_walk64<2, false>), dim3(L.n), dim3(64), 0, h->stream, L.n, c->d_jobs + L.first, c->d_steps, b0, b1, b2, b3, b4, b5);
    }
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

void chain64_stats(const cx_handle *h, int64_t *out8) {
    const Chain64 *c = (const Chain64 *)h->chain64;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    if (!c) return;
    out8[0] = c->K0; out8[1] = c->fan; out8[2] = c->levels; out8[3] = c->n_pot; out8[4] = c->n_compositions; out8[5] = c->n_rules;
    out8[6] = (int64_t)c->launches.size(); out8[7] = c->bytes;
}

}  // namespace cx
