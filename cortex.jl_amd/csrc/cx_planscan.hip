// cx_planscan.hip — a chain of pairs inside a reference-order plan as ONE prefix scan (CX_SCHED_REFERENCE, round 6).
//
// cx_refsched.h (level) finds the chains: pair t is MessageToFactor(x_t, f_t) — the sum of the messages into x_t from its other factors,
// among them the one pair t - 1 stored — followed by MessageToVariable(x_t+1, f_t), the rule of the pairwise factor (or the structured
// variational rule N(mean m, 1 / (var m + 1 / E[precision])), test/inference_engine_tests.jl:1004-1010, which is the additive rule with
// q = 1 / E[precision]).  With the leader's other sources summed into u_t, pair t is the map m -> rule(m + u_t): cx_lin.h's algebra, and a
// run of them is what cx_chain.hip scans — forward only here, links in plan order, every link storing BOTH its messages:
//     v2f[lead_dst] = m + u    (what the reference's compute_message_to_factor! returns: the fold over the dependencies)
//     f2v[fol_dst]  = rule(m + u)
// Two launches: k_pscan_totals gathers each link's inputs once (kept in link order for the second launch), composes a thread's run of
// K links, scans the runs of a wave and stores the prefix of every run within its tile and the tile totals; k_pscan_apply composes the
// totals of the tiles before its own (every workgroup for itself: plans have at most a few thousand tiles) and walks its runs with the
// message rule.  A leader whose variable is observed or a stand-in at run time stores nothing and sends its stored message (a point
// mass): that link is a constant map and starts the scan afresh.
#include "cx_internal.h"
#include "cx_lin.h"
#include "cx_const.h"
#include "cx_refsched.h"

namespace cx {

namespace {
constexpr int kPK = 4, kPT = 256, kPTile = kPK * kPT;

struct PLink { double2 u; double q, a, b; int flags, pad; };      // flags: 1 starts a chain, 2 constant (the leader's variable is observed: u is its stored message)
static_assert(sizeof(PLink) == 48, "PLink is 48 bytes");

struct PScanArgs {
    int nlinks;
    const int32_t *lead_dst, *lead_var, *fol_dst, *prec, *src_off, *src;      // the step's links (src_off indexes `src` absolutely)
    const uint8_t *head, *vinfo;
    const double *q, *a, *b;          // rule parameters per receiving slot (a, b may be null: additive)
    const double2 *marg;              // Gamma marginals (shape, scale) of the precision variables
    const double2 *prod;              // the product store (segment-tree nodes)
    double2 *f2v, *v2f;
    PLink *links;                     // [nlinks] scratch: the gathered inputs
    Lin *run_excl;                    // [tiles * kPT]
    Lin *totals;                      // [tiles]
};

__device__ __forceinline__ PLink gather_link(const PScanArgs &A, int l) {
    PLink r;
    r.pad = 0;
    if (l >= A.nlinks) { r.u = make_double2(0.0, 0.0); r.q = 0.0; r.a = 1.0; r.b = 0.0; r.flags = -1; return r; }
    const int fd = A.fol_dst[l], pv = A.prec[l];
    if (pv >= 0) { const double2 g = A.marg[pv]; r.q = 1.0 / (g.x * g.y); r.a = 1.0; r.b = 0.0; }
    else { r.q = A.q[fd]; r.a = A.a ? A.a[fd] : 1.0; r.b = A.b ? A.b[fd] : 0.0; }
    r.flags = A.head[l] ? 1 : 0;
    if (A.vinfo[A.lead_var[l]] & (kClamped | kGhost)) { r.flags |= 3; r.u = A.v2f[A.lead_dst[l]]; return r; }
    double2 acc = make_double2(0.0, 0.0);      // left to right, the reference's fold over the settled dependencies
    for (int j = A.src_off[l]; j < A.src_off[l + 1]; j++) {
        const int sj = A.src[j];
        const double2 v = sj >= 0 ? A.f2v[sj] : A.prod[~sj];
        acc.x += v.x; acc.y += v.y;
    }
    r.u = acc;
    return r;
}

__device__ __forceinline__ Lin map_of(const PLink &in) {
    if (in.flags & 2) { const double2 o = chain_factor_rule(in.u, in.q, in.a, in.b); return Lin{0.0, 0.0, o.x, 0.0, o.y, 0.0, 1}; }
    return lin_of_link(in.u, in.q, in.a, in.b, in.flags & 1);
}

__global__ __launch_bounds__(kPT) void k_pscan_totals(PScanArgs A) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    __shared__ Lin wave_tot[kPT / 64];
    Lin t = lin_identity();
    const int base = blockIdx.x * kPTile + tid * kPK;
#pragma unroll
    for (int k = 0; k < kPK; k++) {
        const PLink in = gather_link(A, base + k);
        if (in.flags < 0) continue;
        A.links[base + k] = in;
        const Lin m = map_of(in);
        t = k == 0 ? m : lin_compose(t, m);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Lin o = lin_shfl_up(t, d);
        if (lane >= d) t = lin_compose(o, t);
    }
    if (lane == 63) wave_tot[wid] = t;
    __syncthreads();
    Lin ex = lin_shfl_up(t, 1);
    if (lane == 0) ex = lin_identity();
    if (wid > 0) {
        Lin carry = wave_tot[0];
        for (int w = 1; w < wid; w++) carry = lin_compose(carry, wave_tot[w]);
        ex = lin_compose(carry, ex);
    }
    A.run_excl[(size_t)blockIdx.x * kPT + tid] = ex;
    if (tid == 0) {
        Lin tot = wave_tot[0];
        for (int w = 1; w < kPT / 64; w++) tot = lin_compose(tot, wave_tot[w]);
        A.totals[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(kPT) void k_pscan_apply(PScanArgs A) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, pos = blockIdx.x;
    __shared__ Lin wave_tot[kPT / 64];
    __shared__ double2 seam;
    const int base = blockIdx.x * kPTile + tid * kPK;
    PLink in[kPK];
#pragma unroll
    for (int k = 0; k < kPK; k++) { if (base + k < A.nlinks) in[k] = A.links[base + k]; else in[k].flags = -1; }
    const Lin ex = A.run_excl[(size_t)blockIdx.x * kPT + tid];
    // the totals of the tiles before this one, thread t the run [t per, (t + 1) per) of them in order
    const int per = (pos + kPT - 1) / kPT;
    if (per > 0 && wid * 64 * per < pos) {
        const int b = tid * per, e = min(b + per, pos);
        Lin c = lin_identity();
        for (int j = b; j < e; j++) c = j == b ? A.totals[j] : lin_compose(c, A.totals[j]);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            Lin o = lin_shfl_up(c, d);
            if (lane >= d) c = lin_compose(o, c);
        }
        if (lane == 63) wave_tot[wid] = c;
    } else if (lane == 63) wave_tot[wid] = lin_identity();
    __syncthreads();
    if (tid == 0) {
        Lin c = wave_tot[0];
        for (int w = 1; w < kPT / 64; w++) c = lin_compose(c, wave_tot[w]);
        seam = make_double2(c.g, c.B);      // the carry applied to the empty message (every chain starts with a head: the carry is a constant map or the identity)
    }
    __syncthreads();
    double2 m = lin_apply(ex, seam);
#pragma unroll
    for (int k = 0; k < kPK; k++) {
        if (in[k].flags < 0) continue;
        const int l = base + k;
        if (in[k].flags & 2) {           // an observed leader: its stored message through the rule, nothing of the chain before it
            m = chain_factor_rule(in[k].u, in[k].q, in[k].a, in[k].b);
        } else {
            if (in[k].flags & 1) m = make_double2(0.0, 0.0);
            const double2 v = make_double2(m.x + in[k].u.x, m.y + in[k].u.y);
            if (!__builtin_isnan(v.y)) A.v2f[A.lead_dst[l]] = v;
            const double s = 1.0 / (in[k].a * in[k].a + in[k].q * v.y);
            m = make_double2((in[k].a * v.x + in[k].b * v.y) * s, v.y * s);
        }
        if (!__builtin_isnan(m.y)) A.f2v[A.fol_dst[l]] = m;
    }
}
}  // namespace

int64_t plan_scan_scratch_bytes(int64_t nlinks) {
    const int64_t tiles = (nlinks + kPTile - 1) / kPTile;
    return ((nlinks * (int64_t)sizeof(PLink) + 63) / 64) * 64 + tiles * kPT * (int64_t)sizeof(Lin) + tiles * (int64_t)sizeof(Lin) + 256;
}
static_assert((int64_t)2048 * kPTile == refsched::kScanStepMaxLinks, "cx_refsched.h cuts scan steps at this many links");
int64_t plan_scan_max_links() { return (int64_t)2048 * kPTile; }      // every workgroup composes the totals before it: at most 2,048 tiles per step

// one scan step: links [lo, hi) of the plan's link arrays (src_off holds absolute indices into src)
void launch_plan_scan(cx_handle *h, const int32_t *lead_dst, const int32_t *lead_var, const int32_t *fol_dst, const int32_t *prec, const int32_t *src_off,
                      const int32_t *src, const uint8_t *head, int64_t lo, int64_t hi, void *scratch) {
    const int64_t n = hi - lo;
    if (n <= 0) return;
    const int64_t tiles = (n + kPTile - 1) / kPTile;
    PScanArgs A;
    A.nlinks = (int)n;
    A.lead_dst = lead_dst + lo; A.lead_var = lead_var + lo; A.fol_dst = fol_dst + lo; A.prec = prec + lo; A.src_off = src_off + lo; A.src = src; A.head = head + lo;
    A.vinfo = h->d_vinfo; A.q = h->d_q; A.a = h->any_linear ? h->d_a : nullptr; A.b = h->any_linear ? h->d_b : nullptr;
    A.marg = h->d_marg; A.prod = h->d_prod; A.f2v = h->d_f2v; A.v2f = h->d_v2f;
    A.links = (PLink *)scratch;
    A.run_excl = (Lin *)((char *)scratch + ((n * (int64_t)sizeof(PLink) + 63) / 64) * 64);
    A.totals = A.run_excl + tiles * kPT;
    hipLaunchKernelGGL(k_pscan_totals, dim3((unsigned)tiles), dim3(kPT), 0, h->stream, A);
    hipLaunchKernelGGL(k_pscan_apply, dim3((unsigned)tiles), dim3(kPT), 0, h->stream, A);
}

}  // namespace cx
