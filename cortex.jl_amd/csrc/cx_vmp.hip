// cx_vmp.hip — variational message passing with weak dependencies.
//
// The reference runs its second algorithm family through the same engine by wiring messages to depend WEAKLY on
// marginals (custom resolvers of test/inference_engine_tests.jl:599-621 "Mean Field" and :810-897 "Structured") and by
// letting the user rule compute expectations (:633-689, :905-1030).  This file is that family for the model class of
// those tests, on the device:
//
//     factor  CX_FACTOR_NORMAL_PRECISION(out, mean, precision):   out ~ N(mean, 1 / precision)
//     Normal variables (latent or observed) on the OUT / IN roles, Gamma variables on the PRECISION role.
//
// One cx_update_marginals(ids) call is one `update_marginals!(engine, ids)`: the messages into the requested variables
// are computed from the marginals as they stand before the call and the marginals are stored afterwards
// (src/inference_engine.jl:576-628: message rounds first, marginals in the final round).
//
//   CX_FAMILY_VMP_MEAN_FIELD  every variable on its own:
//        f -> Normal v :  N(E[other Normal], E[precision])                                      (:654-664)
//        f -> Gamma  g :  Gamma(3/2, 2 / (var out + var mean + (E out - E mean)^2))             (:666-684)
//   CX_FAMILY_VMP_STRUCTURED  the Normal variables jointly (belief propagation through the factors with the precision
//        replaced by its expectation, :1004-1010, run by an inner scalar handle with the configured schedule), the Gamma
//        variables from the joint marginal of the factor's two Normal variables (:939-967, :1011-1016).
//   marginal = product of the incoming messages (no prior factor, as in the reference's model): Normal in natural
//   parameters, Gamma(shape, scale) as shape = 1 + deg / 2, 1 / scale = sum of the rates.
//
// Layout: marginals as SoA arrays; factors sorted by their Gamma variable so that a Gamma marginal is a deterministic
// two-level reduction (fixed 4096-factor chunks, then one workgroup per Gamma variable) — no atomics, run-to-run identical.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "cx_host.h"

namespace {

constexpr int kChunk = 1024;      // factors per first-level reduction workgroup: four per thread of k_rate_sum, all their loads in flight together
constexpr double kInfD = std::numeric_limits<double>::infinity();

struct Vmp {
    bool structured = false;
    int64_t nN = 0, nG = 0, nF = 0, nChunks = 0;
    // host tables
    std::vector<int64_t> var_ids;           // ascending: all variables
    std::vector<int32_t> var_kind;          // 0 Normal, 1 Gamma
    std::vector<int32_t> var_local;         // index among the Normal / Gamma variables
    std::vector<uint8_t> n_observed;        // per Normal variable
    std::vector<int64_t> normal_ids;        // id of Normal variable k
    std::vector<int64_t> fac_ids;           // per factor (ascending)
    std::vector<int32_t> f_out, f_mean, f_gamma;
    std::vector<int32_t> g_deg, g_chunk_off;    // per Gamma variable: degree, first chunk
    std::vector<uint8_t> g_needs_chain;         // has a factor whose two Normal variables are both latent (set at update time)
    std::vector<int32_t> nb_off, nb_fac;        // CSR: factors of every Normal variable, ascending factor id
    int64_t n_latent = 0;
    bool chain_ready = false;                   // structured: the inner handle has swept at least once
    std::vector<uint8_t> g_fresh;               // structured: the precision variable was updated since the states last were (what a request that
                                                // names states and precisions together depends on: vmp_update_marginals)
    // device
    double *n_mean = nullptr, *n_prec = nullptr, *n_mean_alt = nullptr, *n_prec_alt = nullptr;
    uint8_t *d_observed = nullptr, *d_mask = nullptr;
    double *g_shape = nullptr, *g_scale = nullptr, *g_mean = nullptr, *g_new = nullptr;   // g_new: [2][nG] staged (shape, scale)
    int32_t *d_f_out = nullptr, *d_f_mean = nullptr, *d_f_gamma = nullptr, *d_f_order = nullptr;      // (order: the factors sorted by precision variable)
    int32_t *d_nb_off = nullptr, *d_nb_other = nullptr, *d_nb_gamma = nullptr;            // mean field: CSR per Normal variable
    int32_t *d_chunk_begin = nullptr, *d_chunk_end = nullptr, *d_g_chunk_off = nullptr, *d_g_deg = nullptr;
    int32_t *d_req = nullptr;                                                               // requested Gamma indices
    std::vector<int32_t> req_on_device;                                                     // ... as last uploaded
    double *d_partial = nullptr;
    // structured: inner scalar handle over the Normal variables
    cx_handle *chain = nullptr;
    int32_t *d_slot_out = nullptr, *d_slot_mean = nullptr;   // slot (inner handle) of the factor's OUT / IN edge
    int32_t *d_slot_gamma = nullptr;                         // per inner slot: Gamma variable of its factor, -1 for padding
    std::vector<void *> owned;
    // staging for set / get marginals (grown on demand, freed with the rest)
    int32_t *st_idx = nullptr; double *st_a = nullptr, *st_b = nullptr; int64_t st_cap = 0;
};

int32_t vfail(cx_handle *h, int32_t code, const std::string &msg) { h->err = msg; return code; }

#define VMP_HIP(h, call)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return vfail(h, e_ == hipErrorOutOfMemory ? CX_ERR_OUT_OF_MEMORY : CX_ERR_DEVICE, \
                                           std::string(#call) + ": " + hipGetErrorString(e_));              \
    } while (0)
#define VMP_REQUIRE(h, cond, code, msg) do { if (!(cond)) return vfail(h, code, msg); } while (0)

template <class T>
int32_t up(cx_handle *h, Vmp *s, T **p, const std::vector<T> &v) {
    const size_t n = std::max<size_t>(v.size(), 1);
    VMP_HIP(h, hipMalloc((void **)p, n * sizeof(T)));
    s->owned.push_back(*p);
    h->device_bytes += (int64_t)(n * sizeof(T));
    if (!v.empty()) VMP_HIP(h, hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return CX_OK;
}
template <class T>
int32_t alloc(cx_handle *h, Vmp *s, T **p, int64_t n) {
    n = std::max<int64_t>(n, 1);
    VMP_HIP(h, hipMalloc((void **)p, (size_t)n * sizeof(T)));
    s->owned.push_back(*p);
    h->device_bytes += n * (int64_t)sizeof(T);
    return CX_OK;
}


// device staging of n indices + 2n doubles, reused across calls
int32_t stage(cx_handle *h, Vmp *s, int64_t n) {
    if (n <= s->st_cap) return CX_OK;
    VMP_HIP(h, hipStreamSynchronize(h->stream));
    for (void *p : {(void *)s->st_idx, (void *)s->st_a, (void *)s->st_b}) if (p) (void)hipFree(p);
    s->st_idx = nullptr; s->st_a = s->st_b = nullptr; s->st_cap = 0;
    const int64_t cap = std::max<int64_t>(n, 4096);
    VMP_HIP(h, hipMalloc((void **)&s->st_idx, (size_t)cap * 4));
    VMP_HIP(h, hipMalloc((void **)&s->st_a, (size_t)cap * 16));
    VMP_HIP(h, hipMalloc((void **)&s->st_b, (size_t)cap * 8));
    s->st_cap = cap;
    return CX_OK;
}

// ---- kernels ----------------------------------------------------------------------------------------------------------

// mean field: marginal of every requested latent Normal variable = product of N(E[other], E[precision]) over its factors
__global__ __launch_bounds__(256) void k_mf_normal(int n, const int32_t *__restrict__ nb_off, const int32_t *__restrict__ nb_other,
                                                   const int32_t *__restrict__ nb_gamma, const double *__restrict__ mean_in,
                                                   const double *__restrict__ prec_in, const uint8_t *__restrict__ observed,
                                                   const uint8_t *__restrict__ mask, const double *__restrict__ g_mean,
                                                   double *__restrict__ mean_out, double *__restrict__ prec_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double m = mean_in[i], w = prec_in[i];
    if (!observed[i] && (!mask || mask[i])) {
        double xi = 0.0, ws = 0.0;
        for (int e = nb_off[i]; e < nb_off[i + 1]; e++) {
            const double tau = g_mean[nb_gamma[e]];
            xi += tau * mean_in[nb_other[e]];
            ws += tau;
        }
        if (nb_off[i + 1] > nb_off[i]) { w = ws; m = xi / ws; }
    }
    mean_out[i] = m;
    prec_out[i] = w;
}

// rate = 1 / scale of the Gamma(3/2, .) message of a factor towards its precision variable
template <bool STRUCTURED>
__device__ __forceinline__ double rate_of(int f, const int32_t *__restrict__ f_out, const int32_t *__restrict__ f_mean, const int32_t *__restrict__ f_gamma,
                                          const double *__restrict__ n_mean, const double *__restrict__ n_prec, const uint8_t *__restrict__ observed,
                                          const double *__restrict__ g_mean, const int32_t *__restrict__ slot_out, const int32_t *__restrict__ slot_mean,
                                          const double2 *__restrict__ v2f) {
    const int a = f_out[f], b = f_mean[f];
    double spread;
    if (STRUCTURED && !observed[a] && !observed[b]) {
        // joint marginal of the two latent variables from the messages they send INTO the factor and E[precision]:
        // W = [w1 + t, -t; -t, w2 + t],  mu = W^-1 [xi1; xi2]   (test/inference_engine_tests.jl:958-966)
        const double2 m1 = v2f[slot_out[f]], m2 = v2f[slot_mean[f]];   // natural form (xi, w)
        const double t = g_mean[f_gamma[f]];
        const double p = m1.y + t, q = m2.y + t;
        const double det = p * q - t * t;
        const double v11 = q / det, v22 = p / det, v12 = t / det;
        const double mu1 = v11 * m1.x + v12 * m2.x, mu2 = v12 * m1.x + v22 * m2.x;
        const double dm = mu1 - mu2;
        spread = v11 - v12 - v12 + v22 + dm * dm;                       // :1011-1016
    } else {
        const double va = observed[a] ? 0.0 : 1.0 / n_prec[a], vb = observed[b] ? 0.0 : 1.0 / n_prec[b];
        const double dm = n_mean[a] - n_mean[b];
        spread = va + vb + dm * dm;                                     // :666-684, :990-995
    }
    return 0.5 * spread;     // Gamma(3/2, 2 / spread): rate = spread / 2
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) for (int k = 0; k < (int)(blockDim.x >> 6); k++) r += sh[k];
    __syncthreads();
    return r;   // valid in thread 0
}

// The rates of a chunk of factors — consecutive in the order sorted by precision variable, never spanning two of them — and their sum, in
// one kernel: the rates themselves are nobody's to read (until round 5 they were stored, 8 B a factor, and summed by a kernel of their own:
// 7 us and 32 MB of an iteration at n = 1e6).  A thread takes the chunk's factors t, t + 256, t + 512, t + 768 and adds them in that order,
// the workgroup folds its threads in a fixed tree: the sum does not depend on the launch.
// Measured and NOT kept: the workgroup that finishes last (a counter behind __threadfence()) also finishing the precision variables, to
// save k_gamma_finish's launch (4.6 us) — every workgroup's agent-scope release writes back and invalidates its XCD's L2, and this kernel
// went from 16.2 to 72 us (13.2 to 54 for the mean-field family): profiles/r05_vmp_rocprof.md holds the kept form.
template <bool STRUCTURED>
__global__ __launch_bounds__(256) void k_rate_sum(const int32_t *__restrict__ chunk_begin, const int32_t *__restrict__ chunk_end, const int32_t *__restrict__ order,
                                                  const int32_t *__restrict__ f_out, const int32_t *__restrict__ f_mean, const int32_t *__restrict__ f_gamma,
                                                  const double *__restrict__ n_mean, const double *__restrict__ n_prec, const uint8_t *__restrict__ observed,
                                                  const double *__restrict__ g_mean, const int32_t *__restrict__ slot_out, const int32_t *__restrict__ slot_mean,
                                                  const double2 *__restrict__ v2f, double *__restrict__ partial) {
    __shared__ double sh[4];
    const int c = blockIdx.x, lo = chunk_begin[c], hi = chunk_end[c];
    double r[kChunk / 256];
#pragma unroll
    for (int u = 0; u < kChunk / 256; u++) {
        const int i = lo + threadIdx.x + 256 * u;
        r[u] = i < hi ? rate_of<STRUCTURED>(order[i], f_out, f_mean, f_gamma, n_mean, n_prec, observed, g_mean, slot_out, slot_mean, v2f) : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < kChunk / 256; u++) acc += r[u];
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[c] = s;
}

// one workgroup per requested Gamma variable: sum its chunk partials and store (shape, scale, mean).  Every reader of the OLD
// means in this round (k_mf_normal, k_rate_sum) was launched before this kernel on the same stream, so the Jacobi round needs no
// staging copy and no commit launch of its own.
__global__ __launch_bounds__(256) void k_gamma_finish(const int32_t *__restrict__ req, const int32_t *__restrict__ g_chunk_off,
                                                      const int32_t *__restrict__ g_deg, const double *__restrict__ partial,
                                                      double *__restrict__ shape, double *__restrict__ scale, double *__restrict__ mean) {
    __shared__ double sh[4];
    const int g = req[blockIdx.x];
    double acc = 0.0;
    for (int c = g_chunk_off[g] + threadIdx.x; c < g_chunk_off[g + 1]; c += blockDim.x) acc += partial[c];
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        const double a = 1.0 + 0.5 * (double)g_deg[g];     // product of deg Gamma(3/2, .): shape 3/2 deg - (deg - 1)
        const double th = 1.0 / s;
        shape[g] = a; scale[g] = th; mean[g] = a * th;
    }
}

// structured: the inner handle's factor variance q = 1 / E[precision], for both slots of every factor
__global__ __launch_bounds__(256) void k_set_q(int nslots, const int32_t *__restrict__ slot_gamma, const double *__restrict__ g_mean,
                                               double *__restrict__ q) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const int g = slot_gamma[s];
    if (g >= 0) q[s] = 1.0 / g_mean[g];
}

// structured: marginals of the requested latent variables from the inner handle's (mean, variance)
__global__ __launch_bounds__(256) void k_pull_marginals(int n, const double2 *__restrict__ marg, const uint8_t *__restrict__ observed,
                                                        double *__restrict__ mean, double *__restrict__ prec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || observed[i]) return;
    const double2 mv = marg[i];
    mean[i] = mv.x;
    prec[i] = 1.0 / mv.y;
}

__global__ void k_scatter2(int n, const int32_t *__restrict__ idx, const double *__restrict__ a, const double *__restrict__ b,
                           double *__restrict__ da, double *__restrict__ db, double *__restrict__ dc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    da[idx[i]] = a[i]; db[idx[i]] = b[i];
    if (dc) dc[idx[i]] = a[i] * b[i];
}

__global__ void k_gather2(int n, const int32_t *__restrict__ idx, const double *__restrict__ a, const double *__restrict__ b,
                          double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[2 * i] = a[idx[i]]; out[2 * i + 1] = b[idx[i]];
}

inline int blocks(int64_t n) { return (int)((n + 255) / 256); }

int64_t find_var(const Vmp *s, int64_t id) {
    auto it = std::lower_bound(s->var_ids.begin(), s->var_ids.end(), id);
    return (it == s->var_ids.end() || *it != id) ? -1 : it - s->var_ids.begin();
}

}  // namespace

namespace cx {

void vmp_free(cx_handle *h) {
    Vmp *s = (Vmp *)h->vmp;
    if (!s) return;
    for (void *p : s->owned) (void)hipFree(p);
    for (void *p : {(void *)s->st_idx, (void *)s->st_a, (void *)s->st_b}) if (p) (void)hipFree(p);
    if (s->chain) (void)cx_destroy(s->chain);
    delete s;
    h->vmp = nullptr;
}

int32_t vmp_set_stream(cx_handle *h) {
    Vmp *s = (Vmp *)h->vmp;
    if (s && s->chain) return cx_set_stream(s->chain, (void *)h->stream);
    return CX_OK;
}

int32_t vmp_graph_create(cx_handle *h, int64_t ne, const int64_t *edge_var, const int64_t *edge_fac, const int32_t *edge_role,
                         int64_t nf, const int64_t *factor_ids, const int32_t *factor_kind) {
    VMP_REQUIRE(h, edge_role, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: the variational families need edge_role (OUT / IN / PRECISION)");
    Vmp *s = new (std::nothrow) Vmp();
    VMP_REQUIRE(h, s, CX_ERR_OUT_OF_MEMORY, "cx_graph_create: host allocation failed");
    h->vmp = s;
    s->structured = h->cfg.family == CX_FAMILY_VMP_STRUCTURED;
    try {
        // factors (ascending id) and their three edges
        std::vector<int64_t> ford(nf);
        std::iota(ford.begin(), ford.end(), 0);
        std::sort(ford.begin(), ford.end(), [&](int64_t a, int64_t b) { return factor_ids[a] < factor_ids[b]; });
        s->fac_ids.resize(nf);
        for (int64_t i = 0; i < nf; i++) {
            s->fac_ids[i] = factor_ids[ford[i]];
            VMP_REQUIRE(h, i == 0 || s->fac_ids[i] != s->fac_ids[i - 1], CX_ERR_INVALID_ARGUMENT, "cx_graph_create: duplicate factor id");
            VMP_REQUIRE(h, factor_kind[ford[i]] == CX_FACTOR_NORMAL_PRECISION, CX_ERR_UNSUPPORTED,
                        "cx_graph_create: the variational families take CX_FACTOR_NORMAL_PRECISION factors only");
        }
        s->nF = nf;
        // variables
        s->var_ids.assign(edge_var, edge_var + ne);
        std::sort(s->var_ids.begin(), s->var_ids.end());
        s->var_ids.erase(std::unique(s->var_ids.begin(), s->var_ids.end()), s->var_ids.end());
        const int64_t nv = (int64_t)s->var_ids.size();
        s->var_kind.assign(nv, -1);
        std::vector<int64_t> e_out(nf, -1), e_mean(nf, -1), e_prec(nf, -1);
        for (int64_t e = 0; e < ne; e++) {
            VMP_REQUIRE(h, edge_var[e] >= 1, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: ids are 1-based");
            auto it = std::lower_bound(s->fac_ids.begin(), s->fac_ids.end(), edge_fac[e]);
            VMP_REQUIRE(h, it != s->fac_ids.end() && *it == edge_fac[e], CX_ERR_NOT_FOUND,
                        "cx_graph_create: edge names unknown factor id " + std::to_string(edge_fac[e]));
            const int64_t f = it - s->fac_ids.begin(), v = find_var(s, edge_var[e]);
            const int32_t role = edge_role[e];
            VMP_REQUIRE(h, role == CX_ROLE_OUT || role == CX_ROLE_IN || role == CX_ROLE_PRECISION, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: bad edge role");
            int64_t &dst = role == CX_ROLE_OUT ? e_out[f] : role == CX_ROLE_IN ? e_mean[f] : e_prec[f];
            VMP_REQUIRE(h, dst < 0, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: factor " + std::to_string(edge_fac[e]) + " has two edges with the same role");
            dst = v;
            const int32_t kind = role == CX_ROLE_PRECISION ? 1 : 0;
            VMP_REQUIRE(h, s->var_kind[v] < 0 || s->var_kind[v] == kind, CX_ERR_INVALID_ARGUMENT,
                        "cx_graph_create: variable " + std::to_string(edge_var[e]) + " is used both as a precision and as a Normal variable");
            s->var_kind[v] = kind;
        }
        for (int64_t f = 0; f < nf; f++)
            VMP_REQUIRE(h, e_out[f] >= 0 && e_mean[f] >= 0 && e_prec[f] >= 0 && e_out[f] != e_mean[f], CX_ERR_INVALID_ARGUMENT,
                        "cx_graph_create: factor " + std::to_string(s->fac_ids[f]) + " needs one OUT, one IN and one PRECISION edge");
        s->var_local.assign(nv, 0);
        for (int64_t v = 0; v < nv; v++) {
            if (s->var_kind[v] == 0) { s->var_local[v] = (int32_t)s->nN++; s->normal_ids.push_back(s->var_ids[v]); }
            else s->var_local[v] = (int32_t)s->nG++;
        }
        VMP_REQUIRE(h, s->nG > 0 && s->nN > 0, CX_ERR_INVALID_ARGUMENT, "cx_graph_create: no precision variable");
        s->n_observed.assign(s->nN, 0);
        s->f_out.resize(nf); s->f_mean.resize(nf); s->f_gamma.resize(nf);
        for (int64_t f = 0; f < nf; f++) {
            s->f_out[f] = s->var_local[e_out[f]]; s->f_mean[f] = s->var_local[e_mean[f]]; s->f_gamma[f] = s->var_local[e_prec[f]];
        }
        // factors sorted by Gamma variable (stable: ascending factor id inside), cut into chunks that never span two variables
        std::vector<int32_t> order(nf);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return s->f_gamma[a] < s->f_gamma[b]; });
        s->g_deg.assign(s->nG, 0);
        for (int64_t f = 0; f < nf; f++) s->g_deg[s->f_gamma[f]]++;
        std::vector<int32_t> chunk_begin, chunk_end;
        s->g_chunk_off.assign(s->nG + 1, 0);
        int64_t start = 0;
        for (int64_t g = 0; g < s->nG; g++) {
            s->g_chunk_off[g] = (int32_t)chunk_begin.size();
            for (int64_t b = start; b < start + s->g_deg[g]; b += kChunk) {
                chunk_begin.push_back((int32_t)b);
                chunk_end.push_back((int32_t)std::min<int64_t>(b + kChunk, start + s->g_deg[g]));
            }
            start += s->g_deg[g];
        }
        s->g_chunk_off[s->nG] = (int32_t)chunk_begin.size();
        s->nChunks = (int64_t)chunk_begin.size();
        s->g_needs_chain.assign(s->nG, 0);
        s->g_fresh.assign(s->nG, 0);

        VMP_HIP(h, hipSetDevice(h->cfg.device));
        int32_t rc;
#define TRY(x) do { rc = (x); if (rc != CX_OK) return rc; } while (0)
        TRY(up(h, s, &s->d_f_out, s->f_out)); TRY(up(h, s, &s->d_f_mean, s->f_mean)); TRY(up(h, s, &s->d_f_gamma, s->f_gamma));
        TRY(up(h, s, &s->d_f_order, order));
        TRY(up(h, s, &s->d_chunk_begin, chunk_begin)); TRY(up(h, s, &s->d_chunk_end, chunk_end));
        TRY(up(h, s, &s->d_g_chunk_off, s->g_chunk_off)); TRY(up(h, s, &s->d_g_deg, s->g_deg));
        TRY(alloc(h, s, &s->d_partial, s->nChunks));
        TRY(alloc(h, s, &s->n_mean, s->nN)); TRY(alloc(h, s, &s->n_prec, s->nN));
        TRY(alloc(h, s, &s->n_mean_alt, s->nN)); TRY(alloc(h, s, &s->n_prec_alt, s->nN));
        TRY(alloc(h, s, &s->d_observed, s->nN)); TRY(alloc(h, s, &s->d_mask, s->nN));
        TRY(alloc(h, s, &s->g_shape, s->nG)); TRY(alloc(h, s, &s->g_scale, s->nG)); TRY(alloc(h, s, &s->g_mean, s->nG));
        TRY(alloc(h, s, &s->g_new, 2 * s->nG)); TRY(alloc(h, s, &s->d_req, s->nG));
        // every marginal starts as UndefValue(): NaN
        VMP_HIP(h, hipMemsetAsync(s->n_mean, 0xff, (size_t)s->nN * 8, h->stream));
        VMP_HIP(h, hipMemsetAsync(s->n_prec, 0xff, (size_t)s->nN * 8, h->stream));
        VMP_HIP(h, hipMemsetAsync(s->g_shape, 0xff, (size_t)s->nG * 8, h->stream));
        VMP_HIP(h, hipMemsetAsync(s->g_scale, 0xff, (size_t)s->nG * 8, h->stream));
        VMP_HIP(h, hipMemsetAsync(s->g_mean, 0xff, (size_t)s->nG * 8, h->stream));
        VMP_HIP(h, hipMemsetAsync(s->d_observed, 0, (size_t)s->nN, h->stream));

        // CSR per Normal variable over its factors, ascending factor id (the order of the reference's product fold)
        s->nb_off.assign(s->nN + 1, 0);
        for (int64_t f = 0; f < nf; f++) { s->nb_off[s->f_out[f] + 1]++; s->nb_off[s->f_mean[f] + 1]++; }
        for (int64_t i = 0; i < s->nN; i++) s->nb_off[i + 1] += s->nb_off[i];
        s->nb_fac.resize(s->nb_off[s->nN]);
        {
            std::vector<int32_t> fill(s->nb_off.begin(), s->nb_off.end() - 1);
            for (int64_t f = 0; f < nf; f++) { s->nb_fac[fill[s->f_out[f]]++] = (int32_t)f; s->nb_fac[fill[s->f_mean[f]]++] = (int32_t)f; }
        }
        s->n_latent = s->nN;
        h->n_messages_per_sweep = 3 * nf;
        if (!s->structured) {
            std::vector<int32_t> other(s->nb_fac.size()), gam(s->nb_fac.size());
            for (int64_t i = 0; i < s->nN; i++)
                for (int32_t e = s->nb_off[i]; e < s->nb_off[i + 1]; e++) {
                    const int32_t f = s->nb_fac[e];
                    other[e] = s->f_out[f] == i ? s->f_mean[f] : s->f_out[f];
                    gam[e] = s->f_gamma[f];
                }
            TRY(up(h, s, &s->d_nb_off, s->nb_off)); TRY(up(h, s, &s->d_nb_other, other)); TRY(up(h, s, &s->d_nb_gamma, gam));
        } else {
            // inner scalar handle over the Normal variables: the same factors as additive-noise links x_out = x_mean + N(0, q)
            cx_config cc = h->cfg;
            // k_rate reads the two variable→factor messages of every latent-latent factor from the inner handle: with the chain
            // scan the scan's own kernel stores exactly those (and the marginals), so no variable phase runs per iteration;
            // any other schedule materialises every variable→factor message
            const bool scan = cc.schedule == CX_SCHED_CHAIN_SCAN;
            cc.family = CX_FAMILY_GAUSSIAN; cc.dim = 1; cc.compute_marginals_in_sweep = 1; cc.materialize_messages_to_factor = scan ? 0 : 1;
            rc = cx_create(&cc, &s->chain);
            VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_graph_create (inner handle): ") + cx_last_error(nullptr));
            s->chain->chain_v2f_from_scan = scan;
            std::vector<int64_t> cev(2 * nf), cef(2 * nf);
            std::vector<int32_t> ckind(nf, CX_FACTOR_GAUSS_ADDITIVE);
            std::vector<double> cpar((size_t)nf * CX_NPARAM, 0.0);
            for (int64_t f = 0; f < nf; f++) {
                cev[2 * f] = s->normal_ids[s->f_out[f]]; cev[2 * f + 1] = s->normal_ids[s->f_mean[f]];
                cef[2 * f] = cef[2 * f + 1] = s->fac_ids[f];
                cpar[(size_t)f * CX_NPARAM] = 1.0;
            }
            (void)cx_set_stream(s->chain, (void *)h->stream);
            rc = cx_graph_create(s->chain, 2 * nf, cev.data(), cef.data(), nullptr, nf, s->fac_ids.data(), ckind.data(), cpar.data());
            VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_graph_create (inner handle): ") + cx_last_error(s->chain));
            VMP_REQUIRE(h, s->chain->nv == s->nN, CX_ERR_STATE, "cx_graph_create: inner handle lost a variable");
            std::vector<int64_t> eidx(2 * nf);
            rc = cx_edge_index(s->chain, 2 * nf, cev.data(), cef.data(), eidx.data());
            VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_graph_create (inner handle): ") + cx_last_error(s->chain));
            std::vector<int32_t> so(nf), sm(nf), sg(s->chain->nslots, -1);
            for (int64_t f = 0; f < nf; f++) {
                so[f] = cx::slot_of_edge(s->chain, eidx[2 * f]); sm[f] = cx::slot_of_edge(s->chain, eidx[2 * f + 1]);
                sg[so[f]] = sg[sm[f]] = s->f_gamma[f];
            }
            TRY(up(h, s, &s->d_slot_out, so)); TRY(up(h, s, &s->d_slot_mean, sm)); TRY(up(h, s, &s->d_slot_gamma, sg));
        }
#undef TRY
        VMP_HIP(h, hipStreamSynchronize(h->stream));
        h->nv = nv; h->nf = nf; h->ne = ne;
        h->has_graph = true;
        return CX_OK;
    } catch (const std::bad_alloc &) {
        return vfail(h, CX_ERR_OUT_OF_MEMORY, "cx_graph_create: host allocation failed");
    }
}

int32_t vmp_set_marginals(cx_handle *h, int64_t n, const int64_t *ids, int32_t form, const double *payload) {
    Vmp *s = (Vmp *)h->vmp;
    VMP_REQUIRE(h, form == CX_FORM_POINT || form == CX_FORM_MEAN_PRECISION || form == CX_FORM_GAMMA || form == CX_FORM_MOMENT,
                CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: form must be POINT, MOMENT, MEAN_PRECISION or GAMMA");
    try {
        const int64_t per = form == CX_FORM_POINT ? 1 : 2;
        std::vector<int32_t> idx(n);
        std::vector<double> a(n), b(n);
        std::vector<int64_t> obs_ids;
        std::vector<double> obs_y;
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = find_var(s, ids[i]);
            VMP_REQUIRE(h, v >= 0, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(ids[i]));
            const bool is_gamma = s->var_kind[v] == 1;
            VMP_REQUIRE(h, is_gamma == (form == CX_FORM_GAMMA), CX_ERR_INVALID_ARGUMENT,
                        "cx_set_marginals: variable " + std::to_string(ids[i]) + (is_gamma ? " is a precision variable: use CX_FORM_GAMMA" : " is a Normal variable"));
            idx[i] = s->var_local[v];
            const double p0 = payload[per * i], p1 = per == 2 ? payload[per * i + 1] : 0.0;
            if (form == CX_FORM_GAMMA) {
                VMP_REQUIRE(h, p0 > 0 && p1 > 0 && std::isfinite(p0) && std::isfinite(p1), CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: Gamma needs shape > 0 and scale > 0");
                a[i] = p0; b[i] = p1;
            } else if (form == CX_FORM_POINT) {
                VMP_REQUIRE(h, std::isfinite(p0), CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: non-finite datum");
                a[i] = p0; b[i] = kInfD;
                if (!s->n_observed[idx[i]]) {
                    s->n_observed[idx[i]] = 1;
                    s->n_latent--;
                    h->n_messages_per_sweep -= s->nb_off[idx[i] + 1] - s->nb_off[idx[i]];   // no message is computed towards data
                }
                obs_ids.push_back(ids[i]); obs_y.push_back(p0);
            } else {
                VMP_REQUIRE(h, !s->n_observed[idx[i]], CX_ERR_STATE, "cx_set_marginals: variable " + std::to_string(ids[i]) + " is observed");
                const double w = form == CX_FORM_MOMENT ? 1.0 / p1 : p1;
                VMP_REQUIRE(h, std::isfinite(p0) && w > 0 && std::isfinite(w), CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: Normal needs a finite mean and a positive finite precision");
                a[i] = p0; b[i] = w;
            }
        }
        VMP_HIP(h, hipSetDevice(h->cfg.device));
        { int32_t rc = stage(h, s, n); if (rc != CX_OK) return rc; }
        int32_t *d_idx = s->st_idx; double *d_a = s->st_a, *d_b = s->st_b;
        VMP_HIP(h, hipMemcpyAsync(d_idx, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        VMP_HIP(h, hipMemcpyAsync(d_a, a.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
        VMP_HIP(h, hipMemcpyAsync(d_b, b.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
        if (form == CX_FORM_GAMMA) hipLaunchKernelGGL(k_scatter2, dim3(blocks(n)), dim3(256), 0, h->stream, (int)n, d_idx, d_a, d_b, s->g_shape, s->g_scale, s->g_mean);
        else hipLaunchKernelGGL(k_scatter2, dim3(blocks(n)), dim3(256), 0, h->stream, (int)n, d_idx, d_a, d_b, s->n_mean, s->n_prec, (double *)nullptr);
        if (form == CX_FORM_POINT)
            VMP_HIP(h, hipMemcpyAsync(s->d_observed, s->n_observed.data(), (size_t)s->nN, hipMemcpyHostToDevice, h->stream));
        VMP_HIP(h, hipStreamSynchronize(h->stream));     // the host vectors above go out of scope
        if (s->structured && !obs_ids.empty()) {
            // an observed variable is data for the inner handle: its messages into every factor are the datum
            std::vector<int64_t> ev, ef;
            std::vector<double> ey;
            for (size_t k = 0; k < obs_ids.size(); k++) {
                const int32_t ni = s->var_local[find_var(s, obs_ids[k])];
                for (int32_t e = s->nb_off[ni]; e < s->nb_off[ni + 1]; e++) { ev.push_back(obs_ids[k]); ef.push_back(s->fac_ids[s->nb_fac[e]]); ey.push_back(obs_y[k]); }
            }
            int32_t rc = cx_set_messages(s->chain, (int64_t)ev.size(), ev.data(), ef.data(), CX_TO_FACTOR, CX_FORM_POINT, ey.data());
            VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_set_marginals (inner handle): ") + cx_last_error(s->chain));
        }
        return CX_OK;
    } catch (const std::bad_alloc &) {
        return vfail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_marginals: host allocation failed");
    }
}

int32_t vmp_get_marginals(cx_handle *h, int64_t n, const int64_t *ids, double *out) {
    Vmp *s = (Vmp *)h->vmp;
    try {
        std::vector<int32_t> in, ig;
        std::vector<int64_t> pn, pg;
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = find_var(s, ids[i]);
            VMP_REQUIRE(h, v >= 0, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(ids[i]));
            if (s->var_kind[v] == 0) { in.push_back(s->var_local[v]); pn.push_back(i); } else { ig.push_back(s->var_local[v]); pg.push_back(i); }
        }
        VMP_HIP(h, hipSetDevice(h->cfg.device));
        for (int pass = 0; pass < 2; pass++) {
            const std::vector<int32_t> &idx = pass == 0 ? in : ig;
            const std::vector<int64_t> &where = pass == 0 ? pn : pg;
            const int64_t m = (int64_t)idx.size();
            if (m == 0) continue;
            { int32_t rc = stage(h, s, m); if (rc != CX_OK) return rc; }
            int32_t *d_idx = s->st_idx; double *d_out = s->st_a;      // st_a holds 2 doubles per entry
            VMP_HIP(h, hipMemcpyAsync(d_idx, idx.data(), (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
            if (pass == 0) hipLaunchKernelGGL(k_gather2, dim3(blocks(m)), dim3(256), 0, h->stream, (int)m, d_idx, s->n_mean, s->n_prec, d_out);
            else hipLaunchKernelGGL(k_gather2, dim3(blocks(m)), dim3(256), 0, h->stream, (int)m, d_idx, s->g_shape, s->g_scale, d_out);
            std::vector<double> tmp(2 * m);
            VMP_HIP(h, hipMemcpyAsync(tmp.data(), d_out, (size_t)m * 16, hipMemcpyDeviceToHost, h->stream));
            VMP_HIP(h, hipStreamSynchronize(h->stream));
            for (int64_t k = 0; k < m; k++) { out[2 * where[k]] = tmp[2 * k]; out[2 * where[k] + 1] = tmp[2 * k + 1]; }
        }
        return CX_OK;
    } catch (const std::bad_alloc &) {
        return vfail(h, CX_ERR_OUT_OF_MEMORY, "cx_get_marginals: host allocation failed");
    }
}

// update_marginals!(engine, ids), src/inference_engine.jl:559-632, under the weak-dependency wiring described at the top
int32_t vmp_update_marginals(cx_handle *h, int64_t n, const int64_t *ids) {
    Vmp *s = (Vmp *)h->vmp;
    try {
        std::vector<uint8_t> mask;
        std::vector<int32_t> req;
        int64_t n_normal = 0;
        const int64_t n_latent = s->n_latent;
        if (n < 0) {   // whole classes, without an id list
            VMP_REQUIRE(h, n == CX_VMP_ALL_NORMAL || n == CX_VMP_ALL_PRECISION, CX_ERR_INVALID_ARGUMENT, "cx_update_marginals: n < 0 must be CX_VMP_ALL_NORMAL or CX_VMP_ALL_PRECISION");
            if (n == CX_VMP_ALL_NORMAL) n_normal = n_latent;
            else { req.resize(s->nG); std::iota(req.begin(), req.end(), 0); }
        } else {
            VMP_REQUIRE(h, ids, CX_ERR_INVALID_ARGUMENT, "cx_update_marginals: null id list");
            mask.assign(s->nN, 0);
            for (int64_t i = 0; i < n; i++) {
                const int64_t v = find_var(s, ids[i]);
                VMP_REQUIRE(h, v >= 0, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(ids[i]));
                if (s->var_kind[v] == 1) { if (std::find(req.begin(), req.end(), s->var_local[v]) == req.end()) req.push_back(s->var_local[v]); }
                else if (!s->n_observed[s->var_local[v]] && !mask[s->var_local[v]]) { mask[s->var_local[v]] = 1; n_normal++; }
            }
        }
        VMP_HIP(h, hipSetDevice(h->cfg.device));
        const bool do_normal = n_normal > 0, do_gamma = !req.empty();
        if (s->structured) {
            VMP_REQUIRE(h, !do_normal || n_normal == n_latent, CX_ERR_UNSUPPORTED,
                        "cx_update_marginals (structured): the latent Normal variables are updated together (one belief-propagation pass)");
            if (do_normal && do_gamma) {
                // A request that names states AND precisions (the last call of the reference's own experiment,
                // test/inference_engine_tests.jl:1113).  In the reference its order of evaluation emerges from the lazy readiness flags
                // (src/inference_engine.jl:575-608); pinned against the restated engine (tests/test_vmp_restatement.py) it is, whenever
                //   (a) the states have been updated before (the joint marginals exist) and the chain precisions have degree > 5,
                //   (b) every precision of a factor between two latent states (a "chain" precision) is named BEFORE the first state,
                //   (c) every other requested precision was updated since the states last were,
                // exactly three calls, class by class: the chain precisions (found pending by the first chain message and computed on
                // the fly), the states, the other precisions (in the final round, from the new states).  Anything else interleaves per
                // variable in the reference (some messages read the old expectation, some the new one): refused, as before.
                const char *why = "cx_update_marginals (structured): a request that names states and precisions together is accepted when the states "
                                  "were updated before, every precision of a transition factor (degree > 5) comes before the first state in the "
                                  "request and every other requested precision was updated since the last state update — otherwise the reference's "
                                  "order of evaluation interleaves per variable: request the classes in separate calls";
                VMP_REQUIRE(h, s->chain_ready && n >= 0, CX_ERR_UNSUPPORTED, why);
                std::vector<uint8_t> chain_g(s->nG, 0);
                for (int64_t f = 0; f < s->nF; f++)
                    if (!s->n_observed[s->f_out[f]] && !s->n_observed[s->f_mean[f]]) chain_g[s->f_gamma[f]] = 1;
                std::vector<int64_t> ids_chain, ids_state, ids_other;
                bool seen_state = false;
                for (int64_t i = 0; i < n; i++) {
                    const int64_t v = find_var(s, ids[i]);
                    if (s->var_kind[v] != 1) { if (!s->n_observed[s->var_local[v]]) { seen_state = true; ids_state.push_back(ids[i]); } continue; }
                    const int32_t g = s->var_local[v];
                    if (chain_g[g]) {
                        VMP_REQUIRE(h, !seen_state && s->g_deg[g] > 5, CX_ERR_UNSUPPORTED, why);
                        ids_chain.push_back(ids[i]);
                    } else {
                        VMP_REQUIRE(h, s->g_fresh[g], CX_ERR_UNSUPPORTED, why);
                        ids_other.push_back(ids[i]);
                    }
                }
                const int64_t before = h->sweeps_done;
                int32_t rc = CX_OK;
                if (!ids_chain.empty()) rc = vmp_update_marginals(h, (int64_t)ids_chain.size(), ids_chain.data());
                if (rc == CX_OK) rc = vmp_update_marginals(h, (int64_t)ids_state.size(), ids_state.data());
                if (rc == CX_OK && !ids_other.empty()) rc = vmp_update_marginals(h, (int64_t)ids_other.size(), ids_other.data());
                if (rc == CX_OK) h->sweeps_done = before + 1;      // one call of the caller's
                return rc;
            }
        }
        if (do_gamma) {
            // which precision variables need the chain messages (a factor with two latent Normal variables)
            if (s->structured && !s->chain_ready) {
                std::fill(s->g_needs_chain.begin(), s->g_needs_chain.end(), 0);
                for (int64_t f = 0; f < s->nF; f++)
                    if (!s->n_observed[s->f_out[f]] && !s->n_observed[s->f_mean[f]]) s->g_needs_chain[s->f_gamma[f]] = 1;
                // their messages depend on joint marginals that are not computed yet: not pending, nothing changes
                req.erase(std::remove_if(req.begin(), req.end(), [&](int32_t g) { return s->g_needs_chain[g] != 0; }), req.end());
            }
        }
        if (do_normal && !s->structured) {
            const bool all = n_normal == n_latent;
            if (!all) VMP_HIP(h, hipMemcpyAsync(s->d_mask, mask.data(), (size_t)s->nN, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_mf_normal, dim3(blocks(s->nN)), dim3(256), 0, h->stream, (int)s->nN, s->d_nb_off, s->d_nb_other, s->d_nb_gamma,
                               s->n_mean, s->n_prec, s->d_observed, all ? (const uint8_t *)nullptr : s->d_mask, s->g_mean, s->n_mean_alt, s->n_prec_alt);
        }
        if (!req.empty()) {
            if (req != s->req_on_device) {      // the same request round after round (the usual case): the list is already there
                VMP_HIP(h, hipMemcpyAsync(s->d_req, req.data(), req.size() * 4, hipMemcpyHostToDevice, h->stream));
                VMP_HIP(h, hipStreamSynchronize(h->stream));      // `req` is a local: the copy has to have read it
                s->req_on_device = req;
            }
            if (s->structured)
                hipLaunchKernelGGL(k_rate_sum<true>, dim3((unsigned)s->nChunks), dim3(256), 0, h->stream, s->d_chunk_begin, s->d_chunk_end, s->d_f_order, s->d_f_out, s->d_f_mean,
                                   s->d_f_gamma, s->n_mean, s->n_prec, s->d_observed, s->g_mean, s->d_slot_out, s->d_slot_mean, s->chain->d_v2f, s->d_partial);
            else
                hipLaunchKernelGGL(k_rate_sum<false>, dim3((unsigned)s->nChunks), dim3(256), 0, h->stream, s->d_chunk_begin, s->d_chunk_end, s->d_f_order, s->d_f_out, s->d_f_mean,
                                   s->d_f_gamma, s->n_mean, s->n_prec, s->d_observed, s->g_mean, (const int32_t *)nullptr, (const int32_t *)nullptr, (const double2 *)nullptr, s->d_partial);
            hipLaunchKernelGGL(k_gamma_finish, dim3((unsigned)req.size()), dim3(256), 0, h->stream, s->d_req, s->d_g_chunk_off, s->d_g_deg, s->d_partial,
                               s->g_shape, s->g_scale, s->g_mean);
        }
        // the final round: store
        if (do_normal && !s->structured) { std::swap(s->n_mean, s->n_mean_alt); std::swap(s->n_prec, s->n_prec_alt); }
        if (do_normal && s->structured) {
            cx_handle *c = s->chain;
            // The factor variance q = 1 / E[precision] is one number per precision variable: under the chain-scan schedule the inner
            // handle's kernels read it through the slot's precision index (slot_q, cx_chain.hip) instead of from a per-slot table that
            // k_set_q rewrote on every call (13 us, 48 MB at n = 1e6: profiles/r03_vmp_rocprof.md), and the scan's second kernel stores
            // the states' (mean, precision) straight into this family's arrays instead of k_pull_marginals copying them (8 us, 34 MB).
            // Any other schedule, and a sweep that has to run the general variable phase, keep the two kernels.
            bool scan = c->cfg.schedule == CX_SCHED_CHAIN_SCAN;
            if (scan) {      // a reader of messages off the chains would get its leaf messages from the general factor phase, which reads the table
                int32_t rcb = cxh::build_chains(c);
                VMP_REQUIRE(h, rcb == CX_OK, rcb, std::string("cx_update_marginals (inner handle): ") + cx_last_error(c));
                scan = c->chain_covers_all;
            }
            c->d_q_gamma = scan ? s->d_slot_gamma : nullptr; c->d_q_gmean = scan ? s->g_mean : nullptr;
            c->d_split_mean = scan ? s->n_mean : nullptr; c->d_split_prec = scan ? s->n_prec : nullptr;
            c->chain_msgs_unread = scan;     // k_rate reads the links' variable→factor messages, nobody the factor→variable ones (32 MB of stores)
            if (!scan) hipLaunchKernelGGL(k_set_q, dim3(blocks(c->nslots)), dim3(256), 0, h->stream, (int)c->nslots, s->d_slot_gamma, s->g_mean, c->d_q);
            c->chain_side_dirty = true;      // the leaf messages N(y, q) change with q
            c->split_marg_written = false;
            int32_t rc = cx_sweep(c, 1);
            VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_update_marginals (inner handle): ") + cx_last_error(c));
            if (!c->split_marg_written)
                hipLaunchKernelGGL(k_pull_marginals, dim3(blocks(s->nN)), dim3(256), 0, h->stream, (int)s->nN, c->d_marg, s->d_observed, s->n_mean, s->n_prec);
            s->chain_ready = true;
        }
        if (s->structured) {
            if (do_normal) std::fill(s->g_fresh.begin(), s->g_fresh.end(), 0);
            for (int32_t g : req) s->g_fresh[g] = 1;
        }
        VMP_HIP(h, hipGetLastError());
        h->sweeps_done++;
        return CX_OK;
    } catch (const std::bad_alloc &) {
        return vfail(h, CX_ERR_OUT_OF_MEMORY, "cx_update_marginals: host allocation failed");
    }
}

}  // namespace cx

// ---- checkpoint (cx_state_*): marginals, observed flags, and — structured family — the inner handle's own blob -------------
namespace {
struct VmpStateHeader {
    char magic[8];
    int32_t abi, family, schedule, chain_ready;
    int64_t nN, nG, nF, n_latent, sweeps_done, n_messages, inner_bytes;
    uint64_t fingerprint;
};
const char kVmpMagic[8] = {'C', 'X', 'V', 'M', 'P', 'S', 'T', '1'};

uint64_t vmp_fingerprint(const Vmp *s) {
    uint64_t f = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { f ^= b[i]; f *= 1099511628211ull; } };
    mix(s->var_ids.data(), s->var_ids.size() * 8); mix(s->var_kind.data(), s->var_kind.size() * 4);
    mix(s->fac_ids.data(), s->fac_ids.size() * 8); mix(s->f_out.data(), s->f_out.size() * 4);
    mix(s->f_mean.data(), s->f_mean.size() * 4); mix(s->f_gamma.data(), s->f_gamma.size() * 4);
    return f;
}
int64_t vmp_own_bytes(const Vmp *s) { return (int64_t)sizeof(VmpStateHeader) + s->nN * 17 + s->nG * 24; }
}  // namespace

namespace cx {

int32_t vmp_state_bytes(cx_handle *h, int64_t *bytes) {
    Vmp *s = (Vmp *)h->vmp;
    int64_t inner = 0;
    if (s->chain) { int32_t rc = cx_state_bytes(s->chain, &inner); VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_state_bytes (inner handle): ") + cx_last_error(s->chain)); }
    *bytes = vmp_own_bytes(s) + inner;
    return CX_OK;
}

int32_t vmp_state_export(cx_handle *h, void *buf, int64_t bytes) {
    Vmp *s = (Vmp *)h->vmp;
    int64_t need = 0, inner = 0;
    int32_t rc = vmp_state_bytes(h, &need);
    if (rc != CX_OK) return rc;
    VMP_REQUIRE(h, buf && bytes >= need, CX_ERR_INVALID_ARGUMENT, "cx_state_export: buffer smaller than cx_state_bytes");
    if (s->chain) (void)cx_state_bytes(s->chain, &inner);
    VMP_HIP(h, hipSetDevice(h->cfg.device));
    VMP_HIP(h, hipStreamSynchronize(h->stream));
    VmpStateHeader hd{};
    std::memcpy(hd.magic, kVmpMagic, 8);
    hd.abi = CX_ABI_VERSION; hd.family = h->cfg.family; hd.schedule = h->cfg.schedule; hd.chain_ready = s->chain_ready ? 1 : 0;
    hd.nN = s->nN; hd.nG = s->nG; hd.nF = s->nF; hd.n_latent = s->n_latent; hd.sweeps_done = h->sweeps_done;
    hd.n_messages = h->n_messages_per_sweep; hd.inner_bytes = inner; hd.fingerprint = vmp_fingerprint(s);
    char *o = (char *)buf;
    std::memcpy(o, &hd, sizeof hd); o += sizeof hd;
    VMP_HIP(h, hipMemcpy(o, s->n_mean, (size_t)s->nN * 8, hipMemcpyDeviceToHost)); o += s->nN * 8;
    VMP_HIP(h, hipMemcpy(o, s->n_prec, (size_t)s->nN * 8, hipMemcpyDeviceToHost)); o += s->nN * 8;
    std::memcpy(o, s->n_observed.data(), (size_t)s->nN); o += s->nN;
    VMP_HIP(h, hipMemcpy(o, s->g_shape, (size_t)s->nG * 8, hipMemcpyDeviceToHost)); o += s->nG * 8;
    VMP_HIP(h, hipMemcpy(o, s->g_scale, (size_t)s->nG * 8, hipMemcpyDeviceToHost)); o += s->nG * 8;
    VMP_HIP(h, hipMemcpy(o, s->g_mean, (size_t)s->nG * 8, hipMemcpyDeviceToHost)); o += s->nG * 8;
    if (s->chain) {
        rc = cx_state_export(s->chain, o, inner);
        VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_state_export (inner handle): ") + cx_last_error(s->chain));
    }
    return CX_OK;
}

int32_t vmp_state_import(cx_handle *h, const void *buf, int64_t bytes) {
    Vmp *s = (Vmp *)h->vmp;
    VMP_REQUIRE(h, buf && bytes >= (int64_t)sizeof(VmpStateHeader), CX_ERR_INVALID_ARGUMENT, "cx_state_import: blob too short");
    VmpStateHeader hd;
    std::memcpy(&hd, buf, sizeof hd);
    VMP_REQUIRE(h, std::memcmp(hd.magic, kVmpMagic, 8) == 0 && hd.abi == CX_ABI_VERSION, CX_ERR_INVALID_ARGUMENT, "cx_state_import: not a state blob of a variational handle of this ABI version");
    VMP_REQUIRE(h, hd.family == h->cfg.family && hd.schedule == h->cfg.schedule, CX_ERR_INVALID_ARGUMENT, "cx_state_import: the blob was exported with a different family / schedule");
    VMP_REQUIRE(h, hd.nN == s->nN && hd.nG == s->nG && hd.nF == s->nF && hd.fingerprint == vmp_fingerprint(s), CX_ERR_INVALID_ARGUMENT, "cx_state_import: the blob belongs to a different graph");
    VMP_REQUIRE(h, bytes >= vmp_own_bytes(s) + hd.inner_bytes && (hd.inner_bytes > 0) == (s->chain != nullptr), CX_ERR_INVALID_ARGUMENT, "cx_state_import: truncated or foreign blob");
    VMP_HIP(h, hipSetDevice(h->cfg.device));
    VMP_HIP(h, hipStreamSynchronize(h->stream));
    const char *o = (const char *)buf + sizeof hd;
    if (s->chain) {   // the inner handle validates its own part before anything is written
        int32_t rc = cx_state_import(s->chain, o + s->nN * 17 + s->nG * 24, hd.inner_bytes);
        VMP_REQUIRE(h, rc == CX_OK, rc, std::string("cx_state_import (inner handle): ") + cx_last_error(s->chain));
    }
    VMP_HIP(h, hipMemcpy(s->n_mean, o, (size_t)s->nN * 8, hipMemcpyHostToDevice)); o += s->nN * 8;
    VMP_HIP(h, hipMemcpy(s->n_prec, o, (size_t)s->nN * 8, hipMemcpyHostToDevice)); o += s->nN * 8;
    std::memcpy(s->n_observed.data(), o, (size_t)s->nN);
    VMP_HIP(h, hipMemcpy(s->d_observed, o, (size_t)s->nN, hipMemcpyHostToDevice)); o += s->nN;
    VMP_HIP(h, hipMemcpy(s->g_shape, o, (size_t)s->nG * 8, hipMemcpyHostToDevice)); o += s->nG * 8;
    VMP_HIP(h, hipMemcpy(s->g_scale, o, (size_t)s->nG * 8, hipMemcpyHostToDevice)); o += s->nG * 8;
    VMP_HIP(h, hipMemcpy(s->g_mean, o, (size_t)s->nG * 8, hipMemcpyHostToDevice)); o += s->nG * 8;
    s->n_latent = hd.n_latent; s->chain_ready = hd.chain_ready != 0;
    std::fill(s->g_fresh.begin(), s->g_fresh.end(), 0);      // (not in the blob: a request of states and precisions together waits for the precisions' next update)
    h->sweeps_done = hd.sweeps_done; h->n_messages_per_sweep = hd.n_messages;
    return CX_OK;
}

}  // namespace cx
