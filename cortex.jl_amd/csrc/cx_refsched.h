// cx_refsched.h — the GPU-free part of CX_SCHED_REFERENCE: ONE cx_sweep on ANY graph — loops included — is what ONE
// update_marginals! of the reference leaves there.
//
// On a graph with cycles the reference's call is a sequential pass that always reads the newest values
// (/root/reference/src/inference_engine.jl:575-608: forward / reverse passes over the requested variables;
// /root/reference/src/signal.jl:466-490: per marginal a depth-first walk through the dependencies flagged intermediate, a pending
// signal computed the moment it is met).  WHICH signals are computed, and in which order, is decided by the readiness nibbles alone
// (/root/reference/src/signal.jl:141-154,232-253,339-356,668-730) — never by a value.  So the order of a call can be found without
// computing anything: this header keeps a shadow of every signal's readiness state, wired as the default resolver wires it
// (/root/reference/src/dependencies.jl:17-173), runs the reference's scheduler on the shadow and RECORDS the executions; the recorded
// list is then levelled — an execution's stage is one more than the latest stage of (a) the executions of this call whose result it
// reads, (b) the executions that read the value it overwrites, (c) its own previous execution — and the stages are replayed on the
// device as item lists, the machinery CX_SCHED_TREE has (cx_api_sweep.hip).  Every value an item reads is then exactly the value the
// reference's rule call read, and the results are the reference's up to the rounding of the natural-form arithmetic.
//
// The shadow is driven by whatever changes a value through the ABI: cx_set_messages / cx_seed_messages (the user's set_value!),
// cx_update_batch (a plug-in's process!) and the sweeps themselves.
//
// Signals, by number:   [0, ne)            MessageToFactor of CSR edge e        (inference_signal.jl:29-32)
//                       [ne, 2 ne)         MessageToVariable of CSR edge e      (:45-48)
//                       [2 ne, 2 ne + nv)  IndividualMarginal of variable v     (:78-80)
//                       [.., + nj)         JointMarginal(factor)                (:93-96), user wirings only (nj = factors, else 0)
//                       [.., nsig)         ProductOfMessages(variable, range)   (:62-66), the segment-tree nodes of variables of degree > 5
// Pure host C++ over any struct H with cx_handle's host fields (cx_flatten.h); tests/test_refsched.py runs it against the test suite's
// restated reference engine, execution by execution, also under -fsanitize=address,undefined.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "cx_flatten.h"

namespace cx {
namespace refsched {

using flat::fail_;

constexpr uint8_t kPot = 1, kPend = 2, kComputed = 4;          // SignalProps (signal.jl:47-60) + "value !== UndefValue()"
constexpr uint64_t kInter = 0x1, kWeak = 0x2, kComp = 0x4, kFresh = 0x8;      // signal.jl:507-510
constexpr uint64_t kAllWeak = 0x2222222222222222ull, kAllComp = 0x4444444444444444ull, kAllFresh = 0x8888888888888888ull, kAllPass = 0x1111111111111111ull;

struct Prod { int32_t var, lo, hi; };      // local variable, 1-based inclusive range over its factors in ascending id order (the variant's `range`)

// what computes a MessageToVariable signal (Wiring::vrule).  The sum-product rule of the factor reads the stored messages of the factor's
// other edges; the variational rules of a CX_FACTOR_NORMAL_PRECISION factor (out ~ N(in, 1 / precision)) read exactly their dependency
// list, which is also how they are told apart — the reference's user rule dispatches on the dependencies it is handed the same way
// (test/inference_engine_tests.jl:647-689, 969-1030):
//   kRuleMfNormal   to a Normal variable <- {marginal(other Normal), marginal(precision)}:  N(E[other], E[precision])                    (:654-664)
//   kRuleMfGamma    to the precision     <- {marginal(out), marginal(in)}:  Gamma(3/2, 2 / (var out + var in + (E out - E in)^2))        (:666-684)
//   kRuleStNormal   to a Normal variable <- {MessageToFactor(other Normal), marginal(precision)}:  N(mean m, 1 / (var m + 1 / E[precision]))  (:1004-1010)
//   kRuleStGamma    to the precision     <- {JointMarginal(factor)}:  Gamma(3/2, 2 / (V11 - 2 V12 + V22 + (m1 - m2)^2))                 (:1011-1016)
// and a JointMarginal(factor) <- {MessageToFactor(out), MessageToFactor(in), marginal(precision)} is the 2-d Gaussian of :939-967.
constexpr uint8_t kRuleBP = 0, kRuleNone = 1, kRuleMfNormal = 2, kRuleMfGamma = 3, kRuleStNormal = 4, kRuleStGamma = 5;

struct Wiring {
    int64_t ne = 0, nv = 0, nj = 0, nsig = 0;
    std::vector<int64_t> dep_off, lis_off, chunk_off;      // [nsig + 1]
    std::vector<int32_t> dep;                              // dependencies in the resolver's order (add_dependency! appends)
    std::vector<uint8_t> dep_inter;                        // 1: flagged intermediate
    std::vector<uint8_t> dep_weak;                         // 1: flagged weak (a user wiring, cx_graph_wire; the default resolver sets none)
    std::vector<int32_t> lis;                              // listeners
    std::vector<int32_t> lis_idx;                          // position of this signal in the listener's dependency list (notify_listener!: first match)
    std::vector<uint8_t> lis_listen;                       // listenmask (add_dependency!(...; listen)): 1 everywhere under the default wiring
    bool custom = false;                                   // a user wiring (cx_graph_wire): products of messages are "sum of the dependency list" items
    std::vector<Prod> prods;                               // signal 2 ne + nv + nj + i
    std::vector<uint8_t> vrule;                            // per MessageToVariable signal (index e): kRule*
    std::vector<uint8_t> jrule;                            // per JointMarginal signal (index f): 1 = wired as the variational joint of a NORMAL_PRECISION factor
    std::vector<int64_t> link_off;                         // [nv + 1]: link_signal_to_variable! (model_engine.jl:64-71), in link order; empty = none
    std::vector<int32_t> link;
    std::vector<int32_t> efac, foff, fedge;                // local factor of every CSR edge; the edges of a factor in ascending variable order
    int64_t sig_v2f(int64_t e) const { return e; }
    int64_t sig_f2v(int64_t e) const { return ne + e; }
    int64_t sig_marg(int64_t v) const { return 2 * ne + v; }
    int64_t sig_joint(int64_t f) const { return 2 * ne + nv + f; }
    int64_t sig_prod(int64_t i) const { return 2 * ne + nv + nj + i; }
    bool is_v2f(int64_t s) const { return s < ne; }
    bool is_f2v(int64_t s) const { return s >= ne && s < 2 * ne; }
    bool is_marg(int64_t s) const { return s >= 2 * ne && s < 2 * ne + nv; }
    bool is_joint(int64_t s) const { return s >= 2 * ne + nv && s < 2 * ne + nv + nj; }
    bool is_prod(int64_t s) const { return s >= 2 * ne + nv + nj; }
};

struct State {
    std::vector<uint64_t> chunks;       // 4 bits per dependency, 16 per word (SignalDependenciesProps, signal.jl:36-45)
    std::vector<uint8_t> flags;         // kPot | kPend | kComputed per signal
    uint64_t hash = 0;                  // incremental (Zobrist) fingerprint of chunks + flags: equal states have equal fingerprints
};

inline uint64_t mix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31);
}
inline uint64_t zob_chunk(int64_t c, uint64_t value) { return mix64(value ^ mix64((uint64_t)c * 2 + 1)); }
inline uint64_t zob_flag(int64_t s, uint8_t f) { return mix64((uint64_t)f ^ mix64((uint64_t)s * 2)); }

// ---- the default resolver's wiring (dependencies.jl:5-173), emitted as (signal, dependency, intermediate) in add_dependency! order ----
// resolve_variable_dependencies! (dependencies.jl:33-126) with form_segment_tree_dependency! (:128-173) for ONE variable.  hl(e): does the
// MessageToFactor of edge e have listeners at this point — the reference asks get_listeners while it wires (:73,107,117,149,159), so under a
// user wiring the answer depends on what was wired before
template <class H, class Emit, class HasL>
struct VarWirer {
    const H *h; int64_t ne; std::vector<Prod> *prods; int64_t *next_prod; Emit &emit; HasL &hl;
    VarWirer(const H *h_, std::vector<Prod> *p, int64_t *np, Emit &e, HasL &l) : h(h_), ne(h_->ne), prods(p), next_prod(np), emit(e), hl(l) {}
    // over the 0-based half-open range [lo, hi) of the variable's edges
    int64_t tree(int32_t v, int32_t lo, int32_t hi) {
        const int32_t e0 = h->var_off[v];
        if (hi - lo == 1) return ne + e0 + lo;
        const int32_t mid = lo + (hi - lo) / 2;
        const int64_t left = tree(v, lo, mid), right = tree(v, mid, hi);
        cross(v, lo, mid, right); cross(v, mid, hi, left);
        const int64_t inter = (*next_prod)++;
        if (prods) prods->push_back(Prod{v, lo + 1, hi});
        emit(inter, left, 1); emit(inter, right, 1);
        return inter;
    }
    void cross(int32_t v, int32_t lo, int32_t hi, int64_t other) {
        const int32_t e0 = h->var_off[v];
        for (int32_t k = lo; k < hi; k++) if (hl(e0 + k)) emit((int64_t)(e0 + k), other, 1);
    }
    void variable(int64_t v) {
        const int32_t e0 = h->var_off[v], n = h->var_off[v + 1] - e0;
        const int64_t marg = 2 * ne + v;
        if (n == 0) return;
        if (n < 2) { emit(marg, ne + e0, 1); return; }
        if (n <= 5) {
            for (int32_t k = 0; k < n; k++) {
                emit(marg, ne + e0 + k, 1);
                if (hl(e0 + k))
                    for (int32_t j = 0; j < n; j++) if (j != k) emit((int64_t)(e0 + k), ne + e0 + j, 1);
            }
            return;
        }
        const int32_t mid = n / 2;
        const int64_t left = tree((int32_t)v, 0, mid), right = tree((int32_t)v, mid, n);
        cross((int32_t)v, 0, mid, right); cross((int32_t)v, mid, n, left);
        emit(marg, left, 1); emit(marg, right, 1);
    }
};

template <class H, class Emit>
void wire(const H *h, const std::vector<int32_t> &efac, const std::vector<int32_t> &foff, const std::vector<int32_t> &fedge, std::vector<Prod> *prods,
          Emit &&emit) {
    const int64_t ne = h->ne, nv = h->nv, nf = h->nf;
    // resolve_factor_dependencies!, dependencies.jl:17-31: every message out of a factor depends on the messages INTO it from its other
    // variables, in the order of the factor's variables (ascending id)
    for (int64_t f = 0; f < nf; f++)
        for (int32_t k1 = foff[f]; k1 < foff[f + 1]; k1++)
            for (int32_t k2 = foff[f]; k2 < foff[f + 1]; k2++)
                if (k1 != k2) emit(ne + fedge[k1], (int64_t)fedge[k2], 0);
    auto has_listeners = [&](int64_t e) { return foff[efac[e] + 1] - foff[efac[e]] >= 2; };      // someone listens: a factor of two or more variables
    int64_t next_prod = 2 * ne + nv;
    VarWirer<H, Emit, decltype(has_listeners)> vw(h, prods, &next_prod, emit, has_listeners);
    for (int64_t v = 0; v < nv; v++) vw.variable(v);
}

// the factor side of the graph as CSR (edges of one factor in ascending variable order: the CSR edge order is (variable, factor) ascending)
template <class H>
int32_t factor_csr(const H *h, Wiring &W, std::string &err, const char *who) {
    const int64_t ne = h->ne, nf = h->nf;
    W.efac.assign(ne, 0); W.foff.assign(nf + 1, 0); W.fedge.assign(ne, 0);
    for (int64_t e = 0; e < ne; e++) {
        auto it = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), h->edge_fac_id[e]);
        if (it == h->fac_ids.end() || *it != h->edge_fac_id[e]) return fail_(err, CX_ERR_STATE, std::string(who) + ": edge names an unknown factor");
        W.efac[e] = (int32_t)(it - h->fac_ids.begin());
        W.foff[W.efac[e] + 1]++;
    }
    for (int64_t f = 0; f < nf; f++) W.foff[f + 1] += W.foff[f];
    std::vector<int32_t> fill(W.foff.begin(), W.foff.end() - 1);
    for (int64_t e = 0; e < ne; e++) W.fedge[fill[W.efac[e]]++] = (int32_t)e;
    return CX_OK;
}

// listeners (in add_dependency! order over the whole wiring) and chunk offsets from the dependency lists; lis_listen[q] must have been
// sized; a caller with per-dependency listen flags passes them as `listen_of_dep`
inline void finish_wiring(Wiring &W, const std::vector<uint8_t> *listen_of_dep = nullptr) {
    std::vector<int64_t> lcnt(W.nsig + 1, 0);
    for (int64_t p = 0; p < (int64_t)W.dep.size(); p++) lcnt[W.dep[p] + 1]++;
    W.lis_off.assign(W.nsig + 1, 0);
    for (int64_t s = 0; s < W.nsig; s++) W.lis_off[s + 1] = W.lis_off[s] + lcnt[s + 1];
    std::vector<int64_t> fill(W.lis_off.begin(), W.lis_off.end() - 1);
    for (int64_t s = 0; s < W.nsig; s++)
        for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++) {
            const int64_t q = fill[W.dep[p]]++;
            W.lis[q] = (int32_t)s; W.lis_idx[q] = (int32_t)(p - W.dep_off[s]);
            if (listen_of_dep) W.lis_listen[q] = (*listen_of_dep)[p];
        }
    W.chunk_off.assign(W.nsig + 1, 0);
    for (int64_t s = 0; s < W.nsig; s++) {
        const int64_t n = W.dep_off[s + 1] - W.dep_off[s];
        W.chunk_off[s + 1] = W.chunk_off[s] + std::max<int64_t>(1, (n + 15) / 16);      // a SignalDependenciesProps always owns one chunk (signal.jl:36-45)
    }
}

template <class H>
int32_t build_wiring(const H *h, Wiring &W, std::string &err) {
    const int64_t ne = h->ne, nv = h->nv;
    W = Wiring();
    W.ne = ne; W.nv = nv;
    { const int32_t rc = factor_csr(h, W, err, "reference schedule"); if (rc != CX_OK) return rc; }
    const std::vector<int32_t> &efac = W.efac, &foff = W.foff, &fedge = W.fedge;
    // pass 1: count; the segment-tree nodes get their numbers in creation order
    std::vector<Prod> prods;
    std::vector<int64_t> cnt;
    {
        int64_t nprod = 0;
        for (int64_t v = 0; v < nv; v++) { const int32_t n = h->var_off[v + 1] - h->var_off[v]; if (n > 5) nprod += n - 2; }      // a binary tree over n leaves under two roots
        W.nsig = 2 * ne + nv + nprod;
        if (W.nsig >= (int64_t)0x7fffffff) return fail_(err, CX_ERR_UNSUPPORTED, "reference schedule: more than 2^31 signals");
        cnt.assign(W.nsig + 1, 0);
        wire(h, efac, foff, fedge, &prods, [&](int64_t s, int64_t, int) { cnt[s + 1]++; });
        if ((int64_t)prods.size() != nprod) return fail_(err, CX_ERR_STATE, "reference schedule: segment-tree node count");
    }
    W.prods = std::move(prods);
    W.dep_off.assign(W.nsig + 1, 0);
    for (int64_t s = 0; s < W.nsig; s++) W.dep_off[s + 1] = W.dep_off[s] + cnt[s + 1];
    const int64_t nd = W.dep_off[W.nsig];
    W.dep.assign(nd, -1); W.dep_inter.assign(nd, 0); W.dep_weak.assign(nd, 0);
    {   // pass 2: fill
        std::vector<int64_t> fill(W.dep_off.begin(), W.dep_off.end() - 1);
        wire(h, efac, foff, fedge, nullptr, [&](int64_t s, int64_t d, int inter) {
            const int64_t p = fill[s]++;
            W.dep[p] = (int32_t)d; W.dep_inter[p] = (uint8_t)inter;
        });
    }
    W.lis.assign(nd, -1); W.lis_idx.assign(nd, 0); W.lis_listen.assign(nd, 1);
    finish_wiring(W);
    // messages the device cannot compute: out of a factor that has no sum-product rule here and other variables (the reference's processor
    // would call the user's rule; an opaque factor has none — inference_engine.jl:358 error(...); a CX_FACTOR_NORMAL_PRECISION factor has
    // variational rules only, which the default wiring does not feed)
    W.vrule.assign(ne, kRuleBP);
    for (int64_t e = 0; e < ne; e++) {
        const int32_t f = efac[e], deg = foff[f + 1] - foff[f];
        if (deg >= 2 && (h->fac_kind[f] == CX_FACTOR_OPAQUE || h->fac_kind[f] == CX_FACTOR_NORMAL_PRECISION)) W.vrule[e] = kRuleNone;
    }
    return CX_OK;
}

// ---- a user wiring (cx_graph_wire): the reference's add_dependency!(signal, dependency; weak, listen, intermediate), signal.jl:286-337,
// call by call, in place of the default resolver's (a user resolver, dependencies.jl:1-15).  Signals: messages of both directions,
// marginals and the joint marginals of factors.  What a signal may depend on follows from what computes it on the device:
//   MessageToFactor(v, f)   <- MessageToVariable(v, f'), f' != f                its value = the product of its dependency list
//   IndividualMarginal(v)   <- MessageToVariable(v, f')                         its value = the product of its dependency list
//   MessageToVariable(v, f) <- MessageToFactor(v', f), v' != v                  the factor's sum-product rule on the stored messages of the
//                                                                               factor's other edges (all of them, listed or not)
//   MessageToVariable / JointMarginal of a CX_FACTOR_NORMAL_PRECISION factor    the variational rules, chosen by the dependency list (kRule* above)
// flags: bit 0 weak, bit 1 intermediate, bit 2 "do not listen"; bit 3: signal = IndividualMarginal(v), the dependency is ignored — the
// DEFAULT resolver's resolve_variable_dependencies!(v) at this point of the call order (a user resolver that delegates the variable side,
// test/inference_engine_tests.jl:812-814; it asks which MessageToFactor signals have listeners by now); bit 4: signal = JointMarginal(f),
// dependency = IndividualMarginal(v): link_signal_to_variable!(v, signal) (model_engine.jl:64-71).  Self-dependencies are skipped as
// add_dependency! skips them; a signal may not list a dependency twice.  sig / dep: signal numbers (Wiring::sig_*, joint marginals
// numbered 2 ne + nv + f), in call order.
constexpr int32_t kWireWeak = 1, kWireIntermediate = 2, kWireNoListen = 4, kWireDefaultVariable = 8, kWireLink = 16;

template <class H>
int32_t build_user_wiring(const H *h, int64_t n, const int64_t *sig, const int64_t *dep, const int32_t *flags, Wiring &W, std::string &err) {
    const int64_t ne = h->ne, nv = h->nv, nf = h->nf;
    W = Wiring();
    W.ne = ne; W.nv = nv; W.nj = nf; W.custom = true;
    { const int32_t rc = factor_csr(h, W, err, "cx_graph_wire"); if (rc != CX_OK) return rc; }
    const std::vector<int32_t> &efac = W.efac;
    const int64_t n_named = 2 * ne + nv + nf;      // what a caller can name; the segment-tree nodes come after
    auto var_of = [&](int64_t s) -> int64_t { return s < ne ? h->edge_var[s] : (s < 2 * ne ? h->edge_var[s - ne] : s - 2 * ne); };
    for (int64_t i = 0; i < n; i++) {
        const int64_t s = sig[i], d = dep[i];
        if (s < 0 || s >= n_named) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_wire: unknown signal");
        if (flags[i] & kWireDefaultVariable) {
            if (!W.is_marg(s)) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_wire: CX_WIRE_DEFAULT_VARIABLE names the variable by its IndividualMarginal signal");
            continue;
        }
        if (d < 0 || d >= n_named) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_wire: unknown signal");
        if (flags[i] & kWireLink) {
            if (!W.is_marg(d)) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_wire: CX_WIRE_LINK names the variable by its IndividualMarginal signal");
            continue;
        }
    }
    // the calls, replayed twice (count, fill): dependencies in call order, the default variable wiring where the caller asked for it
    std::vector<int32_t> nlis;
    std::vector<Prod> prods;
    auto replay = [&](auto &&emit, std::vector<Prod> *pr) {
        nlis.assign(n_named, 0);
        int64_t next_prod = n_named;
        auto counted = [&](int64_t s, int64_t d, int fl) { if (d < n_named) nlis[d]++; emit(s, d, fl); };
        auto hl = [&](int64_t e) { return nlis[e] > 0; };
        auto from_default = [&](int64_t s, int64_t d, int inter) { counted(s, d, inter ? kWireIntermediate : 0); };
        VarWirer<H, decltype(from_default), decltype(hl)> vw(h, pr, &next_prod, from_default, hl);
        for (int64_t i = 0; i < n; i++) {
            if (flags[i] & kWireLink) continue;
            if (flags[i] & kWireDefaultVariable) { vw.variable(sig[i] - 2 * ne); continue; }
            if (sig[i] == dep[i]) continue;                                      // signal.jl:295
            counted(sig[i], dep[i], flags[i]);
        }
        return next_prod;
    };
    std::vector<int64_t> cnt(n_named + 1, 0);
    W.nsig = replay([&](int64_t s, int64_t, int) { if (s + 2 > (int64_t)cnt.size()) cnt.resize(s + 2, 0); cnt[s + 1]++; }, &prods);
    if (W.nsig >= (int64_t)0x7fffffff) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_wire: more than 2^31 signals");
    cnt.resize(W.nsig + 1, 0);
    W.prods = std::move(prods);
    W.dep_off.assign(W.nsig + 1, 0);
    for (int64_t s = 0; s < W.nsig; s++) W.dep_off[s + 1] = W.dep_off[s] + cnt[s + 1];
    const int64_t nd = W.dep_off[W.nsig];
    W.dep.assign(nd, -1); W.dep_inter.assign(nd, 0); W.dep_weak.assign(nd, 0);
    std::vector<uint8_t> listen(nd, 1);
    {
        std::vector<int64_t> fill(W.dep_off.begin(), W.dep_off.end() - 1);
        (void)replay([&](int64_t s, int64_t d, int fl) {
            const int64_t p = fill[s]++;
            W.dep[p] = (int32_t)d;
            W.dep_weak[p] = (fl & kWireWeak) ? 1 : 0; W.dep_inter[p] = (fl & kWireIntermediate) ? 1 : 0; listen[p] = (fl & kWireNoListen) ? 0 : 1;
        }, nullptr);
    }
    {   // a dependency listed twice: notify_listener! (signal.jl:339-356) would only ever refresh the first
        std::vector<int32_t> tmp;
        for (int64_t s = 0; s < W.nsig; s++) {
            const int64_t lo = W.dep_off[s], hi = W.dep_off[s + 1];
            if (hi - lo < 2) continue;
            tmp.assign(W.dep.begin() + lo, W.dep.begin() + hi);
            std::sort(tmp.begin(), tmp.end());
            if (std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end()) return fail_(err, CX_ERR_INVALID_ARGUMENT, "cx_graph_wire: a signal lists a dependency twice");
        }
    }
    W.lis.assign(nd, -1); W.lis_idx.assign(nd, 0); W.lis_listen.assign(nd, 1);
    finish_wiring(W, &listen);
    {   // linked signals per variable, in link order
        std::vector<int64_t> lc(nv + 1, 0);
        bool any = false;
        for (int64_t i = 0; i < n; i++) if (flags[i] & kWireLink) { lc[dep[i] - 2 * ne + 1]++; any = true; }
        if (any) {
            W.link_off.assign(nv + 1, 0);
            for (int64_t v = 0; v < nv; v++) W.link_off[v + 1] = W.link_off[v] + lc[v + 1];
            W.link.assign(W.link_off[nv], -1);
            std::vector<int64_t> fill(W.link_off.begin(), W.link_off.end() - 1);
            for (int64_t i = 0; i < n; i++) if (flags[i] & kWireLink) W.link[fill[dep[i] - 2 * ne]++] = (int32_t)sig[i];
        }
    }
    // ---- what computes each signal: the dependency lists must be ones a device rule can serve -----------------------------------------
    const bool np = !h->np_role.empty();
    W.vrule.assign(ne, kRuleBP); W.jrule.assign(nf, 0);
    for (int64_t s = 0; s < W.nsig; s++) {
        const int64_t lo = W.dep_off[s], hi = W.dep_off[s + 1], nd_s = hi - lo;
        if (W.is_v2f(s) || W.is_marg(s) || W.is_prod(s)) {
            const int64_t v = W.is_prod(s) ? W.prods[s - W.sig_prod(0)].var : var_of(s);
            for (int64_t p = lo; p < hi; p++) {
                const int64_t d = W.dep[p];
                const bool ok = (W.is_f2v(d) && var_of(d) == v && !(W.is_v2f(s) && d - ne == s)) || (W.is_prod(d) && W.prods[d - W.sig_prod(0)].var == v);
                if (!ok) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_wire: a MessageToFactor / IndividualMarginal signal takes MessageToVariable signals of its own variable (other factors) as "
                                                               "dependencies — its value is their product; other inputs would need a rule the device does not have");
            }
            continue;
        }
        if (W.is_f2v(s)) {
            const int64_t e = s - ne;
            const int32_t f = efac[e], deg = W.foff[f + 1] - W.foff[f], kind = h->fac_kind[f];
            if (kind != CX_FACTOR_NORMAL_PRECISION) {
                for (int64_t p = lo; p < hi; p++) {
                    const int64_t d = W.dep[p];
                    if (!(W.is_v2f(d) && efac[d] == f && d != e))
                        return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_wire: a MessageToVariable signal of a sum-product factor takes MessageToFactor signals of its factor's other variables as "
                                                              "dependencies (messages that depend on marginals: the variational rules of CX_FACTOR_NORMAL_PRECISION factors)");
                }
                if (deg >= 2 && kind == CX_FACTOR_OPAQUE) W.vrule[e] = kRuleNone;
                continue;
            }
            // out ~ N(in, 1 / precision): the two other edges of the factor, by role
            W.vrule[e] = kRuleNone;
            if (nd_s == 0) continue;
            if (!np) return fail_(err, CX_ERR_STATE, "cx_graph_wire: a CX_FACTOR_NORMAL_PRECISION factor without edge roles");
            const int role = h->np_role[e];
            int64_t e_other = -1, e_other2 = -1, e_prec = -1;      // the Normal edges other than e (one, or two when e is the precision), the precision edge
            for (int32_t k = W.foff[f]; k < W.foff[f + 1]; k++) {
                const int32_t e2 = W.fedge[k];
                if (e2 == e) continue;
                if (h->np_role[e2] == CX_ROLE_PRECISION) e_prec = e2; else if (e_other < 0) e_other = e2; else e_other2 = e2;
            }
            auto has = [&](int64_t d) { for (int64_t p = lo; p < hi; p++) if (W.dep[p] == d) return true; return false; };
            const std::string who = "cx_graph_wire: the message from factor " + std::to_string(h->edge_fac_id[e]) + " to variable " + std::to_string(h->var_ids[h->edge_var[e]]);
            if (role != CX_ROLE_PRECISION) {
                if (nd_s == 2 && has(W.sig_marg(h->edge_var[e_other])) && has(W.sig_marg(h->edge_var[e_prec]))) W.vrule[e] = kRuleMfNormal;
                else if (nd_s == 2 && has(W.sig_v2f(e_other)) && has(W.sig_marg(h->edge_var[e_prec]))) W.vrule[e] = kRuleStNormal;
                else return fail_(err, CX_ERR_UNSUPPORTED, who + ": a message to a Normal variable of a CX_FACTOR_NORMAL_PRECISION factor depends on the precision's marginal and on either the "
                                                                 "other Normal variable's marginal (mean field) or its MessageToFactor (structured)");
            } else {
                if (nd_s == 2 && has(W.sig_marg(h->edge_var[e_other])) && has(W.sig_marg(h->edge_var[e_other2]))) W.vrule[e] = kRuleMfGamma;
                else if (nd_s == 1 && has(W.sig_joint(f))) W.vrule[e] = kRuleStGamma;
                else return fail_(err, CX_ERR_UNSUPPORTED, who + ": a message to the precision of a CX_FACTOR_NORMAL_PRECISION factor depends on the marginals of the factor's two Normal "
                                                                 "variables (mean field) or on the factor's JointMarginal (structured)");
            }
            continue;
        }
        // JointMarginal(f)
        if (nd_s == 0) continue;
        const int64_t f = s - W.sig_joint(0);
        int64_t e_a = -1, e_b = -1, e_prec = -1;
        if (h->fac_kind[f] == CX_FACTOR_NORMAL_PRECISION && np)
            for (int32_t k = W.foff[f]; k < W.foff[f + 1]; k++) {
                const int32_t e2 = W.fedge[k];
                if (h->np_role[e2] == CX_ROLE_PRECISION) e_prec = e2; else if (e_a < 0) e_a = e2; else e_b = e2;
            }
        bool ok = e_a >= 0 && e_b >= 0 && e_prec >= 0 && nd_s == 3;
        if (ok) {
            int seen = 0;
            for (int64_t p = lo; p < hi; p++) seen |= W.dep[p] == W.sig_v2f(e_a) ? 1 : W.dep[p] == W.sig_v2f(e_b) ? 2 : W.dep[p] == W.sig_marg(h->edge_var[e_prec]) ? 4 : 8;
            ok = seen == 7;
        }
        if (!ok) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_wire: the JointMarginal of factor " + std::to_string(h->fac_ids[f]) + " must be that of a CX_FACTOR_NORMAL_PRECISION factor and depend on "
                                                       "the two MessageToFactor signals of its Normal variables and on its precision's marginal (test/inference_engine_tests.jl:939-967)");
        W.jrule[f] = 1;
    }
    // a linked signal must be one this wiring computes
    for (int32_t ls : W.link) if (!W.is_joint(ls)) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_wire: CX_WIRE_LINK links JointMarginal signals");
    {   // process_dependencies! (signal.jl:466-490) descends through intermediate dependencies without a visited set: a cycle of them is an
        // endless recursion in the reference (a stack overflow), so such a wiring is refused here.  Three-colour depth-first search.
        std::vector<uint8_t> colour(W.nsig, 0);
        std::vector<std::pair<int32_t, int64_t>> stack;
        for (int64_t r = 0; r < W.nsig; r++) {
            if (colour[r]) continue;
            colour[r] = 1; stack.emplace_back((int32_t)r, W.dep_off[r]);
            while (!stack.empty()) {
                auto &top = stack.back();
                const int32_t s = top.first;
                if (top.second == W.dep_off[s + 1]) { colour[s] = 2; stack.pop_back(); continue; }
                const int64_t p = top.second++;
                if (!W.dep_inter[p]) continue;
                const int32_t d = W.dep[p];
                if (colour[d] == 1) return fail_(err, CX_ERR_UNSUPPORTED, "cx_graph_wire: the intermediate dependencies form a cycle: process_dependencies! (src/signal.jl:466-490) would never return");
                if (colour[d] == 0) { colour[d] = 1; stack.emplace_back(d, W.dep_off[d]); }
            }
        }
    }
    return CX_OK;
}

inline void init_state(const Wiring &W, State &S) {
    S.chunks.assign(W.chunk_off[W.nsig], 0);
    S.flags.assign(W.nsig, 0);
    for (int64_t s = 0; s < W.nsig; s++)
        for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++)
            if (W.dep_inter[p] || W.dep_weak[p]) {
                const int64_t i = p - W.dep_off[s];
                S.chunks[W.chunk_off[s] + (i >> 4)] |= ((W.dep_inter[p] ? kInter : 0) | (W.dep_weak[p] ? kWeak : 0)) << ((i & 15) << 2);
            }
    S.hash = 0;
    for (int64_t c = 0; c < (int64_t)S.chunks.size(); c++) S.hash ^= zob_chunk(c, S.chunks[c]);
    for (int64_t s = 0; s < W.nsig; s++) S.hash ^= zob_flag(s, 0);
}

inline void put_chunk(State &S, int64_t c, uint64_t value) {
    const uint64_t old = S.chunks[c];
    if (old == value) return;
    S.hash ^= zob_chunk(c, old) ^ zob_chunk(c, value);
    S.chunks[c] = value;
}
inline void put_flags(State &S, int64_t s, uint8_t f) {
    const uint8_t old = S.flags[s];
    if (old == f) return;
    S.hash ^= zob_flag(s, old) ^ zob_flag(s, f);
    S.flags[s] = f;
}

// set_value!, signal.jl:232-253 (+ notify_listener!, :339-356)
inline void set_value(const Wiring &W, State &S, int64_t s) {
    for (int64_t c = W.chunk_off[s]; c < W.chunk_off[s + 1]; c++) put_chunk(S, c, S.chunks[c] & ~kAllFresh);      // unset_all_dependencies_fresh!, :653-655
    put_flags(S, s, kComputed);                                                                                // props = (false, false)
    for (int64_t q = W.lis_off[s]; q < W.lis_off[s + 1]; q++) {
        const int64_t l = W.lis[q], i = W.lis_idx[q];
        if (W.lis_listen[q]) put_flags(S, l, (uint8_t)((S.flags[l] & kComputed) | kPot));                      // listening: potentially pending, not pending
        const int64_t c = W.chunk_off[l] + (i >> 4);
        put_chunk(S, c, S.chunks[c] | ((kFresh | kComp) << ((i & 15) << 2)));
    }
}

// is_meeting_pending_criteria, signal.jl:668-730: every dependency Computed & (Weak | Fresh), chunk-parallel, the last chunk padded with ones
inline bool meets_pending_criteria(const Wiring &W, const State &S, int64_t s) {
    const int64_t n = W.dep_off[s + 1] - W.dep_off[s];
    if (n == 0) return false;
    const int64_t c0 = W.chunk_off[s], nc = W.chunk_off[s + 1] - c0;
    for (int64_t c = 0; c + 1 < nc; c++) {
        const uint64_t ch = S.chunks[c0 + c];
        if ((((ch & kAllComp) >> 2) & (((ch & kAllWeak) >> 1) | ((ch & kAllFresh) >> 3))) != kAllPass) return false;
    }
    const int off = (int)(((n - 1) & 15) << 2);
    const uint64_t fill = off + 4 >= 64 ? 0 : (~0ull << (off + 4));       // Julia's UInt64 << 64 is 0
    const uint64_t ch = S.chunks[c0 + nc - 1] | fill;
    return (((ch & kAllComp) >> 2) & (((ch & kAllWeak) >> 1) | ((ch & kAllFresh) >> 3))) == kAllPass;
}

// is_pending, signal.jl:141-154: the cached answer, or a lazy evaluation when potentially pending
inline bool is_pending(const Wiring &W, State &S, int64_t s) {
    const uint8_t f = S.flags[s];
    if (f & kPend) return true;
    if (f & kPot) {
        const bool p = meets_pending_criteria(W, S, s);
        put_flags(S, s, (uint8_t)((f & kComputed) | (p ? kPend : 0)));
        return p;
    }
    return false;
}

// ---- update_marginals!, inference_engine.jl:559-632, on the shadow: records the executions instead of computing ---------------------
struct Call {
    std::vector<int32_t> order;         // signals in execution order
    std::vector<int32_t> round_of;      // the pass each execution belongs to (0-based; the final marginal round is the last)
    int64_t rounds = 0;
};

struct Runner {
    const Wiring &W; State &S; Call &out; int32_t round = 0; int32_t bad = -1;
    bool f(int64_t d) {                 // the closure of process_inference_request, inference_engine.jl:512-525
        if (!is_pending(W, S, d)) return false;
        if (W.is_f2v(d) && W.vrule[d - W.ne] == kRuleNone && bad < 0) bad = (int32_t)d;
        out.order.push_back((int32_t)d); out.round_of.push_back(round);
        set_value(W, S, d);             // compute! = rule + set_value!, signal.jl:392-410
        return true;
    }
    bool process_dependencies(int64_t s) {      // signal.jl:466-490, retry = true
        bool any = false;
        for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++) {
            const int64_t d = W.dep[p];
            bool processed = f(d);
            if (!processed && W.dep_inter[p]) {
                const bool sub = process_dependencies(d);
                if (sub) processed = f(d);
                any = any || sub;
            }
            any = any || processed;
        }
        return any;
    }
};

// req: local variable numbers in request order.  Returns -1, or the MessageToVariable signal the device has no rule for.
inline int32_t update_marginals(const Wiring &W, State &S, const int32_t *req, int64_t n, Call &out) {
    out = Call();
    // request_inference_for, inference_engine.jl:298-323: the dependencies of the marginal and the variable's linked signals become
    // potentially pending
    const bool links = !W.link_off.empty();
    for (int64_t i = 0; i < n; i++) {
        const int64_t m = W.sig_marg(req[i]);
        for (int64_t p = W.dep_off[m]; p < W.dep_off[m + 1]; p++) { const int64_t d = W.dep[p]; put_flags(S, d, (uint8_t)((S.flags[d] & kComputed) | kPot)); }
        if (links) for (int64_t q = W.link_off[req[i]]; q < W.link_off[req[i] + 1]; q++) { const int64_t l = W.link[q]; put_flags(S, l, (uint8_t)((S.flags[l] & kComputed) | kPot)); }
    }
    std::vector<uint8_t> ready(n, 0);
    Runner R{W, S, out};
    bool cont = true, reverse = false;
    while (cont) {
        cont = false;
        for (int64_t k = 0; k < n; k++) {
            const int64_t i = reverse ? n - 1 - k : k;
            if (ready[i]) continue;
            const int64_t m = W.sig_marg(req[i]);
            const bool did = R.process_dependencies(m);
            if (is_pending(W, S, m)) ready[i] = 1;
            cont = cont || did;
        }
        reverse = !reverse;
        R.round++;
    }
    // the final round, :610-628: per requested variable its marginal, then its linked signals, each if pending
    for (int64_t i = 0; i < n; i++) {
        const int64_t m = W.sig_marg(req[i]);
        if (is_pending(W, S, m)) { out.order.push_back((int32_t)m); out.round_of.push_back(R.round); set_value(W, S, m); }
        if (links)
            for (int64_t q = W.link_off[req[i]]; q < W.link_off[req[i] + 1]; q++) {
                const int64_t l = W.link[q];
                if (is_pending(W, S, l)) { out.order.push_back((int32_t)l); out.round_of.push_back(R.round); set_value(W, S, l); }
            }
    }
    out.rounds = R.round + 1;
    return R.bad;
}

// ---- the recorded executions as stages of device items -------------------------------------------------------------------------------
// internal item kinds of the reference plans (cx_batch.hip: batch_item).  64-66, 72: the value is the SUM of a list of sources — the
// dependencies in the reference's order, exactly what its rule call folds (`reduce(product, get_value.(deps))` in natural form) — for
// the signals of variables of degree > 5, whose dependencies are segment-tree nodes, and for every product under a user wiring: a list
// entry >= 0 is a factor→variable slot, ~entry the index of a node in the product store.  rec = {kind, destination, variable, first list
// entry, entries}.  72: the marginal of a precision variable, stored as Gamma(shape, scale).  67-71: the variational rules (kRule*) of a
// CX_FACTOR_NORMAL_PRECISION factor; list entries: marginal = local variable, message = slot, joint = index in the joint store
//   67 {slot, variable, [marginal other, marginal precision]}     68 {slot, variable, [marginal a, marginal b]}
//   69 {slot, variable, [message slot other, marginal precision]} 70 {joint index, -, [message slot a, message slot b, marginal precision]}   (a, b by ascending variable id)
//   71 {slot, variable, [joint index]}
constexpr int32_t kItemSumToFactor = 64, kItemSumToProduct = 65, kItemSumToMarginal = 66, kItemMfNormal = 67, kItemMfGamma = 68, kItemStNormal = 69,
                  kItemVmpJoint = 70, kItemStGamma = 71, kItemSumToGammaMarginal = 72;

// a list item of more than kWideList sources (a flat product over a hub's messages: the marginal of a precision under a mean-field wiring
// lists every factor) leaves the stage's thread-per-item launch: one workgroup sums it (cx_batch.hip: k_wide_sum)
constexpr int64_t kWideList = 1024;

// Two executions as ONE item of a stage.  On a chain or a grid the reference's order alternates MessageToFactor(x, f) and
// MessageToVariable(y, f): the second reads the first and nothing else that this call computes, so its stage would be the first's plus
// one — half of a plan's depth is such pairs.  The pair is levelled as one item instead: the same thread computes the first, waits for its
// store, and computes the second (cx_batch.hip: the record that LEADS is followed, in the stage's list, by the record that FOLLOWS; the
// follower's own thread skips it).  The reference's order is kept (the second still comes after the first) and every value read is the one
// the reference's rule call read; a pair is only formed when no earlier execution of the call reads or writes the second's slot in that
// stage or later.
constexpr int32_t kRecLeads = 0x40000000, kRecFollows = 0x20000000, kRecKindMask = 0x0fffffff;

// A path inside a plan as ONE step (round 6).  On a state-space model the reference's call is a forward and a backward chain of pairs —
// MessageToFactor(x_t, f_t) then MessageToVariable(x_t+1, f_t), each pair reading the one before it — and levelling gives every pair a
// stage of its own: the wired variational SSM of the reference's tests at n = 10^5 is 100,001 stages of two pairs and a handful of
// dependants, 1.1 us each.  But a pair is a projective-linear map of its incoming message (cx_chain.hip), so a chain of pairs whose
// OTHER inputs are settled before it starts is a prefix scan.  level() finds such chains (kChainMin pairs or more), checks that nothing
// of the call stands between their executions (below), gives ALL their executions the stage of the first pair, levels everything
// else again around them, and hands the chains to the caller as scan steps: links in chain order with the list of settled sources
// each leader adds to the message that travels.  The executions, their order and the values they read are the reference's; only the
// association of the arithmetic differs (a scan composes maps where the sequence applies them one by one: equal to rounding).
struct ScanStep { int32_t stage; int64_t lo, hi; };      // beside the items of stage `stage` (1-based): links [lo, hi)
constexpr int64_t kChainMin = 128;
constexpr int64_t kScanStepMaxLinks = (int64_t)2048 * 1024;      // cx_planscan.hip: every workgroup of a step composes the totals of the tiles before it
struct Plan {
    std::vector<int32_t> rec;            // 5 per item, by stage
    std::vector<int64_t> stage_off;
    std::vector<int32_t> wide_rec;       // the wide list items, 5 per item, by stage
    std::vector<int64_t> wide_off;       // [stages + 1]
    std::vector<int32_t> list;           // sources of the list items
    int64_t n_exec = 0, n_messages = 0, n_marginals = 0, n_products = 0, n_joints = 0, rounds = 0;
    // the chains that run as scans: per link the leader's MessageToFactor slot and variable, the follower's MessageToVariable slot, the
    // precision variable whose marginal the follower's rule reads (-1: the sum-product rule of a pairwise factor), the first link of a chain,
    // and the leader's settled sources (a factor→variable slot, or ~index of a node of the product store) in the reference's fold order
    std::vector<ScanStep> scans;
    std::vector<int32_t> sl_lead_dst, sl_lead_var, sl_fol_dst, sl_prec, sl_src_off, sl_src;
    std::vector<uint8_t> sl_head;
    int64_t n_chain_exec = 0;            // executions that run inside scan steps (not in `rec`)
};

// prod_slot(i): the index of segment-tree node i in the handle's product store; joint_slot(f): of factor f's joint marginal in the joint
// store (the caller registers both)
template <class H, class ProdSlot, class JointSlot>
int32_t level(const H *h, const Wiring &W, const Call &call, ProdSlot &&prod_slot, JointSlot &&joint_slot, Plan &P, std::string &err, int64_t wide_list = kWideList,
              int64_t chain_min = -1) {
    const int64_t ne = W.ne, n = (int64_t)call.order.size();
    P = Plan();
    P.n_exec = n; P.rounds = call.rounds;
    std::vector<int32_t> w_stage(W.nsig, 0), r_stage(W.nsig, 0), stage(n, 0);
    std::vector<int32_t> w_exec(W.nsig, -1), follower(n, -1);      // the execution that last wrote a signal in this call; the execution fused behind one
    std::vector<uint8_t> follows(n, 0);
    int32_t n_stages = 0;
    if (chain_min < 0) { const char *e = std::getenv("CX_REF_CHAIN_MIN"); chain_min = e ? std::atoll(e) : kChainMin; }      // (0: no chains as scans — A/B, tests)
    auto is_wide = [&](int64_t s) { return !W.is_f2v(s) && !W.is_joint(s) && W.dep_off[s + 1] - W.dep_off[s] > wide_list; };
    // the MessageToFactor signal a MessageToVariable execution can be fused behind: the one message its rule reads (a pairwise sum-product
    // factor; the structured variational rule, which reads one message and the precision's marginal); -1: none
    auto pair_source = [&](int64_t s) -> int64_t {
        if (!W.is_f2v(s)) return -1;
        const int64_t e = s - ne;
        const int32_t f = W.efac[e];
        if (W.vrule[e] == kRuleBP) {
            if (W.foff[f + 1] - W.foff[f] != 2 || h->partner[flat::slot_of_edge_t(h, e)] < 0) return -1;
            return W.fedge[W.foff[f]] == e ? W.fedge[W.foff[f] + 1] : W.fedge[W.foff[f]];
        }
        if (W.vrule[e] == kRuleStNormal)
            for (int32_t k = W.foff[f]; k < W.foff[f + 1]; k++) { const int32_t e2 = W.fedge[k]; if (e2 != e && h->np_role[e2] != CX_ROLE_PRECISION) return e2; }
        return -1;
    };
    // dim 2 .. 4 (cx_mvbatch.hip: k_batch_mv): the compact items — records {kind, slot | variable, variable, rule table of the sending
    // slot} — and the list sums of the segment-tree signals (degrees above 5); the default wiring only, and no pairs (that kernel takes one
    // record per thread)
    const bool mv = h->cfg.dim > 1;
    static const bool fuse_env = [] { const char *e = std::getenv("CX_REF_FUSE_PAIRS"); return !(e && e[0] == '0'); }();      // (A/B)
    const bool fuse_pairs = fuse_env && !mv;
    // chains as scans: scalar Gaussian messages (the map algebra of cx_chain.hip), plans of a size whose bookkeeping (three ints per execution) is cheap
    const bool contract = fuse_pairs && chain_min > 0 && n >= 2 * chain_min && n <= ((int64_t)1 << 23) && h->cfg.family != CX_FAMILY_NATURAL2;
    std::vector<int32_t> prev_w, prev_r, cpred, leader_of, member;      // member[i] >= 0: execution i runs inside a scan step of that stage (second pass)
    if (contract) { prev_w.assign(n, 0); prev_r.assign(n, 0); cpred.assign(n, -1); leader_of.assign(n, -1); }
    bool second_pass = false;
    // what an execution reads beyond its dependency list: a sum-product rule reads the stored messages of ALL the factor's other edges
    auto unlisted_reads = [&](int64_t s, auto &&fn) {
        if (!W.is_f2v(s) || W.vrule[s - ne] != kRuleBP) return;
        const int32_t f = W.efac[s - ne];
        if (W.foff[f + 1] - W.foff[f] <= 2 && !W.custom) return;      // (pairwise, default wiring: the other edge is the dependency)
        for (int32_t k = W.foff[f]; k < W.foff[f + 1]; k++) if (W.fedge[k] != s - ne) fn((int64_t)W.fedge[k]);
    };
    auto run_pass = [&]() {
    std::fill(w_stage.begin(), w_stage.end(), 0); std::fill(r_stage.begin(), r_stage.end(), 0); std::fill(stage.begin(), stage.end(), 0);
    std::fill(w_exec.begin(), w_exec.end(), -1); std::fill(follower.begin(), follower.end(), -1); std::fill(follows.begin(), follows.end(), 0);
    n_stages = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t s = call.order[i];
        // (the executions wander over tables of tens of millions of signals: what the ones to come will read is asked for ahead)
        if (i + 24 < n) { const int64_t a = call.order[i + 24]; __builtin_prefetch(&W.dep_off[a]); __builtin_prefetch(&w_stage[a]); __builtin_prefetch(&r_stage[a]); }
        if (i + 12 < n) { const int64_t a = call.order[i + 12]; __builtin_prefetch(&W.dep[W.dep_off[a]]); }
        if (i + 6 < n) {
            const int64_t a = call.order[i + 6];
            for (int64_t p = W.dep_off[a]; p < W.dep_off[a + 1] && p < W.dep_off[a] + 6; p++) { __builtin_prefetch(&w_stage[W.dep[p]]); __builtin_prefetch(&r_stage[W.dep[p]]); }
        }
        if (second_pass && member[i] >= 0) {      // inside a scan step: every execution of the chain at the stage of its first pair
            const int32_t stm = member[i];
            const int64_t srcm = pair_source(s);
            stage[i] = stm; w_stage[s] = stm; w_exec[s] = (int32_t)i;
            for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++) r_stage[W.dep[p]] = std::max(r_stage[W.dep[p]], stm);
            if (srcm >= 0) r_stage[srcm] = std::max(r_stage[srcm], stm);
            n_stages = std::max(n_stages, stm);
            continue;
        }
        if (contract && !second_pass) { prev_w[i] = w_stage[s]; prev_r[i] = r_stage[s]; }
        int32_t st = std::max(w_stage[s], r_stage[s]) + 1;
        const int64_t src = fuse_pairs ? pair_source(s) : -1;
        if (src >= 0 && w_exec[src] >= 0 && follower[w_exec[src]] < 0 && !is_wide(src)) {
            // behind the execution that wrote the message it reads, in that execution's stage — if nothing of this call reads or writes the
            // destination there or later, and every other dependency was written before
            const int32_t j = w_exec[src], sy = stage[j];
            bool ok = std::max(w_stage[s], r_stage[s]) < sy;
            for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1] && ok; p++) if (W.dep[p] != src) ok = w_stage[W.dep[p]] < sy;
            if (ok) { st = sy; follower[j] = (int32_t)i; follows[i] = 1; if (contract && !second_pass) leader_of[i] = j; }
        }
        if (!follows[i]) {
            for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++) st = std::max(st, w_stage[W.dep[p]] + 1);
            unlisted_reads(s, [&](int64_t d) { st = std::max(st, w_stage[d] + 1); });
            // (chains) a MessageToFactor whose stage is set by ONE message a fused pair of this call stored: that pair comes before it in a chain
            if (contract && !second_pass && W.is_v2f(s))
                for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++) {
                    const int32_t k = w_exec[W.dep[p]];
                    if (k >= 0 && follows[k] && stage[k] == st - 1) cpred[i] = cpred[i] == -1 ? k : -2;
                }
        }
        stage[i] = st; w_stage[s] = st; w_exec[s] = (int32_t)i;
        for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++) r_stage[W.dep[p]] = std::max(r_stage[W.dep[p]], st);
        if (src >= 0) r_stage[src] = std::max(r_stage[src], st);      // (a rule reads its message whether the wiring lists it or not)
        unlisted_reads(s, [&](int64_t d) { r_stage[d] = std::max(r_stage[d], st); });
        n_stages = std::max(n_stages, st);
    }
    };
    run_pass();
    // ---- chains of pairs as scan steps --------------------------------------------------------------------------------------------------
    struct ChainRun { int32_t stage; std::vector<int32_t> leaders; };
    std::vector<ChainRun> chain_runs;
    if (contract) {
        const std::vector<int32_t> w_final(w_stage);      // the stage of the LAST write of every signal in this call
        std::vector<int32_t> succ(n, -1), predp(n, -1);
        for (int64_t j = 0; j < n; j++) {
            if (follower[j] < 0 || cpred[j] < 0) continue;
            const int32_t lp = leader_of[cpred[j]];
            if (lp >= 0 && succ[lp] < 0) { succ[lp] = (int32_t)j; predp[j] = lp; }
        }
        member.assign(n, -1);
        for (int64_t j0 = 0; j0 < n; j0++) {
            if (follower[j0] < 0 || predp[j0] >= 0) continue;
            std::vector<int32_t> c;
            for (int32_t j = (int32_t)j0; j >= 0; j = succ[j]) c.push_back(j);
            if ((int64_t)c.size() < chain_min || (int64_t)c.size() > kScanStepMaxLinks) continue;
            // Nothing of the call may stand between the chain's executions: what they read from outside is last written before the first
            // pair's stage (so no execution that depends on the chain feeds it), and what they overwrite was last read and written before
            // it (so nobody waits for the OLD value while the chain runs ahead).  Then every other execution either comes before the
            // chain or can come after ALL of it.
            const int32_t s0 = stage[c[0]];
            bool ok = true;
            for (size_t t = 0; t < c.size() && ok; t++) {
                const int32_t Lx = c[t], Fx = follower[Lx];
                const int64_t sL = call.order[Lx], sF = call.order[Fx], chain_in = t ? (int64_t)call.order[follower[c[t - 1]]] : -1;
                ok = prev_w[Lx] < s0 && prev_r[Lx] < s0 && prev_w[Fx] < s0 && prev_r[Fx] < s0 && !is_wide(sL);
                // (the slot must take part in the sum: the leader of an observed variable stores nothing — not a link)
                ok = ok && !(h->vinfo[h->edge_var[sL]] & (cx::kClamped | cx::kGhost));
                for (int64_t p = W.dep_off[sL]; p < W.dep_off[sL + 1] && ok; p++) if (W.dep[p] != chain_in) ok = w_final[W.dep[p]] < s0;
                for (int64_t p = W.dep_off[sF]; p < W.dep_off[sF + 1] && ok; p++) if (W.dep[p] != sL) ok = w_final[W.dep[p]] < s0;
                if (ok && t) {      // the message that travels must be a listed source of the next leader
                    bool listed = false;
                    for (int64_t p = W.dep_off[sL]; p < W.dep_off[sL + 1]; p++) listed = listed || W.dep[p] == chain_in;
                    ok = listed;
                }
            }
            if (!ok) continue;
            for (int32_t Lx : c) { member[Lx] = s0; member[follower[Lx]] = s0; }
            chain_runs.push_back(ChainRun{s0, std::move(c)});
        }
        if (!chain_runs.empty()) {
            // the pairs keep their records' order; everything else is levelled again around the chains
            std::vector<int32_t> fol1(follower);
            second_pass = true;
            run_pass();
            for (auto &cr : chain_runs) for (int32_t Lx : cr.leaders) { follower[Lx] = -1; (void)fol1; }
            // stages that the chains' executions left empty disappear
            std::vector<int32_t> used(n_stages + 1, 0), remap(n_stages + 1, 0);
            for (int64_t i = 0; i < n; i++) used[stage[i]] = 1;
            int32_t m = 0;
            for (int32_t st = 1; st <= n_stages; st++) if (used[st]) remap[st] = ++m;
            for (int64_t i = 0; i < n; i++) stage[i] = remap[stage[i]];
            for (auto &cr : chain_runs) cr.stage = remap[cr.stage];
            n_stages = m;
            // the links, by stage and chain
            std::stable_sort(chain_runs.begin(), chain_runs.end(), [](const ChainRun &x, const ChainRun &y) { return x.stage < y.stage; });
            P.sl_src_off.push_back(0);
            for (auto &cr : chain_runs) {
                if (P.scans.empty() || P.scans.back().stage != cr.stage || P.scans.back().hi - P.scans.back().lo + (int64_t)cr.leaders.size() > kScanStepMaxLinks) P.scans.push_back(ScanStep{cr.stage, (int64_t)P.sl_head.size(), (int64_t)P.sl_head.size()});
                for (size_t t = 0; t < cr.leaders.size(); t++) {
                    const int32_t Lx = cr.leaders[t], Fx = fol1[Lx];
                    const int64_t sL = call.order[Lx], sF = call.order[Fx], eF = sF - ne, chain_in = t ? (int64_t)call.order[fol1[cr.leaders[t - 1]]] : -1;
                    P.sl_lead_dst.push_back(flat::slot_of_edge_t(h, sL)); P.sl_lead_var.push_back(h->edge_var[sL]);
                    P.sl_fol_dst.push_back(flat::slot_of_edge_t(h, eF));
                    int32_t prec = -1;
                    if (W.vrule[eF] == kRuleStNormal) {
                        const int32_t f = W.efac[eF];
                        for (int32_t k = W.foff[f]; k < W.foff[f + 1]; k++) if (h->np_role[W.fedge[k]] == CX_ROLE_PRECISION) prec = h->edge_var[W.fedge[k]];
                    }
                    P.sl_prec.push_back(prec);
                    P.sl_head.push_back(t == 0 ? 1 : 0);
                    for (int64_t p = W.dep_off[sL]; p < W.dep_off[sL + 1]; p++) {
                        const int64_t d = W.dep[p];
                        if (d == chain_in) continue;
                        P.sl_src.push_back(W.is_f2v(d) ? flat::slot_of_edge_t(h, d - ne) : ~(int32_t)prod_slot(d - W.sig_prod(0)));
                    }
                    P.sl_src_off.push_back((int32_t)P.sl_src.size());
                    P.n_messages += 2; P.n_chain_exec += 2;
                }
                P.scans.back().hi = (int64_t)P.sl_head.size();
            }
        } else member.clear();
    }
    const bool have_chains = !chain_runs.empty();
    auto in_chain = [&](int64_t i) { return have_chains && member[i] >= 0; };
    // (only products — MessageToFactor, marginals, segment-tree nodes — can be wide: the rules read at most three sources)
    P.stage_off.assign(n_stages + 1, 0); P.wide_off.assign(n_stages + 1, 0);
    int64_t n_wide = 0;
    for (int64_t i = 0; i < n; i++) { if (in_chain(i)) continue; if (is_wide(call.order[i])) { P.wide_off[stage[i]]++; n_wide++; } else P.stage_off[stage[i]]++; }
    for (int32_t s = 0; s < n_stages; s++) { P.stage_off[s + 1] += P.stage_off[s]; P.wide_off[s + 1] += P.wide_off[s]; }
    if (mv && n_wide) return fail_(err, CX_ERR_UNSUPPORTED, "reference schedule, dim > 1: a product of more than 1,024 sources is summed by a workgroup for scalar messages only");
    P.rec.assign(5 * (n - n_wide - P.n_chain_exec), 0); P.wide_rec.assign(5 * n_wide, 0);
    std::vector<int64_t> fill(P.stage_off.begin(), P.stage_off.end() - 1), wfill(P.wide_off.begin(), P.wide_off.end() - 1);
    auto source = [&](int64_t d) -> int32_t {       // a dependency of a list item as a list entry
        if (W.is_f2v(d)) return flat::slot_of_edge_t(h, d - ne);
        return ~(int32_t)prod_slot(d - W.sig_prod(0));
    };
    const bool gammas = !h->var_gamma.empty();
    // in execution order, a follower right behind its leader (both fill the same stage's list, one after the other)
    std::vector<int32_t> emit_order;
    emit_order.reserve(n);
    for (int64_t i = 0; i < n; i++) { if (follows[i] || in_chain(i)) continue; emit_order.push_back((int32_t)i); if (follower[i] >= 0) emit_order.push_back(follower[i]); }
    for (int64_t q = 0; q < (int64_t)emit_order.size(); q++) {
        const int64_t i = emit_order[q];
        const int64_t s = call.order[i];
        int32_t *r = is_wide(s) ? &P.wide_rec[5 * wfill[stage[i] - 1]++] : &P.rec[5 * fill[stage[i] - 1]++];
        auto list_of_deps = [&]() {
            r[3] = (int32_t)P.list.size(); r[4] = (int32_t)(W.dep_off[s + 1] - W.dep_off[s]);
            for (int64_t p = W.dep_off[s]; p < W.dep_off[s + 1]; p++) P.list.push_back(source(W.dep[p]));
        };
        if (W.is_v2f(s)) {                               // MessageToFactor
            const int32_t v = h->edge_var[s], deg = h->var_off[v + 1] - h->var_off[v], slot = flat::slot_of_edge_t(h, s);
            P.n_messages++;
            if (deg <= 5 && !W.custom) { r[0] = CX_ITEM_MESSAGE_TO_FACTOR; r[1] = slot; r[2] = v; }
            else { r[0] = kItemSumToFactor; r[1] = slot; r[2] = v; list_of_deps(); }
        } else if (W.is_f2v(s)) {                        // MessageToVariable
            const int64_t e = s - ne;
            const int32_t slot = flat::slot_of_edge_t(h, e), v = h->edge_var[e];
            const uint8_t rule = W.vrule[e];
            P.n_messages++;
            if (rule >= kRuleMfNormal) {
                // the variational rules read exactly their dependencies; entries in a fixed order, whatever order the wiring listed them in
                const int32_t f = W.efac[e];
                int32_t e_other = -1, e_other2 = -1, e_prec = -1;
                for (int32_t k = W.foff[f]; k < W.foff[f + 1]; k++) {
                    const int32_t e2 = W.fedge[k];
                    if (e2 == e) continue;
                    if (h->np_role[e2] == CX_ROLE_PRECISION) e_prec = e2; else if (e_other < 0) e_other = e2; else e_other2 = e2;
                }
                r[1] = slot; r[2] = v; r[3] = (int32_t)P.list.size();
                if (rule == kRuleMfNormal) { r[0] = kItemMfNormal; r[4] = 2; P.list.push_back(h->edge_var[e_other]); P.list.push_back(h->edge_var[e_prec]); }
                else if (rule == kRuleMfGamma) { r[0] = kItemMfGamma; r[4] = 2; P.list.push_back(h->edge_var[e_other]); P.list.push_back(h->edge_var[e_other2]); }
                else if (rule == kRuleStNormal) { r[0] = kItemStNormal; r[4] = 2; P.list.push_back(flat::slot_of_edge_t(h, e_other)); P.list.push_back(h->edge_var[e_prec]); }
                else { r[0] = kItemStGamma; r[4] = 1; P.list.push_back((int32_t)joint_slot(f)); }
            }
            else if (rule == kRuleBP && !h->slot_kary.empty() && h->slot_kary[slot] >= 0) {
                r[0] = 32; r[1] = h->slot_kary[slot];      // kItemKaryEntry (cx_kary_core.h; dim 2 .. 4: cx_kary_mv_core.h)
            }
            else if (rule == kRuleBP && h->partner[slot] >= 0) { r[0] = CX_ITEM_MESSAGE_TO_VARIABLE; r[1] = slot; r[2] = v; if (mv) { r[3] = h->spdir[h->partner[slot]]; r[4] = h->partner[slot] + 1; } }      // (dim > 1: the rule table of the sending slot, and the sending slot itself + 1 — one look-up less on the chain)
            else return fail_(err, CX_ERR_UNSUPPORTED, "reference schedule: the message from factor " + std::to_string(h->edge_fac_id[e]) + " to variable " +
                                                           std::to_string(h->var_ids[v]) + " is pending, and the factor has no rule on the device for it "
                                                           "(an opaque factor of two or more variables: the reference's processor would raise, inference_engine.jl:358; "
                                                           "a CX_FACTOR_NORMAL_PRECISION factor: wire its variational dependencies, cx_graph_wire)");
        } else if (W.is_marg(s)) {                       // IndividualMarginal
            const int32_t v = (int32_t)(s - 2 * ne), deg = h->var_off[v + 1] - h->var_off[v];
            const bool gamma = gammas && h->var_gamma[v];
            P.n_marginals++;
            if (deg <= 5 && !W.custom && !gamma) { r[0] = CX_ITEM_INDIVIDUAL_MARGINAL; r[1] = v; r[2] = v; }
            else { r[0] = gamma ? kItemSumToGammaMarginal : kItemSumToMarginal; r[1] = v; r[2] = v; list_of_deps(); }
        } else if (W.is_joint(s)) {                      // JointMarginal of a NORMAL_PRECISION factor
            const int64_t f = s - W.sig_joint(0);
            if (!W.jrule[f]) return fail_(err, CX_ERR_UNSUPPORTED, "reference schedule: a JointMarginal without a rule is pending");
            int32_t e_a = -1, e_b = -1, e_prec = -1;
            for (int32_t k = W.foff[f]; k < W.foff[f + 1]; k++) {
                const int32_t e2 = W.fedge[k];
                if (h->np_role[e2] == CX_ROLE_PRECISION) e_prec = e2; else if (e_a < 0) e_a = e2; else e_b = e2;
            }
            P.n_joints++;
            r[0] = kItemVmpJoint; r[1] = (int32_t)joint_slot(f); r[3] = (int32_t)P.list.size(); r[4] = 3;
            P.list.push_back(flat::slot_of_edge_t(h, e_a)); P.list.push_back(flat::slot_of_edge_t(h, e_b)); P.list.push_back(h->edge_var[e_prec]);
        } else {                                         // ProductOfMessages
            const int64_t pi = s - W.sig_prod(0);
            P.n_products++;
            r[0] = kItemSumToProduct; r[1] = (int32_t)prod_slot(pi); r[2] = W.prods[pi].var; list_of_deps();
        }
        if (follower[i] >= 0) r[0] |= kRecLeads;
        if (follows[i]) r[0] |= kRecFollows;
    }
    return CX_OK;
}

// the signal of a batch item / a message as the shadow numbers it: -1 when there is none (a ProductOfMessages range that is not a node of
// the variable's segment tree)
inline int64_t find_prod(const Wiring &W, int32_t var, int32_t lo, int32_t hi) {
    for (int64_t i = 0; i < (int64_t)W.prods.size(); i++)      // (hubs are rare; a map would pay only for graphs of many of them)
        if (W.prods[i].var == var && W.prods[i].lo == lo && W.prods[i].hi == hi) return W.sig_prod(i);
    return -1;
}

}  // namespace refsched
}  // namespace cx
