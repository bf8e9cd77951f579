// cx_chain64_plan.h — the work plan of the chain-scan schedule for wide messages (dim 64; cx_mv64chain.hip executes it).
// Pure host C++: no HIP, no device pointers — it compiles with gcc under -fsanitize=address,undefined and is driven from a
// CPU test (tests/test_chain64_plan.py executes the records with numpy against the exact smoother).
//
// What ONE update_marginals! of the reference computes on a chain is the exact forward/backward pass
// (/root/reference/src/inference_engine.jl:575-608; the SSM of test/inference_engine_tests.jl:436-487).  Sequentially that is
// T dependent 64 x 64 rules per direction; here the chain is cut into blocks whose POTENTIALS are composed first:
//
//   A segment [a, b] of a path (positions a..b, links a..b-1) has the pairwise potential of its two end variables with
//   everything in between summed out — interior side information (likelihoods, priors) included, the ends' excluded:
//       psi(x_a, x_b) = exp( -1/2 x_a' P x_a - 1/2 x_b' C x_b + x_b' B x_a + h' x_a + c' x_b ).
//   A single link is the factor's rule table (P, B, C) with h = c = 0.  Two adjacent segments compose over their joint m
//   (side information (eta_m, Lambda_m)):  M = C1 + Lambda_m + P2,  g = c1 + eta_m + h2,
//       P = P1 - B1' M^-1 B1,  B = B2 M^-1 B1,  C = C2 - B2 M^-1 B2',  h = h1 + B1' M^-1 g,  c = c2 + B2 M^-1 g.
//   The SAME potential serves both directions: forwards it is the rule (P, B, C, h, c) on the message entering a, backwards
//   the rule (C, B', P, c, h) on the message entering b — one composition tree for both passes.
//
// Plan: level 0 = blocks of K0 consecutive links; level j + 1 = groups of FAN level-j potentials, while a level has more than
// FAN of them.  Launches, in order: compose level 0, 1, ... (a wave per potential, its children composed left to right);
// then walks (a wave per job, its steps in sequence, every step one rule application — the device runs step s of all jobs of a
// walk launch as one kernel launch): the top level of every path from its two ends, the groups of each lower level (each walk
// hands every child the message that enters it), and finally every level-0 block along its links, forwards and backwards,
// writing the messages into their factor→variable slots.
//
// Records name their operands by HANDLES (space << 56 | offset in doubles); the device resolves them against the base
// pointers of the moment, so a plan survives reallocation and can be executed anywhere.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace cx {
namespace plan64 {

enum Space : int64_t { kZero = 0, kF2V = 1, kPtab = 2, kBtab = 3, kPot = 4, kEnt = 5, kAux = 6, kSpaces = 7 };
constexpr int64_t kOffMask = ((int64_t)1 << 56) - 1;
inline int64_t H(Space s, int64_t off) { return ((int64_t)s << 56) | off; }

struct Child {          // one operand of a composition
    int64_t P, B, Bt, C, h, c;      // the potential's six parts (Bt = B')
    int64_t side[3];                // side information of the joint BEFORE this child (unused for a job's first child)
    int64_t pad;
};
struct Step {           // one rule application of a walk
    int64_t src[3];                 // what is summed into the rule's input: the entering message and/or side information
    int64_t P, Bt, C, h, c;         // the rule: Lambda_out = C - B M^-1 B', eta_out = c + B M^-1 (eta_in + h), M = Lambda_in + P
    int64_t dst;
    int64_t pad;
};
struct Job { int64_t out; int32_t first, n; };      // compose: children [first, first + n) -> potential `out`; walk: steps [first, first + n)

struct Input {
    int d = 64;
    int64_t npos = 0, nlinks = 0;
    // per link (path order; the right end of link l is position link_pos[l] + 1)
    const int32_t *link_pos = nullptr, *from = nullptr, *to = nullptr, *tab_fwd = nullptr, *tab_bwd = nullptr;
    const uint8_t *head_fwd = nullptr, *head_bwd = nullptr;     // first / last link of its path
    // per position: up to three slots of side information, -1 = none
    const int32_t *side = nullptr;      // [npos][3]
    // per position (optional): the index of a message in the "aux" arena that holds the position's side information already summed (the
    // caller sums it before every sweep: a position with three or more side slots), -1: the slots of `side` as they are
    const int32_t *side_aux = nullptr;  // [npos]
    int K0 = 0;                         // links per level-0 block (0: chosen from the chain length and `lanes`)
    int fan = 2;                        // potentials per group of the upper levels
    int64_t lanes = 1024;               // waves the composition launch should fill (SIMDs of the device)
    bool root = false;                  // every path also gets the ONE potential of its two end variables (a time block of a partitioned
                                        // chain hands it to the other blocks: cx_chain_block_maps); the tree then has a single top segment
};

struct Plan {
    int d = 64;
    int64_t msg = 0, pot = 0;           // doubles per message record / potential record
    int64_t n_pot = 0, n_ent = 0;       // records in the potential arena / the entry-message arena
    int K0 = 0, fan = 0, levels = 0;
    std::vector<Child> children;
    std::vector<Step> steps;
    std::vector<std::vector<Job>> compose_launches, walk_launches;
    int64_t n_compositions = 0, n_rules = 0;     // pairwise compositions / rule applications per sweep (the MFMA work count)
    std::vector<int64_t> root_pot;               // Input.root: per path, the handle of its whole potential (P part)
    std::vector<int64_t> end_pos;                // Input.root: per path, its first and last position
};

namespace detail {
struct Seg { int64_t pos0, pos1; int64_t pot; int64_t ent_f, ent_b; int64_t in_f, in_b; int64_t child0, nchild; };
}

inline int choose_k0(int64_t nlinks, int64_t lanes) {
    // one block per lane when the chain is long enough; never fewer than 4 links per block (a composition costs 2.5 rules:
    // tiny blocks would spend more on the tree than on the walks)
    int64_t k = (nlinks + lanes - 1) / lanes;
    if (k < 4) k = 4;
    return (int)k;
}

inline Plan build(const Input &in) {
    using detail::Seg;
    if (in.d < 1 || in.fan < 2 || in.npos < 0 || in.nlinks < 0) throw std::invalid_argument("plan64: bad input");
    Plan p;
    p.d = in.d;
    const int64_t dd = (int64_t)in.d * in.d;
    p.msg = in.d + dd;
    p.pot = 4 * dd + 2 * in.d;
    p.fan = in.fan;
    p.K0 = in.K0 > 0 ? in.K0 : choose_k0(in.nlinks, in.lanes);
    const int64_t zero = H(kZero, 0);
    auto pot_part = [&](int64_t id, int part) { return H(kPot, id * p.pot + (part < 4 ? part * dd : 4 * dd + (part - 4) * in.d)); };   // 0 P, 1 B, 2 Bt, 3 C, 4 h, 5 c
    auto ent = [&](int64_t id) { return H(kEnt, id * p.msg); };
    auto slot = [&](int32_t s) { return H(kF2V, (int64_t)s * p.msg); };
    auto sides_of = [&](int64_t pos, int64_t out[3]) {
        int n = 0;
        if (in.side_aux && in.side_aux[pos] >= 0) out[n++] = H(kAux, (int64_t)in.side_aux[pos] * p.msg);
        else
        for (int k = 0; k < 3; k++) { const int32_t s = in.side[3 * pos + k]; if (s >= 0) out[n++] = slot(s); }
        for (int k = n; k < 3; k++) out[k] = zero;
        return n;
    };
    // sources of a rule input: the entering message (unless it is the empty message) + the side information of `pos`
    auto sources = [&](int64_t entering, int64_t pos, int64_t src[3]) {
        int64_t sd[3];
        const int ns = sides_of(pos, sd);
        int n = 0;
        if (entering != zero) src[n++] = entering;
        if (n + ns > 3) throw std::runtime_error("plan64: position " + std::to_string(pos) + " has more than three inputs besides its outgoing link");
        for (int k = 0; k < ns; k++) src[n++] = sd[k];
        for (int k = n; k < 3; k++) src[k] = zero;
    };
    std::vector<std::vector<Job>> down;     // down[j]: walks over the children of level-j segments (j >= 1)
    std::vector<Job> top_jobs, link_jobs;
    int max_levels = 0;
    for (int64_t l0 = 0; l0 < in.nlinks;) {
        if (!in.head_fwd[l0]) throw std::runtime_error("plan64: link " + std::to_string(l0) + " should start a path");
        int64_t l1 = l0;
        while (!in.head_bwd[l1]) { l1++; if (l1 >= in.nlinks) throw std::runtime_error("plan64: a path without an end"); }
        const int64_t n = l1 - l0 + 1, p0 = in.link_pos[l0];
        for (int64_t l = l0; l <= l1; l++)
            if (in.link_pos[l] != p0 + (l - l0)) throw std::runtime_error("plan64: positions of a path must be consecutive");
        // ---- levels -------------------------------------------------------------------------------------------------
        std::vector<std::vector<Seg>> lev(1);
        for (int64_t i = 0; i * p.K0 < n; i++) {
            const int64_t a = i * p.K0, b = std::min<int64_t>(n, a + p.K0);
            lev[0].push_back(Seg{p0 + a, p0 + b, -1, -1, -1, zero, zero, l0 + a, b - a});
        }
        while ((int64_t)lev.back().size() > (in.root ? 1 : in.fan)) {
            const auto &lo = lev.back();
            std::vector<Seg> up;
            for (int64_t i = 0; i < (int64_t)lo.size(); i += in.fan) {
                const int64_t e = std::min<int64_t>((int64_t)lo.size(), i + in.fan);
                up.push_back(Seg{lo[i].pos0, lo[e - 1].pos1, -1, -1, -1, zero, zero, i, e - i});
            }
            lev.push_back(std::move(up));
        }
        const int top = (int)lev.size() - 1;
        max_levels = std::max(max_levels, top + 1);
        const bool composed = in.root || lev[0].size() >= 2;       // a path of ONE block needs no potential at all (unless it is asked for)
        if (composed)
            for (auto &L : lev) for (auto &s : L) { s.pot = p.n_pot++; s.ent_f = p.n_ent++; s.ent_b = p.n_ent++; }
        // ---- compose jobs -------------------------------------------------------------------------------------------
        if (composed) {
            if ((int)p.compose_launches.size() < top + 1) p.compose_launches.resize(top + 1);
            for (const Seg &s : lev[0]) {
                Job j{pot_part(s.pot, 0), (int32_t)p.children.size(), (int32_t)s.nchild};
                for (int64_t l = s.child0; l < s.child0 + s.nchild; l++) {
                    const int64_t t = in.tab_fwd[l];
                    Child c{H(kPtab, t * 3 * dd), H(kPtab, t * 3 * dd + dd), H(kBtab, t * dd), H(kPtab, t * 3 * dd + 2 * dd), zero, zero, {zero, zero, zero}, 0};
                    if (l > s.child0) sides_of(in.link_pos[l], c.side);
                    p.children.push_back(c);
                }
                p.n_compositions += s.nchild - 1;
                p.compose_launches[0].push_back(j);
            }
            for (int lv = 1; lv <= top; lv++)
                for (const Seg &s : lev[lv]) {
                    Job j{pot_part(s.pot, 0), (int32_t)p.children.size(), (int32_t)s.nchild};
                    for (int64_t k = 0; k < s.nchild; k++) {
                        const Seg &ch = lev[lv - 1][s.child0 + k];
                        Child c{pot_part(ch.pot, 0), pot_part(ch.pot, 1), pot_part(ch.pot, 2), pot_part(ch.pot, 3), pot_part(ch.pot, 4), pot_part(ch.pot, 5), {zero, zero, zero}, 0};
                        if (k > 0) sides_of(ch.pos0, c.side);
                        p.children.push_back(c);
                    }
                    p.n_compositions += s.nchild - 1;
                    p.compose_launches[lv].push_back(j);
                }
        }
        if (in.root) { p.root_pot.push_back(pot_part(lev[top][0].pot, 0)); p.end_pos.push_back(p0); p.end_pos.push_back(p0 + n); }
        // ---- walks over potentials: step k applies child k to the message that enters it --------------------------------
        auto fwd_step = [&](const Seg &ch, int64_t dst) {
            Step st{{zero, zero, zero}, pot_part(ch.pot, 0), pot_part(ch.pot, 2), pot_part(ch.pot, 3), pot_part(ch.pot, 4), pot_part(ch.pot, 5), dst, 0};
            sources(ch.in_f, ch.pos0, st.src);
            p.steps.push_back(st);
        };
        auto bwd_step = [&](const Seg &ch, int64_t dst) {      // the same potential read from its other end: (C, B', P, c, h); the transpose of B' is B
            Step st{{zero, zero, zero}, pot_part(ch.pot, 3), pot_part(ch.pot, 1), pot_part(ch.pot, 0), pot_part(ch.pot, 5), pot_part(ch.pot, 4), dst, 0};
            sources(ch.in_b, ch.pos1, st.src);
            p.steps.push_back(st);
        };
        auto walk_children = [&](std::vector<Seg> &kids, int64_t i0, int64_t nk, int64_t in_f, int64_t in_b, std::vector<Job> &jobs) {
            for (int64_t k = 0; k < nk; k++) {
                kids[i0 + k].in_f = k == 0 ? in_f : ent(kids[i0 + k].ent_f);
                kids[i0 + k].in_b = k == nk - 1 ? in_b : ent(kids[i0 + k].ent_b);
            }
            if (nk < 2) return;
            Job jf{0, (int32_t)p.steps.size(), (int32_t)(nk - 1)};
            for (int64_t k = 0; k + 1 < nk; k++) fwd_step(kids[i0 + k], kids[i0 + k + 1].in_f);
            jobs.push_back(jf);
            Job jb{0, (int32_t)p.steps.size(), (int32_t)(nk - 1)};
            for (int64_t k = nk - 1; k >= 1; k--) bwd_step(kids[i0 + k], kids[i0 + k - 1].in_b);
            jobs.push_back(jb);
            p.n_rules += 2 * (nk - 1);
        };
        walk_children(lev[top], 0, (int64_t)lev[top].size(), zero, zero, top_jobs);
        if ((int)down.size() < top + 1) down.resize(top + 1);
        for (int lv = top; lv >= 1; lv--)
            for (const Seg &s : lev[lv]) walk_children(lev[lv - 1], s.child0, s.nchild, s.in_f, s.in_b, down[lv]);
        // ---- walks along the links of every level-0 block: these write the messages ---------------------------------------
        for (const Seg &s : lev[0]) {
            const int64_t la = s.child0, lb = s.child0 + s.nchild - 1;
            Job jf{0, (int32_t)p.steps.size(), (int32_t)s.nchild};
            for (int64_t l = la; l <= lb; l++) {
                const int64_t t = in.tab_fwd[l];
                Step st{{zero, zero, zero}, H(kPtab, t * 3 * dd), H(kBtab, t * dd), H(kPtab, t * 3 * dd + 2 * dd), zero, zero, slot(in.to[l]), 0};
                sources(l == la ? s.in_f : slot(in.to[l - 1]), in.link_pos[l], st.src);
                p.steps.push_back(st);
            }
            link_jobs.push_back(jf);
            Job jb{0, (int32_t)p.steps.size(), (int32_t)s.nchild};
            for (int64_t l = lb; l >= la; l--) {
                const int64_t t = in.tab_bwd[l];
                Step st{{zero, zero, zero}, H(kPtab, t * 3 * dd), H(kBtab, t * dd), H(kPtab, t * 3 * dd + 2 * dd), zero, zero, slot(in.from[l]), 0};
                sources(l == lb ? s.in_b : slot(in.from[l + 1]), (int64_t)in.link_pos[l] + 1, st.src);
                p.steps.push_back(st);
            }
            link_jobs.push_back(jb);
            p.n_rules += 2 * s.nchild;
        }
        l0 = l1 + 1;
    }
    p.levels = max_levels;
    if (p.steps.size() > (size_t)0x7fffffff || p.children.size() > (size_t)0x7fffffff) throw std::runtime_error("plan64: too many records");
    if (!top_jobs.empty()) p.walk_launches.push_back(std::move(top_jobs));
    for (int lv = (int)down.size() - 1; lv >= 1; lv--)
        if (!down[lv].empty()) p.walk_launches.push_back(std::move(down[lv]));
    if (!link_jobs.empty()) p.walk_launches.push_back(std::move(link_jobs));
    // (compose launches of levels no path reaches stay empty and are skipped by the executor)
    return p;
}

}  // namespace plan64
}  // namespace cx
