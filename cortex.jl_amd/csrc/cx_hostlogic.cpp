// cx_hostlogic.cpp — the GPU-free host logic of libcortex_hip.so as a plain C++ translation unit: it compiles with g++
// (no HIP), also under -fsanitize=address,undefined, and exports a small C interface for the CPU tests
// (tests/hostlogic.py).  The product does not link this file: the .hip translation units include the same headers.
//
//   cxh_plan64_*   the work plan of the chain-scan schedule for dim 64 (cx_chain64_plan.h)
#include <cstdio>
#include <cstring>
#include <new>

#include "cx_chain64_plan.h"

using cx::plan64::Plan;

extern "C" {

void *cxh_plan64_create(int32_t d, int64_t npos, int64_t nlinks, const int32_t *link_pos, const int32_t *from, const int32_t *to,
                        const int32_t *tab_fwd, const int32_t *tab_bwd, const uint8_t *head_fwd, const uint8_t *head_bwd,
                        const int32_t *side, int32_t K0, int32_t fan, int64_t lanes, char *err, int32_t errlen) {
    try {
        cx::plan64::Input in;
        in.d = d; in.npos = npos; in.nlinks = nlinks; in.link_pos = link_pos; in.from = from; in.to = to;
        in.tab_fwd = tab_fwd; in.tab_bwd = tab_bwd; in.head_fwd = head_fwd; in.head_bwd = head_bwd; in.side = side;
        in.K0 = K0; in.fan = fan; in.lanes = lanes;
        return new Plan(cx::plan64::build(in));
    } catch (const std::exception &e) {
        if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.what());
        return nullptr;
    }
}

void cxh_plan64_destroy(void *p) { delete (Plan *)p; }

// what: 0 n_pot, 1 n_ent, 2 children, 3 steps, 4 compose launches, 5 walk launches, 6 msg, 7 pot, 8 K0, 9 levels,
//       10 compositions per sweep, 11 rule applications per sweep
int64_t cxh_plan64_info(const void *pv, int32_t what) {
    const Plan *p = (const Plan *)pv;
    switch (what) {
    case 0: return p->n_pot;
    case 1: return p->n_ent;
    case 2: return (int64_t)p->children.size();
    case 3: return (int64_t)p->steps.size();
    case 4: return (int64_t)p->compose_launches.size();
    case 5: return (int64_t)p->walk_launches.size();
    case 6: return p->msg;
    case 7: return p->pot;
    case 8: return p->K0;
    case 9: return p->levels;
    case 10: return p->n_compositions;
    case 11: return p->n_rules;
    }
    return -1;
}

// jobs of launch `idx` (kind 0: compose, 1: walk) as rows {out, first, n}; returns the count (out may be NULL)
int64_t cxh_plan64_jobs(const void *pv, int32_t kind, int32_t idx, int64_t *out) {
    const Plan *p = (const Plan *)pv;
    const auto &L = kind == 0 ? p->compose_launches : p->walk_launches;
    if (idx < 0 || idx >= (int32_t)L.size()) return -1;
    if (out)
        for (size_t i = 0; i < L[idx].size(); i++) { out[3 * i] = L[idx][i].out; out[3 * i + 1] = L[idx][i].first; out[3 * i + 2] = L[idx][i].n; }
    return (int64_t)L[idx].size();
}

// children (what = 0) or steps (what = 1) as rows of ten int64
void cxh_plan64_records(const void *pv, int32_t what, int64_t *out) {
    const Plan *p = (const Plan *)pv;
    static_assert(sizeof(cx::plan64::Child) == 80 && sizeof(cx::plan64::Step) == 80, "ten words per record");
    if (what == 0) std::memcpy(out, p->children.data(), p->children.size() * sizeof(cx::plan64::Child));
    else std::memcpy(out, p->steps.data(), p->steps.size() * sizeof(cx::plan64::Step));
}

}  // extern "C"
